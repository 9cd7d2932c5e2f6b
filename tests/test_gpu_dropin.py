"""Drop-in boundary checks against artefacts of the REAL reference (tests/golden/gen_golden.py: gen_result):
the key sets / value types of what `fit` returns (vlgp/api.py:18-76), a result file written by the
reference's util.save read back and used (8 f4, vlgp/util.py:181-208), and the reference-side ctypes stub that
INTEGRATION.md section B shows, executed as written."""
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def V():
    import vlgp_amd

    return vlgp_amd


def _describe(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, np.ndarray):
            out[k] = ["ndarray", str(v.dtype), list(v.shape)]
        else:
            out[k] = [type(v).__name__, None, None]
    return out


def test_fit_returns_the_references_dicts(V):
    """Same inputs as gen_result (4 trials x 100 bins x 8 channels, 2 latents, FactorAnalysis initialisation):
    the returned dicts carry exactly the reference's keys, with the reference's value types, dtypes and shapes."""
    from vlgp_amd import synth

    with open(os.path.join(GOLDEN, "fit_keys.json")) as f:
        want = json.load(f)
    trials = [{"ID": t["ID"], "y": t["y"].copy()} for t in synth.make_trials(4, 100, 8, 2, seed=3)]
    np.random.seed(9)
    res = V.fit(trials, 2, max_iter=2, min_iter=2, verbose=False)
    assert sorted(res.keys()) == want["top"]
    assert type(res["trials"]).__name__ == want["trials_type"] and res["trials"] is trials
    got = {"trial": _describe(res["trials"][0]), "params": _describe(res["params"]),
           "config": _describe(res["config"]), "runtime": _describe(res["config"]["runtime"])}
    for sec in ("trial", "params", "config", "runtime"):
        extra = set(got[sec]) - set(want[sec])
        if sec == "config":
            extra -= {"ichol"}  # the build's one extra option
        assert not extra and not (set(want[sec]) - set(got[sec])), (sec, extra, set(want[sec]) - set(got[sec]))
        for k, w in want[sec].items():
            assert got[sec][k] == w, (sec, k, got[sec][k], w)
    assert sorted(res["params"]["initial"].keys()) == want["initial"]
    assert sorted(int(k) for k in res["params"]["cholesky"].keys()) == want["cholesky_keys"]
    assert res["config"]["runtime"]["it"] == 2 and len(res["config"]["runtime"]["em_elapsed"]) == 2


def test_initial_cholesky_without_a_window(V):
    """window=None: vlgp/api.py:60 deep-copies params after make_cholesky(trials), so params["initial"]["cholesky"]
    holds the full-length factors of the INITIAL omega, sigma, keyed by trial length.  Same keys, and the arrays are
    what the oracle's restatement of math.ichol_gauss (pinned to the reference's own ichol.npz) builds from
    params["initial"]["omega"] / ["sigma"]."""
    from oracle import vlgp_oracle as O
    from vlgp_amd import synth

    trials = [{"ID": t["ID"], "y": t["y"].copy()} for t in synth.make_trials(3, 120, 8, 2, seed=4)]
    trials[1]["y"] = trials[1]["y"][:90].copy()  # two distinct lengths
    np.random.seed(2)
    # (Hstep=False: the reference's own H-step cannot run without a window -- gp.optimize builds np.arange(window))
    res = V.fit(trials, 2, max_iter=2, min_iter=2, window=None, Hstep=False, verbose=False)
    init = res["params"]["initial"]
    assert sorted(int(k) for k in init["cholesky"].keys()) == [90, 120]
    assert sorted(int(k) for k in res["params"]["cholesky"].keys()) == [90, 120]
    want = O.build_prior([90, 120], init["omega"], init["sigma"], init["rank"])
    for T in (90, 120):
        G = init["cholesky"][T]
        assert G.shape == (2, T, init["rank"])
        assert relerr(G, want[T]) < 1e-12, T
    # the copy is detached from the live params, as a deepcopy is
    assert init["cholesky"] is not res["params"]["cholesky"]
    # ... and a saved result loads where vlgp_amd is not installed: the lazy dict pickles as a plain one
    import pickle
    import pickletools

    # (only params["transform"] still names this package -- the estimator's bound method, as the reference's names scikit-learn)
    blob = pickle.dumps({"cholesky": res["params"]["cholesky"], "initial": {"cholesky": init["cholesky"]}})
    assert b"vlgp_amd" not in blob, [arg for op, arg, _ in pickletools.genops(blob) if arg and "vlgp" in str(arg)]
    back = pickle.loads(blob)
    assert type(back["initial"]["cholesky"]) is dict and np.array_equal(back["initial"]["cholesky"][90], init["cholesky"][90])


@pytest.mark.parametrize("ext", ["npy", "npz"])
def test_reference_written_result_file_is_usable(V, ext):
    """util.save of the reference wrote tests/golden/ref_result.{npy,npz}; vlgp_amd.load reads it and
    vlgp_amd.transform continues from it: the latents it infers for the stored trials under the stored
    parameters equal the oracle's inference from the same file."""
    from oracle import vlgp_oracle as O

    res = V.load(os.path.join(GOLDEN, "ref_result." + ext))
    assert sorted(res.keys()) == ["config", "params", "trials"]
    trials = [dict(t) for t in res["trials"]]
    params, config = res["params"], res["config"]
    assert params["a"].shape == (2, 8) and trials[0]["mu"].shape == (100, 2)
    mine = [{"ID": t["ID"], "y": t["y"].copy(), "mu": t["mu"].copy()} for t in trials]
    params = dict(params)
    params["transform"] = lambda y: np.zeros((y.shape[0], 2))  # the stored file carries no estimator; mu is supplied
    V.transform(mine, params, config)
    ref = [{"y": t["y"].copy(), "mu": t["mu"].copy(), "x": np.ones((100, 1, 8)), "w": np.zeros((100, 2)),
            "v": np.zeros((100, 2)), "dmu": np.zeros((100, 2))} for t in trials]
    p2 = {k: params[k] for k in ("ydim", "zdim", "xdim", "a", "b", "noise", "sigma", "omega", "rank", "gp_noise",
                                 "dt", "likelihood")}
    p2["cholesky"] = {int(k): v for k, v in res["params"]["cholesky"].items()}
    cfg = O.make_config(**{k: v for k, v in config.items() if k != "runtime"})
    O.infer(ref, p2, cfg)   # api.transform: w = v = 0, then core.infer (vlgp/api.py:181-183)
    for a, b in zip(mine, ref):
        for k in ("mu", "v", "w"):
            assert relerr(a[k], b[k]) < 1e-9, k


def test_integration_md_stub_runs_as_written(V, golden):
    """the `vlgp/_hip.py` ctypes stub a maintainer of the reference would add, taken
    from the document verbatim (only the library path is made absolute) and run on the reference's own E-step
    fixture: core.estep's outputs to 1e-9."""
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        doc = f.read()
    blocks = re.findall(r"```python\n(# vlgp/_hip\.py.*?)```", doc, flags=re.S)
    assert len(blocks) == 1
    from vlgp_amd import _lib

    src = blocks[0].replace('"libvlgp_hip.so"', repr(_lib.LIB_PATH))
    ns = {}
    exec(compile(src, "INTEGRATION.md:vlgp/_hip.py", "exec"), ns)
    g = golden("estep_pois")
    units = [{k: g[k + "0"][m].copy() for k in ("y", "x", "mu", "v", "w")} for m in range(4)]
    for u in units:
        u["dmu"] = np.zeros_like(u["mu"])
    params = {"ydim": 20, "zdim": 3, "xdim": 1, "rank": 50, "a": g["a"].copy(), "b": g["b"].copy(),
              "noise": g["noise"].copy(), "likelihood": np.where(g["gauss"], "gaussian", "poisson"),
              "cholesky": {50: g["G"]}}
    mu_ids = [id(u["mu"]) for u in units]
    ns["estep"](units, params, {"Eniter": 25, "dmu_bound": 5.0, "method": "VB"})
    for m, u in enumerate(units):
        assert id(u["mu"]) == mu_ids[m]
        for k in ("mu", "v", "w"):
            assert relerr(u[k], g["%s_VB_25" % k][m]) < 1e-9, (k, m)


def test_graft_entry_smoke_runs():
    """__graft_entry__.smoke() -- what the driver runs on the GPU box before the bench -- is itself under test (it toggles
    a VLGP_HSTEP_* switch, and those are read when the handle is created: round 5 broke it once, unnoticed)."""
    import importlib
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.insert(0, root)
    g = importlib.import_module("__graft_entry__")
    g.smoke()
