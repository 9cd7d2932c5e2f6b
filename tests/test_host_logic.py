"""CPU-side checks: the C-ABI library loads and exports every declared symbol,
and the host mirror of the reference interface behaves like the reference."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT, relerr
from oracle import vlgp_oracle as O

HEADER = os.path.join(ROOT, "include", "vlgp_hip.h")


def _declared():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(vlgp_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_header_symbol():
    from vlgp_amd import _lib

    lib = _lib.load()
    names = _declared()
    assert len(names) >= 30
    for name in names:
        assert hasattr(lib, name), "libvlgp_hip.so lacks %s" % name
    assert sorted(_lib.EXPORTS) == names, set(names) ^ set(_lib.EXPORTS)
    assert lib.vlgp_abi_version() == _lib.ABI_VERSION


def test_no_gpu_fails_loudly():
    import vlgp_amd
    from vlgp_amd import _lib

    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(vlgp_amd.VlgpError, match="no HIP device"):
        vlgp_amd.Engine(4, 2, 1, 50)
    trials = [{"y": np.zeros((50, 4))}]
    with pytest.raises(vlgp_amd.VlgpError):
        vlgp_amd.fit(trials, 2, verbose=False)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "vlgp_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(base, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_config_defaults_are_the_references():
    from vlgp_amd import get_config

    ref = {"constrain_loading": "fro", "constrain_latent": False, "use_hessian": True, "eps": 1e-8,
           "tol": 1e-8, "min_iter": 5, "method": "VB", "learning_rate": 1.0, "max_iter": 20, "Eniter": 25,
           "Mniter": 25, "Hstep": True, "da_bound": 5.0, "db_bound": 5.0, "dmu_bound": 5.0,
           "omega_bound": (5e-4, 5e-2), "window": 50, "saving_interval": 1800, "callbacks": [],
           "parallel": False}  # vlgp/preprocess.py:85-106
    cfg = get_config()
    for k, v in ref.items():
        assert cfg[k] == v, k
    assert get_config(bogus=1, Eniter=3)["Eniter"] == 3 and "bogus" not in get_config(bogus=1)
    assert get_config()["callbacks"] is not get_config()["callbacks"]


def test_params_skeleton():
    from vlgp_amd import get_params

    trials = [{"y": np.zeros((10, 7))}]
    p = get_params(trials, 3, omega_bound=(5e-4, 5e-2), lik=["poisson"] * 5 + ["gaussian"] * 2, history=2)
    assert (p["ydim"], p["zdim"], p["xdim"], p["rank"]) == (7, 3, 2, 50)
    assert np.all(p["omega"] == 5e-2) and np.all(p["sigma"] == 1) and p["gp_noise"] == 1e-4
    assert list(p["likelihood"][-2:]) == ["gaussian", "gaussian"]


def test_segment_starts_match_reference_rng_protocol():
    from vlgp_amd.util import cut_trials, segment_starts

    rng_trials = [{"y": np.arange(T * 2.0).reshape(T, 2), "x": np.ones((T, 1, 2)), "mu": np.zeros((T, 3)),
                   "w": np.zeros((T, 3)), "v": np.zeros((T, 3))} for T in (100, 130, 50, 75)]
    np.random.seed(5)
    mine = cut_trials(rng_trials, None, {"window": 50})
    np.random.seed(5)
    ref = O.cut_trials(rng_trials, 50)
    assert len(mine) == len(ref) == 2 + 3 + 1 + 2
    for a, b in zip(mine, ref):
        assert np.array_equal(a["y"], b["y"])
        assert np.shares_memory(a["mu"], b["mu"])  # views of the parent, as in the reference
    assert list(segment_starts(100, 50)) == [0, 50]


def test_host_ichol_is_the_oracles():
    from vlgp_amd.gp import ichol_gauss_host

    for n, om in ((50, 5e-2), (200, 3e-3), (64, 1e-2)):
        assert np.array_equal(ichol_gauss_host(n, om, 50), O.ichol_gauss(n, om, 50))


def test_shard_bounds_partition():
    from vlgp_amd.dist import shard_bounds

    for n in (1, 7, 200, 203):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(n, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == n
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def test_weighted_shards_balance_rows_of_ragged_trials():
    """Ragged trials (BASELINE configs[4]) are cut into contiguous blocks of about equal ROWS, identically on every
    rank; equal-length trials keep the equal-count split."""
    from vlgp_amd.dist import shard, shard_bounds, shard_bounds_weighted

    rng = np.random.default_rng(0)
    for _ in range(300):
        n, world = int(rng.integers(1, 60)), int(rng.integers(1, 9))
        w = (50 * rng.integers(10, 41, n)).tolist()
        cuts = [shard_bounds_weighted(w, r, world) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
        if n >= world:
            assert all(hi > lo for lo, hi in cuts)
    w = (50 * np.random.default_rng(0).integers(10, 41, 500)).tolist()
    for world in (2, 4, 8):
        rows = [sum(w[lo:hi]) for lo, hi in (shard_bounds_weighted(w, r, world) for r in range(world))]
        assert max(rows) <= 1.02 * sum(w) / world
    trials = [{"y": np.zeros((t, 2))} for t in (100, 100, 100, 100, 400, 400)]
    assert [len(shard(trials, r, 2)) for r in range(2)] == [5, 1]          # 800 | 400 rows, not 3 | 3 trials (300 | 900)
    same = [{"y": np.zeros((50, 2))} for _ in range(7)]
    assert [len(shard(same, r, 3)) for r in range(3)] == [hi - lo for lo, hi in (shard_bounds(7, r, 3) for r in range(3))]


def test_mstep_sufficient_statistics_form_equals_mstep(golden):
    for tag in ("p1", "p3", "mixed"):
        g = golden("mstep_" + tag)
        cat = lambda k: np.concatenate(list(g[k]), axis=0)
        got = O.mstep_sharded(cat("y"), cat("x"), cat("mu"), cat("v"), g["a"], g["b"], g["gauss"], 25)
        for k, arr in zip(("a", "b", "da", "db", "noise"), got):
            if k in ("da", "db"):  # ~1e-7 increments: judge them on the scale of a, b
                assert np.abs(arr - g[k + "_H_25"]).max() < 1e-9 * np.abs(g[k[1] + "_H_25"]).max(), (tag, k)
            else:
                assert relerr(arr, g[k + "_H_25"]) < 1e-9, (tag, k)


def test_synth_is_seeded():
    from vlgp_amd import synth

    a = synth.make_trials(3, 100, 6, 4, seed=0)
    b = synth.make_trials(3, 100, 6, 4, seed=0)
    assert all(np.array_equal(x["y"], y["y"]) for x, y in zip(a, b))
    assert a[0]["y"].shape == (100, 6) and a[0]["y"].min() >= 0


def test_initialize_matches_reference(golden):
    """preprocess.initialize (FactorAnalysis on the seeded subsample) against the reference's."""
    from vlgp_amd import get_config, get_params, synth
    from vlgp_amd.preprocess import initialize

    g = golden("init_c1")
    n_trials, n_bins, N, L = synth.CONFIGS["C1"]
    trials = synth.make_trials(n_trials, n_bins, N, L, seed=0)
    cfg = get_config()
    params = get_params(trials, L, omega_bound=cfg["omega_bound"])
    np.random.seed(7)
    initialize(trials, params, cfg)
    assert relerr(params["a"], g["a"]) < 1e-10 and relerr(params["b"], g["b"]) < 1e-12
    assert relerr(np.stack([t["mu"] for t in trials]), g["mu"]) < 1e-9
    assert trials[0]["x"].shape == tuple(g["x_shape"]) and np.all(trials[0]["w"] == 0)


@pytest.mark.parametrize("shape", [(10, 200, 20, 3), (20, 300, 40, 10), (30, 400, 60, 5)])
def test_factor_analysis_matches_sklearn(shape):
    """vlgp_amd/fa.py restates sklearn's FactorAnalysis(random_state=0) from the second-moment matrix alone;
    scikit-learn is the reference's own dependency for this step (vlgp/preprocess.py:18-19)."""
    sk = pytest.importorskip("sklearn.decomposition")
    from vlgp_amd import synth
    from vlgp_amd.fa import fit_factor_analysis

    n_trials, n_bins, N, L = shape
    trials = synth.make_trials(n_trials, n_bins, N, L, seed=1)
    y = np.concatenate([t["y"] for t in trials], axis=0)
    rng = np.random.RandomState(3)
    sub = y[rng.choice(y.shape[0], y.shape[0] // 10)]
    fa = sk.FactorAnalysis(n_components=L, random_state=0)
    z_ref = fa.fit_transform(sub)
    own = fit_factor_analysis(sub, L, seed=0)
    assert own.n_iter == fa.n_iter_
    assert relerr(own.components, fa.components_) < 1e-9
    assert relerr(own.noise_variance, fa.noise_variance_) < 1e-9
    assert relerr(own.transform(sub), z_ref) < 1e-9
    assert relerr(own.transform(y[:500]), fa.transform(y[:500])) < 1e-9


def test_sample_posterior_lowrank_form_has_the_reference_covariance():
    """The low-rank form of api.sample_posterior that the device kernel implements (oracle restatement):
    G (I + G'WG)^-1 G' equals the reference's inv(inv(K + reg I) + W) (vlgp/api.py:156-166) up to reg."""
    rng = np.random.default_rng(0)
    T, L = 40, 2
    omega, sigma = np.array([4e-3, 2e-2]), np.array([1.0, 0.7])
    chol = O.build_prior([T], omega, sigma, 50)
    w = rng.random((T, L)) * 3.0
    for l in range(L):
        G = chol[T][l]
        r = int(np.flatnonzero(np.any(G != 0, axis=0))[-1]) + 1
        Gl = G[:, :r]
        H = Gl.T @ (w[:, [l]] * Gl)
        cov = Gl @ np.linalg.solve(np.eye(r) + H, Gl.T)
        want = O.posterior_covariance_reference(G, w[:, l])
        assert np.abs(cov - want).max() < 1e-4 * np.abs(want).max()
        # and the draws of the restatement have exactly that covariance: (G Lc^-T)(G Lc^-T)' = cov
        eye = [np.eye(r) if k == l else np.zeros((int(np.flatnonzero(np.any(chol[T][k] != 0, axis=0))[-1]) + 1, r))
               for k in range(L)]
        dev = O.sample_posterior_lowrank(np.zeros((T, L)), w, chol[T], eye)[:, :, l].T  # columns: G Lc^-T e_i
        assert np.abs(dev @ dev.T - cov).max() < 1e-12


def test_save_load_round_trip(tmp_path):
    from vlgp_amd import load, save

    result = {"params": {"a": np.arange(6.0).reshape(2, 3), "omega": np.array([1e-3, 2e-3])},
              "config": {"window": 50, "method": "VB"}, "trials": [{"ID": 0, "mu": np.ones((4, 2))}]}
    save(result, tmp_path / "fit", "npy")
    back = load(tmp_path / "fit.npy")
    assert back["config"] == result["config"] and np.array_equal(back["params"]["a"], result["params"]["a"])
    assert np.array_equal(back["trials"][0]["mu"], np.ones((4, 2)))
    save({"a": result["params"]["a"], "omega": result["params"]["omega"]}, tmp_path / "arrs", "npz")
    arrs = load(tmp_path / "arrs.npz")
    assert set(arrs) == {"a", "omega"} and np.array_equal(arrs["omega"], result["params"]["omega"])
    with pytest.raises(FileNotFoundError):
        load(tmp_path / "missing.npy")


def test_load_reads_the_references_own_result_files():
    """tests/golden/ref_result.{npy,npz} were written by the REFERENCE's util.save (vlgp/util.py:181-190, gen_golden.py:
    gen_result) from a real fit: `vlgp_amd.load` hands back the same dict from either format."""
    from conftest import GOLDEN

    from vlgp_amd import util as U

    one = U.load(os.path.join(GOLDEN, "ref_result.npy"))
    two = U.load(os.path.join(GOLDEN, "ref_result.npz"))
    for res in (one, two):
        assert sorted(res.keys()) == ["config", "params", "trials"]
        assert len(res["trials"]) == 4 and res["trials"][0]["y"].shape == (100, 8)
        assert res["params"]["a"].shape == (2, 8) and set(res["params"]["cholesky"]) == {100}
        assert res["config"]["runtime"]["it"] == 2 and res["config"]["window"] == 50
    for k in ("a", "b", "noise", "omega", "sigma"):
        assert np.array_equal(one["params"][k], two["params"][k])
    for t1, t2 in zip(one["trials"], two["trials"]):
        for k in ("y", "mu", "v", "w", "dmu", "x"):
            assert np.array_equal(t1[k], t2[k])
    with pytest.raises(FileNotFoundError):
        U.load(os.path.join(GOLDEN, "no_such_result.npy"))
    with pytest.raises(NotImplementedError):
        U.load(os.path.join(GOLDEN, "gen_golden.py"))


def test_blas_thread_limit_only_lowers():
    """preprocess._few_blas_threads caps OpenBLAS for the factor analysis but must never RAISE the thread count: under
    torchrun (OMP_NUM_THREADS=1) an OpenBLAS sized for one thread crashes when asked for eight (seen on the MI355X
    hosts: SIGSEGV in numpy.linalg inside bench.py --gpus 2)."""
    import subprocess
    import sys

    from conftest import ROOT

    code = ("import numpy as np\n"
            "from threadpoolctl import threadpool_info\n"
            "from vlgp_amd.preprocess import _few_blas_threads\n"
            "n = lambda: max([m['num_threads'] for m in threadpool_info() if m.get('user_api') == 'blas'] or [1])\n"
            "before = n()\n"
            "with _few_blas_threads(8):\n"
            "    inside = n()\n"
            "    np.linalg.svd(np.random.default_rng(0).standard_normal((60, 40)))\n"
            "assert inside <= before and inside <= 8, (before, inside)\n"
            "assert n() == before\n"
            "print('ok', before, inside)\n")
    for env_threads in ("1", "4"):
        env = dict(os.environ, OMP_NUM_THREADS=env_threads, OPENBLAS_NUM_THREADS=env_threads)
        done = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert done.returncode == 0 and "ok" in done.stdout, done.stderr[-1500:]


def test_bench_refuses_a_world_that_is_not_the_gpus_asked_for():
    """`bench.py --gpus 8` inside a one-rank launch must not print an n_gpus: 1 line."""
    import subprocess
    import sys

    from conftest import ROOT

    env = dict(os.environ, WORLD_SIZE="1", RANK="0")
    done = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                          env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert done.returncode != 0 and "--gpus 8 but WORLD_SIZE=1" in done.stderr
    assert not [ln for ln in done.stdout.splitlines() if ln.startswith("{")]


def test_initial_prior_behaves_like_the_dict_it_stands_for():
    """engine._InitialPrior (params["initial"]["cholesky"] of a fit without a window): get() answers for a length
    `in` reports, a deep copy stays lazy, and a pickle is a plain {T: ndarray} dict with no vlgp_amd class in it."""
    import copy
    import pickle

    from vlgp_amd import engine as E

    d = E._InitialPrior([40, 60], [1e-2, 5e-3], [1.0, 1.0], 8)
    assert 40 in d and 41 not in d and d.get(41) is None and d.get(41, 7) == 7
    G = d.get(40)
    assert G is not None and G.shape == (2, 40, 8) and np.array_equal(G, d[40])
    want = O.build_prior([40, 60], np.array([1e-2, 5e-3]), np.ones(2), 8)
    assert relerr(G, want[40]) < 1e-12
    c = copy.deepcopy(d)
    assert isinstance(c, E._InitialPrior) and np.array_equal(c[60], d[60])
    blob = pickle.dumps({"initial": {"cholesky": d}})
    assert b"vlgp_amd" not in blob
    back = pickle.loads(blob)["initial"]["cholesky"]
    assert type(back) is dict and sorted(back) == [40, 60] and np.array_equal(back[60], d[60])
