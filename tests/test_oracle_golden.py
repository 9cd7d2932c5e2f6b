"""The CPU oracle against golden vectors captured from the real reference
(tests/golden/gen_golden.py) -- this is what pins the oracle.

Tolerances: the restatement follows the reference's operation order, so stage
outputs agree to ~1e-13; anything that passes through L-BFGS-B is held to 1e-9.
"""
import numpy as np
import pytest
from scipy.linalg import toeplitz

import os
import sys

from conftest import relerr
from oracle import vlgp_oracle as O

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))  # golden_cases

STAGE_TOL = 1e-12


def test_reference_own_ichol_assertion():
    # the reference's only numeric test on this path: tests/test_math.py:7-14
    n, omega = 500, 1
    K = toeplitz(np.exp(-omega * np.arange(n) ** 2))
    G = O.ichol_gauss(n, omega, n)
    assert np.allclose(K, G @ G.T)


def test_ichol_golden(golden):
    g = golden("ichol")
    for i, (n, om, r) in enumerate(g["cases"]):
        G = O.ichol_gauss(int(n), om, int(r))
        if n > 200:
            assert np.array_equal(G[::8], g["G%d_rows" % i])
            assert relerr(np.sum(G * G, axis=1), g["G%d_diag" % i]) < 1e-14
            assert relerr(G.sum(axis=0), g["G%d_colsum" % i]) < 1e-13
        else:
            assert np.array_equal(G, g["G%d" % i])


@pytest.mark.parametrize("tag", ["pois", "mixed"])
def test_estep_golden(golden, tag):
    g = golden("estep_" + tag)
    for method in ("VB", "MAP"):
        for n_it in (1, 25):
            for m in range(g["y0"].shape[0]):
                out = O.estep_unit(g["y0"][m], g["x0"][m], g["mu0"][m], g["v0"][m],
                                   g["w0"][m], g["a"], g["b"], g["noise"], g["gauss"],
                                   g["G"], n_it, 5.0, method == "VB")
                assert out[4] == 0
                for k, arr in zip(("mu", "v", "w", "dmu"), out):
                    assert relerr(arr, g["%s_%s_%d" % (k, method, n_it)][m]) < STAGE_TOL


@pytest.mark.parametrize("tag", ["pois", "mixed"])
def test_update_w_v_golden(golden, tag):
    g = golden("estep_" + tag)
    for m in range(g["y0"].shape[0]):
        zero = np.zeros_like(g["mu0"][m])
        w = O.curvature_unit(g["y0"][m], g["x0"][m], g["mu0"][m], zero, g["a"], g["b"],
                             g["noise"], g["gauss"])
        assert relerr(w, g["w_stage"][m]) < STAGE_TOL
        v, bad = O.variance_unit(w, zero, g["G"])
        assert bad == 0 and relerr(v, g["v_stage"][m]) < STAGE_TOL


def test_estep_long_golden(golden):
    g = golden("estep_long")
    out = O.estep_unit(g["y0"][0], g["x0"][0], g["mu0"][0], g["v0"][0], g["w0"][0], g["a"],
                       g["b"], g["noise"], g["gauss"], g["G"], 5)
    for k, arr in zip(("mu", "v", "w", "dmu"), out):
        assert relerr(arr, g[k + "_VB_5"][0]) < STAGE_TOL


@pytest.mark.parametrize("tag", ["p1", "p3", "mixed"])
def test_mstep_golden(golden, tag):
    g = golden("mstep_" + tag)
    cat = lambda k: np.concatenate(list(g[k]), axis=0)
    for key in ("H_1", "H_25", "G_1", "G_25"):
        out = O.mstep_arrays(cat("y"), cat("x"), cat("mu"), cat("v"), g["a"], g["b"],
                             g["gauss"], int(key[2:]), key[0] == "H", 1e-8,
                             float(g["lr_" + key]))
        for k, arr in zip(("a", "b", "da", "db", "noise"), out):
            assert relerr(arr, g["%s_%s" % (k, key)]) < STAGE_TOL


def test_hstep_objective_golden(golden):
    g = golden("hstep")
    t = np.arange(50.0)
    for l in range(3):
        for i, p in enumerate(g["logp"]):
            ll, dll = O.gp_objective(p, t, g["mu"][:, :, l].T, g["w"][:, :, l].T)
            assert abs(ll - g["ll"][l, i]) <= 1e-12 * abs(g["ll"][l, i])
            assert relerr(dll, g["dll"][l, i]) < 1e-11


def test_hstep_optimize_golden(golden):
    g = golden("hstep")
    sig, om = O.hstep_arrays(g["mu"], g["w"], g["sigma0"], g["omega0"], 1e-4,
                             (5e-4, 5e-2), 50)
    assert relerr(om, g["omega_opt"]) < 1e-9
    assert relerr(sig, g["sigma_opt"]) < 1e-12


def _c1_trials(g):
    y = g["y"].astype(float)
    n, T, N = y.shape
    L = g["mu0"].shape[-1]
    return [{"ID": i, "y": y[i].copy(), "mu": g["mu0"][i].copy(), "x": np.ones((T, 1, N)),
             "w": np.zeros((T, L)), "v": np.zeros((T, L))} for i in range(n)]


@pytest.mark.parametrize("tag,hs", [("H0", False), ("H1", True)])
def test_vem_trajectory_golden(golden, tag, hs):
    g = golden("vem_c1")
    trials = _c1_trials(g)
    cfg = O.make_config(Hstep=hs, max_iter=6, min_iter=6)
    params = O.make_params(trials, 3, a=g["a0"].copy(), b=g["b0"].copy())
    params["da"] = np.zeros_like(params["a"])
    params["db"] = np.zeros_like(params["b"])
    O.fill_trials(trials)
    O.make_cholesky(trials, params)
    O.update_w(trials, params)
    O.update_v(trials, params, cfg)
    segs = O.cut_trials(trials, 50)
    O.make_cholesky(segs, params)
    O.fill_trials(segs)
    traj = []
    cfg["callbacks"] = [lambda t_, p_, c_: traj.append(
        (np.linalg.norm(np.concatenate([s["mu"] for s in t_])), np.linalg.norm(p_["a"]),
         np.array(p_["omega"])))]
    O.vem(segs, params, cfg)
    tol = 1e-9 if hs else STAGE_TOL
    assert cfg["runtime"]["it"] == int(g["it_" + tag])
    assert relerr([t[0] for t in traj], g["norm_mu_" + tag]) < tol
    assert relerr([t[1] for t in traj], g["norm_a_" + tag]) < tol
    assert relerr(np.array([t[2] for t in traj]), g["omega_" + tag]) < tol
    assert relerr(params["a"], g["a_" + tag]) < tol
    assert relerr(params["b"], g["b_" + tag]) < tol
    assert relerr(np.stack([s["mu"] for s in segs]), g["seg_mu_" + tag]) < tol
    assert relerr(np.stack([s["v"] for s in segs]), g["seg_v_" + tag]) < tol


def test_c5_small_vem_golden(golden):
    """The C5 combination (four distinct trial lengths, 30 Poisson + 10 Gaussian channels, ten latents), two EM iterations
    with the H-step on: the oracle against what the REAL reference produced (gen_golden.py c5_small)."""
    import golden_cases

    g = golden("c5_small")
    trials0, a0, b0, mu0, lik = golden_cases.c5_small_inputs()
    assert float(np.concatenate([t["y"] for t in trials0]).sum()) == float(g["y_checksum"][0])
    L, N = a0.shape
    trials = [{"ID": t["ID"], "y": t["y"].copy(), "mu": m.copy(), "x": np.ones((t["y"].shape[0], 1, N)),
               "w": np.zeros((t["y"].shape[0], L)), "v": np.zeros((t["y"].shape[0], L))} for t, m in zip(trials0, mu0)]
    cfg = O.make_config(max_iter=2, min_iter=2)
    params = O.make_params(trials, L, a=a0.copy(), b=b0.copy(), lik=lik)
    params["da"] = np.zeros_like(params["a"])
    params["db"] = np.zeros_like(params["b"])
    O.fill_trials(trials)
    O.make_cholesky(trials, params)
    O.update_w(trials, params)
    O.update_v(trials, params, cfg)
    segs = O.cut_trials(trials, 50)
    O.make_cholesky(segs, params)
    O.fill_trials(segs)
    traj = []
    cfg["callbacks"] = [lambda t_, p_, c_: traj.append(
        (np.linalg.norm(np.concatenate([s["mu"] for s in t_])), np.linalg.norm(p_["a"]), np.array(p_["omega"]),
         np.array(p_["noise"])))]
    O.vem(segs, params, cfg)
    tol = 1e-9
    assert cfg["runtime"]["it"] == int(g["it"]) == 2
    assert relerr([t[0] for t in traj], g["norm_mu"]) < tol
    assert relerr([t[1] for t in traj], g["norm_a"]) < tol
    assert relerr(np.array([t[2] for t in traj]), g["omega"]) < tol
    assert relerr(np.array([t[3] for t in traj]), g["noise_traj"]) < tol
    for k in ("a", "b", "noise"):
        assert relerr(params[k], g[k]) < tol, k
    for k in ("mu", "v", "w"):
        assert relerr(np.stack([segs[i][k] for i in g["pick"]]), g["seg_" + k]) < tol, k


def test_fit_golden(golden):
    g = golden("fit_c1")
    trials = _c1_trials(g)
    cfg = O.make_config(Hstep=False, max_iter=5, min_iter=5)
    params = O.make_params(trials, 3, a=g["a0"].copy(), b=g["b0"].copy())
    O.fit_given_init(trials, params, cfg)
    for k in ("mu", "v", "w", "dmu"):
        assert relerr(np.stack([t[k] for t in trials]), g[k]) < STAGE_TOL
    assert relerr(params["a"], g["a"]) < STAGE_TOL
    assert np.array_equal(params["cholesky"][200], g["G200"])


# ---- the non-default branches, pinned to the real reference (tests/golden/branches.npz, tests/golden_cases.py) ----
def _oracle_branch_fit(name):
    import golden_cases

    fresh, a0, b0, lik, dims, history, run = golden_cases.case_inputs(name)
    n_trials, n_bins, N, L = dims
    ref = fresh()
    xdim = max(history, 1)
    for t in ref:
        t["x"] = np.ones((n_bins, xdim, N))
        t["w"] = np.zeros((n_bins, L))
        t["v"] = np.zeros((n_bins, L))
    cfg = O.make_config(**run)
    params = O.make_params(ref, L, a=a0.copy(), b=b0.copy(), lik=lik, history=history, omega_bound=cfg["omega_bound"])
    O.fit_given_init(ref, params, cfg)
    return ref, params, cfg, dims


BRANCHES = ["loading_svd", "loading_1", "loading_2", "loading_inf", "loading_off", "latent_location", "latent_scale",
            "latent_both", "svd_and_both", "window_25", "window_40", "window_100", "window_200", "all_gaussian", "history_2"]


@pytest.mark.parametrize("name", BRANCHES)
def test_branches_golden(golden, name):
    """core.constrain_loading / constrain_latent in every mode (vlgp/core.py:366-416), windows 25 / 40 / 100,
    all-Gaussian channels and history = 2: `fit` of the oracle against `fit` of the real reference."""
    g = golden("branches")
    ref, params, cfg, dims = _oracle_branch_fit(name)
    assert cfg["runtime"]["it"] == int(g[name + "__it"])
    # through L-BFGS-B: 1e-9 (measured 1e-14 ... 1e-11).  Two stated exceptions: at window 25 one latent's line search
    # ends one evaluation apart (SciPy stops on a relative decrease of 2.2e-9 of the objective, which pins omega to a
    # few 1e-6 at best; measured 4.4e-6 on omega, 6e-7 on a); with history = 2 the two regressors are both columns of
    # ones, the 2 x 2 Newton system of b is singular up to the 1e-8 jitter and LAPACK's posv (the reference, through
    # the sym_pos shim) and potrf + potrs (the oracle) differ by 6e-8 on b
    tol = {"window_25": {"omega": 1e-5, "a": 1e-5, "b": 1e-5, "noise": 1e-5}, "history_2": {"b": 1e-6}}.get(name, {})
    for k in ("a", "b", "noise", "omega", "sigma"):
        assert relerr(params[k], g["%s__%s" % (name, k)]) < tol.get(k, 1e-9), k
    G = params["cholesky"][dims[1]]
    if np.array_equal(G[:, ::10], g[name + "__G_rows"]):  # same pivots -> the end-to-end posterior is comparable
        for k in ("mu", "v"):
            assert relerr(np.stack([ref[i][k] for i in (0, -1)]), g["%s__%s" % (name, k)]) < (1e-4 if tol else 1e-8), k
    else:
        # a near-tie in the pivot search of a rank-exhausted factor (200-bin trials at rank 50) broken the other way by an
        # omega that differs in its last digits: another, equally valid, incomplete factor ("pivot
        # hazard"); ichol_gauss itself is pinned bit for bit by test_ichol_golden, omega above.  Nothing to compare.
        assert G.shape[0] == g[name + "__G_rows"].shape[0]


def test_transform_golden(golden):
    """api.transform (vlgp/api.py:171-184) = initialize (mu = transform(y), w = v = 0) + one core.infer."""
    import golden_cases
    from vlgp_amd import synth

    g = golden("branches")
    new = synth.make_trials(3, 100, 12, 3, seed=8)
    N, L = 12, 3
    gauss = np.zeros(N, bool)
    for i, t in enumerate(new):
        out = O.estep_unit(t["y"], np.ones((100, 1, N)), g["transform__mu0"][i], np.zeros((100, L)), np.zeros((100, L)),
                           g["transform__a"], g["transform__b"], g["transform__noise"], gauss, g["transform__G"],
                           int(g["transform__max_iter"]))
        for k, arr in zip(("mu", "v", "w"), out):
            assert relerr(arr, g["transform__" + k][i]) < STAGE_TOL, (k, i)


@pytest.mark.parametrize("n_it", [1, 3])
def test_mstep_singular_golden(golden, n_it):
    """A Newton system that does not factor -> the gradient step learning_rate * grad (vlgp/core.py:191-198)."""
    import golden_cases

    d = golden_cases.singular_mstep_inputs()
    g = golden("mstep_singular")
    N = d["y"].shape[1]
    a, b, da, db, noise = O.mstep_arrays(d["y"], d["x"], d["mu"], d["v"], d["a"].copy(), d["b"].copy(),
                                         np.zeros(N, bool), n_it, use_hessian=True, eps=0.0, learning_rate=d["lr"])
    for k, arr in zip(("a", "b", "da", "db", "noise"), (a, b, da, db, noise)):
        assert relerr(arr, g["%s_%d" % (k, n_it)]) < STAGE_TOL, k


def test_ragged_window_golden(golden):
    """Trial lengths that are not multiples of the window (vlgp/util.py:482-496): overlapping segments as views of the
    trial arrays, updated one after the other in place, with the reference's own multinomial draw of the overlaps."""
    import golden_cases

    g = golden("ragged_window")
    fresh, a0, b0, (lengths, N, L), run = golden_cases.ragged_window_inputs()
    ref = fresh()
    for t in ref:
        n = t["y"].shape[0]
        t["x"] = np.ones((n, 1, N))
        t["w"] = np.zeros((n, L))
        t["v"] = np.zeros((n, L))
    cfg = O.make_config(**run)
    params = O.make_params(ref, L, a=a0.copy(), b=b0.copy(), omega_bound=cfg["omega_bound"])
    np.random.seed(4)
    np.random.choice(sum(lengths), max(sum(lengths) // 10, 50))  # initialize() draws its subsample first (preprocess.py:14)
    O.fit_given_init(ref, params, cfg)
    for k in ("a", "b", "noise", "omega", "sigma"):
        assert relerr(params[k], g[k]) < 1e-9, k
    assert relerr(ref[0]["mu"], g["mu0"]) < 1e-8 and relerr(ref[3]["mu"], g["mu3"]) < 1e-8
