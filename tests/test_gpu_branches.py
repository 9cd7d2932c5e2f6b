"""The branches of the hot path that the default configuration never takes, each against the oracle:
constrain_loading in {svd, 1, 2, inf}, constrain_latent in {location, scale, both} (vlgp/core.py:366-416),
the omega += log 10 retry of a K that does not factor (vlgp/gp.py:128-135), the exp clamp at 10
(vlgp/math.py:24-38), history = 2 (two regressors) through fit, and api.transform (vlgp/api.py:171-184)."""
import numpy as np
import pytest

from conftest import relerr
from oracle import vlgp_oracle as O

pytestmark = pytest.mark.gpu

STAGE = 1e-9
TRAJ = 1e-6


@pytest.fixture(scope="module")
def V():
    import vlgp_amd

    return vlgp_amd


def _small_problem(seed=3, n_trials=6, n_bins=150, N=14, L=3, n_gauss=0):
    from vlgp_amd import synth

    trials = synth.make_trials(n_trials, n_bins, N, L, seed=seed, n_gauss=n_gauss)
    rng = np.random.default_rng(seed + 100)
    a0 = 0.3 * rng.standard_normal((L, N))
    ycat = np.concatenate([t["y"] for t in trials])
    b0 = np.zeros((1, N))
    npois = N - n_gauss
    b0[0, :npois] = np.log(np.maximum(ycat[:, :npois].mean(0), 1e-8))
    mu0 = [0.2 * rng.standard_normal((n_bins, L)) for _ in trials]
    lik = ["poisson"] * npois + ["gaussian"] * n_gauss
    fresh = lambda: [{"ID": i, "y": t["y"].copy(), "mu": m.copy()} for i, (t, m) in enumerate(zip(trials, mu0))]
    return fresh, a0, b0, lik, (n_trials, n_bins, N, L)


def _oracle_fit(fresh, a0, b0, lik, dims, xdim=1, **cfg_kw):
    n_trials, n_bins, N, L = dims
    ref = fresh()
    for t in ref:
        t.setdefault("x", np.ones((n_bins, xdim, N)))
        t["w"] = np.zeros((n_bins, L))
        t["v"] = np.zeros((n_bins, L))
    cfg = O.make_config(**cfg_kw)
    params = O.make_params(ref, L, a=a0.copy(), b=b0.copy(), lik=lik, history=xdim if xdim > 1 else 0,
                           omega_bound=cfg["omega_bound"])
    O.fit_given_init(ref, params, cfg)
    return ref, params, cfg


@pytest.mark.parametrize("kw", [
    dict(constrain_loading="svd"), dict(constrain_loading=1), dict(constrain_loading=2),
    dict(constrain_loading=np.inf), dict(constrain_latent="location"), dict(constrain_latent="scale"),
    dict(constrain_latent="both"), dict(constrain_loading="svd", constrain_latent="both"),
    dict(constrain_loading=False),
], ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()))
def test_constraint_modes_two_em_iterations_vs_oracle(V, kw):
    """core.constrain_loading / core.constrain_latent in every mode, two EM iterations with the H-step on
    (100-bin trials and omega <= 1e-2: the full-length factors converge before the rank budget, so the
    end-to-end posterior is comparable whenever the two runs pick the same pivots)."""
    fresh, a0, b0, lik, dims = _small_problem(n_bins=100)
    run = dict(max_iter=2, min_iter=2, omega_bound=(1e-3, 1e-2), **kw)
    got = V.fit(fresh(), dims[3], a=a0.copy(), b=b0.copy(), lik=lik, verbose=False, **run)
    ref, params, _ = _oracle_fit(fresh, a0, b0, lik, dims, **run)
    for k in ("a", "b", "noise", "omega"):  # 1e-5: where the L-BFGS-B line searches stop (as in the other fit-vs-oracle tests)
        assert relerr(got["params"][k], params[k]) < 1e-5, k
    # same pivots in both runs (omega differs at ~1e-10) -> the factors agree to that level and so does the posterior;
    # a flipped pivot moves G G' by the 1e-6 truncation remainder of a converged factor
    same = relerr(got["params"]["cholesky"][dims[1]], params["cholesky"][dims[1]]) < 1e-7
    for tg, tr in zip(got["trials"], ref):
        assert relerr(tg["mu"], tr["mu"]) < (1e-5 if same else 1e-4)
        assert relerr(tg["v"], tr["v"]) < (1e-5 if same else 1e-4)


@pytest.mark.parametrize("T", [50, 100])
def test_hstep_objective_retry_when_K_does_not_factor(V, T):
    """gp.construct_posterior_cov (vlgp/gp.py:128-135): a kernel matrix that fails its Cholesky gets log 10 ADDED
    to omega (the reference's quirk: it adds to the exponentiated parameter) until it factors; gp.elbo then
    sees the modified omega too.  Provoked with a jitter far below rounding at a tiny omega.  T = 50: the round kernel
    hands a failed K to the generic kernels; T = 100: the retry loop of hstep_prep_big."""
    rng = np.random.default_rng(4)
    M, L = 37, 2
    units = [{"y": np.zeros((T, 2)), "mu": rng.standard_normal((T, L)), "w": rng.uniform(0.05, 3.0, (T, L)),
              "v": np.zeros((T, L))} for _ in range(M)]
    pts = np.log(np.array([[1.0, 1e-9, 1e-22],      # K = ones + 1e-22 I: not positive definite in fp64 -> retry
                           [0.5, 3e-10, 1e-25],
                           [1.0, 5e-3, 1e-4]]))     # an ordinary point in the same call
    t = np.arange(T) * 1.0
    with V.Engine(2, L, 1, 50) as eng:
        eng.upload(0, units)
        for l in range(L):
            lat = np.full(len(pts), l, dtype=np.int32)
            ll, dll = eng.hstep_objective(0, T, 1.0, lat, pts)
            for i, lp in enumerate(pts):
                want_ll, want_dll = O.gp_objective(lp, t, np.stack([u["mu"][:, l] for u in units], 1),
                                                   np.stack([u["w"][:, l] for u in units], 1))
                assert np.isfinite(want_ll)
                assert abs(ll[i] - want_ll) <= STAGE * abs(want_ll), (l, i, ll[i], want_ll)
                assert abs(dll[i, 1] - want_dll[1]) <= STAGE * max(abs(want_dll[1]), 1e-3 * abs(want_ll)), (l, i)
    # the first two points really took the retry: at the stated omega K has no Cholesky factor
    from scipy.linalg import LinAlgError, cholesky
    for lp in pts[:2]:
        s2, om, eps = np.exp(lp)
        with pytest.raises(LinAlgError):
            cholesky(O.se_kernel(t, s2, om, eps)[0], lower=True)


@pytest.mark.parametrize("split", [False, True])
def test_exp_clamp_at_ten_in_e_and_m_step(V, split, monkeypatch):
    """math.trunc_exp = exp(min(x, 10)) (vlgp/math.py:24-38): loadings large enough that eta + v a^2 / 2 passes
    10 on a good part of the (bin, channel) pairs -- E-step (25 sweeps) and M-step (3 Newton iterations)."""
    rng = np.random.default_rng(9)
    T, N, L, M = 50, 16, 3, 5
    a = 2.5 * rng.standard_normal((L, N))
    b = rng.uniform(0.5, 2.0, (1, N))
    noise = np.ones(N)
    gauss = np.zeros(N, bool)
    G = O.build_prior([T], np.array([1e-2, 5e-3, 2e-3]), np.ones(L), 50)[T]
    units = []
    n_clamped = 0
    for _ in range(M):
        mu = 1.5 * rng.standard_normal((T, L))
        v = rng.uniform(0.0, 0.5, (T, L))
        y = rng.poisson(3.0, (T, N)).astype(float)
        x = np.ones((T, 1, N))
        n_clamped += int(((mu @ a + b + 0.5 * v @ a ** 2) > 10).sum())
        w = O.curvature_unit(y, x, mu, v, a, b, noise, gauss)
        units.append({"y": y, "x": x, "mu": mu, "v": v, "w": w, "dmu": np.zeros((T, L))})
    assert n_clamped > 0.1 * M * T * N  # the clamp branch is taken, often
    params = {"ydim": N, "zdim": L, "xdim": 1, "rank": 50, "a": a.copy(), "b": b.copy(), "noise": noise.copy(),
              "likelihood": np.array(["poisson"] * N), "cholesky": {T: G}, "gp_noise": 1e-4, "dt": 1}
    want = [O.estep_unit(u["y"], u["x"], u["mu"], u["v"], u["w"], a, b, noise, gauss, G, 25) for u in units]
    mine = [{k: np.array(val) for k, val in u.items()} for u in units]
    if split:  # the table-driven exp of the split E-step's passes has its own clamp
        monkeypatch.setenv("VLGP_ESTEP_SPLIT", "1")
    V.estep(mine, params, V.get_config())
    from vlgp_amd import engine as E
    assert E.TRACE["estep"] in (("split", "split_mixed") if split else ("fast",))
    for u, wv in zip(mine, want):
        for k, arr in zip(("mu", "v", "w"), wv):
            # curvatures up to e^10 a^2: the reference's v = rowsum(G o (G - G H + G H M)) cancels at cond(I + H) ~ 1e5
            # (DESIGN.md section 2); the oracle inherits that error, hence 1e-8 here instead of 1e-9
            assert relerr(u[k], arr) < 1e-8, k
    cat = lambda k: np.concatenate([u[k] for u in units], axis=0)
    wm = O.mstep_arrays(cat("y"), cat("x"), cat("mu"), cat("v"), a.copy(), b.copy(), gauss, 3)
    p2 = dict(params, a=a.copy(), b=b.copy())
    V.mstep([{k: np.array(val) for k, val in u.items()} for u in units], p2, V.get_config(Mniter=3))
    assert relerr(p2["a"], wm[0]) < STAGE and relerr(p2["b"], wm[1]) < STAGE


def test_fit_with_history_two_regressors(V):
    """history = 2 -> xdim = 2 (vlgp/preprocess.py:53,43-44: x defaults to ones of shape (T, 2, N)): the
    general-x kernels (x.b product, regressor Hessian of the M-step) through two EM iterations."""
    fresh, a0, b0, lik, dims = _small_problem(seed=5, n_trials=4, n_bins=100, N=10)
    b2 = np.vstack([b0, np.zeros_like(b0)])
    run = dict(max_iter=2, min_iter=2, omega_bound=(1e-3, 2e-2))
    got = V.fit(fresh(), dims[3], a=a0.copy(), b=b2.copy(), lik=lik, history=2, verbose=False, **run)
    assert got["params"]["xdim"] == 2 and got["params"]["b"].shape == (2, dims[2])
    assert got["trials"][0]["x"].shape == (dims[1], 2, dims[2])
    ref, params, _ = _oracle_fit(fresh, a0, b2, lik, dims, xdim=2, **run)
    for k in ("a", "b", "noise", "omega"):
        assert relerr(got["params"][k], params[k]) < TRAJ, k


def test_fit_twenty_latents_vs_oracle(V):
    """zdim = 20: the reference loops `for l in range(zdim)` with no bound (vlgp/core.py:76,106; gp.py:82).  Beyond
    sixteen latents the loop-based fallbacks run (generic E-step with spilled per-latent arrays, mstep_*_gen, the
    H-step's evaluations in slices of sixteen): two EM iterations with the H-step on against the oracle."""
    fresh, a0, b0, lik, dims = _small_problem(seed=11, n_trials=4, n_bins=100, N=30, L=20)
    run = dict(max_iter=2, min_iter=2, omega_bound=(1e-3, 1e-2))
    got = V.fit(fresh(), 20, a=a0.copy(), b=b0.copy(), lik=lik, verbose=False, **run)
    ref, params, _ = _oracle_fit(fresh, a0, b0, lik, dims, **run)
    for k in ("a", "b", "noise", "omega", "sigma"):
        assert relerr(got["params"][k], params[k]) < TRAJ, k
    for tg, tr in zip(got["trials"], ref):
        assert relerr(tg["mu"], tr["mu"]) < 1e-5
        assert relerr(tg["w"], tr["w"]) < 1e-5


def test_fit_with_eleven_regressors_and_gaussian_channels(V):
    """history = 11 -> xdim = 11 > 8 (vlgp/preprocess.py:53), the trials bringing their own x (a constant column and
    ten random regressors; preprocess.py:43-44 keeps a trial's x): the M-step's loop-based fallback with a general x --
    Poisson Newton systems of 11 unknowns for b and the Gaussian channels' least squares, solved in global memory."""
    fresh0, a0, b0, lik, dims = _small_problem(seed=12, n_trials=4, n_bins=100, N=10, n_gauss=3)
    n_trials, n_bins, N, L = dims
    xs = []
    rng = np.random.default_rng(99)
    for _ in range(n_trials):
        x = 0.1 * rng.standard_normal((n_bins, 11, N))
        x[:, 0, :] = 1.0
        xs.append(x)

    def fresh():
        out = fresh0()
        for t, x in zip(out, xs):
            t["x"] = x.copy()
        return out

    b11 = np.vstack([b0] + [np.zeros_like(b0)] * 10)
    run = dict(max_iter=2, min_iter=2, omega_bound=(1e-3, 2e-2))
    got = V.fit(fresh(), L, a=a0.copy(), b=b11.copy(), lik=lik, history=11, verbose=False, **run)
    assert got["params"]["xdim"] == 11 and got["params"]["b"].shape == (11, N)
    ref, params, _ = _oracle_fit(fresh, a0, b11, lik, dims, xdim=11, **run)
    for k in ("a", "b", "noise", "omega"):
        assert relerr(got["params"][k], params[k]) < TRAJ, k


def test_transform_value_parity(V):
    """api.transform (vlgp/api.py:171-184): new trials get mu = params["transform"](y), w = v = 0 and one
    core.infer (E-step with Eniter := max_iter) under the fitted parameters and prior factors."""
    fresh, a0, b0, lik, dims = _small_problem(seed=7, n_trials=5, n_bins=100, N=12)
    n_trials, n_bins, N, L = dims
    np.random.seed(2)
    fit = V.fit([{"ID": t["ID"], "y": t["y"]} for t in fresh()], L, max_iter=3, min_iter=3, verbose=False)
    params, config = fit["params"], fit["config"]
    from vlgp_amd import synth

    new = synth.make_trials(3, n_bins, N, L, seed=8)
    new.append({"ID": 3, "y": new[0]["y"][:60].copy()})  # a length the fit has no factor for (the reference raises KeyError)
    np.random.seed(3)
    got = V.transform([{"ID": t["ID"], "y": t["y"].copy()} for t in new], params, config)
    gauss = np.zeros(N, bool)
    for tg, t in zip(got, new):
        T = t["y"].shape[0]
        mu0 = params["transform"](t["y"])
        G = params["cholesky"][T]
        assert np.array_equal(G, O.build_prior([T], params["omega"], params["sigma"], 50)[T])
        want = O.estep_unit(t["y"], np.ones((T, 1, N)), mu0, np.zeros((T, L)), np.zeros((T, L)), params["a"],
                            params["b"], params["noise"], gauss, G, config["max_iter"])
        for k, arr in zip(("mu", "v", "w"), want):
            assert relerr(tg[k], arr) < STAGE, (k, T)


def test_sample_posterior_on_the_device(V):
    """api.sample_posterior (vlgp/api.py:142-168) through vlgp_sample_posterior: bit-for-bit the same normal draws
    as the oracle's low-rank restatement -> same samples to rounding; mean and covariance of many draws equal
    the reference's inv(inv(K + reg I) + W) to sampling error."""
    rng = np.random.default_rng(0)
    T, L = 40, 3
    omega, sigma = np.array([4e-3, 2e-2, 9e-3]), np.array([1.0, 0.7, 1.2])
    chol = O.build_prior([T], omega, sigma, 50)
    trial = {"mu": rng.standard_normal((T, L)), "w": rng.random((T, L)) * 3.0}
    n = 64
    got = V.sample_posterior(trial, {"cholesky": chol}, n, rng=np.random.default_rng(1))
    assert got.shape == (n, T, L)
    g2 = np.random.default_rng(1)
    eps = []
    for l in range(L):
        r = int(np.flatnonzero(np.any(chol[T][l] != 0, axis=0))[-1]) + 1
        eps.append(g2.standard_normal((r, n)))
    want = O.sample_posterior_lowrank(trial["mu"], trial["w"], chol[T], eps)
    assert relerr(got, want) < 1e-10
    n = 100000
    draws = V.sample_posterior(trial, {"cholesky": chol}, n, rng=np.random.default_rng(2))
    for l in range(L):
        cov = O.posterior_covariance_reference(chol[T][l], trial["w"][:, l])
        assert np.abs(np.cov(draws[:, :, l].T) - cov).max() < 0.03 * np.abs(cov).max() + 2e-3
        assert np.abs(draws[:, :, l].mean(0) - trial["mu"][:, l]).max() < 0.02


def test_sample_posterior_more_latents_than_a_handle_and_legacy_rng(V):
    """More than 16 latents go through the device in groups (the latents are independent), and a generator without
    standard_normal (numpy.random.RandomState-like: normal(loc, scale, size)) draws (r, n) arrays, not shape-as-loc."""
    rng = np.random.default_rng(5)
    T, L, n = 30, 19, 16
    omega = 10 ** rng.uniform(-3, -1.7, L)
    chol = O.build_prior([T], omega, np.ones(L), 50)
    trial = {"mu": rng.standard_normal((T, L)), "w": rng.random((T, L)) * 2.0}

    class Legacy:  # only normal(loc, scale, size)
        def __init__(self, seed):
            self.g = np.random.default_rng(seed)

        def normal(self, loc=0.0, scale=1.0, size=None):
            return loc + scale * self.g.standard_normal(size)

    got = V.sample_posterior(trial, {"cholesky": chol}, n, rng=Legacy(3))
    g2 = np.random.default_rng(3)
    eps = []
    for l in range(L):
        r = int(np.flatnonzero(np.any(chol[T][l] != 0, axis=0))[-1]) + 1
        eps.append(g2.standard_normal((r, n)))
    want = O.sample_posterior_lowrank(trial["mu"], trial["w"], chol[T], eps)
    assert got.shape == (n, T, L) and relerr(got, want) < 1e-10


def test_command_line_fit_and_result_file(V, tmp_path):
    """python -m vlgp_amd FIN FOUT N_FACTORS --max_iter --min_iter (vlgp/__main__.py:6-22): the saved result is
    the dict fit returns, loadable with util.load, and equals the in-process fit on the same input."""
    import subprocess
    import sys

    from conftest import ROOT
    from vlgp_amd import synth, util

    trials = synth.make_trials(4, 100, 10, 2, seed=3)
    fin, fout = tmp_path / "trials.npy", tmp_path / "result"
    np.save(fin, np.array(trials, dtype=object), allow_pickle=True)
    code = "import numpy as np, sys; np.random.seed(1); from vlgp_amd.__main__ import cli; sys.exit(cli(sys.argv[1:]))"
    done = subprocess.run([sys.executable, "-c", code, str(fin), str(fout), "2", "--max_iter", "3", "--min_iter", "3"],
                          cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert done.returncode == 0, done.stderr[-2000:]
    assert "Loading" in done.stdout and "saved" in done.stdout
    res = util.load(str(fout) + ".npy")
    assert set(res) == {"trials", "params", "config"} and res["config"]["runtime"]["it"] == 3
    np.random.seed(1)
    want = V.fit([{"ID": t["ID"], "y": t["y"].copy()} for t in trials], 2, max_iter=3, min_iter=3, verbose=False)
    assert np.array_equal(res["params"]["a"], want["params"]["a"])
    assert np.array_equal(np.stack([t["mu"] for t in res["trials"]]), np.stack([t["mu"] for t in want["trials"]]))


# ---- the same branches against the REAL reference (tests/golden/branches.npz, generated by gen_golden.py) ----
import os as _os
import sys as _sys

_sys.path.insert(0, _os.path.dirname(_os.path.abspath(__file__)))  # golden_cases
import golden_cases  # noqa: E402


@pytest.mark.parametrize("name", sorted(golden_cases.CASES))
def test_branches_against_reference_golden(V, golden, name):
    """`fit` on the HIP path against `fit` of the real reference for every constrain_loading / constrain_latent mode
    (vlgp/core.py:366-416), windows 25 / 40 / 100 (the generic, the padded and the long-window H-step kernels),
    all-Gaussian channels and history = 2.  1e-6 as SURVEY 8(c) states for multi-iteration runs; window 25: 1e-5
    (one latent's L-BFGS-B line search ends on rounding noise: the oracle itself is 4e-6 off the reference there,
    tests/test_oracle_golden.py)."""
    g = golden("branches")
    fresh, a0, b0, lik, dims, history, run = golden_cases.case_inputs(name)
    kw = dict(a=a0.copy(), b=b0.copy(), lik=lik, verbose=False, **run)
    if history:
        kw["history"] = history
    got = V.fit(fresh(), dims[3], **kw)
    assert got["config"]["runtime"]["it"] == int(g[name + "__it"])
    tol = 1e-5 if name == "window_25" else TRAJ
    for k in ("a", "b", "noise", "omega", "sigma"):
        # omega: SciPy's L-BFGS-B stops on a relative decrease of 2.2e-9 of the objective, which pins the minimiser to a
        # few 1e-6 at best: a last-bit change of (ll, dll) can end a line search one evaluation earlier or later
        # (measured 2.6e-6 on one latent of "svd_and_both" through the dense round, 3e-10 through the low-rank round)
        assert relerr(got["params"][k], g["%s__%s" % (name, k)]) < (max(tol, 1e-5) if k == "omega" else tol), k
    G = got["params"]["cholesky"][dims[1]]
    if np.array_equal(G[:, ::10], g[name + "__G_rows"]):  # same pivots in the full-length factors
        for k in ("mu", "v"):
            assert relerr(np.stack([got["trials"][i][k] for i in (0, -1)]), g["%s__%s" % (name, k)]) < 1e-5, k


def test_transform_against_reference_golden(V, golden):
    """api.transform (vlgp/api.py:171-184) with the reference's own fitted parameters and prior factors: the
    posterior of three new trials against what the reference's transform returned."""
    from vlgp_amd import synth

    g = golden("branches")
    N, L, T = 12, 3, 100
    new = synth.make_trials(3, T, N, L, seed=8)
    params = {"ydim": N, "zdim": L, "xdim": 1, "rank": 50, "gp_noise": 1e-4, "dt": 1,
              "a": g["transform__a"].copy(), "b": g["transform__b"].copy(), "noise": g["transform__noise"].copy(),
              "omega": g["transform__omega"].copy(), "sigma": g["transform__sigma"].copy(),
              "likelihood": np.array(["poisson"] * N), "cholesky": {T: g["transform__G"].copy()},
              "transform": lambda y: (_ for _ in ()).throw(AssertionError("mu is given"))}
    config = V.get_config(max_iter=int(g["transform__max_iter"]))
    trials = [{"ID": t["ID"], "y": t["y"].copy(), "mu": g["transform__mu0"][i].copy()} for i, t in enumerate(new)]
    got = V.transform(trials, params, config)
    for k in ("mu", "v", "w"):
        assert relerr(np.stack([t[k] for t in got]), g["transform__" + k]) < STAGE, k


@pytest.mark.parametrize("n_it", [1, 3])
def test_mstep_gradient_step_when_the_newton_system_is_singular(V, golden, n_it):
    """core.mstep's fallback (vlgp/core.py:191-198): a channel whose Hessian does not factor takes the step
    learning_rate * grad.  Third latent mu = v = 0, jitter 0 -> a zero pivot in every Poisson channel."""
    d = golden_cases.singular_mstep_inputs()
    g = golden("mstep_singular")
    T, N = d["y"].shape
    L = d["mu"].shape[1]
    units = [{"y": d["y"].copy(), "x": d["x"].copy(), "mu": d["mu"].copy(), "v": d["v"].copy(),
              "w": np.zeros((T, L)), "dmu": np.zeros((T, L))}]
    params = {"ydim": N, "zdim": L, "xdim": 1, "rank": 50, "a": d["a"].copy(), "b": d["b"].copy(), "noise": np.ones(N),
              "likelihood": np.array(["poisson"] * N), "gp_noise": 1e-4, "dt": 1}
    V.mstep(units, params, V.get_config(Mniter=n_it, eps=0.0, learning_rate=d["lr"]))
    for k in ("a", "b", "da", "db", "noise"):
        assert relerr(params[k], g["%s_%d" % (k, n_it)]) < STAGE, k


def test_ragged_window_against_reference_golden(V, golden):
    """Trial lengths that are not multiples of the window.  The reference's overlapping segments are VIEWS of the same
    trial rows (vlgp/util.py:482-496): core.estep visits them one after the other, a shared row is advanced by both
    segments in turn, and the default constrain_loading scales it once per segment that holds it (core.py:413-416).
    The device keeps independent copies, stored stage-major, runs the E-step stage by stage with the shared rows handed
    over in between and applies in-place constraints to shared rows twice (vlgp_set_overlaps): five trials of
    90 ... 230 bins at window 50 against the real reference, same multinomial draw of the overlaps."""
    g = golden("ragged_window")
    fresh, a0, b0, (lengths, N, L), run = golden_cases.ragged_window_inputs()
    np.random.seed(4)
    got = V.fit(fresh(), L, a=a0.copy(), b=b0.copy(), verbose=False, **run)
    assert got["config"]["runtime"]["it"] == int(g["it"])
    for k in ("a", "b", "noise", "omega", "sigma"):
        assert relerr(got["params"][k], g[k]) < (1e-5 if k == "omega" else TRAJ), k
    assert relerr(got["trials"][0]["mu"], g["mu0"]) < 1e-5 and relerr(got["trials"][0]["v"], g["v0"]) < 1e-5
    assert relerr(got["trials"][3]["mu"], g["mu3"]) < 1e-5


@pytest.mark.parametrize("kw", [dict(constrain_loading="svd"), dict(constrain_loading=2),
                                dict(constrain_latent="both"), dict(constrain_loading="svd", constrain_latent="scale"),
                                dict(window=40)],
                         ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()))
def test_ragged_window_other_modes_vs_oracle(V, kw):
    """The overlapping-segment semantics under the other constraints: "svd" REBINDS every segment's mu (the segments
    stop sharing mu from the first iteration on, v stays shared), the in-place ones touch shared rows twice; and another
    window (other overlaps).  Against the oracle, whose views ARE the reference's (test_ragged_window_golden pins it)."""
    fresh, a0, b0, (lengths, N, L), run = golden_cases.ragged_window_inputs()
    run = dict(run, **kw)
    np.random.seed(4)
    got = V.fit(fresh(), L, a=a0.copy(), b=b0.copy(), verbose=False, **run)
    ref = fresh()
    for t in ref:
        n = t["y"].shape[0]
        t["x"] = np.ones((n, 1, N))
        t["w"] = np.zeros((n, L))
        t["v"] = np.zeros((n, L))
    cfg = O.make_config(**run)
    params = O.make_params(ref, L, a=a0.copy(), b=b0.copy(), omega_bound=cfg["omega_bound"])
    np.random.seed(4)
    np.random.choice(sum(lengths), max(sum(lengths) // 10, 50))  # initialize() draws its subsample first
    O.fit_given_init(ref, params, cfg)
    # 1e-5 on the parameters: omega is where L-BFGS-B's own stopping rule (ftol 2.2e-9 on an objective that is flat to
    # second order at the optimum) leaves it, and a, b, noise follow it through the next E-step; 1e-4 on the final
    # full-length posterior: its prior factor is ichol_gauss of that omega, whose pivot order on rank-exhausted trials is
    # discontinuous in omega.  The reference-captured trajectories hold 1e-6 (test_branches_against_reference_golden).
    for k in ("a", "b", "noise", "omega"):
        assert relerr(got["params"][k], params[k]) < 1e-5, k
    for tg, tr in zip(got["trials"], ref):
        assert relerr(tg["mu"], tr["mu"]) < 1e-4 and relerr(tg["v"], tr["v"]) < 1e-4
