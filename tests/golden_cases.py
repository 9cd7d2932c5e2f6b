"""Inputs and run arguments of the non-default branches pinned by tests/golden/branches.npz.

Shared by the generator (tests/golden/gen_golden.py, which runs the REAL reference on them in the development
container) and by the tests (which run the oracle / the HIP path on the same inputs and compare with the stored
reference outputs).  Data only: seeded synthetic trials from vlgp_amd.synth and keyword arguments of `fit`.
"""
import numpy as np


def small_problem(seed=3, n_trials=6, n_bins=150, N=14, L=3, n_gauss=0):
    """Seeded trials with injected a0, b0, mu0 (so that no FactorAnalysis result enters the comparison)."""
    from vlgp_amd import synth

    trials = synth.make_trials(n_trials, n_bins, N, L, seed=seed, n_gauss=n_gauss)
    rng = np.random.default_rng(seed + 100)
    a0 = 0.3 * rng.standard_normal((L, N))
    ycat = np.concatenate([t["y"] for t in trials])
    b0 = np.zeros((1, N))
    npois = N - n_gauss
    if npois:
        b0[0, :npois] = np.log(np.maximum(ycat[:, :npois].mean(0), 1e-8))
    mu0 = [0.2 * rng.standard_normal((n_bins, L)) for _ in trials]
    lik = ["poisson"] * npois + ["gaussian"] * n_gauss

    def fresh():
        return [{"ID": i, "y": t["y"].copy(), "mu": m.copy()} for i, (t, m) in enumerate(zip(trials, mu0))]

    return fresh, a0, b0, lik, (n_trials, n_bins, N, L)


# name -> (small_problem keywords, history, fit keywords).  Two EM iterations each (three for the window / Gaussian
# cases), H-step on; omega_bound keeps the 100-bin full-length factors below the rank budget where stated.
_CONSTR = dict(problem=dict(n_bins=100), history=0)
_RUN = dict(max_iter=2, min_iter=2, omega_bound=(1e-3, 1e-2))
CASES = {
    "loading_svd": dict(_CONSTR, run=dict(_RUN, constrain_loading="svd")),
    "loading_1": dict(_CONSTR, run=dict(_RUN, constrain_loading=1)),
    "loading_2": dict(_CONSTR, run=dict(_RUN, constrain_loading=2)),
    "loading_inf": dict(_CONSTR, run=dict(_RUN, constrain_loading=np.inf)),
    "loading_off": dict(_CONSTR, run=dict(_RUN, constrain_loading=False)),
    "latent_location": dict(_CONSTR, run=dict(_RUN, constrain_latent="location")),
    "latent_scale": dict(_CONSTR, run=dict(_RUN, constrain_latent="scale")),
    "latent_both": dict(_CONSTR, run=dict(_RUN, constrain_latent="both")),
    "svd_and_both": dict(_CONSTR, run=dict(_RUN, constrain_loading="svd", constrain_latent="both")),
    "window_25": dict(problem=dict(seed=9, n_bins=200, N=12), history=0, run=dict(max_iter=3, min_iter=3, window=25)),
    "window_40": dict(problem=dict(seed=9, n_bins=200, N=12), history=0, run=dict(max_iter=3, min_iter=3, window=40)),
    "window_100": dict(problem=dict(seed=9, n_bins=200, N=12), history=0, run=dict(max_iter=3, min_iter=3, window=100)),
    "all_gaussian": dict(problem=dict(seed=9, n_bins=200, N=12, n_gauss=12), history=0,
                         run=dict(max_iter=3, min_iter=3)),
    "window_200": dict(problem=dict(seed=9, n_trials=3, n_bins=400, N=8), history=0,
                       run=dict(max_iter=2, min_iter=2, window=200, omega_bound=(5e-4, 5e-3))),
    "history_2": dict(problem=dict(seed=5, n_trials=4, n_bins=100, N=10), history=2,
                      run=dict(max_iter=2, min_iter=2, omega_bound=(1e-3, 2e-2))),
}


def case_inputs(name):
    c = CASES[name]
    fresh, a0, b0, lik, dims = small_problem(**c["problem"])
    if c["history"] > 1:
        b0 = np.vstack([b0] + [np.zeros_like(b0)] * (c["history"] - 1))
    return fresh, a0, b0, lik, dims, c["history"], dict(c["run"])


def singular_mstep_inputs():
    """M-step statistics whose Newton system is exactly singular: the third latent has mu = v = 0 and the jitter
    (config eps) is 0, so the Hessian of every Poisson channel has a zero row -> the reference's Cholesky solve
    raises and core.mstep takes the gradient step learning_rate * grad (vlgp/core.py:191-198)."""
    rng = np.random.default_rng(123)
    T, N, L = 240, 9, 3
    mu = 0.8 * rng.standard_normal((T, L))
    v = rng.uniform(0.0, 0.3, (T, L))
    mu[:, 2] = 0.0
    v[:, 2] = 0.0
    a = 0.3 * rng.standard_normal((L, N))
    b = np.log(0.4) + 0.1 * rng.standard_normal((1, N))
    x = np.ones((T, 1, N))
    y = rng.poisson(np.exp(mu @ a + b)).astype(float)
    return dict(y=y, x=x, mu=mu, v=v, a=a, b=b, lr=1e-3)


def ragged_window_inputs():
    """Trial lengths that are NOT multiples of the window (vlgp/util.py:482-496: the surplus becomes random overlaps of
    neighbouring segments, which in the reference are NumPy views of the same trial rows)."""
    from vlgp_amd import synth

    lengths = [130, 170, 230, 90, 110]
    N, L = 12, 3
    trials = synth.make_trials(len(lengths), max(lengths), N, L, seed=21, lengths=lengths)
    rng = np.random.default_rng(121)
    a0 = 0.3 * rng.standard_normal((L, N))
    ycat = np.concatenate([t["y"] for t in trials])
    b0 = np.log(np.maximum(ycat.mean(0, keepdims=True), 1e-8))
    mu0 = [0.2 * rng.standard_normal((n, L)) for n in lengths]

    def fresh():
        return [{"ID": i, "y": t["y"].copy(), "mu": m.copy()} for i, (t, m) in enumerate(zip(trials, mu0))]

    return fresh, a0, b0, (lengths, N, L), dict(max_iter=2, min_iter=2, omega_bound=(1e-3, 1e-2))


# ---- the C5 combination in small (tests/golden/c5_small.npz: gen_golden.py c5_small runs the real reference on it) ----
C5S = dict(n_trials=24, lengths=[100, 150, 200, 250], N=40, n_gauss=10, L=10)


def c5_small_inputs():
    """Several distinct trial lengths, mixed Poisson / Gaussian channels, ten latents; injected a, b, mu."""
    import numpy as np

    from vlgp_amd import synth

    c = C5S
    lengths = [c["lengths"][i % len(c["lengths"])] for i in range(c["n_trials"])]
    trials = synth.make_trials(c["n_trials"], max(lengths), c["N"], c["L"], seed=5, n_gauss=c["n_gauss"], lengths=lengths)
    rng = np.random.default_rng(51)
    a0 = 0.3 * rng.standard_normal((c["L"], c["N"]))
    b0 = np.mean(np.concatenate([t["y"] for t in trials]), axis=0, keepdims=True)
    b0[:, :c["N"] - c["n_gauss"]] = np.log(np.maximum(b0[:, :c["N"] - c["n_gauss"]], 1e-8))
    mu0 = [0.2 * rng.standard_normal((t["y"].shape[0], c["L"])) for t in trials]
    lik = ["poisson"] * (c["N"] - c["n_gauss"]) + ["gaussian"] * c["n_gauss"]
    return trials, a0, b0, mu0, lik
