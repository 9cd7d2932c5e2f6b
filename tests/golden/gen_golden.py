"""Generate the golden vectors under tests/golden/ from the REAL reference.

Run in the development container only (the reference does not exist on the
GPU box):

    cd /tmp && python /root/repo/tests/golden/gen_golden.py

It imports ``/root/reference`` (read-only), rebinding ``vlgp.core.solve`` so the
removed SciPy keyword ``sym_pos=True`` maps to ``assume_a='pos'`` -- without
that shim the reference raises TypeError at vlgp/core.py:465 under SciPy>=1.11
(SURVEY.md section 0).  The outputs are data only: seeded inputs and the arrays
the reference produced from them.
"""
import copy
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)
warnings.filterwarnings("ignore")

import scipy.linalg as sl  # noqa: E402
import vlgp  # noqa: E402,F401
import vlgp.core as core  # noqa: E402
import vlgp.gp as gp  # noqa: E402
from vlgp import api as ref_api  # noqa: E402
from vlgp.math import ichol_gauss  # noqa: E402
from vlgp.preprocess import get_config, get_params, fill_trials, fill_params  # noqa: E402
from vlgp.simulation import lorenz  # noqa: E402
from vlgp.util import cut_trials  # noqa: E402

from vlgp_amd import synth  # noqa: E402

sys.path.insert(0, os.path.join(REPO, "tests"))
import golden_cases  # noqa: E402


def _solve(a, b, sym_pos=False, **kw):
    if sym_pos:
        kw.setdefault("assume_a", "pos")
    return sl.solve(a, b, **kw)


core.solve = _solve


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("%-28s %7.1f KB" % (name, os.path.getsize(path) / 1024))


# ---------------------------------------------------------------- 0. generator
def check_generator():
    mine = synth.lorenz_path(3000, x0=(0.3, 0.6, 0.9))
    ref = lorenz(3000, dt=5e-3, s=10, r=28, b=2.667, x0=(0.3, 0.6, 0.9))
    assert np.array_equal(mine, ref), np.abs(mine - ref).max()
    print("synth.lorenz_path == reference simulation.lorenz (bitwise)")


# ---------------------------------------------------------------- 1. ichol
def gen_ichol():
    cases = [(50, 5e-2, 50), (50, 5e-3, 50), (50, 5e-4, 50), (50, 1.3e-2, 50),
             (200, 5e-3, 50), (200, 4e-2, 50), (1000, 5e-4, 50), (1000, 5e-3, 50),
             (64, 2e-2, 20), (37, 7e-3, 50)]
    out = {"cases": np.array(cases, dtype=float)}
    for i, (n, om, r) in enumerate(cases):
        G = ichol_gauss(n, om, r)
        if n > 200:  # keep fixtures small: every 8th row + the full Gram diagonal
            out["G%d_rows" % i] = G[::8]
            out["G%d_diag" % i] = np.sum(G * G, axis=1)
            out["G%d_colsum" % i] = G.sum(axis=0)
        else:
            out["G%d" % i] = G
    save("ichol", **out)


# ---------------------------------------------------------------- helpers
def unit_inputs(rng, M, T, N, L, P=1, n_gauss=0, omega=None, rank=50):
    """Seeded stage inputs: M units of T bins."""
    a = 0.4 * rng.standard_normal((L, N))
    b = np.log(0.3) + 0.2 * rng.standard_normal((P, N))
    if P > 1:
        b[1:] *= 0.1
    noise = 0.5 + rng.random(N)
    lik = np.array(["poisson"] * (N - n_gauss) + ["gaussian"] * n_gauss)
    omega = np.array([4e-2, 6e-3, 1.2e-3, 2e-2, 9e-4][:L]) if omega is None else omega
    sigma = np.array([1.0, 0.9, 1.1, 0.8, 1.0][:L])
    G = np.array([ichol_gauss(T, omega[l], rank) * sigma[l] for l in range(L)])
    units = []
    for _ in range(M):
        z = np.stack([np.sin(np.linspace(0, (2 + l) * np.pi, T) + rng.random() * 6)
                      for l in range(L)], axis=1)
        x = np.ones((T, P, N))
        if P > 1:
            x[:, 1:, :] = rng.standard_normal((T, P - 1, N)) * 0.3
        eta = z @ a + np.einsum("tpn,pn->tn", x, b)
        y = rng.poisson(np.exp(np.minimum(eta, 3))).astype(float)
        if n_gauss:
            y[:, N - n_gauss:] = eta[:, N - n_gauss:] + 0.7 * rng.standard_normal((T, n_gauss))
        mu = z + 0.3 * rng.standard_normal((T, L))
        units.append({"y": y, "x": x, "mu": mu})
    params = {"ydim": N, "zdim": L, "xdim": P, "a": a, "b": b, "noise": noise,
              "sigma": sigma, "omega": omega, "rank": rank, "gp_noise": 1e-4, "dt": 1,
              "likelihood": lik, "cholesky": {T: G}}
    return units, params


def ref_init_wv(units, params, config):
    """w, v as api.fit prepares them (api.py:52-54)."""
    for u in units:
        u["w"] = np.zeros_like(u["mu"])
        u["v"] = np.zeros_like(u["mu"])
    core.update_w(units, params, config)
    core.update_v(units, params, config)


def stack(units, key):
    return np.stack([u[key] for u in units])


# ---------------------------------------------------------------- 2/3. E-step
def gen_estep():
    rng = np.random.default_rng(20240901)
    for tag, n_gauss in (("pois", 0), ("mixed", 8)):
        units0, params = unit_inputs(rng, 4, 50, 20, 3, n_gauss=n_gauss)
        config = get_config()
        ref_init_wv(units0, params, config)
        out = {k: params[k] for k in ("a", "b", "noise", "omega", "sigma")}
        out["gauss"] = params["likelihood"] == "gaussian"
        out["G"] = params["cholesky"][50]
        for k in ("y", "x", "mu", "w", "v"):
            out[k + "0"] = stack(units0, k)
        # update_w / update_v stage outputs from mu0 with w=v=0 (fixture 3)
        out["w_stage"] = out["w0"]
        out["v_stage"] = out["v0"]
        for method in ("VB", "MAP"):
            for n_it in (1, 25):
                units = copy.deepcopy(units0)
                p = copy.deepcopy(params)
                cfg = get_config(method=method, Eniter=n_it)
                fill_trials(units)
                core.estep(units, p, cfg)
                for k in ("mu", "w", "v", "dmu"):
                    out["%s_%s_%d" % (k, method, n_it)] = stack(units, k)
        save("estep_" + tag, **out)

    # one long unit in the truncated-rank regime, G injected
    units0, params = unit_inputs(rng, 1, 300, 20, 3, n_gauss=4,
                                 omega=np.array([3e-2, 8e-3, 2e-3]))
    config = get_config()
    ref_init_wv(units0, params, config)
    out = {k: params[k] for k in ("a", "b", "noise")}
    out["gauss"] = params["likelihood"] == "gaussian"
    out["G"] = params["cholesky"][300]
    for k in ("y", "x", "mu", "w", "v"):
        out[k + "0"] = stack(units0, k)
    units = copy.deepcopy(units0)
    fill_trials(units)
    core.estep(units, copy.deepcopy(params), get_config(Eniter=5))
    for k in ("mu", "w", "v", "dmu"):
        out[k + "_VB_5"] = stack(units, k)
    save("estep_long", **out)


# ---------------------------------------------------------------- 4. M-step
def gen_mstep():
    rng = np.random.default_rng(77)
    for tag, P, n_gauss in (("p1", 1, 0), ("p3", 3, 0), ("mixed", 1, 6)):
        units, params = unit_inputs(rng, 4, 50, 20, 3, P=P, n_gauss=n_gauss)
        config = get_config()
        ref_init_wv(units, params, config)
        fill_trials(units)
        out = {k: params[k] for k in ("a", "b", "noise")}
        out["gauss"] = params["likelihood"] == "gaussian"
        for k in ("y", "x", "mu", "v"):
            out[k] = stack(units, k)
        for hess in (True, False):
            for n_it in (1, 25):
                if not hess and n_it == 25:
                    lr = 1e-4  # plain gradient ascent diverges at lr=1
                elif not hess:
                    lr = 1e-3
                else:
                    lr = 1.0
                p = copy.deepcopy(params)
                fill_params(p)
                cfg = get_config(Mniter=n_it, use_hessian=hess, learning_rate=lr)
                core.mstep(copy.deepcopy(units), p, cfg)
                key = "%s_%d" % ("H" if hess else "G", n_it)
                out["lr_" + key] = lr
                for k in ("a", "b", "da", "db", "noise"):
                    out["%s_%s" % (k, key)] = p[k]
        save("mstep_" + tag, **out)


# ---------------------------------------------------------------- 5. H-step
def gen_hstep():
    rng = np.random.default_rng(5)
    units, params = unit_inputs(rng, 8, 50, 20, 3)
    config = get_config()
    ref_init_wv(units, params, config)
    fill_trials(units)
    core.estep(units, params, get_config(Eniter=3))
    mu = stack(units, "mu")
    w = stack(units, "w")
    t = np.arange(50) * 1.0
    mask = np.array([0, 1, 0])
    pts = np.log(np.array([
        [1.0, 5e-2, 1e-4], [1.0, 5e-4, 1e-4], [0.81, 6e-3, 1e-4],
        [1.0, 2.3e-3, 2e-4], [0.5, 1.7e-2, 5e-5]]))
    lls, dlls = [], []
    for l in range(3):
        for p in pts:
            ex = np.exp(p)
            S = gp.construct_posterior_cov(t, w[:, :, l].T, ex)
            ll, dll = gp.elbo(ex, mask, t, mu[:, :, l].T, S)
            lls.append(ll)
            dlls.append(dll)
    p2 = copy.deepcopy(params)
    for u in units:
        u["y"] = u["y"]
    gp.optimize(units, p2, get_config())
    save("hstep", mu=mu, w=w, logp=pts, ll=np.array(lls).reshape(3, -1),
         dll=np.array(dlls).reshape(3, len(pts), 3), sigma0=params["sigma"],
         omega0=params["omega"], omega_opt=p2["omega"], sigma_opt=p2["sigma"],
         G_opt=p2["cholesky"][50])


# ---------------------------------------------------------------- 6/7. vem, fit
def c1_inputs():
    n_trials, n_bins, N, L = synth.CONFIGS["C1"]
    trials = synth.make_trials(n_trials, n_bins, N, L, seed=0)
    rng = np.random.default_rng(11)
    a0 = 0.3 * rng.standard_normal((L, N))
    b0 = np.log(np.maximum(np.mean(np.concatenate([t["y"] for t in trials]), axis=0,
                                   keepdims=True), 1e-8))
    mu0 = [0.2 * rng.standard_normal((n_bins, L)) for _ in trials]
    return trials, a0, b0, mu0


def gen_vem():
    trials0, a0, b0, mu0 = c1_inputs()
    base = {"y": np.stack([t["y"] for t in trials0]).astype(np.uint8), "a0": a0, "b0": b0,
            "mu0": np.stack(mu0)}
    assert np.array_equal(base["y"].astype(float), np.stack([t["y"] for t in trials0]))
    out = dict(base)
    for hs in (True, False):
        trials = [{"ID": t["ID"], "y": t["y"].copy(), "mu": m.copy()}
                  for t, m in zip(trials0, mu0)]
        cfg = get_config(Hstep=hs, max_iter=6, min_iter=6)
        params = get_params(trials, 3, a=a0.copy(), b=b0.copy(), omega_bound=cfg["omega_bound"])
        for tr in trials:  # what initialize leaves behind when a, b, mu are given
            T = tr["y"].shape[0]
            tr["x"] = np.ones((T, 1, 20))
            tr["w"] = np.zeros((T, 3))
            tr["v"] = np.zeros((T, 3))
        fill_params(params)
        fill_trials(trials)
        gp.make_cholesky(trials, params, cfg)
        core.update_w(trials, params, cfg)
        core.update_v(trials, params, cfg)
        segs = cut_trials(trials, params, cfg)
        gp.make_cholesky(segs, params, cfg)
        fill_trials(segs)
        traj = {"mu": [], "a": [], "b": [], "omega": []}

        def spy(tr_, p_, c_):
            traj["mu"].append(sl.norm(np.concatenate([s["mu"] for s in tr_])))
            traj["a"].append(sl.norm(p_["a"]))
            traj["b"].append(sl.norm(p_["b"]))
            traj["omega"].append(np.array(p_["omega"]))

        cfg["callbacks"] = [spy]
        core.vem(segs, params, cfg)
        tag = "H1" if hs else "H0"
        out["norm_mu_" + tag] = np.array(traj["mu"])
        out["norm_a_" + tag] = np.array(traj["a"])
        out["norm_b_" + tag] = np.array(traj["b"])
        out["omega_" + tag] = np.array(traj["omega"])
        out["a_" + tag] = params["a"]
        out["b_" + tag] = params["b"]
        out["noise_" + tag] = params["noise"]
        out["it_" + tag] = cfg["runtime"]["it"]
        out["seg_mu_" + tag] = np.stack([s["mu"] for s in segs])
        out["seg_v_" + tag] = np.stack([s["v"] for s in segs])
        out["seg_w_" + tag] = np.stack([s["w"] for s in segs])
    save("vem_c1", **out)


def gen_fit():
    trials0, a0, b0, mu0 = c1_inputs()
    trials = [{"ID": t["ID"], "y": t["y"].copy(), "mu": m.copy()}
              for t, m in zip(trials0, mu0)]
    np.random.seed(3)
    res = ref_api.fit(trials, 3, a=a0.copy(), b=b0.copy(), Hstep=False, max_iter=5, min_iter=5)
    p = res["params"]
    save("fit_c1", a0=a0, b0=b0, mu0=np.stack(mu0),
         y=np.stack([t["y"] for t in trials0]).astype(np.uint8),
         a=p["a"], b=p["b"], noise=p["noise"], omega=p["omega"], sigma=p["sigma"],
         G200=p["cholesky"][200], it=res["config"]["runtime"]["it"],
         mu=np.stack([t["mu"] for t in res["trials"]]),
         v=np.stack([t["v"] for t in res["trials"]]),
         w=np.stack([t["w"] for t in res["trials"]]),
         dmu=np.stack([t["dmu"] for t in res["trials"]]))


def gen_fit_h1():
    """fit end to end with the H-step ON (the default) and with every default left alone."""
    trials0, a0, b0, mu0 = c1_inputs()
    trials = [{"ID": t["ID"], "y": t["y"].copy(), "mu": m.copy()} for t, m in zip(trials0, mu0)]
    np.random.seed(3)
    res = ref_api.fit(trials, 3, a=a0.copy(), b=b0.copy(), max_iter=5, min_iter=5)
    p = res["params"]
    out = dict(a0=a0, b0=b0, mu0=np.stack(mu0), y=np.stack([t["y"] for t in trials0]).astype(np.uint8),
               a=p["a"], b=p["b"], noise=p["noise"], omega=p["omega"], sigma=p["sigma"],
               G200=p["cholesky"][200], it=res["config"]["runtime"]["it"])
    for k in ("mu", "v", "w", "dmu"):
        out[k] = np.stack([t[k] for t in res["trials"]])
    # nothing injected: FactorAnalysis initialisation on the seeded subsample, default iteration counts
    trials = [{"ID": t["ID"], "y": t["y"].copy()} for t in trials0]
    np.random.seed(5)
    res = ref_api.fit(trials, 3, max_iter=8)
    p = res["params"]
    out.update(d_a=p["a"], d_b=p["b"], d_noise=p["noise"], d_omega=p["omega"], d_sigma=p["sigma"],
               d_G200=p["cholesky"][200], d_it=res["config"]["runtime"]["it"])
    for k in ("mu", "v", "w"):
        out["d_" + k] = np.stack([t[k] for t in res["trials"]])
    save("fit_c1_h1", **out)


def gen_vem_c2():
    """BASELINE.json configs[1] at full size: 50 trials x 500 bins x 50 channels, 3 latents -> 500 segments;
    three EM iterations of the real reference with every default (H-step on), from injected a, b, mu."""
    n_trials, n_bins, N, L = synth.CONFIGS["C2"]
    trials0 = synth.make_trials(n_trials, n_bins, N, L, seed=0)
    rng = np.random.default_rng(21)
    a0 = 0.3 * rng.standard_normal((L, N))
    b0 = np.log(np.maximum(np.mean(np.concatenate([t["y"] for t in trials0]), axis=0, keepdims=True), 1e-8))
    mu0 = [0.2 * rng.standard_normal((n_bins, L)) for _ in trials0]
    trials = [{"ID": t["ID"], "y": t["y"].copy(), "mu": m.copy()} for t, m in zip(trials0, mu0)]
    cfg = get_config(max_iter=3, min_iter=3)
    params = get_params(trials, L, a=a0.copy(), b=b0.copy(), omega_bound=cfg["omega_bound"])
    for tr in trials:
        tr["x"] = np.ones((n_bins, 1, N))
        tr["w"] = np.zeros((n_bins, L))
        tr["v"] = np.zeros((n_bins, L))
    fill_params(params)
    fill_trials(trials)
    gp.make_cholesky(trials, params, cfg)
    core.update_w(trials, params, cfg)
    core.update_v(trials, params, cfg)
    segs = cut_trials(trials, params, cfg)
    gp.make_cholesky(segs, params, cfg)
    fill_trials(segs)
    traj = {"mu": [], "a": [], "b": [], "omega": []}

    def spy(tr_, p_, c_):
        traj["mu"].append(sl.norm(np.concatenate([s["mu"] for s in tr_])))
        traj["a"].append(sl.norm(p_["a"]))
        traj["b"].append(sl.norm(p_["b"]))
        traj["omega"].append(np.array(p_["omega"]))

    cfg["callbacks"] = [spy]
    core.vem(segs, params, cfg)
    pick = np.arange(0, len(segs), 20)
    save("vem_c2", norm_mu=np.array(traj["mu"]), norm_a=np.array(traj["a"]), norm_b=np.array(traj["b"]),
         omega=np.array(traj["omega"]), a=params["a"], b=params["b"], noise=params["noise"],
         it=cfg["runtime"]["it"], pick=pick, seg_mu=np.stack([segs[i]["mu"] for i in pick]),
         seg_v=np.stack([segs[i]["v"] for i in pick]), seg_w=np.stack([segs[i]["w"] for i in pick]),
         y_checksum=np.array([float(np.concatenate([t["y"] for t in trials0]).sum())]))


c5_small_inputs = golden_cases.c5_small_inputs
C5S = golden_cases.C5S


def gen_c5_small():
    """BASELINE.json configs[4]'s COMBINATION against the real reference (VERDICT round 5, item 7 i): 24 trials of four
    distinct lengths (100 ... 250 bins) x 40 channels (30 Poisson + 10 Gaussian), ten latents; two EM iterations with every
    default (H-step on: vlgp/gp.py:65-97 on the 50-bin segments of ALL lengths at once, make_cholesky per distinct
    length, vlgp/gp.py:150-162; mixed likelihood in E- and M-step, vlgp/core.py:36-37,82-83,222-235)."""
    trials0, a0, b0, mu0, lik = c5_small_inputs()
    c = C5S
    trials = [{"ID": t["ID"], "y": t["y"].copy(), "mu": m.copy()} for t, m in zip(trials0, mu0)]
    cfg = get_config(max_iter=2, min_iter=2)
    params = get_params(trials, c["L"], a=a0.copy(), b=b0.copy(), lik=lik, omega_bound=cfg["omega_bound"])
    for tr in trials:
        T = tr["y"].shape[0]
        tr["x"] = np.ones((T, 1, c["N"]))
        tr["w"] = np.zeros((T, c["L"]))
        tr["v"] = np.zeros((T, c["L"]))
    fill_params(params)
    fill_trials(trials)
    gp.make_cholesky(trials, params, cfg)
    assert sorted(params["cholesky"]) == sorted(set(c["lengths"]))
    core.update_w(trials, params, cfg)
    core.update_v(trials, params, cfg)
    segs = cut_trials(trials, params, cfg)
    gp.make_cholesky(segs, params, cfg)
    fill_trials(segs)
    traj = {"mu": [], "a": [], "b": [], "omega": [], "noise": []}

    def spy(tr_, p_, c_):
        traj["mu"].append(sl.norm(np.concatenate([s["mu"] for s in tr_])))
        traj["a"].append(sl.norm(p_["a"]))
        traj["b"].append(sl.norm(p_["b"]))
        traj["omega"].append(np.array(p_["omega"]))
        traj["noise"].append(np.array(p_["noise"]))

    cfg["callbacks"] = [spy]
    core.vem(segs, params, cfg)
    save("c5_small", norm_mu=np.array(traj["mu"]), norm_a=np.array(traj["a"]), norm_b=np.array(traj["b"]),
         omega=np.array(traj["omega"]), noise_traj=np.array(traj["noise"]), a=params["a"], b=params["b"],
         noise=params["noise"], it=cfg["runtime"]["it"], pick=np.arange(0, len(segs), 3),
         seg_mu=np.stack([s_["mu"] for s_ in segs[::3]]),
         seg_v=np.stack([s_["v"] for s_ in segs[::3]]), seg_w=np.stack([s_["w"] for s_ in segs[::3]]),
         y_checksum=np.array([float(np.concatenate([t["y"] for t in trials0]).sum())]))


def gen_vem_c3():
    """BASELINE.json configs[2] (the headline) at full size: 200 trials x 1000 bins x 100 channels, 5 latents ->
    4000 segments; two EM iterations of the real reference with every default (H-step on), from injected a, b, mu
    (vlgp/core.py:269-359).  About a quarter of an hour of one core in the development container."""
    n_trials, n_bins, N, L = synth.CONFIGS["C3"]
    trials0 = synth.make_trials(n_trials, n_bins, N, L, seed=0)
    rng = np.random.default_rng(31)
    a0 = 0.3 * rng.standard_normal((L, N))
    b0 = np.log(np.maximum(np.mean(np.concatenate([t["y"] for t in trials0]), axis=0, keepdims=True), 1e-8))
    mu0 = [0.2 * rng.standard_normal((n_bins, L)) for _ in trials0]
    trials = [{"ID": t["ID"], "y": t["y"].copy(), "mu": m.copy()} for t, m in zip(trials0, mu0)]
    cfg = get_config(max_iter=2, min_iter=2)
    params = get_params(trials, L, a=a0.copy(), b=b0.copy(), omega_bound=cfg["omega_bound"])
    for tr in trials:
        tr["x"] = np.ones((n_bins, 1, N))
        tr["w"] = np.zeros((n_bins, L))
        tr["v"] = np.zeros((n_bins, L))
    fill_params(params)
    fill_trials(trials)
    gp.make_cholesky(trials, params, cfg)
    core.update_w(trials, params, cfg)
    core.update_v(trials, params, cfg)
    segs = cut_trials(trials, params, cfg)
    gp.make_cholesky(segs, params, cfg)
    fill_trials(segs)
    pick = np.arange(0, len(segs), 20)
    traj = {"mu": [], "a": [], "b": [], "omega": [], "seg_mu": []}

    def spy(tr_, p_, c_):
        traj["mu"].append(sl.norm(np.concatenate([s["mu"] for s in tr_])))
        traj["a"].append(sl.norm(p_["a"]))
        traj["b"].append(sl.norm(p_["b"]))
        traj["omega"].append(np.array(p_["omega"]))
        traj["seg_mu"].append(np.stack([tr_[i]["mu"] for i in pick[::4]]))
        print("  iteration done", len(traj["mu"]), flush=True)

    cfg["callbacks"] = [spy]
    core.vem(segs, params, cfg)
    save("vem_c3", norm_mu=np.array(traj["mu"]), norm_a=np.array(traj["a"]), norm_b=np.array(traj["b"]),
         omega=np.array(traj["omega"]), a=params["a"], b=params["b"], noise=params["noise"],
         it=cfg["runtime"]["it"], pick=pick, seg_mu_it1=traj["seg_mu"][0],
         seg_mu=np.stack([segs[i]["mu"] for i in pick]),
         seg_v=np.stack([segs[i]["v"] for i in pick]), seg_w=np.stack([segs[i]["w"] for i in pick]),
         y_checksum=np.array([float(np.concatenate([t["y"] for t in trials0]).sum())]))


def _describe(d):
    """key -> [type name, dtype or None, shape or None] of a returned dict."""
    out = {}
    for k, v in d.items():
        if isinstance(v, np.ndarray):
            out[k] = ["ndarray", str(v.dtype), list(v.shape)]
        else:
            out[k] = [type(v).__name__, None, None]
    return out


def gen_result():
    """What `fit` hands back and what `util.save` writes (vlgp/api.py:18-76, vlgp/util.py:181-190): the key sets /
    value types of trials[0], params, config and config["runtime"] as JSON, and the reference's own result files
    (.npy: one pickled object; .npz: the three top-level keys) for `vlgp_amd.load` to read."""
    import json

    from vlgp import util as ref_util

    trials0 = synth.make_trials(4, 100, 8, 2, seed=3)
    trials = [{"ID": t["ID"], "y": t["y"].copy()} for t in trials0]
    np.random.seed(9)
    res = ref_api.fit(trials, 2, max_iter=2, min_iter=2)
    keys = {"trial": _describe(res["trials"][0]), "params": _describe(res["params"]),
            "config": _describe(res["config"]), "runtime": _describe(res["config"]["runtime"]),
            "initial": sorted(res["params"]["initial"].keys()),
            "cholesky_keys": sorted(int(k) for k in res["params"]["cholesky"].keys()),
            "top": sorted(res.keys()), "trials_type": type(res["trials"]).__name__}
    with open(os.path.join(HERE, "fit_keys.json"), "w") as f:
        json.dump(keys, f, indent=1, sort_keys=True)
    # the bound scikit-learn method in params["transform"] (preprocess.py:21) would drag a pickled estimator into
    # the file; the stored result keeps everything else exactly as util.save writes it
    res["params"]["transform"] = None
    res["params"]["initial"]["transform"] = None
    ref_util.save(res, os.path.join(HERE, "ref_result"), ext="npy")
    ref_util.save(res, os.path.join(HERE, "ref_result"), ext="npz")
    for ext in ("npy", "npz"):
        print("ref_result.%s %7.1f KB" % (ext, os.path.getsize(os.path.join(HERE, "ref_result." + ext)) / 1024))


def gen_init():
    """preprocess.initialize on C1 (FactorAnalysis on the seeded 10 % subsample)."""
    from vlgp.preprocess import initialize

    n_trials, n_bins, N, L = synth.CONFIGS["C1"]
    trials = synth.make_trials(n_trials, n_bins, N, L, seed=0)
    cfg = get_config()
    params = get_params(trials, L, omega_bound=cfg["omega_bound"])
    np.random.seed(7)
    initialize(trials, params, cfg)
    save("init_c1", a=params["a"], b=params["b"], noise=params["noise"],
         mu=np.stack([t["mu"] for t in trials]), x_shape=np.array(trials[0]["x"].shape))


def gen_branches():
    """The branches the default configuration never takes (vlgp/core.py:366-416 constrain_loading / constrain_latent,
    window != 50, all-Gaussian channels, history = 2): `fit` of the real reference on the seeded problems of
    tests/golden_cases.py with a, b, mu injected; and api.transform (vlgp/api.py:171-184) on new trials."""
    out = {}
    for name in golden_cases.CASES:
        fresh, a0, b0, lik, dims, history, run = golden_cases.case_inputs(name)
        trials = fresh()
        np.random.seed(1)  # initialize() draws the FactorAnalysis subsample (its result is overridden by a, b, mu)
        kw = dict(a=a0.copy(), b=b0.copy(), lik=lik, **run)
        if history:
            kw["history"] = history
        res = ref_api.fit(trials, dims[3], **kw)
        p = res["params"]
        for k in ("a", "b", "noise", "omega", "sigma"):
            out["%s__%s" % (name, k)] = np.asarray(p[k])
        out[name + "__it"] = res["config"]["runtime"]["it"]
        out[name + "__G_rows"] = p["cholesky"][dims[1]][:, ::10]  # every tenth bin of the full-length factors
        for k in ("mu", "v"):  # first and last trial (the fixture stays small)
            out["%s__%s" % (name, k)] = np.stack([res["trials"][i][k] for i in (0, -1)])
        print("  branch %-16s it %d omega %s" % (name, out[name + "__it"], np.array2string(p["omega"], precision=4)))
    # transform: fit (FactorAnalysis initialisation, seeded) then infer new trials of a length the fit knows
    fresh, a0, b0, lik, dims = golden_cases.small_problem(seed=7, n_trials=5, n_bins=100, N=12)
    np.random.seed(2)
    res = ref_api.fit([{"ID": t["ID"], "y": t["y"]} for t in fresh()], dims[3], max_iter=3, min_iter=3)
    p, cfg = res["params"], res["config"]
    new = synth.make_trials(3, 100, 12, dims[3], seed=8)
    np.random.seed(3)
    got = ref_api.transform([{"ID": t["ID"], "y": t["y"].copy()} for t in new], p, cfg)
    for k in ("a", "b", "noise", "omega", "sigma"):
        out["transform__" + k] = np.asarray(p[k])
    out["transform__G"] = p["cholesky"][100]
    out["transform__max_iter"] = cfg["max_iter"]
    out["transform__mu0"] = np.stack([p["transform"](t["y"]) for t in new])
    for k in ("mu", "v", "w"):
        out["transform__" + k] = np.stack([t[k] for t in got])
    save("branches", **out)


def gen_mstep_singular():
    """core.mstep's fallback when the Newton system of a channel does not factor (vlgp/core.py:191-198)."""
    d = golden_cases.singular_mstep_inputs()
    T, N = d["y"].shape
    L = d["mu"].shape[1]
    out = {}
    for n_it in (1, 3):
        units = [{"y": d["y"].copy(), "x": d["x"].copy(), "mu": d["mu"].copy(), "v": d["v"].copy()}]
        params = {"ydim": N, "zdim": L, "xdim": 1, "a": d["a"].copy(), "b": d["b"].copy(), "noise": np.ones(N),
                  "likelihood": np.array(["poisson"] * N), "rank": 50, "gp_noise": 1e-4, "dt": 1,
                  "sigma": np.ones(L), "omega": np.full(L, 5e-3)}
        fill_params(params)
        cfg = get_config(Mniter=n_it, eps=0.0, learning_rate=d["lr"])
        core.mstep(units, params, cfg)
        for k in ("a", "b", "da", "db", "noise"):
            out["%s_%d" % (k, n_it)] = params[k]
    # the solve really failed: the delta is learning_rate * grad for every channel at the first iteration
    mu, v, y, a, b = d["mu"], d["v"], d["y"], d["a"], d["b"]
    r = np.exp(mu @ a + b + 0.5 * v @ a ** 2)
    grad = np.stack([mu.T @ y[:, n] - (mu + v * a[:, n]).T @ r[:, n] for n in range(N)], axis=1)
    assert np.allclose(out["da_1"], np.clip(d["lr"] * grad, -5, 5), rtol=1e-12, atol=0), "the Newton solve did not fail"
    save("mstep_singular", **out)


def gen_ragged():
    """Trial lengths that are not multiples of the window: the reference's segments are overlapping VIEWS of the trial
    arrays (vlgp/util.py:482-496), updated one after the other in place."""
    fresh, a0, b0, (lengths, N, L), run = golden_cases.ragged_window_inputs()
    np.random.seed(4)
    res = ref_api.fit(fresh(), L, a=a0.copy(), b=b0.copy(), **run)
    p = res["params"]
    save("ragged_window", a=p["a"], b=p["b"], noise=p["noise"], omega=p["omega"], sigma=p["sigma"],
         it=res["config"]["runtime"]["it"], mu0=res["trials"][0]["mu"], v0=res["trials"][0]["v"],
         mu3=res["trials"][3]["mu"])


ALL = ["ichol", "estep", "mstep", "mstep_singular", "ragged", "hstep", "vem", "fit", "init", "fit_h1", "branches", "result",
       "c5_small", "vem_c2", "vem_c3"]

if __name__ == "__main__":
    os.chdir("/tmp")  # the reference writes vlgp.log into the cwd at import
    check_generator()
    todo = sys.argv[1:] or ALL  # a bare run regenerates EVERY fixture (vem_c3 last: a quarter of an hour)
    for name in todo:  # `gen_golden.py fit_h1` regenerates one fixture
        globals()["gen_" + name]()
