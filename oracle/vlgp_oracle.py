"""NumPy/SciPy restatement of the reference's variational-EM hot path.

TEST INFRASTRUCTURE ONLY -- this module is the *checker* (and the timed CPU
baseline of ``bench.py``); the product package ``vlgp_amd`` never imports it.

Parity status: PINNED.  Every function here is checked against golden vectors
captured from the real reference (``/root/reference``, imported with the
``sym_pos -> assume_a='pos'`` shim) by ``tests/golden/gen_golden.py``; see
``tests/test_oracle_golden.py``.  The reference's own test-suite holds exactly
one numeric assertion on this path (``tests/test_math.py:7-14``,
``K ~= G G^T``), which is reproduced in ``tests/test_oracle_golden.py`` too.

Third-party arithmetic the reference delegates to (not vendored, unpinned in
``requirements.txt:1-4``): LAPACK ``posv/potrf/potrs`` via ``scipy.linalg`` and
L-BFGS-B via ``scipy.optimize.minimize``.  The oracle calls the same SciPy
entry points.

Layout conventions (all float64, C order):
    y   (T, N)        observations of one unit (trial or 50-bin segment)
    x   (T, P, N)     regressors (P = xdim)
    mu  (T, L)        posterior mean,  v (T, L) marginal variance,
    w   (T, L)        likelihood curvature, dmu (T, L) last mean step
    a   (L, N)        loading,  b (P, N) bias/regression,  noise (N,)
    G   (L, T, R)     low-rank prior factor, K_l ~= G_l G_l^T
    gauss (N,) bool   True where the channel likelihood is Gaussian
"""
import logging
import math
import time

import numpy as np
from scipy import linalg as sla
from scipy.optimize import minimize

log = logging.getLogger("vlgp_oracle")

EXP_CAP = 10.0


# --------------------------------------------------------------------------
# elementary pieces
# --------------------------------------------------------------------------
def capped_exp(z, cap=EXP_CAP):
    """exp(min(z, cap)).  Reference: vlgp/math.py:24-38 (trunc_exp)."""
    return np.exp(np.minimum(z, cap))


def ichol_gauss(n, omega, rank, dt=1.0, tol=1e-6, return_pivots=False):
    """Pivoted incomplete Cholesky of K_ij = exp(-omega (i-j)^2 dt^2).

    Reference: vlgp/math.py:76-126.  Stops after ``rank`` columns or as soon
    as the residual diagonal mass drops to ``tol * n``; untouched columns stay
    zero; rows are returned in natural (un-pivoted) order.

    The operation order below is the reference's (BLAS ``dot`` for the
    Schur-complement term, ``np.sum`` of squares for the residual diagonal
    recomputed from scratch at each step) because the arg-max pivot choice is
    sensitive to the last bit whenever the rank budget is exhausted.
    """
    pos = np.arange(n) * dt
    resid = np.ones(n)
    order = np.arange(n)
    fac = np.zeros((n, rank))
    k = 0
    while k < rank and resid[k:].sum() > tol * n:
        p = k + int(np.argmax(resid[k:])) if k > 0 else 0
        if p != k:
            order[[k, p]] = order[[p, k]]
            fac[[k, p], : k + 1] = fac[[p, k], : k + 1]
        piv = math.sqrt(resid[p])
        fac[k, k] = piv
        tail = order[k + 1:]
        col = np.exp(-omega * (pos[tail] - pos[order[k]]) ** 2)
        fac[k + 1:, k] = (col - np.dot(fac[k + 1:, :k], fac[k, :k])) / piv
        resid[k + 1:] = 1 - np.sum(np.square(fac[k + 1:, : k + 1]), axis=1)
        k += 1
    out = fac[np.argsort(order), :]
    if return_pivots:
        return out, order[:k].copy()
    return out


def build_prior(lengths, omega, sigma, rank, dt=1.0):
    """{T: (L, T, rank)} for every distinct unit length.

    Reference: vlgp/gp.py:150-162 (make_cholesky); the reference ignores
    ``params['dt']`` here (ichol_gauss default dt=1) and so does this.
    """
    out = {}
    for T in np.unique(np.asarray(lengths)):
        T = int(T)
        out[T] = np.stack(
            [ichol_gauss(T, omega[l], rank) * sigma[l] for l in range(len(omega))]
        )
    return out


def linear_predictor(x, mu, a, b):
    """eta = mu a + sum_p x[:, p, :] * b[p, :].   (vlgp/core.py:66,69)"""
    return mu @ a + np.einsum("tpn,pn->tn", x, b)


def _spd_solve(A, B):
    return sla.solve(A, B, assume_a="pos")


# --------------------------------------------------------------------------
# E-step  (vlgp/core.py:22-120)
# --------------------------------------------------------------------------
def estep_unit(y, x, mu, v, w, a, b, noise, gauss, G, n_iter, dmu_bound=5.0, vb=True):
    """Run ``n_iter`` inner iterations of the posterior update for one unit.

    Returns new (mu, v, w, dmu, n_failed); inputs are not modified.
    Reference: vlgp/core.py:22-120 (infer_single_trial).
    """
    y = np.asarray(y, dtype=float)
    mu = np.array(mu, dtype=float)
    v = np.array(v, dtype=float)
    w = np.array(w, dtype=float)
    dmu = np.zeros_like(mu)
    if n_iter < 1:
        return mu, v, w, dmu, 0
    L = mu.shape[1]
    R = G.shape[-1]
    eye = np.eye(R)
    pois = ~gauss
    asq = a ** 2
    xb = np.einsum("tpn,pn->tn", x, b)
    gnoise = noise[gauss]
    n_failed = 0

    for _ in range(n_iter):
        eta = mu @ a + xb
        rate = capped_exp(eta + 0.5 * (v @ asq))
        # working residual: identical for every latent inside this sweep
        res = np.empty_like(y)
        res[:, pois] = y[:, pois] - rate[:, pois]
        res[:, gauss] = (y[:, gauss] - eta[:, gauss]) / gnoise
        for l in range(L):
            Gl = G[l]
            WG = w[:, l:l + 1] * Gl
            H = Gl.T @ WG
            u = Gl @ (Gl.T @ (res @ a[l])) - mu[:, l]
            try:
                rhs = WG.T @ u
                sol = _spd_solve(eye + H, rhs)
                step = u - Gl @ rhs + Gl @ (H @ sol)
                np.clip(step, -dmu_bound, dmu_bound, out=step)
            except Exception as exc:  # reference: core.py:92-94
                log.warning("mean update failed: %r", exc)
                step = np.zeros(mu.shape[0])
                n_failed += 1
            dmu[:, l] = step
            mu[:, l] += step

        eta = mu @ a + xb
        rate = capped_exp(eta + 0.5 * (v @ asq))
        curv = np.empty_like(y)
        curv[:, pois] = rate[:, pois]
        curv[:, gauss] = 1.0 / gnoise
        w = curv @ asq.T
        if vb:
            for l in range(L):
                Gl = G[l]
                H = Gl.T @ (w[:, l:l + 1] * Gl)
                try:
                    sol = _spd_solve(eye + H, H)
                    v[:, l] = np.sum(Gl * (Gl - Gl @ H + Gl @ (H @ sol)), axis=1)
                except Exception as exc:  # reference: core.py:112-113
                    log.warning("variance update failed: %r", exc)
                    n_failed += 1
    return mu, v, w, dmu, n_failed


def curvature_unit(y, x, mu, v, a, b, noise, gauss):
    """w = U (a^T)^2 with U = rate (Poisson) | 1/noise (Gaussian).  core.py:419-442."""
    asq = a ** 2
    eta = linear_predictor(x, mu, a, b)
    rate = capped_exp(eta + 0.5 * (v @ asq))
    curv = np.empty_like(rate)
    curv[:, ~gauss] = rate[:, ~gauss]
    curv[:, gauss] = 1.0 / noise[gauss]
    return curv @ asq.T


def variance_unit(w, v, G):
    """v_l = diag(G_l (I + G_l^T W_l G_l)^-1 G_l^T).  core.py:445-471.

    Returns (v_new, n_failed); a failed latent keeps its old column.
    """
    v = np.array(v, dtype=float)
    L, _, R = G.shape
    eye = np.eye(R)
    bad = 0
    for l in range(L):
        Gl = G[l]
        H = Gl.T @ (w[:, l:l + 1] * Gl)
        try:
            sol = _spd_solve(eye + H, H)
            v[:, l] = np.sum(Gl * (Gl - Gl @ H + Gl @ (H @ sol)), axis=1)
        except sla.LinAlgError:
            bad += 1
    return v, bad


# --------------------------------------------------------------------------
# M-step  (vlgp/core.py:129-249)
# --------------------------------------------------------------------------
def mstep_arrays(y, x, mu, v, a, b, gauss, n_iter, use_hessian=True, eps=1e-8,
                 learning_rate=1.0, da_bound=5.0, db_bound=5.0, noise=None):
    """Per-channel Newton / least-squares update of loading and bias.

    y (Ttot,N), x (Ttot,P,N), mu,v (Ttot,L): all units concatenated in time.
    Returns (a, b, da, db, noise).  Inputs are not modified.
    Reference: vlgp/core.py:129-249.  ``noise`` is returned untouched if
    ``n_iter < 1`` (the reference returns before computing it).
    """
    a = np.array(a, dtype=float)
    b = np.array(b, dtype=float)
    L, N = a.shape
    P = b.shape[0]
    da = np.zeros_like(a)
    db = np.zeros_like(b)
    if n_iter < 1:
        return a, b, da, db, noise
    sum_v = v.sum(axis=0)
    for _ in range(n_iter):
        eta = mu @ a + np.einsum("tpn,pn->tn", x, b)
        rate = capped_exp(eta + 0.5 * (v @ a ** 2))
        noise = np.var(y - eta, axis=0)
        for n in range(N):
            xn = x[:, :, n]
            if not gauss[n]:
                shifted = mu + v * a[:, n]
                g_a = mu.T @ y[:, n] - shifted.T @ rate[:, n]
                step = learning_rate * g_a
                if use_hessian:
                    h_a = shifted.T @ (rate[:, n:n + 1] * shifted)
                    h_a[np.diag_indices(L)] += rate[:, n] @ v
                    try:
                        step = _spd_solve(h_a + eps * np.eye(L), g_a)
                    except Exception as exc:  # core.py:194-196
                        log.warning("loading Newton step failed: %r", exc)
                np.clip(step, -da_bound, da_bound, out=step)
                da[:, n] = step
                a[:, n] += step

                g_b = xn.T @ (y[:, n] - rate[:, n])
                step = learning_rate * g_b
                if use_hessian:
                    h_b = xn.T @ (rate[:, n:n + 1] * xn)
                    try:
                        step = _spd_solve(h_b + eps * np.eye(P), g_b)
                    except Exception as exc:  # core.py:212-214
                        log.warning("bias Newton step failed: %r", exc)
                np.clip(step, -db_bound, db_bound, out=step)
                db[:, n] = step
                b[:, n] += step
            else:
                # closed-form alternating least squares (core.py:224-235);
                # not guarded by try/except in the reference either
                gram = mu.T @ mu
                gram[np.diag_indices(L)] += sum_v
                a[:, n] = _spd_solve(gram, mu.T @ (y[:, n] - xn @ b[:, n]))
                b[:, n] = _spd_solve(xn.T @ xn, xn.T @ (y[:, n] - mu @ a[:, n]))
                b[1:, n] = 0
    return a, b, da, db, noise


# --------------------------------------------------------------------------
# H-step  (vlgp/gp.py:12-147)
# --------------------------------------------------------------------------
def se_kernel(t, sigmasq, omega, eps):
    """K = sigmasq exp(-omega D^2) + eps I and dK/dln(omega).  gp.py:46-62.

    Only the ln-omega derivative is returned: the reference masks the other
    two gradient components to zero (gp.py:16,85).
    """
    d2 = (t[:, None] - t[None, :]) ** 2
    K = sigmasq * np.exp(-omega * d2)
    dK = -K * d2 * omega
    K = K + eps * np.eye(t.size)
    return K, dK


def gp_objective(logp, t, mu, w):
    """Per-latent GP term (ll, dll[3]) summed over segments.

    mu, w: (T, M) -- one column per segment.  Reference: gp.py:100-123
    (obj_func) = construct_posterior_cov (gp.py:126-147) + elbo (gp.py:12-43)
    with mask = [0, 1, 0].  Returns the *un-negated* (ll, dll).

    Quirk kept on purpose: when K fails to factor, the reference *adds*
    log(10) to omega itself (gp.py:135 operates on the exponentiated vector,
    in place), and ``elbo`` then sees that modified omega as well.
    """
    sigmasq, omega, eps = np.exp(np.asarray(logp, dtype=float))
    T, M = mu.shape
    while True:
        K, dK = se_kernel(t, sigmasq, omega, eps)
        try:
            Lk = sla.cholesky(K, lower=True)
            break
        except sla.LinAlgError:
            omega = omega + math.log(10)
    K_inv = sla.cho_solve((Lk, True), np.eye(T))
    alpha = sla.cho_solve((Lk, True), mu)
    ll_seg = -0.5 * np.einsum("tm,tm->m", mu, alpha)
    outer = np.einsum("im,jm->ijm", alpha, alpha)
    outer -= K_inv[:, :, None]
    for i in range(M):
        Li = sla.cholesky(K_inv + np.diag(w[:, i]), lower=True)
        S = sla.cho_solve((Li, True), np.eye(T))
        KiS = sla.cho_solve((Lk, True), S)
        ll_seg[i] -= 0.5 * np.trace(KiS)
        outer[:, :, i] += KiS @ K_inv
    ll_seg -= np.log(np.diag(Lk)).sum()
    dll = np.zeros(3)
    dll[1] = 0.5 * np.einsum("ijm,ij->m", outer, dK).sum()
    return ll_seg.sum(), dll


def hstep_arrays(mu, w, sigma, omega, gp_noise, omega_bound, window, dt=1.0):
    """L-BFGS-B on log(sigma^2, omega, eps) per latent, omega-only gradient.

    mu, w: (M, T, L).  Returns (sigma_new, omega_new).  Reference: gp.py:65-123.
    """
    sigma = np.array(sigma, dtype=float)
    omega = np.array(omega, dtype=float)
    t = np.arange(window) * dt
    for l in range(omega.size):
        start = np.log(np.array([sigma[l] ** 2, omega[l], gp_noise]))
        bounds = np.log(np.array([(1e-3, 1.0), tuple(omega_bound),
                                  (gp_noise / 2, gp_noise * 2)]))
        m_l = np.ascontiguousarray(mu[:, :, l].T)
        w_l = np.ascontiguousarray(w[:, :, l].T)

        def neg(logp):
            ll, dll = gp_objective(logp, t, m_l, w_l)
            return -ll, -dll

        res = minimize(neg, start, jac=True, bounds=bounds)
        sig2, om, _ = np.exp(res.x)
        if not np.any(np.isclose(om, omega_bound)):
            omega[l] = om
        sigma[l] = math.sqrt(sig2)
    return sigma, omega


# --------------------------------------------------------------------------
# dict-level driver mirroring the reference's (trials, params, config) seam
# --------------------------------------------------------------------------
DEFAULT_CONFIG = {
    "constrain_loading": "fro", "constrain_latent": False, "use_hessian": True,
    "eps": 1e-8, "tol": 1e-8, "min_iter": 5, "method": "VB", "learning_rate": 1.0,
    "max_iter": 20, "Eniter": 25, "Mniter": 25, "Hstep": True, "da_bound": 5.0,
    "db_bound": 5.0, "dmu_bound": 5.0, "omega_bound": (5e-4, 5e-2), "window": 50,
    "saving_interval": 60 * 30, "callbacks": [], "parallel": False,
}


def make_config(**kw):
    """preprocess.py:84-112: known keys only, unknown keys dropped."""
    cfg = {k: (list(v) if isinstance(v, list) else v) for k, v in DEFAULT_CONFIG.items()}
    cfg.update({k: v for k, v in kw.items() if k in cfg})
    return cfg


def make_params(trials, n_factors, **kw):
    """preprocess.py:49-81."""
    N = trials[0]["y"].shape[-1]
    lik = kw.get("lik", "poisson")
    if not isinstance(lik, list):
        lik = [lik] * N
    ob = kw.get("omega_bound", DEFAULT_CONFIG["omega_bound"])
    return {
        "ydim": N, "zdim": n_factors, "xdim": max(kw.get("history", 0), 1),
        "a": kw.get("a"), "b": kw.get("b"),
        "noise": kw.get("noise", np.ones(N)),
        "sigma": kw.get("sigma", np.ones(n_factors)),
        "omega": kw.get("omega", np.full(n_factors, ob[1])),
        "rank": 50, "gp_noise": 1e-4, "dt": 1, "likelihood": np.asarray(lik),
    }


def _gauss_mask(params):
    return np.asarray(params["likelihood"]) == "gaussian"


def make_cholesky(trials, params, config=None):
    params["cholesky"] = build_prior([tr["y"].shape[0] for tr in trials],
                                     params["omega"], params["sigma"], params["rank"])


def update_w(trials, params, config=None):
    g = _gauss_mask(params)
    for tr in trials:
        mu = tr["mu"]
        tr.setdefault("w", np.zeros_like(mu))
        v = tr.setdefault("v", np.zeros_like(mu))
        tr["w"] = curvature_unit(tr["y"], tr["x"], mu, v, params["a"], params["b"],
                                 params["noise"], g)


def update_v(trials, params, config):
    if config["method"] != "VB":
        return
    for tr in trials:
        mu = tr["mu"]
        w = tr.setdefault("w", np.zeros_like(mu))
        v = tr.setdefault("v", np.zeros_like(mu))
        v[...], _ = variance_unit(w, v, params["cholesky"][mu.shape[0]])


def estep(trials, params, config):
    g = _gauss_mask(params)
    for tr in trials:
        if config["Eniter"] < 1:
            continue
        mu, v, w, dmu, _ = estep_unit(
            tr["y"], tr["x"], tr["mu"], tr["v"], tr["w"], params["a"], params["b"],
            params["noise"], g, params["cholesky"][tr["y"].shape[0]],
            config["Eniter"], config["dmu_bound"], config["method"] == "VB")
        tr["mu"][...] = mu          # in place (segments are views of the parent)
        tr["v"][...] = v
        tr["dmu"][...] = dmu
        tr["w"] = w                 # rebound, as the reference does (core.py:118)


def mstep(trials, params, config):
    if config["Mniter"] < 1:
        return
    cat = lambda k: np.concatenate([tr[k] for tr in trials], axis=0)
    a, b, da, db, noise = mstep_arrays(
        cat("y"), cat("x"), cat("mu"), cat("v"), params["a"], params["b"],
        _gauss_mask(params), config["Mniter"], config["use_hessian"], config["eps"],
        config["learning_rate"], config["da_bound"], config["db_bound"],
        noise=params["noise"])
    params["a"][...] = a
    params["b"][...] = b
    params["da"][...] = da
    params["db"][...] = db
    params["noise"] = noise


def hstep(trials, params, config):
    if not config["Hstep"]:
        return
    mu = np.stack([tr["mu"] for tr in trials])
    w = np.stack([tr["w"] for tr in trials])
    params["sigma"], params["omega"] = hstep_arrays(
        mu, w, params["sigma"], params["omega"], params["gp_noise"],
        config["omega_bound"], config["window"], params["dt"])
    make_cholesky(trials, params, config)


def constrain_loading(trials, params, config):
    """core.py:392-416; 'fro' (default), per-row norms, or 'svd'."""
    kind = config["constrain_loading"]
    if not kind or kind == "none":
        return
    a = params["a"]
    if kind == "svd":
        _, _, vt = sla.svd(a, full_matrices=False)
        us = a @ vt.T
        for tr in trials:
            tr["mu"] = tr["mu"] @ us
        params["a"] = vt
        return
    if kind == "fro":
        s = sla.norm(a, ord="fro") + config["eps"]
        params["a"] /= s
        for tr in trials:
            tr["mu"] *= s
    else:
        s = np.linalg.norm(a, ord=kind, axis=1, keepdims=True) + config["eps"]
        params["a"] /= s
        for tr in trials:
            tr["mu"] *= s.T


def constrain_latent(trials, params, config):
    """core.py:366-389 (off by default)."""
    kind = config["constrain_latent"]
    if not kind or kind == "none":
        return
    mu = np.concatenate([tr["mu"] for tr in trials], axis=0)
    mean = mu.mean(axis=0, keepdims=True)
    std = mu.std(axis=0, keepdims=True)
    if kind in ("location", "both"):
        for tr in trials:
            tr["mu"] -= mean
        params["b"][0, :] += np.squeeze(mean @ params["a"])
    if kind in ("scale", "both"):
        for tr in trials:
            tr["mu"] /= std
        params["a"] *= std.T


def vem(trials, params, config, echo=None):
    """EM loop with the reference's timers and stopping rule.  core.py:269-359."""
    rt = {"it": 0, "e_elapsed": [], "m_elapsed": [], "h_elapsed": [], "em_elapsed": []}
    for it in range(config["max_iter"]):
        rt["it"] += 1
        n_mu = sla.norm(np.concatenate([tr["mu"] for tr in trials], axis=0))
        n_a = sla.norm(params["a"])
        n_b = sla.norm(params["b"])
        t0 = time.perf_counter()
        constrain_loading(trials, params, config)
        estep(trials, params, config)
        t1 = time.perf_counter()
        constrain_latent(trials, params, config)
        mstep(trials, params, config)
        t2 = time.perf_counter()
        hstep(trials, params, config)
        t3 = time.perf_counter()
        rt["e_elapsed"].append(t1 - t0)
        rt["m_elapsed"].append(t2 - t1)
        rt["h_elapsed"].append(t3 - t2)
        rt["em_elapsed"].append(t3 - t0)
        config["runtime"] = rt
        if echo:
            echo("Iteration {:4d}, E-step {:.2f}s, M-step {:.2f}s".format(
                rt["it"], rt["e_elapsed"][-1], rt["m_elapsed"][-1]))
        for cb in config["callbacks"]:
            try:
                cb(trials, params, config)
            except RuntimeError:
                log.error("callback %r failed", cb)
        n_dmu = sla.norm(np.concatenate([tr["dmu"] for tr in trials], axis=0))
        tol = config["tol"]
        done = (n_dmu < tol * n_mu and sla.norm(params["da"]) < tol * n_a
                and sla.norm(params["db"]) < tol * n_b)
        if done and it + 1 >= config["min_iter"]:
            break


def infer(trials, params, config):
    """core.py:260-266: E-step with Eniter := max_iter."""
    keep = config["Eniter"]
    config["Eniter"] = config["max_iter"]
    try:
        estep(trials, params, config)
    finally:
        config["Eniter"] = keep


def cut_trials(trials, window):
    """util.py:457-499.  Segments are *views* of the parent arrays; when a
    length is not a multiple of ``window`` the start offsets are drawn from the
    global NumPy RNG exactly as the reference draws them."""
    if not window:
        return trials
    segs = []
    for tr in trials:
        T = tr["y"].shape[0]
        k = math.ceil(T / window)
        extra = k * window - T
        starts = np.arange(k) * window
        shift = np.cumsum(np.append([0], np.random.multinomial(
            extra, np.ones(k - 1) / (k - 1))))
        starts = starts - shift
        for s in starts:
            sl = slice(int(s), int(s) + window)
            segs.append({key: tr[key][sl] for key in ("y", "x", "mu", "w", "v")})
    return segs


def fill_trials(trials):
    for i, tr in enumerate(trials):
        tr["cut"] = i
        for key in ("w", "v", "dmu"):
            tr.setdefault(key, np.zeros_like(tr["mu"]))


def fit_given_init(trials, params, config):
    """api.py:47-71 from *after* ``initialize``: every trial already holds
    y, x, mu and zero w, v; params holds a, b, noise.  Mutates in place."""
    params.setdefault("da", np.zeros_like(params["a"]))
    params.setdefault("db", np.zeros_like(params["b"]))
    fill_trials(trials)
    make_cholesky(trials, params, config)
    update_w(trials, params, config)
    update_v(trials, params, config)
    segs = cut_trials(trials, config["window"])
    make_cholesky(segs, params, config)
    fill_trials(segs)
    vem(segs, params, config)
    make_cholesky(trials, params, config)
    update_w(trials, params, config)
    update_v(trials, params, config)
    infer(trials, params, config)
    return {"trials": trials, "params": params, "config": config}


# --------------------------------------------------------------------------
# Sufficient-statistics form of the M-step (the multi-GPU protocol)
# --------------------------------------------------------------------------
def mstep_sharded(y, x, mu, v, a, b, gauss, n_iter, allreduce=None, use_hessian=True, eps=1e-8,
                  learning_rate=1.0, da_bound=5.0, db_bound=5.0):
    """M-step on ONE shard of the rows, exchanging only sums with the other shards.

    Mathematically ``mstep_arrays`` on the concatenation of all shards: every
    quantity that couples rows is a sum over rows, so each rank accumulates its
    part and ``allreduce(buf)`` (in-place sum over ranks; identity when None)
    completes it -- one fused buffer per Newton iteration, exactly what
    libvlgp_hip.so sends through RCCL.  Returns (a, b, da, db, noise).
    """
    if allreduce is None:
        allreduce = lambda buf: buf
    a = np.array(a, dtype=float)
    b = np.array(b, dtype=float)
    L, N = a.shape
    P = b.shape[0]
    da, db = np.zeros_like(a), np.zeros_like(b)
    pois = ~gauss
    # sweep-invariant moments: one fused buffer
    pre = np.concatenate([
        (mu.T @ y).ravel(),                          # MtY (L, N)
        np.einsum("tpn,tn->pn", x, y).ravel(),       # XtY (P, N)
        np.einsum("tpn,tl->npl", x, mu).ravel(),     # XtM (N, P, L)
        np.einsum("tpn,tqn->npq", x, x).ravel(),     # XtX (N, P, P)
        (mu.T @ mu).ravel(), v.sum(0), [float(y.shape[0])]])
    pre = allreduce(pre)
    o = 0
    MtY = pre[o:o + L * N].reshape(L, N); o += L * N
    XtY = pre[o:o + P * N].reshape(P, N); o += P * N
    XtM = pre[o:o + N * P * L].reshape(N, P, L); o += N * P * L
    XtX = pre[o:o + N * P * P].reshape(N, P, P); o += N * P * P
    gram = pre[o:o + L * L].reshape(L, L); o += L * L
    sumv = pre[o:o + L]; o += L
    count = pre[o]
    noise = None
    for it in range(n_iter):
        eta = mu @ a + np.einsum("tpn,pn->tn", x, b)
        if it == n_iter - 1:  # noise = var(y - eta) entering the last iteration
            s1 = allreduce((y - eta).sum(0))
            mean = s1 / count
            s2 = allreduce(((y - eta - mean) ** 2).sum(0))
            noise = s2 / count
        rate = capped_exp(eta + 0.5 * (v @ a ** 2))
        shifted = mu[:, :, None] + v[:, :, None] * a[None, :, :]          # (T, L, N)
        stats = np.concatenate([
            np.einsum("tln,tn->ln", shifted, rate).ravel(),                # g1
            np.einsum("tln,tn,tkn->nlk", shifted, rate, shifted).ravel(),  # H
            (v.T @ rate).ravel(),                                          # rv (L, N)
            np.einsum("tpn,tn->pn", x, rate).ravel(),                      # gb
            np.einsum("tpn,tn,tqn->npq", x, rate, x).ravel()])             # Hb
        stats = allreduce(stats)
        o = 0
        g1 = stats[o:o + L * N].reshape(L, N); o += L * N
        H = stats[o:o + N * L * L].reshape(N, L, L); o += N * L * L
        rv = stats[o:o + L * N].reshape(L, N); o += L * N
        gb = stats[o:o + P * N].reshape(P, N); o += P * N
        Hb = stats[o:o + N * P * P].reshape(N, P, P)
        for n in range(N):
            if pois[n]:
                g_a = MtY[:, n] - g1[:, n]
                step = learning_rate * g_a
                if use_hessian:
                    try:
                        step = _spd_solve(H[n] + np.diag(rv[:, n]) + eps * np.eye(L), g_a)
                    except Exception:
                        pass
                step = np.clip(step, -da_bound, da_bound)
                da[:, n] = step
                a[:, n] += step
                g_b = XtY[:, n] - gb[:, n]
                step = learning_rate * g_b
                if use_hessian:
                    try:
                        step = _spd_solve(Hb[n] + eps * np.eye(P), g_b)
                    except Exception:
                        pass
                step = np.clip(step, -db_bound, db_bound)
                db[:, n] = step
                b[:, n] += step
            else:
                a[:, n] = _spd_solve(gram + np.diag(sumv), MtY[:, n] - XtM[n].T @ b[:, n])
                b[:, n] = _spd_solve(XtX[n], XtY[:, n] - XtM[n] @ a[:, n])
                b[1:, n] = 0
    return a, b, da, db, noise


# --------------------------------------------------------------------------
# api.sample_posterior  (vlgp/api.py:142-168)
# --------------------------------------------------------------------------
def posterior_covariance_reference(G_l, w_l, reg=1e-6):
    """The covariance the reference samples from: inv(inv(K + reg I) + W), K = G G'.  api.py:160-163."""
    K = G_l @ G_l.T
    return np.linalg.inv(np.linalg.inv(K + reg * np.eye(K.shape[0])) + np.diag(w_l))


def sample_posterior_lowrank(mu, w, G, eps):
    """Draws mu_l + G_l Lc^-T eps_l with Lc Lc' = I + G_l' W_l G_l: the same Gaussian as the reference's
    (reg -> 0) through the low-rank factor.  eps: list of (r_l, n) standard normal arrays, r_l = number of
    leading non-zero columns of G_l.  Returns (n, T, L)."""
    T, L = mu.shape
    n = eps[0].shape[1]
    out = np.empty((n, T, L))
    for l in range(L):
        r = eps[l].shape[0]
        Gl = G[l][:, :r]
        H = Gl.T @ (w[:, [l]] * Gl)
        Lc = np.linalg.cholesky(np.eye(r) + H)
        out[:, :, l] = mu[:, l][None, :] + (Gl @ sla.solve_triangular(Lc.T, eps[l], lower=False)).T
    return out
