"""Configuration, parameter skeleton and initialisation (vlgp/preprocess.py).

Host-side and tiny; runs once per fit.  Semantics follow the reference:
unknown keyword arguments are dropped silently, rank is fixed at 50, and the
initial loading / latent come from a FactorAnalysis fit on a random 10 %
subsample drawn from the global NumPy RNG.
"""
import numpy as np

_ONE = np.ones(())

_DEFAULTS = (
    ("constrain_loading", "fro"), ("constrain_latent", False), ("use_hessian", True),
    ("eps", 1e-8), ("tol", 1e-8), ("min_iter", 5), ("method", "VB"), ("learning_rate", 1.0),
    ("max_iter", 20), ("Eniter", 25), ("Mniter", 25), ("Hstep", True), ("da_bound", 5.0),
    ("db_bound", 5.0), ("dmu_bound", 5.0), ("omega_bound", (5e-4, 5e-2)), ("window", 50),
    ("saving_interval", 60 * 30), ("callbacks", None), ("parallel", False),
    # build-specific: "device" (HIP ichol kernel) or "host" (NumPy, reference pivots)
    ("ichol", "device"),
)


def get_config(**kwargs):
    """vlgp/preprocess.py:84-112."""
    config = {k: ([] if k == "callbacks" else v) for k, v in _DEFAULTS}
    config.update({k: v for k, v in kwargs.items() if k in config})
    return config


def get_params(trials, zdim, **kwargs):
    """vlgp/preprocess.py:49-81."""
    ydim = trials[0]["y"].shape[-1]
    lik = kwargs.get("lik", "poisson")
    if not isinstance(lik, list):
        lik = [lik] * ydim
    return {
        "ydim": ydim, "zdim": zdim, "xdim": max(kwargs.get("history", 0), 1),
        "a": kwargs.get("a", None), "b": kwargs.get("b", None),
        "noise": kwargs.get("noise", np.full(ydim, fill_value=1.0)),
        "sigma": kwargs.get("sigma", np.full(zdim, fill_value=1.0)),
        "omega": kwargs.get("omega", np.full(zdim, fill_value=kwargs["omega_bound"][1])),
        "rank": 50, "gp_noise": 1e-4, "dt": 1, "likelihood": np.asarray(lik),
    }


def _gather_rows(trials, pick):
    """Rows ``pick`` of the row-wise concatenation of every trial's y, without building it."""
    off = np.zeros(len(trials) + 1, dtype=np.int64)
    np.cumsum([tr["y"].shape[0] for tr in trials], out=off[1:])
    owner = np.searchsorted(off, pick, side="right") - 1
    order = np.argsort(owner, kind="stable")
    out = np.empty((pick.size, trials[0]["y"].shape[-1]))
    bounds = np.searchsorted(owner[order], np.arange(len(trials) + 1))
    for i, tr in enumerate(trials):
        sel = order[bounds[i]:bounds[i + 1]]
        if sel.size:
            out[sel] = tr["y"][pick[sel] - off[i]]
    return out


def _few_blas_threads(limit=8):
    """The factor analysis works on a (rows / 10, N) sample and N x N matrices.  Left alone OpenBLAS runs it on every
    hardware thread (256 on the MI355X hosts), which is no faster -- and the idle workers then spin for ~100 ms,
    during which everything else the process does (the 160 MB concatenation and upload of y that follow) runs 5-10x
    slower: measured 82-130 ms instead of 25 ms for the upload at C3."""
    import contextlib

    try:
        from threadpoolctl import threadpool_info, threadpool_limits

        now = [m.get("num_threads", 1) for m in threadpool_info() if m.get("user_api") == "blas"]
        # only ever LOWER the count: an OpenBLAS that started with OMP_NUM_THREADS=1 (torchrun exports that) has
        # buffers for one thread and crashes when asked for more
        if not now or max(now) <= limit:
            return contextlib.nullcontext()
        return threadpool_limits(limits=limit, user_api="blas")
    except Exception:  # threadpoolctl is optional
        return contextlib.nullcontext()


def initialize(trials, params, config, defer_latent=False, pool=None):
    """vlgp/preprocess.py:4-46.

    With ``defer_latent`` (used by ``fit``, which uploads y anyway) the two operations that touch
    every row of y -- the initial latents ``mu = transform(y)`` and ``b = log mean y`` -- are left
    to the device: the function then returns ``{"proj", "shift", "need_b"}`` for
    ``Engine.project_latent`` and gives every trial a zero ``mu`` placeholder.  It falls back to the
    host path (and returns None) when the caller supplied a transform or any ``mu``.

    ``pool`` (an attached :class:`Engine` of a multi-rank fit: ``allreduce_host``, ``rank``, ``world``):
    ``trials`` is this rank's shard.  The subsample is then drawn over the rows of ALL ranks (rank 0's
    draw), the factor analysis is fitted to the pooled sample (every rank gets the same loading, noise
    and latent map) and ``b`` is the log of the global channel means -- the same initialisation the
    single-process fit of the concatenated trials computes, whatever the sharding."""
    zdim, xdim = params["zdim"], params["xdim"]
    rows_local = int(sum(tr["y"].shape[0] for tr in trials))
    ydim = trials[0]["y"].shape[-1]
    pooled = pool is not None and getattr(pool, "world", 1) > 1
    row0, rows = 0, rows_local
    if pooled:
        counts = np.zeros(pool.world)
        counts[pool.rank] = rows_local
        pool.allreduce_host(counts)
        row0, rows = int(counts[:pool.rank].sum()), int(counts.sum())
    pick = np.random.choice(rows, max(rows // 10, 50))
    if pooled:
        buf = pick.astype(np.float64) if pool.rank == 0 else np.zeros(pick.size)
        pool.allreduce_host(buf)
        pick = buf.astype(np.int64)
        pick = pick[(pick >= row0) & (pick < row0 + rows_local)] - row0
    defer = bool(defer_latent) and params.get("transform") is None and \
        not any(tr.get("mu") is not None for tr in trials)
    y = None if defer else np.concatenate([tr["y"] for tr in trials], axis=0)
    plan = None
    if params.get("transform") is None:
        # FactorAnalysis(n_components=zdim, random_state=0) of preprocess.py:18-19, restated
        # without scikit-learn (vlgp_amd/fa.py: same estimator, same random test vectors)
        from .fa import fit_factor_analysis

        sample = _gather_rows(trials, pick) if defer else y[pick, :]
        with _few_blas_threads():
            if pooled:
                fa = fit_factor_analysis(sample, zdim, seed=0, allreduce=pool.allreduce_host, rank=pool.rank,
                                         world=pool.world)
            else:
                fa = fit_factor_analysis(sample, zdim, seed=0)
            z = fa.transform(sample)
        a = fa.components
        params["transform"] = fa.transform
        if params.get("a") is None:
            params["a"] = a
        need_b = params.get("b") is None
        if need_b and not defer:
            colsum = np.sum(y, axis=0, keepdims=True, dtype=np.float64)  # integer counts must not reach the all-reduce as int64
            if pooled:
                pool.allreduce_host(colsum)
            params["b"] = np.log(np.maximum(colsum / rows, config["eps"])) if pooled else \
                np.log(np.maximum(np.mean(y, axis=0, keepdims=True), config["eps"]))
        if params.get("noise") is None:
            res = sample - z @ a
            if pooled:
                mom = np.concatenate([[float(res.shape[0])], res.sum(axis=0), (res * res).sum(axis=0)])
                pool.allreduce_host(mom)
                m1 = mom[1:1 + ydim] / mom[0]
                params["noise"] = mom[1 + ydim:] / mom[0] - m1 * m1
            else:
                params["noise"] = np.var(res, ddof=0, axis=0)
        if defer:
            plan = {"proj": fa.projection, "shift": fa.shift, "need_b": need_b, "rows": rows}
    to_latent = params["transform"]
    mu_all = None
    if not defer and not any(tr.get("mu") is not None for tr in trials):
        mu_all = np.asarray(to_latent(y), dtype=float)  # one product over all rows instead of one per trial
    row = 0
    for tr in trials:
        T = tr["y"].shape[0]
        if tr.get("mu") is None:
            if defer:
                tr["mu"] = np.zeros((T, zdim))
            else:
                tr["mu"] = mu_all[row:row + T].copy() if mu_all is not None else to_latent(tr["y"])
        row += T
        if tr.get("x") is None:
            # the reference allocates np.ones((T, xdim, ydim)) per trial (preprocess.py:43-44): the same
            # values as a zero-stride read-only view -- 160 MB less to write and re-scan at C3
            tr["x"] = np.broadcast_to(_ONE, (T, xdim, ydim))
        tr["w"] = np.zeros((T, zdim))
        tr["v"] = np.zeros((T, zdim))
    return plan


def fill_trials(trials):
    """vlgp/preprocess.py:115-120."""
    for i, tr in enumerate(trials):
        tr["cut"] = i
        for key in ("w", "v", "dmu"):
            tr.setdefault(key, np.zeros_like(tr["mu"]))


def fill_params(params):
    """vlgp/preprocess.py:123-125."""
    params.setdefault("da", np.zeros_like(params["a"]))
    params.setdefault("db", np.zeros_like(params["b"]))
