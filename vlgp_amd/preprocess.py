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


def initialize(trials, params, config, defer_latent=False):
    """vlgp/preprocess.py:4-46.

    With ``defer_latent`` (used by ``fit``, which uploads y anyway) the two operations that touch
    every row of y -- the initial latents ``mu = transform(y)`` and ``b = log mean y`` -- are left
    to the device: the function then returns ``{"proj", "shift", "need_b"}`` for
    ``Engine.project_latent`` and gives every trial a zero ``mu`` placeholder.  It falls back to the
    host path (and returns None) when the caller supplied a transform or any ``mu``."""
    zdim, xdim = params["zdim"], params["xdim"]
    rows = int(sum(tr["y"].shape[0] for tr in trials))
    ydim = trials[0]["y"].shape[-1]
    pick = np.random.choice(rows, max(rows // 10, 50))
    defer = bool(defer_latent) and params.get("transform") is None and \
        not any(tr.get("mu") is not None for tr in trials)
    y = None if defer else np.concatenate([tr["y"] for tr in trials], axis=0)
    plan = None
    if params.get("transform") is None:
        # FactorAnalysis(n_components=zdim, random_state=0) of preprocess.py:18-19, restated
        # without scikit-learn (vlgp_amd/fa.py: same estimator, same random test vectors)
        from .fa import fit_factor_analysis

        sample = _gather_rows(trials, pick) if defer else y[pick, :]
        fa = fit_factor_analysis(sample, zdim, seed=0)
        z = fa.transform(sample)
        a = fa.components
        params["transform"] = fa.transform
        if params.get("a") is None:
            params["a"] = a
        need_b = params.get("b") is None
        if need_b and not defer:
            params["b"] = np.log(np.maximum(np.mean(y, axis=0, keepdims=True), config["eps"]))
        if params.get("noise") is None:
            params["noise"] = np.var(sample - z @ a, ddof=0, axis=0)
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
