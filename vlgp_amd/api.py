"""``fit`` / ``transform`` with the reference's call contract (vlgp/api.py:18-76,171-184).

``fit(trials, n_factors, **kwargs) -> {"trials", "params", "config"}``: the
caller's trial dicts are mutated in place and returned; mu, v, dmu are updated
in place, w is replaced; numerical failures never raise.  The whole EM loop
runs on the GPU between one upload and one download.
"""
import copy
import logging

import numpy as np

from . import engine as E
from .preprocess import fill_params, fill_trials, get_config, get_params, initialize
from .util import segment_starts

__all__ = ["fit", "transform"]

logger = logging.getLogger(__name__)

SET_TRIALS, SET_SEGMENTS = 0, 1


def _echo(msg):
    print(msg, flush=True)


def _segments(trials, window, eng):
    """Cut the resident trials into window-sized segments (util.cut_trials):
    host-side NumPy views for the dict interface, one device cut for the data."""
    starts, segs, row0 = [], [], 0
    for tr in trials:
        T = tr["y"].shape[0]
        for s in segment_starts(T, window):
            sl = slice(int(s), int(s) + window)
            segs.append({k: tr[k][sl] for k in ("y", "x", "mu", "w", "v")})
            starts.append(row0 + int(s))
        row0 += T
    eng.cut(SET_TRIALS, SET_SEGMENTS, np.asarray(starts, dtype=np.int64), window)
    return E.DeviceTrials(segs, eng, SET_SEGMENTS)


def fit(trials, n_factors, device=0, comm=None, verbose=True, **kwargs):
    """Variational-EM fit of vLGP on one MI355X (or one rank of several).

    Same arguments as the reference: ``lik``, ``history``, ``a``, ``b``,
    ``noise``, ``sigma``, ``omega`` and every ``get_config`` key.  Extra:
    ``device`` (GPU index), ``comm`` (a :class:`vlgp_amd.dist.Comm` when the
    trials are sharded over ranks), ``verbose``.
    """
    echo = _echo if verbose else None
    config = get_config(**kwargs)
    logger.info("\n".join("{} : {}".format(k, v) for k, v in config.items()))
    kwargs["omega_bound"] = config["omega_bound"]
    params = get_params(trials, n_factors, **kwargs)

    if echo:
        echo("Initializing")
    initialize(trials, params, config)
    if echo:
        echo("Initialized")
    fill_params(params)
    fill_trials(trials)
    params["a"] = np.array(params["a"], dtype=float)
    params["b"] = np.array(params["b"], dtype=float)
    params["noise"] = np.array(params["noise"], dtype=float)
    params["omega"] = np.array(params["omega"], dtype=float)
    params["sigma"] = np.array(params["sigma"], dtype=float)

    eng = E.Engine(params["ydim"], params["zdim"], params["xdim"], params["rank"],
                   np.asarray(params["likelihood"]) == "gaussian", device=device)
    try:
        if comm is not None:
            comm.attach(eng)
        eng.set_params(params["a"], params["b"], params["noise"])
        eng.upload(SET_TRIALS, trials)
        dev_trials = E.DeviceTrials(trials, eng, SET_TRIALS)
        E.make_cholesky(dev_trials, params, config)
        E.update_w(dev_trials, params, config)
        E.update_v(dev_trials, params, config)

        window = config["window"]
        if window:
            segs = _segments(trials, window, eng)
            E.make_cholesky(segs, params, config)
            fill_trials(segs)
        else:
            segs = dev_trials

        snapshot = {k: v for k, v in params.items() if k not in ("cholesky", "transform")}
        params["initial"] = copy.deepcopy(snapshot)

        if echo:
            echo("Fitting")
        E.vem(segs, params, config, echo=echo)

        if segs is not dev_trials:
            eng.merge(SET_SEGMENTS)
        E.make_cholesky(dev_trials, params, config)
        E.update_w(dev_trials, params, config)
        E.update_v(dev_trials, params, config)
        if echo:
            echo("Inferring")
        E.infer(dev_trials, params, config, echo=echo)
        dev_trials.pull()
        if isinstance(params["cholesky"], E._LazyPrior):
            params["cholesky"] = params["cholesky"].materialize()
        if echo:
            echo("Done")
    finally:
        eng.close()
    return {"trials": trials, "params": params, "config": config}


def transform(trials, params, config, device=0):
    """Infer latents of new trials with fitted parameters (vlgp/api.py:171-184).

    Unlike the reference (which raises KeyError for a trial length it has no
    factor for), missing prior factors are built on the fly."""
    initialize(trials, params, config)
    fill_trials(trials)
    with E.Engine(params["ydim"], params["zdim"], params["xdim"], params["rank"],
                  np.asarray(params["likelihood"]) == "gaussian", device=device) as eng:
        eng.set_params(params["a"], params["b"], params["noise"])
        eng.upload(SET_TRIALS, trials)
        dev = E.DeviceTrials(trials, eng, SET_TRIALS)
        chol = params.get("cholesky") or {}
        lengths = sorted({int(tr["y"].shape[0]) for tr in trials})
        missing = [T for T in lengths if T not in chol]
        if missing:
            eng.build_prior(missing, params["omega"], params["sigma"])
            for T in missing:
                chol[T] = eng.get_prior(T)
        for T in lengths:
            if T not in missing:
                eng.set_prior(T, chol[T])
        params["cholesky"] = dict(chol)
        E.infer(dev, params, config)
        dev.pull()
    return trials
