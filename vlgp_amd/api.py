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

__all__ = ["fit", "transform", "FitSession"]

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
        if T < window:
            # the reference would cut a shorter segment here and then fail in gp.optimize (np.stack of
            # unequal segments); the device cut needs equal lengths, so say it up front
            raise ValueError("trial of %d bins is shorter than window=%d; pass a smaller window" % (T, window))
        for s in segment_starts(T, window):
            sl = slice(int(s), int(s) + window)
            segs.append({k: tr[k][sl] for k in ("y", "x", "mu", "w", "v")})
            starts.append(row0 + int(s))
        row0 += T
    # the per-segment dmu buffers fill_trials would otherwise allocate one by one (4000 at C3): views of one block
    if segs:
        block = np.zeros((len(segs), window, segs[0]["mu"].shape[1]))
        for i, sg in enumerate(segs):
            sg["dmu"] = block[i]
    starts = np.asarray(starts, dtype=np.int64)
    # Overlapping neighbours (a trial length that is not a multiple of the window, vlgp/util.py:482-496): in the reference
    # they are views of the same rows, visited one after the other.  The device keeps independent copies, stored
    # stage-major (stage = position in a chain of overlapping neighbours) so that the E-step can run stage by stage with
    # the shared rows handed over in between (vlgp_set_overlaps); `unit_of` maps a segment of this list to its unit.
    n = len(segs)
    over = np.zeros(n, dtype=np.int64)          # rows segment i shares with segment i - 1
    if n > 1:
        gap = starts[1:] - starts[:-1]
        over[1:] = np.where((gap > 0) & (gap < window), window - gap, 0)
    if not over.any():
        eng.cut(SET_TRIALS, SET_SEGMENTS, starts, window)
        return E.DeviceTrials(segs, eng, SET_SEGMENTS, parent_set=SET_TRIALS)
    stage = np.zeros(n, dtype=np.int64)
    for i in range(1, n):
        stage[i] = stage[i - 1] + 1 if over[i] else 0
    order = np.argsort(stage, kind="stable")    # unit u holds segment order[u]
    unit_of = np.empty(n, dtype=np.int64)
    unit_of[order] = np.arange(n)
    n_stages = int(stage.max()) + 1
    stage_start = np.searchsorted(stage[order], np.arange(n_stages + 1)).astype(np.int32)
    linked = np.flatnonzero(over)               # segment i shares over[i] rows with segment i - 1
    linked = linked[np.argsort(stage[linked], kind="stable")]
    links = np.stack([unit_of[linked - 1], unit_of[linked], over[linked]], axis=1).astype(np.int32)
    link_start = np.searchsorted(stage[linked], np.arange(n_stages + 1)).astype(np.int32)
    eng.cut(SET_TRIALS, SET_SEGMENTS, starts[order], window)
    eng.set_overlaps(SET_SEGMENTS, stage_start, links, link_start)
    return E.DeviceTrials(segs, eng, SET_SEGMENTS, parent_set=SET_TRIALS, unit_of=unit_of)


class FitSession:
    """The three phases of api.fit as separate calls.

    ``FitSession(...)``  api.py:27-60 -- config, params, initialisation, upload,
                         prior factor, w, v, segmentation
    ``run()``            api.py:64 -- core.vem on the segments
    ``finish()``         api.py:66-76 -- full-length prior, w, v, inference, download

    ``bench.py`` drives ``em_iteration`` directly to time single EM iterations;
    ``fit`` is ``FitSession(...).run().finish()``.
    """

    def __init__(self, trials, n_factors, device=0, comm=None, verbose=True, **kwargs):
        self.echo = _echo if verbose else None
        echo = self.echo
        self.trials = trials
        self.config = config = get_config(**kwargs)
        logger.info("\n".join("{} : {}".format(k, v) for k, v in config.items()))
        kwargs["omega_bound"] = config["omega_bound"]
        self.params = params = get_params(trials, n_factors, **kwargs)
        self.comm = comm
        self.eng = None
        self.runtime = E.new_runtime()
        # the reference leaves a writable np.ones((T, xdim, N)) in every trial that came without regressors
        # (preprocess.py:43-44); the fit itself works on a zero-stride view, the real arrays are made on the way out
        self.materialize_x = bool(kwargs.get("materialize_x", True))

        if echo:
            echo("Initializing")
        eng = E.Engine(params["ydim"], params["zdim"], params["xdim"], params["rank"],
                       np.asarray(params["likelihood"]) == "gaussian", device=device)
        self.eng = eng
        try:
            multi = comm is not None and comm.world > 1
            if multi:
                comm.attach(eng)
            # sharded fit: the subsample, the factor analysis and b = log mean y are pooled over the ranks
            plan = initialize(trials, params, config, defer_latent=True, pool=eng if multi else None)
            fill_trials(trials)
            eng.upload(SET_TRIALS, trials)
            if plan is not None:
                # the two full passes over y of preprocess.initialize, on the device: mu = transform(y), b = log mean y
                colsum = eng.project_latent(SET_TRIALS, plan["proj"], plan["shift"])
                if plan["need_b"]:
                    if multi:
                        eng.allreduce_host(colsum)
                    params["b"] = np.log(np.maximum(colsum[None, :] / plan["rows"], config["eps"]))
            if echo:
                echo("Initialized")
            fill_params(params)
            for key in ("a", "b", "noise", "omega", "sigma"):
                params[key] = np.array(params[key], dtype=float)
            if multi:
                self._replicate_params()
            eng.set_params(params["a"], params["b"], params["noise"])
            self.dev_trials = E.DeviceTrials(trials, eng, SET_TRIALS)
            E.make_cholesky(self.dev_trials, params, config)
            E.update_w(self.dev_trials, params, config)
            E.update_v(self.dev_trials, params, config)
            window = config["window"]
            if window:
                self.segs = _segments(trials, window, eng)
                E.make_cholesky(self.segs, params, config)
                fill_trials(self.segs)
            else:
                self.segs = self.dev_trials
            # params["initial"] = deepcopy(params) (api.py:60): every key, the segment-length factors included.  The
            # resident factor is downloaded for it when a window is set (one (L, window, rank) array); without a
            # window it is every full-length factor: those are rebuilt on the host when somebody reads them
            snapshot = {k: v for k, v in params.items() if k != "cholesky"}
            chol = params["cholesky"]
            if isinstance(chol, E._LazyPrior):
                # no window: the full-length factors of the initial omega, factored on first access (engine._InitialPrior)
                chol = chol.materialize() if window else E._InitialPrior(chol._lengths, params["omega"], params["sigma"],
                                                                         params["rank"])
            snapshot["cholesky"] = chol
            params["initial"] = copy.deepcopy(snapshot)
            E._push_params(eng, params)
        except Exception:
            self.close()
            raise

    def _replicate_params(self):
        """Every rank must start from the same a, b, noise: keep rank 0's."""
        p, eng = self.params, self.eng
        for key in ("a", "b", "noise"):
            buf = np.ascontiguousarray(p[key], dtype=float)
            if eng.rank != 0:
                buf = np.zeros_like(buf)
            eng.allreduce_host(buf)
            p[key] = buf

    def em_iteration(self):
        """One EM iteration on the segments; True when the stopping rule fires."""
        return E.em_iteration(self.segs, self.params, self.config, self.runtime, self.echo)

    def run(self):
        if self.echo:
            self.echo("Fitting")
        for _ in range(self.config["max_iter"]):
            if self.em_iteration():
                break
        return self

    def finish(self):
        eng, params, config, echo = self.eng, self.params, self.config, self.echo
        try:
            if self.segs is not self.dev_trials:
                eng.merge(SET_SEGMENTS)
                if self.segs.detached:  # constrain_loading "svd": the trials kept their pre-rotation mu (core.py:407-408)
                    eng.stash_mu(SET_TRIALS, restore=True)
            E.make_cholesky(self.dev_trials, params, config)
            E.update_w(self.dev_trials, params, config)
            E.update_v(self.dev_trials, params, config)
            if echo:
                echo("Inferring")
            E.infer(self.dev_trials, params, config, echo=echo)
            self.dev_trials.pull()
            if self.materialize_x:
                for tr in self.trials:
                    x = tr.get("x")
                    if isinstance(x, np.ndarray) and not x.flags.writeable and x.size and all(st == 0 for st in x.strides):
                        tr["x"] = np.ones(x.shape)
            if isinstance(params["cholesky"], E._LazyPrior):
                params["cholesky"] = params["cholesky"].materialize()
            if echo:
                echo("Done")
        finally:
            self.close()
        return {"trials": self.trials, "params": params, "config": config}

    def close(self):
        if self.eng is not None:
            self.eng.close()
            self.eng = None


def fit(trials, n_factors, device=0, comm=None, verbose=True, **kwargs):
    """Variational-EM fit of vLGP on one MI355X (or one rank of several).

    Same arguments as the reference: ``lik``, ``history``, ``a``, ``b``,
    ``noise``, ``sigma``, ``omega`` and every ``get_config`` key.  Extra:
    ``device`` (GPU index), ``comm`` (a :class:`vlgp_amd.dist.Comm` when the
    trials are sharded over ranks), ``verbose``, ``ichol`` ("device"/"host": who builds the prior
    factor -- the device kernel is bit-identical to the reference's on the NumPy 2.2 / AVX-512 / OpenBLAS
    stack the golden vectors were captured on, "host" uses this host's own NumPy: INTEGRATION.md),
    ``materialize_x`` (default True: trials that came
    without regressors get a writable ``np.ones((T, xdim, N))`` back, as the reference leaves them).
    """
    return FitSession(trials, n_factors, device=device, comm=comm, verbose=verbose, **kwargs).run().finish()


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
        chol = dict(params.get("cholesky") or {})
        lengths = sorted({int(tr["y"].shape[0]) for tr in trials})
        missing = [T for T in lengths if T not in chol]
        have = [T for T in lengths if T in chol]
        if missing:
            eng.build_prior(missing, params["omega"], params["sigma"])
            for T in missing:
                chol[T] = eng.get_prior(T)
        for T in have:
            eng.set_prior(T, chol[T])
        params["cholesky"] = chol
        E.infer(dev, params, config)
        dev.pull()
    return trials


def sample_posterior(trial, params, nsamples, reg=1e-6, rng=None, device=0):
    """Draw ``nsamples`` latent trajectories from the variational posterior of one trial
    (vlgp/api.py:142-168): independent Gaussians per latent with mean ``mu[:, l]`` and covariance
    ``(K_l^-1 + W_l)^-1``, ``K_l = G_l G_l'`` the low-rank prior of this trial length.

    The reference forms the T x T matrices (two dense inverses per latent plus
    ``multivariate_normal``'s SVD: O(T^3)).  With ``H = G'WG`` (r x r) the same covariance is
    ``G (I + H)^-1 G'`` (the reference's ``reg`` regulariser set to zero; it only exists to make
    ``K`` invertible), so a draw is ``mu + G L^-T eps`` with ``L L' = I + H`` and ``eps ~ N(0, I_r)``:
    O(T r^2) per latent, computed on the GPU (``vlgp_sample_posterior``).  ``reg`` is accepted and ignored.
    The standard normal draws come from ``rng`` (a ``numpy.random.Generator``/``RandomState``) or, like
    the reference, the global NumPy state: ``(r_l, nsamples)`` values per latent, in latent order.

    Returns an array of shape (nsamples, bins, nfactors)."""
    del reg
    mu = np.ascontiguousarray(trial["mu"], dtype=float)
    w = np.ascontiguousarray(trial["w"], dtype=float)
    nbins, nfactors = mu.shape
    G = np.ascontiguousarray(params["cholesky"][nbins], dtype=float)
    R = G.shape[-1]
    if rng is None:
        normal = np.random.standard_normal
    elif hasattr(rng, "standard_normal"):
        normal = rng.standard_normal
    else:  # anything with normal(loc, scale, size)
        normal = lambda shape: rng.normal(size=shape)
    eps = np.zeros((nfactors, R, int(nsamples)))
    for l in range(nfactors):
        nz = np.flatnonzero(np.any(G[l] != 0.0, axis=0))  # columns ichol_gauss left at zero carry nothing
        r = int(nz[-1]) + 1 if nz.size else 1
        eps[l, :r] = normal((r, int(nsamples)))
    out = np.empty((int(nsamples), nbins, nfactors))
    import ctypes as C

    from ._lib import dptr

    # the latents are independent: more than a handle holds (16) go through in groups
    for l0 in range(0, nfactors, 16):
        sl = slice(l0, min(l0 + 16, nfactors))
        nl = sl.stop - sl.start
        mu_c, w_c, G_c, eps_c = (np.ascontiguousarray(arr) for arr in (mu[:, sl], w[:, sl], G[sl], eps[sl]))
        out_c = out if nl == nfactors else np.empty((int(nsamples), nbins, nl))
        with E.Engine(2, nl, 1, R, device=device) as eng:
            bad = C.c_int(0)
            eng._ck(eng.lib.vlgp_sample_posterior(eng.h, nbins, dptr(mu_c), dptr(w_c), dptr(G_c), int(nsamples),
                                                  dptr(eps_c), dptr(out_c), C.byref(bad)))
            if bad.value:
                logger.error("I + G'WG was not positive definite for %d latent(s): their draws equal the mean",
                             bad.value)
        if out_c is not out:
            out[:, :, sl] = out_c
    return out
