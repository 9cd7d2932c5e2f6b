"""GP hyper-parameter step (H-step) and prior-factor helpers.

``optimize`` mirrors gp.optimize (vlgp/gp.py:65-97): one bounded L-BFGS-B run
per latent over log(sigma^2, omega, eps) with the gradient masked to omega.
The optimiser itself is SciPy's (it is control logic, a few dozen scalar
decisions per latent); every objective evaluation -- the T x T factorisations
per segment that make up all of the arithmetic -- runs on the GPU through
``Engine.hstep_objective``.  The L latents' optimisers run in lock step so
that each round of evaluations is ONE kernel launch over all latents.
"""
import logging
import math
import threading

import numpy as np
from scipy.optimize import minimize

logger = logging.getLogger(__name__)


def ichol_gauss_host(n, omega, rank, dt=1.0, tol=1e-6):
    """Host NumPy pivoted incomplete Cholesky of exp(-omega (i-j)^2 dt^2).

    Same algorithm and operation order as math.ichol_gauss (vlgp/math.py:76-126);
    used only when ``config["ichol"] == "host"`` is requested explicitly.
    """
    pos = np.arange(n) * dt
    resid = np.ones(n)
    perm = np.arange(n)
    F = np.zeros((n, rank))
    k = 0
    while k < rank and resid[k:].sum() > tol * n:
        p = k + int(np.argmax(resid[k:])) if k else 0
        if p != k:
            perm[[k, p]] = perm[[p, k]]
            F[[k, p], :k + 1] = F[[p, k], :k + 1]
        piv = math.sqrt(resid[p])
        F[k, k] = piv
        rest = perm[k + 1:]
        F[k + 1:, k] = (np.exp(-omega * (pos[rest] - pos[perm[k]]) ** 2)
                        - np.dot(F[k + 1:, :k], F[k, :k])) / piv
        resid[k + 1:] = 1 - np.sum(np.square(F[k + 1:, :k + 1]), axis=1)
        k += 1
    return F[np.argsort(perm), :]


class _LockStep:
    """Batches the objective evaluations of several concurrently running
    scipy optimisers into single device launches.

    Each optimiser runs in its own Python thread and blocks in ``evaluate``
    until every still-running optimiser has posted a request; the last one to
    arrive launches one batched evaluation for all of them.  Evaluations of
    different latents are independent, so the batching changes no result.
    """

    def __init__(self, n_workers, batch_fn):
        self.cv = threading.Condition()
        self.active = n_workers
        self.pending = {}
        self.results = {}
        self.batch_fn = batch_fn
        self.error = None

    def _flush(self):
        keys = sorted(self.pending)
        try:
            ll, dll = self.batch_fn(keys, np.stack([self.pending[k] for k in keys]))
            for i, k in enumerate(keys):
                self.results[k] = (float(ll[i]), np.array(dll[i]))
        except Exception as exc:  # propagate to every waiting optimiser
            self.error = exc
            for k in keys:
                self.results[k] = None
        self.pending.clear()
        self.cv.notify_all()

    def evaluate(self, key, logp):
        with self.cv:
            self.pending[key] = np.array(logp, dtype=float)
            if len(self.pending) >= self.active:
                self._flush()
            while key not in self.results:
                self.cv.wait()
            res = self.results.pop(key)
        if res is None:
            raise self.error
        return res

    def retire(self):
        with self.cv:
            self.active -= 1
            if self.pending and len(self.pending) >= self.active:
                self._flush()


def optimize(trials, params, config):
    """gp.optimize (vlgp/gp.py:65-97) on a DeviceTrials of equal-length segments."""
    from .engine import DeviceTrials, make_cholesky

    if not isinstance(trials, DeviceTrials):
        raise TypeError("the H-step runs on DeviceTrials (use vlgp_amd.fit)")
    eng, sid = trials.engine, trials.set_id
    L = params["zdim"]
    window = config["window"]
    dt = params["dt"]
    gp_noise = params["gp_noise"]
    sigma = params["sigma"]
    omega = params["omega"]
    bounds = np.log(np.array([(1e-3, 1.0), tuple(config["omega_bound"]),
                              (gp_noise / 2, gp_noise * 2)]))

    def batch(latents, logps):
        return eng.hstep_objective(sid, window, dt, latents, logps)

    lock = _LockStep(L, batch)
    out = [None] * L
    errors = []

    def run(l):
        try:
            x0 = np.log(np.array([sigma[l] ** 2, omega[l], gp_noise]))

            def neg(x):
                ll, dll = lock.evaluate(l, x)
                return -ll, -dll

            out[l] = minimize(neg, x0, jac=True, bounds=bounds)
        except Exception as exc:
            errors.append(exc)
        finally:
            lock.retire()

    if L == 1:
        run(0)
    else:
        threads = [threading.Thread(target=run, args=(l,), daemon=True) for l in range(L)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    for l in range(L):
        sig2, om, _ = np.exp(out[l].x)
        if not np.any(np.isclose(om, config["omega_bound"])):  # gp.py:91-92
            omega[l] = om
        sigma[l] = math.sqrt(sig2)
    params["sigma"] = sigma
    params["omega"] = omega
    make_cholesky(trials, params, config)
