"""GP hyper-parameter step (H-step) and prior-factor helpers.

``optimize`` mirrors gp.optimize (vlgp/gp.py:65-97): one bounded L-BFGS-B run
per latent over log(sigma^2, omega, eps) with the gradient masked to omega.
The optimiser is host control logic (a few dozen scalar decisions per latent):
since round 6 an own reverse-communication L-BFGS-B in C (csrc/lbfgsb.c, inside
``vlgp_amd._lockstep``) that takes the very steps of SciPy's -- bit for bit
when it is bound to the OpenBLAS SciPy ships (``_scipy_blas_addresses``), to
rounding with its own loops; SciPy's own routine remains as the fallback when
the module is not built.  Every objective evaluation -- the T x T
factorisations per segment that make up all of the arithmetic -- runs on the
GPU through ``Engine.hstep_objective``.  The L latents' optimisers run in lock
step so that each round of evaluations is ONE kernel launch over all latents.
"""
import logging
import math
import os
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
            f, G = self.batch_fn(keys, np.stack([self.pending[k] for k in keys]))
            for i, k in enumerate(keys):
                self.results[k] = (float(f[i]), np.array(G[i]))
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


class _Lbfgsb:
    """One bounded L-BFGS-B run as an explicit state machine.

    Same algorithm, constants and stopping rules as
    ``scipy.optimize.minimize(fun, x0, jac=True, bounds=...)`` (which selects
    L-BFGS-B): it drives the very same reverse-communication routine
    (``scipy.optimize._lbfgsb.setulb``) that SciPy's own driver loop
    (``scipy/optimize/_lbfgsb_py.py:_minimize_lbfgsb``) drives, with SciPy's
    defaults (maxcor 10, ftol 2.22e-9, gtol 1e-5, maxls 20, maxiter = maxfun =
    15000).  Exposing the "needs f, g at x" state lets several runs share one
    batched device evaluation per round without any threads.
    """

    MAXCOR, FTOL, GTOL, MAXLS, MAXITER, MAXFUN = 10, 2.2204460492503131e-09, 1e-5, 20, 15000, 15000

    def __init__(self, setulb, x0, log_bounds, xbuf=None, extra=()):
        self.setulb = setulb
        lo, hi = log_bounds[:, 0].copy(), log_bounds[:, 1].copy()
        n = x0.size
        m = self.MAXCOR
        if xbuf is None:
            self.x = np.clip(np.array(x0, dtype=np.float64), lo, hi)
        else:  # the caller's row of a shared (runs, n) array: the batch of pending points needs no gathering
            self.x = xbuf
            np.clip(np.asarray(x0, dtype=np.float64), lo, hi, out=self.x)
        self.lo, self.hi = lo, hi
        self.nbd = np.full(n, 2, dtype=np.int32)  # both bounds finite
        self.f = 0.0
        self.g = np.zeros(n)
        self.factr = self.FTOL / np.finfo(float).eps
        self.wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m)
        self.iwa = np.zeros(3 * n, dtype=np.int32)
        self.task = np.zeros(2, dtype=np.int32)
        self.ln_task = np.zeros(2, dtype=np.int32)
        self.lsave = np.zeros(4, dtype=np.int32)
        self.isave = np.zeros(44, dtype=np.int32)
        self.dsave = np.zeros(29)
        self.nfev = 0
        self.nit = 0
        self.done = False
        # the call's arguments either side of f (the one scalar that changes): bound once, this runs ~2 x per evaluation
        self._head = (self.MAXCOR, self.x, self.lo, self.hi, self.nbd)
        self._tail = (self.g, self.factr, self.GTOL, self.wa, self.iwa, self.task, self.lsave, self.isave, self.dsave,
                      self.MAXLS, self.ln_task) + tuple(extra)  # (extra: the BLAS table of the own routine)
        self._args = self._head + (self.f,) + self._tail  # rebuilt by feed(): f is the only argument passed by value

    def advance(self):
        """Run until the routine asks for (f, g) at self.x or terminates.
        Returns True when an evaluation is wanted."""
        task = self.task
        setulb, args = self.setulb, self._args
        while not self.done:
            setulb(*args)
            t0 = int(task[0])
            if t0 == 3:
                return True
            if t0 == 1:  # new iterate accepted
                self.nit += 1
                if self.nit >= self.MAXITER:
                    task[0], task[1] = 5, 504
                elif self.nfev > self.MAXFUN:
                    task[0], task[1] = 5, 502
            else:
                self.done = True
        return False

    def feed(self, f, g):
        self.f = float(f)
        self.g[:] = g  # in place: setulb reads the array bound in _tail
        self.nfev += 1
        self._args = self._head + (self.f,) + self._tail


def _setulb_or_none():
    """The private reverse-communication entry point, if this SciPy has the
    signature we know (SciPy 1.15: 17 positional arguments)."""
    try:
        from scipy.optimize import _lbfgsb

        doc = _lbfgsb.setulb.__doc__ or ""
        if "setulb(m,x,l,u,nbd,f,g,factr,pgtol,wa,iwa,task,lsave,isave,dsave,maxls,ln_task)" not in doc.replace(" ", ""):
            return None
        return _lbfgsb.setulb
    except Exception:  # pragma: no cover
        return None


_warned = set()


def _warn_once(msg):
    if msg not in _warned:
        _warned.add(msg)
        logger.warning(msg)


def _lockstep_ext():
    """vlgp_amd._lockstep (csrc/lockstep_ext.c), or None when it is not built."""
    try:
        from . import _lockstep

        return _lockstep
    except Exception:  # pragma: no cover
        return None


_BLAS_NAMES = ("ddot", "daxpy", "dscal", "dcopy", "dnrm2", "dpotrf", "dtrtrs")
_blas_cache = {}


def _scipy_blas_addresses():
    """Addresses of the seven BLAS / LAPACK routines L-BFGS-B calls, taken from the OpenBLAS that SciPy itself links
    (``scipy.libs/libscipy_openblas*.so``, symbols ``scipy_ddot_`` ...): bound to them, csrc/lbfgsb.c reproduces
    scipy.optimize's iterates bit for bit on this machine.  None when that library is not there (another SciPy build,
    no SciPy): the optimiser then runs on its own portable loops -- same algorithm, same decisions up to rounding."""
    if "v" in _blas_cache:
        return _blas_cache["v"]
    out = None
    try:
        import ctypes
        import glob

        import scipy

        root = os.path.dirname(os.path.dirname(os.path.abspath(scipy.__file__)))
        for path in sorted(glob.glob(os.path.join(root, "scipy.libs", "libscipy_openblas*.so*"))):
            if "64_" in os.path.basename(path):
                continue  # (the ILP64 build: 8-byte integers)
            lib = ctypes.CDLL(path)
            addr = tuple(ctypes.cast(getattr(lib, "scipy_%s_" % k), ctypes.c_void_p).value for k in _BLAS_NAMES)
            if all(addr):
                _blas_cache["lib"] = lib  # (keeps the handle alive)
                out = addr
                break
    except Exception:
        out = None
    _blas_cache["v"] = out
    return out


def lbfgsb_blas():
    """The BLAS table the own optimiser runs on: SciPy's OpenBLAS when found, else None = the loops of csrc/lbfgsb.c
    (VLGP_LBFGSB_BLAS=own forces the latter)."""
    if os.environ.get("VLGP_LBFGSB_BLAS", "") == "own":
        return None
    return _scipy_blas_addresses()


def lockstep_minimize_own(objective_address, handle_address, set_id, window, dt, latents, x0s, log_bounds):
    """All runs in lock step with the optimiser of csrc/lbfgsb.c and the objective (vlgp_hstep_objective) called through
    its address: no Python object on the path of a round.  Returns (xs, status); None when the module is not built."""
    ext = _lockstep_ext()
    if ext is None or not hasattr(ext, "run_own"):
        return None
    X = np.ascontiguousarray(np.stack([np.asarray(x0, dtype=np.float64) for x0 in x0s]))
    bnd = np.ascontiguousarray(log_bounds, dtype=np.float64)
    status = ext.run_own([int(l) for l in latents], X, bnd, int(objective_address), int(handle_address), int(set_id),
                         int(window), float(dt), _Lbfgsb.MAXITER, _Lbfgsb.MAXFUN, lbfgsb_blas())
    return [X[k].copy() for k in range(len(x0s))], int(status)


def lockstep_minimize_native(objective_address, handle_address, set_id, window, dt, latents, x0s, log_bounds):
    """lockstep_minimize with the loop around SciPy's setulb and the objective call in C (vlgp_amd._lockstep): same
    calls to the same routine with the same arguments, hence the same iterates; the objective is vlgp_hstep_objective
    called through its address.  Returns (xs, status); None when SciPy's routine or the module is not available."""
    setulb, ext = _setulb_or_none(), _lockstep_ext()
    if setulb is None or ext is None:
        return None
    n_runs = len(x0s)
    X = np.empty((n_runs, 3))
    runs = [_Lbfgsb(setulb, x0, log_bounds, xbuf=X[k]) for k, x0 in enumerate(x0s)]
    status = ext.run(setulb, [(r._head, r._tail, int(l)) for r, l in zip(runs, latents)], int(objective_address),
                     int(handle_address), int(set_id), int(window), float(dt), _Lbfgsb.MAXITER, _Lbfgsb.MAXFUN)
    return [r.x.copy() for r in runs], int(status)


def _own_setulb_or_none():
    """One reverse-communication step of csrc/lbfgsb.c with SciPy's argument list (vlgp_amd._lockstep.setulb)."""
    ext = _lockstep_ext()
    return getattr(ext, "setulb", None) if ext is not None else None


def lockstep_minimize(batch_fn, x0s, log_bounds, routine=None):
    """Minimise len(x0s) independent objectives with L-BFGS-B, evaluating all
    pending points of a round with ONE call ``batch_fn(keys, X) -> (f, G)``
    (f, G are the objective values / gradients to minimise).  Returns the list
    of final x.  Single-threaded on a reverse-communication routine -- the own
    one (csrc/lbfgsb.c; ``routine="own"``), else SciPy's (``"scipy"``); without
    either, one ``scipy.optimize.minimize`` per thread with the same batching
    (identical results every way)."""
    setulb, extra = None, ()
    if routine in (None, "own"):
        setulb = _own_setulb_or_none()
        extra = (lbfgsb_blas(),)
    if setulb is None and routine in (None, "scipy"):
        setulb, extra = _setulb_or_none(), ()
    if setulb is None:
        _warn_once("this SciPy does not expose the L-BFGS-B reverse-communication routine with the signature known here "
                   "(scipy.optimize._lbfgsb.setulb, SciPy 1.15): the H-step runs one scipy.optimize.minimize per latent "
                   "in threads (the public API, same iterates, slower)")
        return _lockstep_threads(batch_fn, x0s, log_bounds)
    n_runs = len(x0s)
    X = np.empty((n_runs, np.asarray(x0s[0]).size))  # row k IS run k's iterate (setulb updates it in place)
    runs = [_Lbfgsb(setulb, x0, log_bounds, xbuf=X[k], extra=extra) for k, x0 in enumerate(x0s)]
    active = list(range(n_runs))
    while True:
        active = [k for k in active if runs[k].advance()]
        if not active:
            break
        f, G = batch_fn(active, X if len(active) == n_runs else X[active])
        for i, k in enumerate(active):
            runs[k].feed(f[i], G[i])
    return [r.x.copy() for r in runs]


def _lockstep_threads(batch_fn, x0s, log_bounds):
    n = len(x0s)
    lock = _LockStep(n, batch_fn)
    out = [None] * n
    errors = []

    def run(k):
        try:
            def obj(x):
                return lock.evaluate(k, x)

            out[k] = minimize(obj, x0s[k], jac=True, bounds=log_bounds).x
        except Exception as exc:
            errors.append(exc)
        finally:
            lock.retire()

    if n == 1:
        run(0)
    else:
        threads = [threading.Thread(target=run, args=(k,), daemon=True) for k in range(n)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    if errors:
        raise errors[0]
    return out


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

    def batch(latents, logps):  # scipy minimises: negate ll and its gradient
        ll, dll = eng.hstep_objective(sid, window, dt, latents, logps, copy=False)
        np.negative(ll, out=ll)  # (the engine's own result buffers: consumed before the next call)
        np.negative(dll, out=dll)
        return ll, dll

    x0s = [np.log(np.array([sigma[l] ** 2, omega[l], gp_noise])) for l in range(L)]
    eng.hstep_begin(sid, window)
    try:
        xs = None
        if not os.environ.get("VLGP_LOCKSTEP_PYTHON"):
            # VLGP_LBFGSB=scipy: SciPy's own routine driven from C (the round-4 / 5 path); default: the own optimiser
            drivers = ((lockstep_minimize_native,) if os.environ.get("VLGP_LBFGSB", "") == "scipy"
                       else (lockstep_minimize_own, lockstep_minimize_native))
            for driver in drivers:
                res = driver(eng.hstep_objective_address, eng.handle_address, sid, window, dt, range(L), x0s, bounds)
                if res is not None:
                    xs, status = res
                    eng.check(status)
                    break
        if xs is None:
            xs = lockstep_minimize(batch, x0s, bounds)
    finally:
        eng.hstep_end()
    ob = [float(b) for b in config["omega_bound"]]
    for l in range(L):
        sig2, om, _ = (float(v) for v in np.exp(xs[l]))  # (np.exp on the vector, as the reference: its bits)
        # gp.py:91-92: `not np.any(np.isclose(om, omega_bound))`, i.e. |om - b| <= atol + rtol |b| with NumPy's defaults
        # (the very predicate for finite scalars; np.isclose itself costs ~25 us a call, five times per H-step)
        if not any(abs(om - b) <= 1e-8 + 1e-5 * abs(b) for b in ob):
            omega[l] = om
        sigma[l] = math.sqrt(sig2)
    params["sigma"] = sigma
    params["omega"] = omega
    make_cholesky(trials, params, config)
