"""Host side of the MI355X vEM engine.

Two layers:

* :class:`Engine` -- one opaque handle of ``libvlgp_hip.so`` per GPU; NumPy in,
  NumPy out, every heavy operation is a HIP kernel behind the C ABI.
* the reference's inner seam, with the reference's names, argument meaning
  and in-place mutation semantics (vlgp/core.py, vlgp/gp.py)::

      estep | mstep | hstep | update_w | update_v | make_cholesky | infer
      | constrain_loading | constrain_latent | vem   (trials, params, config) -> None

  Called on plain lists of trial dicts they upload, run and download (that is
  what the stage-wise parity tests do); called on a :class:`DeviceTrials` --
  what :func:`vlgp_amd.fit` builds -- the data stays resident on the GPU and
  nothing crosses PCIe between stages.
"""
import ctypes as C
import logging
import os
import time

import numpy as np

from . import _lib
from ._lib import VlgpError, check, dptr, i64ptr, iptr, u8ptr

logger = logging.getLogger(__name__)

__all__ = ["Engine", "DeviceTrials", "estep", "mstep", "hstep", "update_w", "update_v",
           "make_cholesky", "infer", "vem", "constrain_loading", "constrain_latent", "VlgpError"]


# which kernel family ran the most recent seam call on a temporary engine (the engine is gone by the time the
# caller could ask it): "estep" / "update_w" / "update_v" -> one of _lib.ESTEP_PATHS.  Read by the parity tests.
TRACE = {}


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _all_ones(x):
    """x == 1 everywhere; O(1) for the zero-stride view preprocess.initialize hands out."""
    x = np.asarray(x)
    if x.size and all(st == 0 for st in x.strides):
        return bool(x.flat[0] == 1.0)
    return bool(np.all(x == 1.0))


class Engine:
    """One GPU, one handle (include/vlgp_hip.h)."""

    def __init__(self, n_channels, n_latents, xdim, rank, gauss_mask=None, device=0):
        self._hbuf = None
        self.lib = _lib.load()
        self.N, self.L, self.P, self.R = int(n_channels), int(n_latents), int(xdim), int(rank)
        mask = np.zeros(self.N, dtype=np.uint8) if gauss_mask is None else \
            np.ascontiguousarray(np.asarray(gauss_mask, dtype=bool).astype(np.uint8))
        if mask.shape != (self.N,):
            raise ValueError("gauss_mask must have one entry per channel")
        self.gauss = mask.astype(bool)
        h = C.c_void_p()
        rc = self.lib.vlgp_create(int(device), self.N, self.L, self.P, self.R, u8ptr(mask), C.byref(h))
        if rc != 0:
            check(rc, None)
        self.h = h
        self.sets = {}  # set id -> (M, rows, offsets)
        self.state_epoch = 0  # bumped by every call that can change a set's mu (em_iteration's norm cache keys on it)
        self.rank, self.world = 0, 1
        self.host_exchange = False

    # -- lifetime ---------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.vlgp_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        check(rc, self.h)

    def check(self, rc):
        """Raise VlgpError with the handle's message for a non-zero status returned outside this class."""
        check(rc, self.h)

    @property
    def handle_address(self):
        """The vlgp_ctx* as an integer (for native callers of the C ABI: vlgp_amd._lockstep)."""
        return int(self.h.value)

    @property
    def hstep_objective_address(self):
        """Address of vlgp_hstep_objective in the loaded library."""
        return int(C.cast(self.lib.vlgp_hstep_objective, C.c_void_p).value)

    def synchronize(self, main_only=False):
        """Wait for the device: everything, or (main_only) the main stream without a pending M-step lane."""
        self._ck(self.lib.vlgp_synchronize_main(self.h) if main_only else self.lib.vlgp_synchronize(self.h))

    # -- unit sets --------------------------------------------------------
    def upload(self, set_id, trials):
        """Pack a list of trial dicts (keys y, x, mu, v, w) into set ``set_id``."""
        self.state_epoch += 1
        lengths = np.array([tr["y"].shape[0] for tr in trials], dtype=np.int64)
        off = np.zeros(len(trials) + 1, dtype=np.int64)
        np.cumsum(lengths, out=off[1:])
        y = _f64(np.concatenate([tr["y"] for tr in trials], axis=0))
        if y.shape[1] != self.N:
            raise ValueError("trial y has %d channels, engine was built for %d" % (y.shape[1], self.N))
        x = None
        if any(tr.get("x") is not None for tr in trials):
            given = [tr.get("x") for tr in trials]
            for tr, xi in zip(trials, given):
                if xi is not None and tuple(xi.shape) != (tr["y"].shape[0], self.P, self.N):
                    raise ValueError("trial x must be (T, %d, %d)" % (self.P, self.N))
            # the default regressors of preprocess.initialize are x == 1: detected per trial, never packed
            if not (self.P == 1 and all(xi is None or _all_ones(xi) for xi in given)):
                xs = [xi if xi is not None else np.ones((tr["y"].shape[0], self.P, self.N))
                      for tr, xi in zip(trials, given)]
                x = _f64(np.concatenate(xs, axis=0))
        elif self.P != 1:
            raise ValueError("trials without x need xdim == 1")

        def cat(key):
            if any(tr.get(key) is None for tr in trials):
                return None
            arr = _f64(np.concatenate([tr[key] for tr in trials], axis=0))
            if arr.shape != (y.shape[0], self.L):
                raise ValueError("trial %s must be (T, %d)" % (key, self.L))
            return arr

        mu, v, w = cat("mu"), cat("v"), cat("w")
        self._ck(self.lib.vlgp_upload_units(self.h, set_id, len(trials), i64ptr(off), dptr(y), dptr(x),
                                            dptr(mu), dptr(v), dptr(w)))
        self.sets[set_id] = (len(trials), int(off[-1]), off)

    def cut(self, src, dst, starts, window):
        self.state_epoch += 1
        starts = np.ascontiguousarray(starts, dtype=np.int64)
        self._ck(self.lib.vlgp_cut_units(self.h, src, dst, len(starts), i64ptr(starts), int(window)))
        off = np.arange(len(starts) + 1, dtype=np.int64) * int(window)
        self.sets[dst] = (len(starts), int(off[-1]), off)

    def set_overlaps(self, set_id, stage_start, links, link_start):
        """Overlapping units of a copied cut: see vlgp_set_overlaps (include/vlgp_hip.h)."""
        stage_start = np.ascontiguousarray(stage_start, dtype=np.int32)
        links = np.ascontiguousarray(links, dtype=np.int32).reshape(-1, 3)
        link_start = np.ascontiguousarray(link_start, dtype=np.int32)
        self._ck(self.lib.vlgp_set_overlaps(self.h, set_id, len(stage_start) - 1, iptr(stage_start), len(links),
                                            iptr(links), iptr(link_start)))

    def unshare_mu(self, set_id):
        self._ck(self.lib.vlgp_unshare_mu(self.h, set_id))

    def merge(self, cut_set):
        self.state_epoch += 1
        self._ck(self.lib.vlgp_merge_units(self.h, cut_set))

    def download(self, set_id, keys=("mu", "v", "w", "dmu")):
        """dict key -> (rows, L) array for the requested keys."""
        _, rows, _ = self.sets[set_id]
        out = {k: np.empty((rows, self.L)) for k in keys}
        self._ck(self.lib.vlgp_download_units(self.h, set_id, dptr(out.get("mu")), dptr(out.get("v")),
                                              dptr(out.get("w")), dptr(out.get("dmu"))))
        return out

    def stash_mu(self, set_id, restore=False):
        """Keep (or write back) a device-side copy of the set's mu (see vlgp_stash_mu)."""
        self.state_epoch += 1
        self._ck(self.lib.vlgp_stash_mu(self.h, int(set_id), int(bool(restore))))

    def free_units(self, set_id):
        self.state_epoch += 1
        self._ck(self.lib.vlgp_free_units(self.h, set_id))
        self.sets.pop(set_id, None)

    # -- parameters -------------------------------------------------------
    def set_params(self, a, b, noise):
        a, b, noise = _f64(a), _f64(b), _f64(noise)
        if a.shape != (self.L, self.N) or b.shape != (self.P, self.N) or noise.shape != (self.N,):
            raise ValueError("parameter shapes must be a (L,N), b (P,N), noise (N)")
        self._ck(self.lib.vlgp_set_params(self.h, dptr(a), dptr(b), dptr(noise)))

    def set_loading(self, a):
        """Replace the loading matrix only (constrain_loading rescales a and leaves b, noise alone)."""
        a = _f64(a)
        if a.shape != (self.L, self.N):
            raise ValueError("loading matrix must be (L, N)")
        self._ck(self.lib.vlgp_set_params(self.h, dptr(a), None, None))

    def get_params(self):
        a, da = np.empty((self.L, self.N)), np.empty((self.L, self.N))
        b, db = np.empty((self.P, self.N)), np.empty((self.P, self.N))
        noise = np.empty(self.N)
        self._ck(self.lib.vlgp_get_params(self.h, dptr(a), dptr(b), dptr(noise), dptr(da), dptr(db)))
        return a, b, noise, da, db

    # -- prior ------------------------------------------------------------
    def build_prior(self, lengths, omega, sigma):
        lengths = np.ascontiguousarray(np.unique(np.asarray(lengths)) if len(lengths) > 1 else lengths, dtype=np.int32)
        omega, sigma = _f64(omega), _f64(sigma)
        self._ck(self.lib.vlgp_build_prior(self.h, len(lengths), iptr(lengths), dptr(omega), dptr(sigma)))

    def set_prior(self, T, G):
        G = _f64(G)
        if G.shape != (self.L, int(T), self.R):
            raise ValueError("prior factor for length %d must be (%d, %d, %d), got %r"
                             % (T, self.L, T, self.R, G.shape))
        self._ck(self.lib.vlgp_set_prior(self.h, int(T), dptr(G)))

    def clear_prior(self):
        self._ck(self.lib.vlgp_clear_prior(self.h))

    def prior_ranks(self, T):
        """Effective rank (number of columns ichol_gauss built) per latent for length T; no device traffic."""
        rk = np.zeros(self.L, dtype=np.int32)
        self._ck(self.lib.vlgp_get_prior(self.h, int(T), None, iptr(rk)))
        return rk

    def get_prior(self, T, with_rank=False):
        G = np.empty((self.L, int(T), self.R))
        rk = np.zeros(self.L, dtype=np.int32)
        self._ck(self.lib.vlgp_get_prior(self.h, int(T), dptr(G), iptr(rk)))
        return (G, rk) if with_rank else G

    # -- E / M / H --------------------------------------------------------
    def update_w(self, set_id):
        self.state_epoch += 1
        self._ck(self.lib.vlgp_update_w(self.h, set_id))

    def update_v(self, set_id, vb=True, count=True):
        self.state_epoch += 1
        n = C.c_int(0)
        self._ck(self.lib.vlgp_update_v(self.h, set_id, int(bool(vb)), C.byref(n) if count else None))
        return n.value

    def estep(self, set_id, n_iter, dmu_bound=5.0, vb=True, count=True):
        self.state_epoch += 1
        n = C.c_int(0)
        self._ck(self.lib.vlgp_estep(self.h, set_id, int(n_iter), float(dmu_bound), int(bool(vb)),
                                     C.byref(n) if count else None))
        return n.value

    def estep_wait(self):
        """Wait for the most recent estep call's launches (not for work queued behind them since)."""
        self._ck(self.lib.vlgp_estep_wait(self.h))

    def mstep(self, set_id, n_iter, use_hessian=True, eps=1e-8, learning_rate=1.0, da_bound=5.0,
              db_bound=5.0, count=True):
        n = C.c_int(0)
        self._ck(self.lib.vlgp_mstep(self.h, set_id, int(n_iter), int(bool(use_hessian)), float(eps),
                                     float(learning_rate), float(da_bound), float(db_bound),
                                     C.byref(n) if count else None))
        return n.value

    def mstep_begin(self, set_id, n_iter, use_hessian=True, eps=1e-8, learning_rate=1.0, da_bound=5.0,
                    db_bound=5.0):
        """Enqueue the M-step on the engine's second stream and return at once."""
        self._ck(self.lib.vlgp_mstep_begin(self.h, set_id, int(n_iter), int(bool(use_hessian)), float(eps),
                                           float(learning_rate), float(da_bound), float(db_bound)))

    def mstep_end(self):
        """Wait for the M-step begun earlier; returns (n_failed, device milliseconds)."""
        n, ms = C.c_int(0), C.c_double(0)
        self._ck(self.lib.vlgp_mstep_end(self.h, C.byref(n), C.byref(ms)))
        return n.value, ms.value

    def hstep_objective(self, set_id, window, dt, latents, logp, copy=True):
        """Batched (ll, dll) for evaluations (latents[e], logp[e, :3]).  copy=False hands out views of the call's
        persistent result buffers (valid until the next call)."""
        n = len(latents)
        if n > 16:  # a call takes at most 16 evaluations (more than 16 latents in lock-step): in slices
            logp = np.reshape(logp, (n, 3))
            parts = [self.hstep_objective(set_id, window, dt, latents[i:i + 16], logp[i:i + 16]) for i in range(0, n, 16)]
            return np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts])
        hb = self._hbuf
        if hb is None or hb[0] < n:  # persistent argument buffers: this is called ~40 times per EM iteration
            cap = max(16, n)
            arrs = (np.zeros(cap, dtype=np.int32), np.zeros((cap, 3)), np.zeros(cap), np.zeros((cap, 3)))
            hb = self._hbuf = (cap, arrs, (iptr(arrs[0]), dptr(arrs[1]), dptr(arrs[2]), dptr(arrs[3])))
        (lat, lp, ll, dll), ptrs = hb[1], hb[2]
        lat[:n] = latents
        lp[:n] = logp if getattr(logp, "shape", None) == (n, 3) else np.reshape(logp, (n, 3))
        self._ck(self.lib.vlgp_hstep_objective(self.h, set_id, int(window), float(dt), n, *ptrs))
        if copy:
            return ll[:n].copy(), dll[:n].copy()
        return ll[:n], dll[:n]

    def project_latent(self, set_id, proj, shift):
        """mu = y @ proj - shift on the device for every row of the set; returns the column sums of y."""
        self.state_epoch += 1
        proj = _f64(proj)
        shift = _f64(shift)
        if proj.shape != (self.N, self.L) or shift.shape != (self.L,):
            raise ValueError("projection must be (N, L) with an (L,) shift")
        colsum = np.empty(self.N)
        self._ck(self.lib.vlgp_project_units(self.h, set_id, dptr(proj), dptr(shift), dptr(colsum)))
        return colsum

    def hstep_prepare(self, set_id, window):
        """Enqueue now what the H-step bracket builds from the units alone (see vlgp_hstep_prepare)."""
        self._ck(self.lib.vlgp_hstep_prepare(self.h, set_id, int(window)))

    def hstep_begin(self, set_id, window):
        """Start of one gp.optimize run: mu, w of the set stay fixed until hstep_end."""
        self._ck(self.lib.vlgp_hstep_begin(self.h, set_id, int(window)))

    def hstep_end(self):
        self._ck(self.lib.vlgp_hstep_end(self.h))

    # -- constraints / norms ----------------------------------------------
    def apply_latent_map(self, set_id, mat, shift=None):
        self.state_epoch += 1
        mat = _f64(mat)
        if mat.shape != (self.L, self.L):
            raise ValueError("latent map must be (L, L)")
        self._ck(self.lib.vlgp_apply_latent_map(self.h, set_id, dptr(mat),
                                                dptr(_f64(shift)) if shift is not None else None))

    def norms(self, set_id):
        out = np.empty(2)
        self._ck(self.lib.vlgp_norms(self.h, set_id, dptr(out)))
        return float(np.sqrt(out[0])), float(np.sqrt(out[1]))

    def norms_begin(self, set_id):
        """Enqueue the norms of mu, dmu (they run beside whatever follows on the main stream); norms_end collects."""
        self._ck(self.lib.vlgp_norms_begin(self.h, set_id))

    def norms_end(self):
        out = np.empty(2)
        self._ck(self.lib.vlgp_norms_end(self.h, dptr(out)))
        return float(np.sqrt(out[0])), float(np.sqrt(out[1]))

    def latent_moments(self, set_id):
        s1, s2 = np.empty(self.L), np.empty(self.L)
        cnt = C.c_double(0)
        self._ck(self.lib.vlgp_latent_moments(self.h, set_id, dptr(s1), dptr(s2), C.byref(cnt)))
        return s1, s2, cnt.value

    # -- multi-GPU --------------------------------------------------------
    def comm_init(self, uid, rank, world, uid_aux=None):
        self._ck(self.lib.vlgp_comm_init(self.h, uid, int(rank), int(world)))
        if uid_aux is not None:
            self._ck(self.lib.vlgp_comm_init_aux(self.h, uid_aux))
        self.rank, self.world = int(rank), int(world)
        self.host_exchange = bool(self.lib.vlgp_comm_host_exchange(self.h))

    @property
    def transport(self):
        """"none" (single rank), "rccl" or "shm" (the opt-in host shared-memory test transport)."""
        return ("none", "rccl", "shm")[int(self.lib.vlgp_comm_transport(self.h))]

    @property
    def rccl_ranks(self):
        """(main lane, M-step lane): the rank counts RCCL reports for the two communicators (ncclCommCount); 0 without."""
        a, b = C.c_int(0), C.c_int(0)
        self._ck(self.lib.vlgp_comm_rccl_ranks(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def allreduce_host(self, arr):
        """In-place sum over ranks of a float64 host array (no-op on one GPU)."""
        if not (isinstance(arr, np.ndarray) and arr.dtype == np.float64 and arr.flags["C_CONTIGUOUS"]):
            raise TypeError("allreduce_host needs a C-contiguous float64 array (the buffer is summed in place as doubles)")
        self._ck(self.lib.vlgp_comm_allreduce_host(self.h, dptr(arr), arr.size))
        return arr

    def barrier(self):
        """Drain this handle's stream and rendezvous with the other ranks."""
        self._ck(self.lib.vlgp_comm_allreduce_host(self.h, None, 0))

    # -- measurement ------------------------------------------------------
    def profile(self, on=True):
        self._ck(self.lib.vlgp_profile_enable(self.h, int(bool(on))))

    def profile_reset(self):
        self._ck(self.lib.vlgp_profile_reset(self.h))

    def phase_clock(self, on=True):
        """Read (and re-arm or disable) the E-step per-phase cycle counters."""
        out = (C.c_uint64 * 8)()
        self._ck(self.lib.vlgp_debug_phase_clock(self.h, int(bool(on)), out))
        return [int(v) for v in out]

    @property
    def last_estep_path(self):
        """Kernel family of the most recent E-step / update_w / update_v: one of _lib.ESTEP_PATHS."""
        p = C.c_int(0)
        self._ck(self.lib.vlgp_debug_last_estep_path(self.h, C.byref(p)))
        return _lib.ESTEP_PATHS[p.value]

    @property
    def last_hstep_path(self):
        """Kernel family of the most recent hstep_objective call: one of _lib.HSTEP_PATHS."""
        p = C.c_int(0)
        self._ck(self.lib.vlgp_debug_last_hstep_path(self.h, C.byref(p)))
        return _lib.HSTEP_PATHS[p.value]

    def reload_switches(self):
        """Read the VLGP_HSTEP_* debug switches from the environment again (they are cached at creation)."""
        self._ck(self.lib.vlgp_debug_reload_switches(self.h))

    def hstep_stats(self):
        """(low-rank evaluations, sum of their predicted ranks, dense evaluations, low-rank rounds re-run densely)."""
        out = np.zeros(4)
        self._ck(self.lib.vlgp_debug_hstep_stats(self.h, dptr(out)))
        return out

    def profile_get(self, kind):
        n, ms, units = C.c_int64(0), C.c_double(0), C.c_double(0)
        self._ck(self.lib.vlgp_profile_get(self.h, int(kind), C.byref(n), C.byref(ms), C.byref(units)))
        return n.value, ms.value, units.value


def unique_id():
    """RCCL unique id (rank 0 creates it, every rank passes it to Engine.comm_init)."""
    buf = C.create_string_buffer(_lib.UNIQUE_ID_BYTES)
    check(_lib.load().vlgp_comm_unique_id(buf), None)
    return buf.raw


# ===========================================================================
# the reference's (trials, params, config) seam
# ===========================================================================
class DeviceTrials(list):
    """A list of trial dicts whose arrays are resident on an :class:`Engine`.

    Behaves as the plain list the reference passes around; the host copies of
    mu/v/w/dmu are refreshed only by :meth:`pull` (callbacks, end of fit).
    """

    def __init__(self, trials, engine, set_id, parent_set=None, unit_of=None):
        super().__init__(trials)
        self.engine = engine
        self.set_id = set_id
        self.parent_set = parent_set  # segments: the set of the trials they were cut from
        self.detached = False         # True once the segments' mu stopped writing through (constrain_loading "svd")
        self.unit_of = unit_of        # overlapping segments are stored stage-major: list index -> unit of the set

    def pull(self, keys=("mu", "v", "w", "dmu")):
        """Copy device state into the dicts: mu, v, dmu in place, w rebound
        (exactly the ownership the reference leaves behind, core.py:117-120)."""
        got = self.engine.download(self.set_id, keys)
        _, _, off = self.engine.sets[self.set_id]
        for i, tr in enumerate(self):
            u = i if self.unit_of is None else int(self.unit_of[i])
            sl = slice(int(off[u]), int(off[u + 1]))
            for k in keys:
                if k == "w" or tr.get(k) is None or tr[k].shape != got[k][sl].shape:
                    tr[k] = got[k][sl].copy()
                else:
                    tr[k][...] = got[k][sl]


def _gauss_mask(params):
    return np.asarray(params["likelihood"]) == "gaussian"


def _push_params(eng, params):
    eng.set_params(params["a"], params["b"], params["noise"])


def _pull_params(eng, params):
    a, b, noise, da, db = eng.get_params()
    params["a"][...] = a          # in place, like core.py:202,220
    params["b"][...] = b
    params["noise"] = noise       # rebound, like core.py:244
    if params.get("da") is None or params["da"].shape != da.shape:
        params["da"], params["db"] = da, db
    else:
        params["da"][...] = da
        params["db"][...] = db


class _Bound:
    """Context that yields (engine, set_id) for a trials argument.

    DeviceTrials: the resident engine, parameters pushed.  Plain list: a
    temporary engine -- upload on enter, download into the dicts on exit.
    """

    def __init__(self, trials, params, need_prior=True, pull=("mu", "v", "w", "dmu")):
        self.trials, self.params, self.need_prior, self.keys = trials, params, need_prior, pull
        self.temp = None

    def __enter__(self):
        tr, p = self.trials, self.params
        if isinstance(tr, DeviceTrials):
            return tr.engine, tr.set_id
        tr = list(tr)
        eng = Engine(p["ydim"], p["zdim"], p["xdim"], p["rank"], _gauss_mask(p))
        self.temp = eng
        try:
            _push_params(eng, p)
            eng.upload(0, tr)
            if self.need_prior:
                chol = p.get("cholesky")  # may be a _LazyPrior: an empty-looking dict that answers __contains__
                chol = {} if chol is None else chol
                for T in sorted({t["y"].shape[0] for t in tr}):
                    if T not in chol:
                        raise KeyError("params['cholesky'] has no factor for length %d "
                                       "(call make_cholesky first)" % T)
                    eng.set_prior(T, chol[T])
        except Exception:
            eng.close()
            raise
        self.view = DeviceTrials(tr, eng, 0)
        return eng, 0

    def __exit__(self, et, ev, tb):
        if self.temp is not None:
            try:
                if et is None and self.keys:
                    self.view.pull(self.keys)
            finally:
                self.temp.close()
        return False


def make_cholesky(trials, params, config=None):
    """gp.make_cholesky (vlgp/gp.py:150-162) with math.ichol_gauss on the GPU.

    ``config["ichol"] == "host"`` selects the host NumPy factorisation instead
    (bit-identical pivots to the reference for bit-identical omega; see
    DESIGN.md, "pivot chaos")."""
    from . import gp as _gp

    if isinstance(trials, DeviceTrials):
        # the unit lengths of a resident set are fixed at upload: the engine holds the offsets
        # (cached per upload: np.unique over the offsets of 4000 segments is ~35 us between the H-step's last round and
        # the prior kernel, every EM iteration)
        entry = trials.engine.sets[trials.set_id]
        cached = getattr(trials, "_length_cache", None)
        if cached is None or cached[0] is not entry:
            cached = (entry, sorted({int(t) for t in np.unique(np.diff(np.asarray(entry[2])))}))
            trials._length_cache = cached
        lengths = cached[1]
    else:
        lengths = sorted({int(tr["y"].shape[0]) for tr in trials})
    mode = (config or {}).get("ichol", "device")
    if mode == "host":
        chol = {T: np.stack([_gp.ichol_gauss_host(T, params["omega"][l], params["rank"]) * params["sigma"][l]
                             for l in range(params["zdim"])]) for T in lengths}
        params["cholesky"] = chol
        if isinstance(trials, DeviceTrials):
            trials.engine.clear_prior()
            for T, G in chol.items():
                trials.engine.set_prior(T, G)
        return
    if isinstance(trials, DeviceTrials):
        trials.engine.build_prior(lengths, params["omega"], params["sigma"])
        params["cholesky"] = _LazyPrior(trials.engine, lengths)
        return
    with Engine(params["ydim"], params["zdim"], params["xdim"], params["rank"], _gauss_mask(params)) as eng:
        eng.build_prior(lengths, params["omega"], params["sigma"])
        params["cholesky"] = {T: eng.get_prior(T) for T in lengths}


class _LazyPrior(dict):
    """params["cholesky"] while the factors live on the device: downloads G on
    first access of a length (keeps the EM loop free of PCIe traffic)."""

    def __init__(self, engine, lengths):
        super().__init__()
        self._engine, self._lengths = engine, set(int(t) for t in lengths)

    def __missing__(self, T):
        if int(T) not in self._lengths:
            raise KeyError(T)
        G = self._engine.get_prior(int(T))
        self[int(T)] = G
        return G

    def __contains__(self, T):
        return int(T) in self._lengths

    def get(self, T, default=None):  # (dict.get bypasses __missing__)
        return self[T] if T in self else default

    def materialize(self):
        return {T: np.array(self[T]) for T in sorted(self._lengths)}

    def __reduce__(self):  # a result file must load without vlgp_amd (util.save pickles the dict): a plain dict
        return (dict, (self.materialize(),))


class _InitialPrior(dict):
    """params["initial"]["cholesky"] of a fit without a window (vlgp/api.py:60 deep-copies the full-length factors of
    the INITIAL omega, sigma): the device's copies are overwritten by the first H-step and downloading every
    full-length factor up front costs PCIe time nobody may want, so a length is factored on first access -- on the
    host, in math.ichol_gauss's own operation order (gp.ichol_gauss_host), from the initial hyperparameters kept here."""

    def __init__(self, lengths, omega, sigma, rank):
        super().__init__()
        self._lengths = sorted(int(t) for t in lengths)
        self._omega, self._sigma, self._rank = np.array(omega, dtype=float), np.array(sigma, dtype=float), int(rank)

    def __missing__(self, T):
        from . import gp as _gp

        if int(T) not in self._lengths:
            raise KeyError(T)
        G = np.stack([_gp.ichol_gauss_host(int(T), self._omega[l], self._rank) * self._sigma[l]
                      for l in range(len(self._omega))])
        self[int(T)] = G
        return G

    def __contains__(self, T):
        return int(T) in self._lengths

    def __iter__(self):
        return iter(self._lengths)

    def __len__(self):
        return len(self._lengths)

    def keys(self):
        return list(self._lengths)

    def items(self):
        return [(T, self[T]) for T in self._lengths]

    def values(self):
        return [self[T] for T in self._lengths]

    def get(self, T, default=None):  # (dict.get bypasses __missing__)
        return self[T] if T in self else default

    def __deepcopy__(self, memo):
        return _InitialPrior(self._lengths, self._omega, self._sigma, self._rank)

    def __reduce__(self):  # pickled (util.save) as the plain {T: ndarray} dict the reference's load expects
        return (dict, (dict(self.items()),))


def update_w(trials, params, config=None):
    """core.update_w (vlgp/core.py:419-442)."""
    for tr in trials:  # the reference creates missing w/v (core.py:433-434)
        if not isinstance(trials, DeviceTrials):
            tr.setdefault("w", np.zeros_like(tr["mu"]))
            tr.setdefault("v", np.zeros_like(tr["mu"]))
    with _Bound(trials, params, need_prior=False, pull=("w",)) as (eng, sid):
        eng.update_w(sid)
        TRACE["update_w"] = eng.last_estep_path


def update_v(trials, params, config):
    """core.update_v (vlgp/core.py:445-471)."""
    if config["method"] != "VB":
        return
    with _Bound(trials, params, pull=("v",)) as (eng, sid):
        bad = eng.update_v(sid, True, count=not isinstance(trials, DeviceTrials))
        TRACE["update_v"] = eng.last_estep_path
        if bad:
            logger.error("Singular I + G'WG in %d unit-latent pairs", bad)


def estep(trials, params, config):
    """core.estep (vlgp/core.py:123-126): config["Eniter"] inner iterations per unit."""
    if config["Eniter"] < 1:
        return
    with _Bound(trials, params) as (eng, sid):
        bad = eng.estep(sid, config["Eniter"], config["dmu_bound"], config["method"] == "VB",
                        count=not isinstance(trials, DeviceTrials))
        TRACE["estep"] = eng.last_estep_path
        if bad:
            logger.error("%d posterior updates hit a singular system and were zeroed", bad)


def mstep(trials, params, config):
    """core.mstep (vlgp/core.py:129-249)."""
    if config["Mniter"] < 1:
        return
    if params.get("da") is None:
        params["da"] = np.zeros_like(params["a"])
        params["db"] = np.zeros_like(params["b"])
    with _Bound(trials, params, need_prior=False, pull=()) as (eng, sid):
        bad = eng.mstep(sid, config["Mniter"], config["use_hessian"], config["eps"],
                        config["learning_rate"], config["da_bound"], config["db_bound"],
                        count=not isinstance(trials, DeviceTrials))
        if bad:
            logger.error("%d Newton systems were singular (gradient step taken)", bad)
        _pull_params(eng, params)


def hstep(trials, params, config):
    """core.hstep (vlgp/core.py:252-257)."""
    if not config["Hstep"]:
        return
    from . import gp as _gp

    _gp.optimize(trials, params, config)


def infer(trials, params, config, echo=None):
    """core.infer (vlgp/core.py:260-266): E-step with Eniter := max_iter."""
    keep = config["Eniter"]
    config["Eniter"] = config["max_iter"]
    t0 = time.perf_counter()
    try:
        estep(trials, params, config)
        if isinstance(trials, DeviceTrials):
            trials.engine.synchronize()
    finally:
        config["Eniter"] = keep
    if echo:
        echo("{:.2f}s".format(time.perf_counter() - t0))


def constrain_loading(trials, params, config):
    """core.constrain_loading (vlgp/core.py:392-416)."""
    kind = config["constrain_loading"]
    if not kind or kind == "none":
        return
    a = params["a"]
    L = a.shape[0]
    if kind == "svd":
        _, _, vt = np.linalg.svd(a, full_matrices=False)
        mat = a @ vt.T
        params["a"] = vt
    else:
        if kind == "fro":
            s = np.full(L, np.linalg.norm(a, ord="fro") + config["eps"])
        else:
            s = np.linalg.norm(a, ord=kind, axis=1) + config["eps"]
        params["a"] /= s[:, None]
        mat = np.diag(s)
    if isinstance(trials, DeviceTrials):
        if kind == "svd" and getattr(trials, "parent_set", None) is not None and not trials.detached:
            # the reference REBINDS every segment's mu here (core.py:407-408): from now on the segments no longer
            # write through to their trials, whose mu stays what it is at this moment (fit's final inference
            # starts from it).  Keep that copy; FitSession.finish puts it back after the merge.
            trials.engine.stash_mu(trials.parent_set)
            trials.detached = True
            if trials.unit_of is not None:  # overlapping segments stop sharing their mu rows (v stays shared)
                trials.engine.unshare_mu(trials.set_id)
        trials.engine.apply_latent_map(trials.set_id, mat)
        trials.engine.set_loading(params["a"])  # b and noise are untouched by this constraint
    else:
        for tr in trials:
            tr["mu"] = tr["mu"] @ mat if kind == "svd" else _imul_cols(tr["mu"], np.diag(mat))


def _imul_cols(mu, s):
    mu *= s
    return mu


def constrain_latent(trials, params, config):
    """core.constrain_latent (vlgp/core.py:366-389); off by default."""
    kind = config["constrain_latent"]
    if not kind or kind == "none":
        return
    L = params["zdim"]
    if isinstance(trials, DeviceTrials):
        s1, s2, cnt = trials.engine.latent_moments(trials.set_id)
        mean = s1 / cnt
        std = np.sqrt(np.maximum(s2 / cnt - mean ** 2, 0.0))
    else:
        mu = np.concatenate([tr["mu"] for tr in trials], axis=0)
        mean, std = mu.mean(axis=0), mu.std(axis=0)
    shift = np.zeros(L)
    scale = np.ones(L)
    if kind in ("location", "both"):
        shift = mean
        params["b"][0, :] += mean @ params["a"]
    if kind in ("scale", "both"):
        scale = 1.0 / std
        params["a"] *= std[:, None]
    if isinstance(trials, DeviceTrials):
        if trials.unit_of is not None and kind == "both":
            # overlapping segments: the reference shifts every segment, THEN scales every segment (core.py:377-388); a
            # shared row sees each pass twice, so the two passes cannot be folded into one affine map
            trials.engine.apply_latent_map(trials.set_id, np.eye(L), shift)
            trials.engine.apply_latent_map(trials.set_id, np.diag(scale))
        else:
            trials.engine.apply_latent_map(trials.set_id, np.diag(scale), shift)
        _push_params(trials.engine, params)
    else:
        for tr in trials:
            tr["mu"] -= shift
            tr["mu"] *= scale


def new_runtime():
    return {"it": 0, "e_elapsed": [], "m_elapsed": [], "h_elapsed": [], "em_elapsed": []}


def em_iteration(trials, params, config, runtime, echo=None):
    """One pass of the body of core.vem (vlgp/core.py:298-357): E, M, H, timers,
    callbacks, convergence test.  Returns True when the stopping rule fires."""
    eng, sid = trials.engine, trials.set_id
    tol = config["tol"]
    runtime["it"] += 1
    # pre-iteration norm of mu (core.py:300-305): the value the previous iteration's closing norms call returned,
    # when nothing has touched the set since (one kernel + one device -> host copy less per iteration)
    cached = getattr(trials, "_norm_cache", None)
    if cached is not None and cached[0] == (eng.state_epoch, sid):
        norm_mu = cached[1]
    else:
        norm_mu, _ = eng.norms(sid)
    trials._norm_cache = None
    norm_a = np.linalg.norm(params["a"])
    norm_b = np.linalg.norm(params["b"])

    t0 = time.perf_counter()
    constrain_loading(trials, params, config)
    estep(trials, params, config)
    # M and H are independent given the posterior (M: a, b from mu, v; H: omega from
    # mu, w): the M-step is enqueued on the engine's second stream and runs under the
    # H-step's host-driven rounds.  m_elapsed is the M-step's device time, h_elapsed
    # the wall time of the H-step, em_elapsed the wall time of the whole iteration.
    m_ms = 0.0
    m_async = config["Mniter"] >= 1
    latent_constraint = bool(config["constrain_latent"]) and config["constrain_latent"] != "none"

    def begin_m():
        if params.get("da") is None:
            params["da"] = np.zeros_like(params["a"])
            params["db"] = np.zeros_like(params["b"])
        eng.mstep_begin(sid, config["Mniter"], config["use_hessian"], config["eps"], config["learning_rate"],
                        config["da_bound"], config["db_bound"])

    # the M-step lane waits for the E-step on the device (an event): its ~50 launches are enqueued (one graph launch,
    # ~80 us of host time) while the E-step still runs, unless constrain_latent has to touch mu, a, b in between
    early_m = m_async and not latent_constraint
    if early_m:
        begin_m()
    m_first = m_async and ((eng.world > 1 and not eng.host_exchange) or bool(os.environ.get("VLGP_M_SEQUENTIAL")))

    # mu, dmu and w are final once the E-step (and constrain_latent) are done -- the M-step writes a, b; the H-step omega:
    # the sums of the stopping rule (core.py:350-354) and the H-step's moments of mu / copy of w are queued behind the
    # E-step while it still runs: nothing the host has to do between the E-step's end and the first round
    prepare = bool(config["Hstep"]) and bool(config.get("window")) and not m_first

    def after_estep():
        eng.norms_begin(sid)
        if prepare:
            eng.hstep_prepare(sid, config["window"])

    if not latent_constraint:
        after_estep()
    eng.estep_wait()   # the E-step alone: neither the M-step lane nor what is queued behind it
    t1 = time.perf_counter()
    constrain_latent(trials, params, config)
    if latent_constraint:
        after_estep()
    norms_epoch = eng.state_epoch
    if m_async and not early_m:
        begin_m()

    def finish_m():
        bad, ms = eng.mstep_end()
        if bad:
            logger.error("%d Newton systems were singular (gradient step taken)", bad)
        _pull_params(eng, params)
        return ms

    # With several ranks the M-step lane issues RCCL all-reduces on its own communicator.  That is safe
    # under the H-step only when the H-step issues none itself, i.e. when its round sums are exchanged on
    # the host (Engine.host_exchange); otherwise two communicators would be in flight with no common
    # order across ranks, and the M-step is finished first.
    if m_first:
        m_ms = finish_m()
        m_async = False
    t2 = time.perf_counter()
    hstep(trials, params, config)  # every objective round ends with a device -> host copy: no extra sync
    t3 = time.perf_counter()
    if m_async:
        m_ms = finish_m()
    t4 = time.perf_counter()

    runtime["e_elapsed"].append(t1 - t0)
    runtime["m_elapsed"].append(m_ms * 1e-3)
    runtime["h_elapsed"].append(t3 - t2)
    runtime["em_elapsed"].append(t4 - t0)
    config["runtime"] = runtime
    if echo:
        echo("Iteration {:4d}, E-step {:.2f}s, M-step {:.2f}s".format(
            runtime["it"], runtime["e_elapsed"][-1], runtime["m_elapsed"][-1]))

    norm_mu_now, norm_dmu = eng.norms_end()
    if config["callbacks"]:
        trials.pull()
        for cb in config["callbacks"]:
            try:
                cb(trials, params, config)
            except RuntimeError:
                logger.error("Callback {} failed".format(cb))
        if eng.state_epoch != norms_epoch:  # a callback changed the units: the reference takes the norms after it
            norm_mu_now, norm_dmu = eng.norms(sid)
    trials._norm_cache = ((eng.state_epoch, sid), norm_mu_now)
    converged = (norm_dmu < tol * norm_mu
                 and np.linalg.norm(params["da"]) < tol * norm_a
                 and np.linalg.norm(params["db"]) < tol * norm_b)
    return bool(converged and runtime["it"] >= config["min_iter"])


def vem(trials, params, config, echo=None):
    """core.vem (vlgp/core.py:269-359): the EM loop, timers and stopping rule.

    Timers are wall-clock around device-synchronised phases, so
    ``config["runtime"]`` means what it means in the reference.
    """
    if not isinstance(trials, DeviceTrials):
        raise TypeError("vem runs on DeviceTrials (use vlgp_amd.fit, or Engine + DeviceTrials)")
    runtime = new_runtime()
    _push_params(trials.engine, params)
    for _ in range(config["max_iter"]):
        if em_iteration(trials, params, config, runtime, echo):
            break
