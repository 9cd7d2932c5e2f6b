"""ctypes binding of libvlgp_hip.so (C ABI: include/vlgp_hip.h).

There is no CPU fallback: if the shared library is missing, or no GPU is
visible when a handle is created, this raises.
"""
import ctypes as C
import os

import numpy as np

# the host driver of this pool only supports dmabuf IPC: without this RCCL's (and any cross-process
# device-memory sharing's) hipIpcGetMemHandle fails; it must be in the environment before HIP starts
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvlgp_hip.so")

ABI_VERSION = 1
MAX_SETS = 4
UNIQUE_ID_BYTES = 128
PROF_ESTEP, PROF_MSTEP, PROF_HSTEP, PROF_PRIOR = 0, 1, 2, 3
PROF_ESTEP_RA16, PROF_ESTEP_RA24, PROF_ESTEP_RA32, PROF_ESTEP_LONG, PROF_ESTEP_GENERIC = 4, 5, 6, 7, 8
PROF_ESTEP_PASS, PROF_ESTEP_FACTOR, PROF_ESTEP_MEAN = 9, 10, 11  # split E-step, sampled launches
PROF_HSTEP_LR, PROF_HSTEP_TAB = 12, 13  # low-rank H-step round and its tables kernel
ESTEP_PATHS = ("none", "split", "fast", "long", "generic", "long_split", "split_mixed")  # VLGP_PATH_ESTEP_*
HSTEP_PATHS = ("none", "lowrank", "dense", "big", "generic", "old", "mixed")  # VLGP_PATH_HSTEP_*

_lib = None


class VlgpError(RuntimeError):
    """An entry point of libvlgp_hip.so returned a non-zero status."""


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_i64p = C.POINTER(C.c_int64)
_u8p = C.POINTER(C.c_uint8)
_h = C.c_void_p

_SIGNATURES = {
    "vlgp_abi_version": (C.c_int, []),
    "vlgp_device_count": (C.c_int, [_ip]),
    "vlgp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, C.POINTER(_h)]),
    "vlgp_destroy": (C.c_int, [_h]),
    "vlgp_last_error": (C.c_char_p, [_h]),
    "vlgp_synchronize": (C.c_int, [_h]),
    "vlgp_synchronize_main": (C.c_int, [_h]),
    "vlgp_upload_units": (C.c_int, [_h, C.c_int, C.c_int, _i64p, _dp, _dp, _dp, _dp, _dp]),
    "vlgp_cut_units": (C.c_int, [_h, C.c_int, C.c_int, C.c_int, _i64p, C.c_int]),
    "vlgp_merge_units": (C.c_int, [_h, C.c_int]),
    "vlgp_download_units": (C.c_int, [_h, C.c_int, _dp, _dp, _dp, _dp]),
    "vlgp_stash_mu": (C.c_int, [_h, C.c_int, C.c_int]),
    "vlgp_free_units": (C.c_int, [_h, C.c_int]),
    "vlgp_set_params": (C.c_int, [_h, _dp, _dp, _dp]),
    "vlgp_get_params": (C.c_int, [_h, _dp, _dp, _dp, _dp, _dp]),
    "vlgp_build_prior": (C.c_int, [_h, C.c_int, _ip, _dp, _dp]),
    "vlgp_set_prior": (C.c_int, [_h, C.c_int, _dp]),
    "vlgp_clear_prior": (C.c_int, [_h]),
    "vlgp_get_prior": (C.c_int, [_h, C.c_int, _dp, _ip]),
    "vlgp_update_w": (C.c_int, [_h, C.c_int]),
    "vlgp_update_v": (C.c_int, [_h, C.c_int, C.c_int, _ip]),
    "vlgp_estep": (C.c_int, [_h, C.c_int, C.c_int, C.c_double, C.c_int, _ip]),
    "vlgp_estep_wait": (C.c_int, [_h]),
    "vlgp_mstep": (C.c_int, [_h, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                             C.c_double, _ip]),
    "vlgp_mstep_begin": (C.c_int, [_h, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                   C.c_double]),
    "vlgp_mstep_end": (C.c_int, [_h, _ip, _dp]),
    "vlgp_hstep_objective": (C.c_int, [_h, C.c_int, C.c_int, C.c_double, C.c_int, _ip, _dp, _dp, _dp]),
    "vlgp_sample_posterior": (C.c_int, [_h, C.c_int, _dp, _dp, _dp, C.c_int, _dp, _dp, _ip]),
    "vlgp_comm_host_exchange": (C.c_int, [_h]),
    "vlgp_comm_transport": (C.c_int, [_h]),
    "vlgp_comm_rccl_ranks": (C.c_int, [_h, _ip, _ip]),
    "vlgp_project_units": (C.c_int, [_h, C.c_int, _dp, _dp, _dp]),
    "vlgp_hstep_begin": (C.c_int, [_h, C.c_int, C.c_int]),
    "vlgp_hstep_prepare": (C.c_int, [_h, C.c_int, C.c_int]),
    "vlgp_hstep_end": (C.c_int, [_h]),
    "vlgp_apply_latent_map": (C.c_int, [_h, C.c_int, _dp, _dp]),
    "vlgp_set_overlaps": (C.c_int, [_h, C.c_int, C.c_int, _ip, C.c_int, _ip, _ip]),
    "vlgp_unshare_mu": (C.c_int, [_h, C.c_int]),
    "vlgp_norms": (C.c_int, [_h, C.c_int, _dp]),
    "vlgp_norms_begin": (C.c_int, [_h, C.c_int]),
    "vlgp_norms_end": (C.c_int, [_h, _dp]),
    "vlgp_latent_moments": (C.c_int, [_h, C.c_int, _dp, _dp, _dp]),
    "vlgp_comm_unique_id": (C.c_int, [C.c_char_p]),
    "vlgp_comm_init": (C.c_int, [_h, C.c_char_p, C.c_int, C.c_int]),
    "vlgp_comm_init_aux": (C.c_int, [_h, C.c_char_p]),
    "vlgp_comm_allreduce_host": (C.c_int, [_h, _dp, C.c_int]),
    "vlgp_profile_enable": (C.c_int, [_h, C.c_int]),
    "vlgp_profile_reset": (C.c_int, [_h]),
    "vlgp_profile_get": (C.c_int, [_h, C.c_int, _i64p, _dp, _dp]),
    "vlgp_debug_phase_clock": (C.c_int, [_h, C.c_int, C.POINTER(C.c_uint64)]),
    "vlgp_debug_npx": (C.c_int, [_h, C.c_int, C.c_int64, _dp, _dp, _dp]),
    "vlgp_debug_last_estep_path": (C.c_int, [_h, _ip]),
    "vlgp_debug_last_hstep_path": (C.c_int, [_h, _ip]),
    "vlgp_debug_reload_switches": (C.c_int, [_h]),
    "vlgp_debug_hstep_stats": (C.c_int, [_h, _dp]),
}
EXPORTS = tuple(_SIGNATURES)


def load():
    """Load the shared library once; raise ImportError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("VLGP_LIB_PATH", LIB_PATH)  # (A/B runs of two builds on one box; never set in production)
    if not os.path.exists(path):
        raise ImportError(
            "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C vlgp_amd/csrc` (hipcc, gfx950). There is no CPU fallback." % path)
    lib = C.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.vlgp_abi_version() != ABI_VERSION:
        raise ImportError("libvlgp_hip.so ABI %d != binding ABI %d" % (lib.vlgp_abi_version(), ABI_VERSION))
    _lib = lib
    return lib


def dptr(arr):
    """double* of a C-contiguous float64 array (None -> NULL)."""
    if arr is None:
        return None
    assert arr.dtype == np.float64 and arr.flags["C_CONTIGUOUS"], "need C-contiguous float64"
    return arr.ctypes.data_as(_dp)


def iptr(arr):
    if arr is None:
        return None
    assert arr.dtype == np.int32 and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(_ip)


def i64ptr(arr):
    assert arr.dtype == np.int64 and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(_i64p)


def u8ptr(arr):
    assert arr.dtype == np.uint8 and arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(_u8p)


def check(rc, handle=None):
    if rc != 0:
        msg = load().vlgp_last_error(handle)
        raise VlgpError("libvlgp_hip status %d: %s" % (rc, (msg or b"").decode(errors="replace")))


def device_count():
    n = C.c_int(0)
    rc = load().vlgp_device_count(C.byref(n))
    return n.value if rc == 0 else 0
