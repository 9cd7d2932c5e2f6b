"""vlgp_amd/csrc/np_exact.h -- the arithmetic the device ichol kernel is made of -- compiled for the
CPU (tests/native/np_exact_harness.cpp, g++) and compared BITWISE with the live NumPy / OpenBLAS
and with the reference's golden factors.  No GPU needed: this pins the restatement of np.exp,
np.dot and np.sum that makes the device pivots identical to the reference's (vlgp/math.py:101-126).
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from oracle import vlgp_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def npx():
    out = os.path.join(tempfile.mkdtemp(prefix="npx_"), "libnpx.so")
    subprocess.run(["g++", "-O2", "-mfma", "-ffp-contract=off", "-std=c++17", "-shared", "-fPIC",
                    os.path.join(HERE, "native", "np_exact_harness.cpp"), "-o", out], check=True)
    lib = C.CDLL(out)
    lib.npx_sum_array.restype = C.c_double
    return lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _live_stack_mismatch():
    """np_exact.h restates ONE host stack's arithmetic: NumPy's AVX-512 SVML exp and OpenBLAS's Haswell / SkylakeX /
    Zen ddot / dgemv_t kernels (all four share the accumulator shapes restated).  Returns a reason string when the
    live NumPy / BLAS of this host is another stack -- the comparisons against the LIVE libraries are then
    meaningless and skipped; the golden-vector test below still holds the header to the reference's own factors."""
    try:
        from numpy._core._multiarray_umath import __cpu_features__ as feats
    except Exception:  # pragma: no cover
        return "cannot read NumPy's CPU feature table"
    if not (feats.get("AVX512F") and feats.get("AVX512_SKX")):
        return "NumPy dispatches exp without AVX-512 (no SVML exp8) on this CPU"
    try:
        from threadpoolctl import threadpool_info
        blas = [d for d in threadpool_info() if d.get("user_api") == "blas"]
    except Exception:  # pragma: no cover
        return "threadpoolctl unavailable: BLAS kernel family unknown"
    if not blas or blas[0].get("internal_api") != "openblas":
        return "NumPy is not linked against OpenBLAS"
    if blas[0].get("architecture") not in ("SkylakeX", "Haswell", "Zen", "Cooperlake", "Sapphirerapids"):
        return "OpenBLAS core %r has other ddot/dgemv kernels" % blas[0].get("architecture")
    return None


live = pytest.mark.skipif(_live_stack_mismatch() is not None, reason=str(_live_stack_mismatch()))


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float64).view(np.uint64)


def _ichol(lib, n, omega, r):
    G = np.zeros((n, r))
    piv = np.zeros(n, dtype=np.int32)
    k = lib.npx_ichol_gauss(int(n), C.c_double(float(omega)), int(r), _p(G), _p(piv))
    return G, piv, k


@live
def test_exp_matches_numpy_bitwise(npx):
    rng = np.random.default_rng(0)
    parts = [-rng.uniform(0, 50, 400_000), -rng.uniform(0, 760, 400_000), rng.uniform(-1e-3, 1e-3, 100_000),
             rng.uniform(0, 709.9, 100_000), -np.exp(rng.uniform(-40, 7, 200_000)),
             np.array([0.0, -0.0, -745.2, -745.13, -746.0, -1e4, -np.inf, 1e-300, -1e-300, -708.396, -708.4,
                       -707.7, -707.69])]
    for om in (5e-2, 5e-3, 5e-4, 1.0, 0.0123456789):  # the arguments ichol_gauss actually forms
        parts.append(-om * np.arange(2000.0) ** 2)
    x = np.concatenate(parts)
    y = np.empty_like(x)
    npx.npx_exp_array(_p(x), _p(y), C.c_long(x.size))
    with np.errstate(over="ignore"):
        want = np.exp(x)
    assert np.array_equal(_bits(y), _bits(want))


@live
def test_sum_matches_numpy_bitwise(npx):
    rng = np.random.default_rng(1)
    for n in list(range(0, 300)) + [496, 504, 999, 1000, 1001, 1999, 2000, 5000]:
        a = rng.uniform(0, 1, n)
        assert npx.npx_sum_array(_p(a), n) == np.sum(a), n


@live
def test_dot_matches_openblas_bitwise(npx):
    # every (outputs, length) shape class of G[i+1:, :i] @ G[i, :i] (math.py:117), incl. the single-row ddot case
    rng = np.random.default_rng(2)
    for k in range(0, 64):
        G = np.ascontiguousarray(rng.standard_normal((1100, 64)))
        for mo in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 14, 15, 49, 150, 999):
            A, xv = G[10:10 + mo, :k], np.ascontiguousarray(G[5, :k])
            out = np.empty(mo)
            npx.npx_dot_array(_p(G[10:]), 64, _p(xv), mo, k, _p(out))
            assert np.array_equal(_bits(out), _bits(np.dot(A, xv))), (k, mo)


def test_ichol_golden_bitwise(npx, golden):
    g = golden("ichol")
    for i, (n, om, r) in enumerate(g["cases"]):
        G, _, _ = _ichol(npx, int(n), om, int(r))
        if n > 200:  # rank-exhausted 1000-bin factors: the fixture holds every 8th row and the column sums
            assert np.array_equal(G[::8], g["G%d_rows" % i])
            assert np.array_equal(G.sum(axis=0), g["G%d_colsum" % i])
        else:
            assert np.array_equal(G, g["G%d" % i])


@live
def test_ichol_random_cases_bitwise_vs_oracle(npx):
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = int(rng.integers(1, 260))
        r = int(rng.choice([3, 20, 50, 64, 130]))
        om = float(np.exp(rng.uniform(np.log(1e-5), np.log(10))))
        G, _, _ = _ichol(npx, n, om, r)
        assert np.array_equal(G, O.ichol_gauss(n, om, r)), (n, om, r)
    # full-rank windows (single-row ddot step), the reference's own test case, long rank-exhausted trials
    for n, om, r in [(50, 1.0, 50), (50, 5.0, 50), (51, 2.0, 50), (34, 0.7, 50), (500, 1.0, 500),
                     (2000, 5e-4, 50), (1503, 3e-3, 50)]:
        G, _, _ = _ichol(npx, n, om, r)
        assert np.array_equal(G, O.ichol_gauss(n, om, r)), (n, om, r)
