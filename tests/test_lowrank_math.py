"""The algebra behind the low-rank H-step round (vlgp_amd/csrc/hstep_lr.h), checked on the CPU against the dense
formulas of gp.py:12-43,126-147: for A = I + S K S, K = sigma^2 exp(-omega D^2) + eps I,
    tr(A^-1)  and  sum_jk s_j s_k dK_jk (A^-1)_jk  ( = d log det A / d ln omega )
from an even / odd folded pivoted Cholesky K_s = U U' with its tangent dU / d ln omega, M = I + U' diag(w~) U and a
symmetric Gauss-Jordan inverse of M -- the exact operations of the kernel, in NumPy (tools/lr_proto.py)."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("lr_proto", os.path.join(ROOT, "tools", "lr_proto.py"))
P = importlib.util.module_from_spec(spec)
spec.loader.exec_module(P)


def _dense(t, sigmasq, omega, eps, w):
    T = len(t)
    D2 = (t[:, None] - t[None, :]) ** 2
    Ks = sigmasq * np.exp(-omega * D2)
    dK = -omega * D2 * Ks
    s = np.sqrt(w)
    A = np.eye(T) + s[:, None] * (Ks + eps * np.eye(T)) * s[None, :]
    Ai = np.linalg.inv(A)
    return np.trace(Ai), np.sum(s[:, None] * s[None, :] * dK * Ai)


@pytest.mark.parametrize("T", [4, 7, 24, 33, 50, 64])
@pytest.mark.parametrize("omega", [5e-4, 4e-3, 1.5e-2])
def test_folded_woodbury_terms_equal_the_dense_ones(T, omega):
    rng = np.random.default_rng(T)
    t = np.arange(T, dtype=float)
    U, Ud, re, ro = P.tables(T, 1.0, 0.7, omega, 1e-12)
    assert re + ro <= T
    for trial in range(3):
        w = rng.uniform(0.01, 1.0, T) * 10.0 ** rng.uniform(-1, 1.5)
        tr0, cs0 = _dense(t, 0.7, omega, 1e-4, w)
        tr1, cs1 = P.seg_terms_eo(U, Ud, re, ro, T, 1e-4, w)
        assert abs(tr1 - tr0) <= 1e-10 * abs(tr0), (T, omega, trial)
        assert abs(cs1 - cs0) <= 1e-9 * max(abs(cs0), 1e-3 * abs(tr0)), (T, omega, trial)


def test_rank_grows_with_omega_and_stays_below_the_window():
    ranks = [sum(P.tables(50, 1.0, 1.0, om, 1e-12)[2:]) for om in (5e-4, 2e-3, 8e-3, 2e-2, 5e-2)]
    assert ranks == sorted(ranks) and ranks[0] <= 12 and ranks[-1] <= 50
    assert 20 <= ranks[2] <= 26  # omega = 8e-3: the steady state of C3 sits in the rank-24 class


def test_gauss_jordan_sweeps_invert_spd_matrices():
    rng = np.random.default_rng(0)
    for r in (1, 5, 16, 27):
        B = rng.standard_normal((40, r))
        M = np.eye(r) + B.T @ B
        assert np.abs(P.sweep_inv(M) @ M - np.eye(r)).max() < 1e-11
