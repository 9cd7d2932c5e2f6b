"""Factor-analysis initialisation without scikit-learn.

The reference initialises ``a``, ``noise`` and every trial's ``mu`` with
``sklearn.decomposition.FactorAnalysis(n_components=zdim, random_state=0)``
fitted on a 10 % subsample (vlgp/preprocess.py:14-30).  That estimator is an
EM iteration whose only data-sized operation is a *randomized* truncated SVD
of the rescaled data ``X / (sqrt(psi) sqrt(n))`` (k + 10 Gaussian test vectors
from ``RandomState(0)``, three power iterations, then an exact SVD of the
projected matrix).  On the 20000 x 100 subsample of workload C3 it takes 2.5-7 s
-- ten times the whole EM loop on the GPU -- almost all of it in LAPACK
LU/QR factorizations of tall 20000 x 15 matrices.

Everything the estimator needs from the data is the second-moment matrix
``S = X'X / n``: with ``M = X D^-1 / sqrt(n)`` (``D = diag(sqrt(psi))``),

* the power iteration only ever uses ``M'M = D^-1 S D^-1`` (100 x 100): the
  sampled subspace is ``span((M'M)^3 Omega)`` whichever normaliser is used;
* for an orthonormal basis ``Z`` of that subspace the projected matrix is
  ``B = R^-T Z' M'M`` with ``R'R = Z' M'M Z`` (Cholesky-QR of ``M Z``), and the
  singular values / right singular vectors of ``B`` are those scikit-learn gets
  from ``Q'M``;
* the deterministic sign convention (``svd_flip``: the largest-magnitude entry
  of each *left* singular vector is positive) needs ``U = M Z R^-1 Uhat`` once,
  for the final iterate only, because the noise update uses ``W**2``.

So the fit below draws the same test matrices in the same order, runs the same
EM recursion and stopping rule, and costs one ``X'X`` plus O(n_features^2 k)
per iteration: milliseconds.  Results agree with scikit-learn 1.7 to rounding
(1e-10 relative on ``components_``/``noise_variance_``; pinned by
``tests/test_host_logic.py::test_factor_analysis_matches_sklearn``).
"""
from __future__ import annotations

import math

import numpy as np
import scipy.linalg as sla

__all__ = ["FactorModel", "fit_factor_analysis"]

_SMALL = 1e-12          # floor on psi and guard added to sqrt(psi), as in the estimator being restated
_OVERSAMPLES = 10       # extra test vectors of the randomized range finder
_POWER_ITERATIONS = 3   # FactorAnalysis(iterated_power=3)


class FactorModel:
    """Fitted loading ``components`` (k, n_features), ``noise_variance`` (n_features), ``mean`` (n_features)."""

    def __init__(self, components, noise_variance, mean, loglike):
        self.components = components
        self.noise_variance = noise_variance
        self.mean = mean
        self.loglike = loglike
        self.n_iter = len(loglike)
        wpsi = components / noise_variance
        cov_z = np.linalg.inv(np.eye(components.shape[0]) + wpsi @ components.T)
        self._proj = wpsi.T @ cov_z                 # (n_features, k): posterior-mean map
        self._shift = mean @ self._proj

    @property
    def projection(self):
        """(n_features, k) map P with transform(X) = X @ P - shift."""
        return self._proj

    @property
    def shift(self):
        return self._shift

    def transform(self, X):
        """Posterior mean of the factors, E[z | x] = (x - mean) W'Psi^-1 (I + W Psi^-1 W')^-1."""
        return np.asarray(X, dtype=float) @ self._proj - self._shift


def _projected_svd(C, omega):
    """Top singular pairs of M restricted to span((M'M)^q omega), from C = M'M alone.

    Returns (s, Vt, basis) with ``basis`` (n_features, size) such that the left
    singular vectors are ``U = M @ basis``."""
    Z = omega
    for _ in range(_POWER_ITERATIONS):
        Z, _ = np.linalg.qr(C @ Z)  # re-orthonormalised every pass: the span is what matters
    G = Z.T @ C @ Z
    G = 0.5 * (G + G.T)
    ZC = Z.T @ C
    try:
        R = np.linalg.cholesky(G).T                  # (M Z)'(M Z) = R'R
        B = sla.solve_triangular(R, ZC, trans="T")   # Q'M with Q = M Z R^-1
        Uhat, s, Vt = np.linalg.svd(B, full_matrices=False)
        basis = Z @ sla.solve_triangular(R, Uhat, lower=False)
    except np.linalg.LinAlgError:
        # rank-deficient sample (more test vectors than features, constant channels, ...): orthonormalise
        # M Z through the eigen-decomposition of its Gram matrix and drop the null directions
        lam, V = np.linalg.eigh(np.nan_to_num(G))
        keep = lam > max(lam.max(), 0.0) * 1e-13
        T = V[:, keep] / np.sqrt(lam[keep])
        p = C.shape[0]
        s, Vt, basis = np.zeros(0), np.zeros((0, p)), np.zeros((p, 0))
        if T.shape[1]:
            Uhat, s, Vt = np.linalg.svd(T.T @ ZC, full_matrices=False)
            basis = Z @ (T @ Uhat)
    pad = omega.shape[1] - s.size
    if pad > 0:  # callers index the leading k components
        s = np.concatenate([s, np.zeros(pad)])
        Vt = np.vstack([Vt, np.zeros((pad, Vt.shape[1]))])
        basis = np.hstack([basis, np.zeros((basis.shape[0], pad))])
    return s, Vt, basis


def fit_factor_analysis(X, n_components, seed=0, tol=1e-2, max_iter=1000, allreduce=None, rank=0, world=1):
    """EM factor analysis of the rows of X; the randomized-SVD variant with ``RandomState(seed)``.

    With ``allreduce`` (in-place sum over ranks of a float64 array) X holds only THIS rank's rows of the
    sample: the estimator needs the rows through their count, sum and second-moment matrix alone (plus one
    arg-max for the sign convention), so every rank obtains the fit of the pooled sample."""
    X = np.asarray(X, dtype=float)
    n, p = X.shape
    k = int(n_components)
    if allreduce is None:
        mean = X.mean(axis=0)
        Xc = X - mean
        var = Xc.var(axis=0)
        S = (Xc.T @ Xc) / n
    else:
        mom = np.concatenate([[float(n)], X.sum(axis=0), (X.T @ X).ravel()])
        allreduce(mom)
        n = int(round(mom[0]))
        mean = mom[1:1 + p] / n
        S = mom[1 + p:].reshape(p, p) / n - np.outer(mean, mean)
        S = 0.5 * (S + S.T)
        var = np.diag(S).copy()
        Xc = X - mean
    rng = np.random.RandomState(seed)
    size = k + _OVERSAMPLES
    llconst = p * math.log(2.0 * math.pi) + k
    psi = np.ones(p)
    loglike, old_ll = [], -np.inf
    W = basis = sqrt_psi = None
    for _ in range(max_iter):
        sqrt_psi = np.sqrt(psi) + _SMALL
        C = S / np.outer(sqrt_psi, sqrt_psi)
        omega = rng.normal(size=(p, size))
        s, Vt, basis = _projected_svd(C, omega)
        s2 = s[:k] ** 2
        unexplained = np.trace(C) - s2.sum()
        W = np.sqrt(np.maximum(s2 - 1.0, 0.0))[:, None] * Vt[:k] * sqrt_psi
        with np.errstate(divide="ignore", invalid="ignore"):
            ll = -0.5 * n * (llconst + np.log(s2).sum() + unexplained + np.log(psi).sum())
        loglike.append(ll)
        with np.errstate(invalid="ignore"):
            converged = ll - old_ll < tol
        if converged:
            break
        old_ll = ll
        psi = np.maximum(var - (W ** 2).sum(axis=0), _SMALL)
    # sign convention of the final iterate: largest |entry| of each left singular vector positive
    U = Xc @ ((basis[:, :k] / sqrt_psi[:, None]) / math.sqrt(n))
    if allreduce is None:
        rows = np.argmax(np.abs(U), axis=0)
        signs = np.sign(U[rows, np.arange(k)])
    else:  # the arg-max runs over the rows of every rank: each rank posts its own candidate, lowest rank wins ties
        slots = np.zeros((world, k, 2))
        if U.shape[0]:
            rows = np.argmax(np.abs(U), axis=0)
            slots[rank, :, 0] = np.abs(U[rows, np.arange(k)])
            slots[rank, :, 1] = np.sign(U[rows, np.arange(k)])
        allreduce(slots)
        signs = slots[np.argmax(slots[:, :, 0], axis=0), np.arange(k), 1]
    signs[signs == 0] = 1.0
    return FactorModel(W * signs[:, None], psi, mean, loglike)
