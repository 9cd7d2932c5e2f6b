"""Seeded synthetic Lorenz/Poisson trials (the workload of BASELINE.json).

Recipe (after the reference's tutorial notebook,
``notebook/tutorial.ipynb:143-146,209-216``): latents are a z-scored forward
Euler integration of the Lorenz system cut into trials; for more than three
latents, independent squared-exponential GP draws are appended; counts are
Poisson with log-rate ``z a + log(0.1)`` capped at 3.  Gaussian channels (used
by the mixed-likelihood configuration) observe ``z a + N(0, 0.5^2)``.

This is the build's own generator; its Lorenz integrator is checked once
against ``vlgp/simulation.py:108-151`` in ``tests/golden/gen_golden.py``.
"""
import numpy as np

__all__ = ["lorenz_path", "make_trials", "CONFIGS"]

# name -> (n_trials, n_bins, n_channels, n_latents)
CONFIGS = {
    "C1": (10, 200, 20, 3),
    "C2": (50, 500, 50, 3),
    "C3": (200, 1000, 100, 5),
}


def lorenz_path(n, dt=5e-3, s=10.0, r=28.0, b=2.667, x0=(0.0, 1.0, 1.05)):
    """Forward-Euler Lorenz trajectory, shape (n, 3)."""
    out = np.empty((n, 3))
    px, py, pz = (float(c) for c in x0)
    out[0] = (px, py, pz)
    for i in range(1, n):
        dx = s * (py - px)
        dy = r * px - py - px * pz
        dz = px * py - b * pz
        px, py, pz = px + dx * dt, py + dy * dt, pz + dz * dt
        out[i] = (px, py, pz)
    return out


def _zscore(a, axis=0):
    return (a - a.mean(axis=axis, keepdims=True)) / a.std(axis=axis, keepdims=True)


def _latents(n_trials, n_bins, n_latents, skip=500):
    x0 = np.random.random(3)
    path = lorenz_path(skip + n_trials * n_bins, x0=x0)[skip:]
    z = _zscore(path).reshape(n_trials, n_bins, 3)
    if n_latents <= 3:
        return z[:, :, :n_latents]
    # extra latents: SE-GP draws (omega = 5e-3), one per trial and dimension
    t = np.arange(n_bins)
    K = np.exp(-5e-3 * (t[:, None] - t[None, :]) ** 2) + 1e-6 * np.eye(n_bins)
    C = np.linalg.cholesky(K)
    extra = np.einsum("ts,msl->mtl", C, np.random.randn(n_trials, n_bins, n_latents - 3))
    extra = _zscore(extra.reshape(-1, n_latents - 3)).reshape(n_trials, n_bins, -1)
    return np.concatenate([z, extra], axis=2)


def make_trials(n_trials, n_bins, n_channels, n_latents, seed=0, n_gauss=0,
                lengths=None, return_truth=False):
    """List of ``{"ID": i, "y": (T_i, N) float64}`` trials.

    ``n_gauss`` trailing channels are Gaussian; ``lengths`` (optional, one per
    trial) gives unequal trial lengths (each <= n_bins).
    """
    np.random.seed(seed)
    z = _latents(n_trials, n_bins, n_latents)
    np.random.seed(seed)
    a = 0.5 * (np.random.rand(n_latents, n_channels) + 1.0) * np.sign(
        np.random.randn(n_latents, n_channels))
    bias = np.log(0.1)
    n_pois = n_channels - n_gauss
    trials = []
    for i in range(n_trials):
        T = n_bins if lengths is None else int(lengths[i])
        lin = z[i, :T] @ a
        y = np.empty((T, n_channels))
        y[:, :n_pois] = np.random.poisson(np.exp(np.minimum(lin[:, :n_pois] + bias, 3.0)))
        if n_gauss:
            y[:, n_pois:] = lin[:, n_pois:] + 0.5 * np.random.randn(T, n_gauss)
        trials.append({"ID": i, "y": y})
    if return_truth:
        return trials, {"z": z, "a": a, "b": bias}
    return trials
