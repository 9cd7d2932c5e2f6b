"""Relative error of the H-step objective (ll, dll) against the reference's golden vectors (tests/golden/hstep.npz)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlgp_amd.engine as V
g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "hstep.npz"))
M, T, L = g["mu"].shape
units = [{"y": np.zeros((T, 2)), "mu": g["mu"][m].copy(), "w": g["w"][m].copy(), "v": np.zeros((T, L))} for m in range(M)]
with V.Engine(2, L, 1, 50) as eng:
    eng.upload(0, units)
    lat = np.repeat(np.arange(L), len(g["logp"]))
    logp = np.tile(g["logp"], (L, 1))
    ll, dll = eng.hstep_objective(0, T, 1.0, lat, logp)
ll = ll.reshape(L, -1); dll = dll.reshape(L, -1, 3)
e1 = np.max(np.abs(ll - g["ll"]) / np.abs(g["ll"]))
e2 = np.max(np.abs(dll[:, :, 1] - g["dll"][:, :, 1]) / np.maximum(np.abs(g["dll"][:, :, 1]), 1e-3 * np.abs(g["ll"])))
print("H-step objective vs golden: max rel err ll %.2e, dll %.2e (M=%d segments, T=%d)" % (e1, e2, M, T))
# K-block cost: with 8 segments the round kernel is its K blocks (one per evaluation) plus the launch
import time
with V.Engine(2, L, 1, 50) as eng:
    eng.upload(0, units)
    eng.hstep_begin(0, T)
    for n_eval in (1, 5):
        lat1 = np.arange(n_eval) % L
        lp1 = np.tile(g["logp"][1], (n_eval, 1))
        for _ in range(10):
            eng.hstep_objective(0, T, 1.0, lat1, lp1)
        eng.profile(True); eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(200):
            eng.hstep_objective(0, T, 1.0, lat1, lp1)
        wall = (time.perf_counter() - t0) / 200
        n, ms, _u = eng.profile_get(2)
        eng.profile(False)
        print("K blocks only (M = 8), n_eval %d: kernel %.1f us, wall per call %.1f us" % (n_eval, 1e3 * ms / n, 1e6 * wall))
    eng.hstep_end()
