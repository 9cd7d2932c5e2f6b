"""Final full-length inference at C5 (500 ragged trials of 500 .. 2000 bins, 200 mixed channels, ten latents): the
persistent long-unit kernel (VLGP_ESTEP_LSPLIT=0) against the task-parallel launch sequence (=1), ten sweeps."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlgp_amd as V
from vlgp_amd import synth
n_trials = int(os.environ.get("TRIALS", "500"))
rng = np.random.default_rng(0)
lengths = (50 * rng.integers(10, 41, n_trials)).tolist()
L, N = 10, 200
trials = synth.make_trials(n_trials, 2000, N, L, seed=0, n_gauss=50, lengths=lengths)
gauss = np.array([False] * 150 + [True] * 50)
a = 0.2 * rng.standard_normal((L, N))
b = np.zeros((1, N)); b[0, :150] = np.log(0.1)
omega = np.array([2.7e-3, 5e-3, 6.3e-3, 2.8e-3, 4e-3, 4.9e-3, 5.6e-3, 5.3e-3, 5.6e-3, 3.5e-3])
units = [{"y": t["y"], "mu": 0.2 * rng.standard_normal((t["y"].shape[0], L))} for t in trials]
for mode in ("0", "1"):
    os.environ["VLGP_ESTEP_LSPLIT"] = mode
    with V.Engine(N, L, 1, 50, gauss) as eng:
        eng.set_params(a, b, np.ones(N))
        eng.upload(0, units)
        eng.build_prior(sorted(set(lengths)), omega, np.ones(L))
        eng.update_w(0); eng.update_v(0); eng.synchronize()
        ts = []
        for rep in range(3):
            t0 = time.perf_counter(); eng.estep(0, 10, count=False); eng.synchronize(); ts.append(time.perf_counter() - t0)
        print("LSPLIT=%s path %s: %s ms per 10-sweep call (%d rows)" % (mode, eng.last_estep_path, np.round(1e3 * np.array(ts), 1), sum(lengths)))
