"""Where Engine.upload spends its time at C3 (160 MB of y): concatenation, the x == 1 test, the C call (allocation +
host -> device copies)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlgp_amd import synth, engine as E
from vlgp_amd.preprocess import get_config, get_params, initialize, fill_params, fill_trials
import bench
n_trials, n_bins, N, L = bench.WORKLOADS["C3"]
trials = synth.make_trials(n_trials, n_bins, N, L, seed=0)
np.random.seed(0)
config = get_config()
params = get_params(trials, L, omega_bound=config["omega_bound"])
initialize(trials, params, config)
fill_params(params); fill_trials(trials)
eng = E.Engine(N, L, 1, 50)
for rep in range(3):
    t0 = time.perf_counter()
    y = np.concatenate([tr["y"] for tr in trials], axis=0)
    t1 = time.perf_counter()
    ones = all(E._all_ones(tr["x"]) for tr in trials if tr.get("x") is not None)
    t2 = time.perf_counter()
    eng.upload(0, trials)
    t3 = time.perf_counter()
    print("concatenate y %.1f ms | x == 1 test %.1f ms (%s) | Engine.upload in all %.1f ms" % (
        1e3 * (t1 - t0), 1e3 * (t2 - t1), ones, 1e3 * (t3 - t2)))
eng.close()
