"""Wall time of the phases of vlgp_amd.fit at C3 (max_iter = 10): initialisation, upload, EM loop, final inference."""
import os, sys, time, cProfile, pstats
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlgp_amd import synth
from vlgp_amd.api import FitSession
import bench
n_trials, n_bins, N, L = bench.WORKLOADS[os.environ.get("WL", "C3")]
for rep in range(2):
    trials = synth.make_trials(n_trials, n_bins, N, L, seed=0)
    np.random.seed(0)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    sess = FitSession(trials, L, verbose=False, max_iter=10, min_iter=10)
    pr.disable()
    t1 = time.perf_counter()
    sess.run()
    sess.eng.synchronize()
    t2 = time.perf_counter()
    pr2 = cProfile.Profile(); pr2.enable()
    sess.finish()
    pr2.disable()
    t3 = time.perf_counter()
    print("rep %d: init+upload %.1f ms | vem (10 it) %.1f ms | final infer + download %.1f ms" % (rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)))
pstats.Stats(pr).sort_stats("cumtime").print_stats(22)
pstats.Stats(pr2).sort_stats("cumtime").print_stats(14)
