"""Full-length inference at C3 (200 trials x 1000 bins): wall time and phase anatomy of the long-unit E-step kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=10, min_iter=10)
for _ in range(3):
    sess.em_iteration()
eng, params, config = sess.eng, sess.params, sess.config
eng.merge(1)
E.make_cholesky(sess.dev_trials, params, config)
names = ["staging", "ya+factor0", "residual pass", "mean update", "curvature pass", "factor+variance (all)", "  of which build", "  of which chol+inv"]
for rep in range(3):
    eng.synchronize(); t0 = time.perf_counter()
    E.update_w(sess.dev_trials, params, config); eng.synchronize(); t1 = time.perf_counter()
    E.update_v(sess.dev_trials, params, config); eng.synchronize(); t2 = time.perf_counter()
    eng.phase_clock(True)
    E.infer(sess.dev_trials, params, config); eng.synchronize(); t3 = time.perf_counter()
    clk = eng.phase_clock(rep < 2)
    print("update_w %.2f ms  update_v %.2f ms  infer(%d sweeps) %.2f ms" % (1e3*(t1-t0), 1e3*(t2-t1), config["max_iter"], 1e3*(t3-t2)))
    tot = sum(clk[:6]); M = len(trials)
    print(" | ".join("%s %.0f%% (%.0f cyc/sweep)" % (n, 100.0 * c / max(tot, 1), c / M / config["max_iter"]) for n, c in zip(names, clk)))
sess.close()
