"""E-step phase anatomy at C3 (thread-0 cycle counters summed over workgroups)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
names = ["staging", "ya+factor0", "residual pass", "mean update", "curvature pass", "factor+variance (all)", "  of which build I+GtWG", "  of which chol+inv"]
for it in range(4):
    sess.eng.phase_clock(True)
    sess.em_iteration()
    clk = sess.eng.phase_clock(True)
    tot = sum(clk[:6])
    M = len(sess.segs)
    print("iter", it, "E-step ms %.2f" % (1e3 * sess.runtime["e_elapsed"][-1]), "ranks", sess.eng.get_prior(50, with_rank=True)[1].tolist(),
          " | ".join("%s %.0f%% (%.0f cyc/wg/iter)" % (n, 100.0 * c / tot, c / M / 25) for n, c in zip(names, clk)))
sess.close()
