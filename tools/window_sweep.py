import os, sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs("C3")
for window in [int(w) for w in os.environ.get("WINDOWS", "40,50,100").split(",")]:
    # BINS: truncate the trials so that the window tiles them (e.g. BINS=896 = 16 x 56 = 14 x 64); a window that does not
    # tile the trials runs the staged E-step of overlapping segments (the reference's view semantics), another regime
    nb = int(os.environ.get("BINS", "0")) or trials[0]["y"].shape[0]
    tr = [{"ID": t["ID"], "y": t["y"][:nb].copy(), "mu": t["mu"][:nb].copy()} for t in trials]
    sess = FitSession(tr, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=8, min_iter=8, window=window)
    for _ in range(6):
        sess.em_iteration()
    rt = sess.runtime
    print("window %3d: E %.1f M %.1f H %.1f ms (iteration 6), segments %d" % (window, 1e3 * rt["e_elapsed"][-1], 1e3 * rt["m_elapsed"][-1], 1e3 * rt["h_elapsed"][-1], len(sess.segs)))
    sess.close()
