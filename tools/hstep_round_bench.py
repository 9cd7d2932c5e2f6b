"""Cost of one H-step objective round at C3 (5 evaluations, no M-step overlap): wall per call vs kernel time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
eng = sess.eng
for _ in range(2):
    sess.em_iteration()
sid = sess.segs.set_id
L = dims[3]
for n_eval in (1, 3, 5):
    lat = np.arange(n_eval, dtype=np.int32) % L
    logp = np.log(np.array([[1.0, 2e-3 * (1 + 0.1 * i), 1e-4] for i in range(n_eval)]))
    for mode in ("fused", "fused+bracket", "unfused"):
        os.environ.pop("VLGP_HSTEP_UNFUSED", None)
        if mode == "unfused":
            os.environ["VLGP_HSTEP_UNFUSED"] = "1"
        if mode == "fused+bracket":
            eng.hstep_begin(sid, 50)
        for _ in range(5):
            eng.hstep_objective(sid, 50, 1.0, lat, logp)
        eng.synchronize()
        eng.profile(True); eng.profile_reset()
        t0 = time.perf_counter()
        for _ in range(50):
            ll, dll = eng.hstep_objective(sid, 50, 1.0, lat, logp)
        wall = (time.perf_counter() - t0) / 50
        n, ms, units = eng.profile_get(2)
        eng.profile(False)
        t0 = time.perf_counter()
        for _ in range(50):
            eng.hstep_objective(sid, 50, 1.0, lat, logp)
        wall_np = (time.perf_counter() - t0) / 50
        if mode == "fused+bracket":
            eng.hstep_end()
        print("n_eval %d %-14s wall/call %.1f us (no profiling %.1f us)  kernel %.1f us  ll0 %.12e dll0 %.12e"
              % (n_eval, mode, 1e6 * wall, 1e6 * wall_np, 1e3 * ms / max(n, 1), ll[0], dll[0, 1]))
sess.close()
