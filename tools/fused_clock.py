"""Cycles per phase of the FIRST workgroup of the fused E-step sweeps (estep_fused.h) at C3, every latent at one omega.
    VLGP_LANE_CLOCK=3 OMS=5e-3,1e-2 python tools/fused_clock.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WORKLOAD", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
for _ in range(3):
    sess.em_iteration()
L = dims[3]
names = ["staging", "factor0", "res pass", "mean", "curv pass", "factor", "write-back", "-"]
for om in [float(x) for x in os.environ.get("OMS", "5e-3").split(",")]:
    sess.params["omega"] = np.full(L, om)
    E.make_cholesky(sess.segs, sess.params, sess.config)
    ranks = sess.eng.get_prior(50, with_rank=True)[1].tolist()
    E.estep(sess.segs, sess.params, sess.config)
    sess.eng.synchronize()
    sess.eng.phase_clock(True)
    n = 4
    import time
    t0 = time.perf_counter()
    for _ in range(n):
        E.estep(sess.segs, sess.params, sess.config)
    sess.eng.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    clk = sess.eng.phase_clock(True)
    per = [c / n / (1 if i in (0, 1, 6) else 24) for i, c in enumerate(clk)]
    print("omega %.1e ranks %s path %s  E-step %.3f ms |" % (om, ranks, sess.eng.last_estep_path, ms),
          " | ".join("%s %.0f" % (nm, c) for nm, c in zip(names, per)), "(cycles per phase, per sweep for the four sweep phases)")
sess.close()
