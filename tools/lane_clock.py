"""Cycles per phase of the FIRST wave of the lane-per-task E-step launches (estep_lane.h) at C3, every latent at one omega.
    OMS=5e-3 python tools/lane_clock.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs("C3")
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
for _ in range(3):
    sess.em_iteration()
L = dims[3]
names = {"1": ["staging", "build", "reduction", "chol+inv", "hand-back", "variance", "stores", "-"],
         "2": ["staging", "G's", "reduction", "solve", "hand-back", "expansion", "update", "-"]}[os.environ.get("VLGP_LANE_CLOCK", "1")]
for om in [float(x) for x in os.environ.get("OMS", "5e-3").split(",")]:
    sess.params["omega"] = np.full(L, om)
    E.make_cholesky(sess.segs, sess.params, sess.config)
    ranks = sess.eng.get_prior(50, with_rank=True)[1].tolist()
    E.estep(sess.segs, sess.params, sess.config)
    sess.eng.synchronize()
    sess.eng.phase_clock(True)
    n = 4
    for _ in range(n):
        E.estep(sess.segs, sess.params, sess.config)
    sess.eng.synchronize()
    clk = sess.eng.phase_clock(True)
    lanes = 1 if os.environ.get("VLGP_ESTEP_LANES") == "1" else 2
    launches = n * 25 * lanes
    print("omega %.1e ranks %s" % (om, ranks), " | ".join("%s %.0f" % (nm, c / launches) for nm, c in zip(names, clk)))
sess.close()
