"""Cycles per phase of ONE workgroup of the low-rank round (first segment block), per omega; single evaluation."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
eng = sess.eng
sess.em_iteration()
sid = sess.segs.set_id
eng.hstep_begin(sid, 50)
for om in (1e-3, 4e-3, 8e-3, 1.3e-2):
    for ne in (1, 15):
        lat = np.arange(ne, dtype=np.int32) % dims[3]
        logp = np.log(np.array([[1.0, om, 1e-4]] * ne))
        for _ in range(3):
            eng.hstep_objective(sid, 50, 1.0, lat, logp)
        eng.phase_clock(True)
        n = 20
        for _ in range(n):
            eng.hstep_objective(sid, 50, 1.0, lat, logp)
        c = eng.phase_clock(False)
        print("omega %.1e n_eval %2d: cycles/100 (clock64 = 100 MHz? units raw) setup+weights %d  phase1 %d  phase2 %d  phase3+reduce %d"
              % (om, ne, c[0] / n, c[1] / n, c[2] / n, c[3] / n))
eng.hstep_end(); sess.close()
