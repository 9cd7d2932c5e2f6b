"""Split E-step launch times against the number of rows: does the time follow rows / 64 waves smoothly or in steps of
one wave per SIMD (1024 SIMDs x 64 lanes = 65536 rows)?  Also the M-step Newton launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E, _lib
from vlgp_amd.api import FitSession
for n_trials in [int(s) for s in os.environ.get("TRIALS", "131,160,190,196,197,200,210,230,262").split(",")]:
    bench.WORKLOADS["X"] = (n_trials, 1000, 100, 5)
    trials, a0, b0, dims = bench.build_inputs("X")
    sess = FitSession(trials, 5, verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
    for _ in range(6):
        sess.em_iteration()
    eng = sess.eng
    eng.profile(True); eng.profile_reset()
    for _ in range(4):
        E.estep(sess.segs, sess.params, sess.config)
    eng.synchronize()
    out = []
    for k in (_lib.PROF_ESTEP, _lib.PROF_ESTEP_PASS, _lib.PROF_ESTEP_FACTOR, _lib.PROF_ESTEP_MEAN):
        n, ms, units = eng.profile_get(k)
        out.append(1e3 * ms / max(n, 1))
    eng.profile_reset()
    E.mstep(sess.segs, sess.params, sess.config)
    eng.synchronize()
    n, ms, units = eng.profile_get(_lib.PROF_MSTEP)
    ranks = eng.get_prior(50, with_rank=True)[1].tolist()
    rows = n_trials * 1000
    print("rows %7d (%.3f waves/SIMD)  E-step %.0f us  pass %.1f us  factor %.1f us  mean %.1f us  M-step launch %.1f us (n=%d)  ranks %s"
          % (rows, rows / 65536.0, out[0], out[1], out[2], out[3], 1e3 * ms / max(n, 1), n, ranks), flush=True)
    sess.close()
