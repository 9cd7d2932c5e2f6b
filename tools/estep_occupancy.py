"""E-step fast-kernel time against the number of segments (512 / 768 / 1024 / ...): how many workgroups does a CU hold at once?"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E, _lib
from vlgp_amd.api import FitSession
for n_trials in (13, 26, 39, 52, 64, 77, 103, 200):
    bench.WORKLOADS["X"] = (n_trials, 1000, 100, 5)
    trials, a0, b0, dims = bench.build_inputs("X")
    sess = FitSession(trials, 5, verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
    for _ in range(5):
        sess.em_iteration()
    eng = sess.eng
    eng.profile(True); eng.profile_reset()
    for _ in range(3):
        E.estep(sess.segs, sess.params, sess.config)
    eng.synchronize()
    n, ms, units = eng.profile_get(_lib.PROF_ESTEP)
    ranks = eng.get_prior(50, with_rank=True)[1].tolist()
    print("segments %5d  E-step kernel %.3f ms  (%.2f us per segment)  ranks %s" % (len(sess.segs), ms / n, 1e3 * ms / n / len(sess.segs), ranks))
    sess.close()
