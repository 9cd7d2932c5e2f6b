"""M-step alone at a workload (no H-step beside it): wall per call.  VLGP_MSTEP_WG_PER_CU selects the workgroups per CU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
for _ in range(4):
    sess.em_iteration()
E.estep(sess.segs, sess.params, sess.config)
for _ in range(2):
    E.mstep(sess.segs, sess.params, sess.config)
sess.eng.synchronize()
t = time.perf_counter()
for _ in range(10):
    E.mstep(sess.segs, sess.params, sess.config)
sess.eng.synchronize()
print("M-step alone %.3f ms (VLGP_MSTEP_WG_PER_CU=%s)" % ((time.perf_counter() - t) / 10 * 1e3, os.environ.get("VLGP_MSTEP_WG_PER_CU", "default")))
sess.close()
