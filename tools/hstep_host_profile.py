"""Host-side cost of the H-step's lock-step L-BFGS-B driver (cProfile over a few EM iterations at C3)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs("C3")
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
for _ in range(4):
    sess.em_iteration()
pr = cProfile.Profile()
for _ in range(5):
    E.estep(sess.segs, sess.params, sess.config)
    pr.enable()
    E.hstep(sess.segs, sess.params, sess.config)
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
sess.close()
