"""Per-iteration phase times, effective ranks and omega over a longer C3 run."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
n = int(os.environ.get("ITERS", "40"))
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=n, min_iter=n)
for it in range(n):
    sess.em_iteration()
    rt = sess.runtime
    ranks = sess.eng.get_prior(50, with_rank=True)[1].tolist()
    print("it %2d  E %.2f M %.2f H %.2f ms  ranks %s  omega %s" % (it, 1e3 * rt["e_elapsed"][-1], 1e3 * rt["m_elapsed"][-1],
          1e3 * rt["h_elapsed"][-1], ranks, np.array2string(np.asarray(sess.params["omega"]), precision=4)))
sess.close()
