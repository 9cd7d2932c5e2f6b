"""E / M / H wall time of every EM iteration of the bench workload next to the ranks of the factor each E-step used."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
n = int(os.environ.get("STEPS", "25"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=n, min_iter=n)
ranks = []
for _ in range(n):
    ranks.append([int(r) for r in sess.eng.prior_ranks(sess.config["window"])])
    sess.em_iteration()
rt = sess.runtime
for i in range(n):
    print("it %2d ranks %-22s E %.2f M %.2f H %.2f  total %.2f ms" % (i, ranks[i], 1e3 * rt["e_elapsed"][i], 1e3 * rt["m_elapsed"][i],
                                                                1e3 * rt["h_elapsed"][i], 1e3 * rt["em_elapsed"][i]))
sess.close()
