"""E-step wall time at C3 by rank class of the split kernels: every latent's omega set to the same value.
    python tools/estep_rank_classes.py            wall per E-step call
    (under rocprofv3 --kernel-trace --stats: the per-class kernel times)"""
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
oms = [float(x) for x in os.environ.get("OMS", "5e-3,1.6e-2,3e-2,4.5e-2").split(",")]
mixed = os.environ.get("MIXED")  # only latent 0 at the omega, the others at 5e-3
for om in oms:
    sess.params["omega"] = np.full(L, 5e-3) if mixed else np.full(L, om)
    if mixed:
        sess.params["omega"][0] = om
    E.make_cholesky(sess.segs, sess.params, sess.config)
    ranks = sess.eng.get_prior(50, with_rank=True)[1].tolist()
    E.estep(sess.segs, sess.params, sess.config)
    sess.eng.synchronize()
    t = time.perf_counter()
    for _ in range(5):
        E.estep(sess.segs, sess.params, sess.config)
    sess.eng.synchronize()
    print("omega %.1e ranks %s: E-step %.2f ms (%s)" % (om, ranks, (time.perf_counter() - t) / 5 * 1e3, E.TRACE.get("estep")))
sess.close()
