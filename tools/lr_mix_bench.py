"""A steady-state C3 round (five latents at their own omegas) through the low-rank and the dense round kernels."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs("C3")
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
eng = sess.eng
for _ in range(2):
    sess.em_iteration()
sid = sess.segs.set_id
eng.hstep_begin(sid, 50)
oms = [0.01329, 0.008278, 0.004389, 0.009787, 0.0010188]
lat = np.arange(5, dtype=np.int32)
logp = np.log(np.array([[1.0, om, 1e-4] for om in oms]))
for dense in (False, True):
    if dense: os.environ["VLGP_HSTEP_DENSE"] = "1"
    else: os.environ.pop("VLGP_HSTEP_DENSE", None)
    eng.reload_switches()  # (cached at vlgp_create)
    for _ in range(5):
        eng.hstep_objective(sid, 50, 1.0, lat, logp)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        ll, dll = eng.hstep_objective(sid, 50, 1.0, lat, logp)
    wall = (time.perf_counter() - t0) / 100
    print("%s: wall per round %.1f us   ll %s" % (eng.last_hstep_path, 1e6 * wall, ll[:2]))
os.environ.pop("VLGP_HSTEP_DENSE", None)
eng.hstep_end(); sess.close()
