"""Round kernel at C3 (4000 segments): low-rank vs dense, per omega (rank class) and number of evaluations."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
eng = sess.eng
for _ in range(2):
    sess.em_iteration()
sid = sess.segs.set_id
L = dims[3]
eng.hstep_begin(sid, 50)
os.environ["VLGP_HSTEP_LOWRANK"] = "1"  # (no size rule: this is what calibrates it)
eng.reload_switches()
def timeit(lat, logp, dense):
    if dense: os.environ["VLGP_HSTEP_DENSE"] = "1"
    else: os.environ.pop("VLGP_HSTEP_DENSE", None)
    eng.reload_switches()  # (cached at vlgp_create)
    for _ in range(5):
        eng.hstep_objective(sid, 50, 1.0, lat, logp)
    eng.synchronize()
    eng.profile(True); eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(50):
        ll, dll = eng.hstep_objective(sid, 50, 1.0, lat, logp)
    wall = (time.perf_counter() - t0) / 50
    n, ms, units = eng.profile_get(2)
    eng.profile(False)
    os.environ.pop("VLGP_HSTEP_DENSE", None)
    return 1e6 * wall, 1e3 * ms / max(n, 1), eng.last_hstep_path, ll, dll
for om in [float(x) for x in os.environ.get("OMS", "1e-3,2e-3,4e-3,6e-3,8e-3,1e-2,1.3e-2,1.6e-2").split(",")]:
    for n_eval in [int(x) for x in os.environ.get("NE", "1,5,15").split(",")]:
        lat = np.arange(n_eval, dtype=np.int32) % L
        logp = np.log(np.array([[1.0, om, 1e-4] for i in range(n_eval)]))
        w1, k1, p1, ll1, dll1 = timeit(lat, logp, False)
        w2, k2, p2, ll2, dll2 = timeit(lat, logp, True)
        print("omega %.1e n_eval %2d  %-8s wall %.1f kernel %.1f us | %-6s wall %.1f kernel %.1f us | dll diff %.1e"
              % (om, n_eval, p1, w1, k1, p2, w2, k2, np.max(np.abs(dll1[:, 1] - dll2[:, 1]) / np.abs(dll2[:, 1]))), flush=True)
eng.hstep_end()
sess.close()
