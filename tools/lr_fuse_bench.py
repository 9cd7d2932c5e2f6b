"""Low-rank rounds of C3 with the tables from their own launch against the tables factored by every round workgroup
(VLGP_HSTEP_FUSE_TABLES=1): wall time per round for 1 ... 5 evaluations, and the results compared bit for bit."""
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
oms = [float(x) for x in os.environ.get("OMS", "0.006,0.004,0.003,0.005,0.001").split(",")]
os.environ["VLGP_HSTEP_LOWRANK"] = "1"
res = {}
for n in (1, 2, 5):
    lat = np.arange(n, dtype=np.int32)
    logp = np.log(np.array([[1.0, om, 1e-4] for om in oms[:n]]))
    for fuse in (False, True, False, True):
        if fuse: os.environ["VLGP_HSTEP_FUSE_TABLES"] = "1"
        else: os.environ.pop("VLGP_HSTEP_FUSE_TABLES", None)
        eng.reload_switches()
        for _ in range(5):
            eng.hstep_objective(sid, 50, 1.0, lat, logp)
        eng.synchronize()
        t0 = time.perf_counter()
        for _ in range(200):
            ll, dll = eng.hstep_objective(sid, 50, 1.0, lat, logp)
        wall = (time.perf_counter() - t0) / 200
        key = (n, fuse)
        if key in res:
            assert np.array_equal(res[key][0], ll) and np.array_equal(res[key][1], dll)
        res[key] = (ll.copy(), dll.copy())
        print("%d evaluations, %s (%s): %.1f us per round" % (n, "fused " if fuse else "tables", eng.last_hstep_path, 1e6 * wall))
    same = np.array_equal(res[(n, False)][0], res[(n, True)][0]) and np.array_equal(res[(n, False)][1], res[(n, True)][1])
    print("   identical results: %s   (max rel diff ll %.2e)" % (same, np.max(np.abs(res[(n, False)][0] - res[(n, True)][0]) / np.abs(res[(n, False)][0]))))
eng.hstep_end(); sess.close()
