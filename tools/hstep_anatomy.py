"""Where does an H-step go: device call vs host (scipy + thread hand-offs)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
eng = sess.eng
orig = eng.hstep_objective
acc = {"t": 0.0, "n": 0, "evals": 0}
def timed(*a, **k):
    t0 = time.perf_counter(); r = orig(*a, **k); acc["t"] += time.perf_counter() - t0; acc["n"] += 1; acc["evals"] += len(a[3]); return r
eng.hstep_objective = timed
for it in range(5):
    acc.update(t=0.0, n=0, evals=0)
    sess.em_iteration()
    print("iter", it, "h_elapsed ms %.2f" % (1e3 * sess.runtime["h_elapsed"][-1]), "device calls", acc["n"], "evals", acc["evals"],
          "in-call ms %.2f (%.3f per call)" % (1e3 * acc["t"], 1e3 * acc["t"] / max(acc["n"], 1)))
sess.close()
