"""Host time of one H-step at C3 outside the objective calls (SciPy's setulb + the lock-step driver), and inside them
outside the kernels' span (argument staging, ctypes, two launches, mailbox)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
for _ in range(8):
    sess.em_iteration()
eng = sess.eng
orig = eng.hstep_objective
acc = [0.0, 0]
def timed(*a, **kw):
    t = time.perf_counter()
    r = orig(*a, **kw)
    acc[0] += time.perf_counter() - t
    acc[1] += 1
    return r
eng.hstep_objective = timed
tot = 0.0
for _ in range(10):
    E.estep(sess.segs, sess.params, sess.config)
    eng.synchronize() if hasattr(eng, "synchronize") else None
    t = time.perf_counter()
    E.hstep(sess.segs, sess.params, sess.config)
    tot += time.perf_counter() - t
n = acc[1]
if n == 0:  # the native driver (vlgp_amd._lockstep) calls the C ABI by address: nothing to intercept
    print("H-step %.3f ms with the native lock-step driver (VLGP_LOCKSTEP_PYTHON=1 for the split of the Python driver)" % (tot / 10 * 1e3))
    sess.close()
    sys.exit(0)
print("H-step %.3f ms, %d rounds each; in objective calls %.1f us per round, outside %.1f us per round (+ fixed part)"
      % (tot / 10 * 1e3, n // 10, acc[0] / n * 1e6, (tot - acc[0]) / n * 1e6))
sess.close()
