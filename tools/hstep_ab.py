"""H-step objective of one fixed state (C3 after one E-step, no H-step before it: the state does not depend on the
round kernel) through the round kernel selected by the environment (default one-set routine, VLGP_HSTEP_TWOSET=1,
VLGP_HSTEP_LEAN=1): repeatability over 20 calls (bitwise) and the values, saved to OUT for comparison across variants."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
eng = sess.eng
E.estep(sess.segs, sess.params, sess.config)
sid = sess.segs.set_id
L = dims[3]
window = int(os.environ.get("WINDOW", "50"))
lat = np.arange(5, dtype=np.int32) % L
logp = np.log(np.array([[1.0 + 0.05 * i, 2e-3 * (1 + 0.7 * i), 1e-4 * (1 + i)] for i in range(5)]))
eng.hstep_begin(sid, window)
ref = None
same = True
for _ in range(20):
    ll, dll = eng.hstep_objective(sid, window, 1.0, lat, logp)
    cur = np.concatenate([ll.ravel(), dll.ravel()])
    if ref is None:
        ref = cur.copy()
    same &= bool(np.array_equal(ref, cur))
eng.hstep_end()
print("repeatable bitwise:", same)
print("ll", ll)
out = os.environ.get("OUT")
if out:
    np.save(out, ref)
sess.close()
