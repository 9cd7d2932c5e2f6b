"""What the per-launch HIP events of eng.profile(True) cost an EM iteration (bench.py keeps them on in its timed region):
alternating blocks of EM iterations with the events off / on, same session, same box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
n = 46
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=n, min_iter=n)
for _ in range(6):
    sess.em_iteration()
rt = sess.runtime
out = {False: [], True: []}
for blk in range(8):
    on = bool(blk & 1)
    sess.eng.profile(on)
    sess.eng.profile_reset()
    i0 = len(rt["em_elapsed"])
    for _ in range(5):
        sess.em_iteration()
    out[on].append([1e3 * np.mean(rt[k + "_elapsed"][i0:]) for k in ("e", "m", "h", "em")])
for on in (False, True):
    a = np.array(out[on])
    print("events %-3s  E %.3f  M %.3f  H %.3f  EM %.3f ms   (blocks: %s)" % ("on" if on else "off", *a.mean(0), np.round(a[:, 3], 3).tolist()))
sess.close()
