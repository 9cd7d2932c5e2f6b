"""Anatomy of the H-step at C3: rounds per EM iteration by number of evaluations, wall time per round, host time between rounds."""
import os, sys, time, collections
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=40, min_iter=40)
eng = sess.eng
for _ in range(5):
    sess.em_iteration()
rec = []
orig = eng.hstep_objective
def spy(sid, window, dt, latents, logp):
    t0 = time.perf_counter()
    out = orig(sid, window, dt, latents, logp)
    t1 = time.perf_counter()
    rec.append((len(latents), t0, t1))
    return out
eng.hstep_objective = spy
os.environ["VLGP_M_SEQUENTIAL"] = os.environ.get("VLGP_M_SEQUENTIAL", "1")
n_it = 10
for _ in range(n_it):
    sess.em_iteration()
by = collections.defaultdict(list)
gaps = []
for i, (n, t0, t1) in enumerate(rec):
    by[n].append(t1 - t0)
    if i and rec[i][1] - rec[i - 1][2] < 1e-3:
        gaps.append(rec[i][1] - rec[i - 1][2])
print("rounds per EM iteration %.1f, evaluations per EM iteration %.1f" % (len(rec) / n_it, sum(r[0] for r in rec) / n_it))
for n in sorted(by):
    print("n_eval %d: %.1f rounds/iter, call wall %.1f us (min %.1f)" % (n, len(by[n]) / n_it, 1e6 * np.mean(by[n]), 1e6 * np.min(by[n])))
print("host time between consecutive rounds: mean %.1f us, median %.1f us; total per iteration %.2f ms" % (1e6 * np.mean(gaps), 1e6 * np.median(gaps), 1e3 * np.sum(gaps) / n_it))
print("h_elapsed ms", 1e3 * np.mean(sess.runtime["h_elapsed"][-n_it:]), "em", 1e3 * np.mean(sess.runtime["em_elapsed"][-n_it:]))
sess.close()
