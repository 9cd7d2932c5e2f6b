"""Wall time of the host-visible pieces of one EM iteration (engine.em_iteration), per workload (WL=C1|C2|C3|C3s8):
norms, constrain_loading, E-step, M-step enqueue, H-step (rounds, time inside the device call vs the L-BFGS-B
driver), M-step join + parameter pull.  Median over the iterations after the warm-up."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E
from vlgp_amd.api import FitSession

wl = os.environ.get("WL", "C3")
n_it = int(os.environ.get("ITERS", "12"))
trials, a0, b0, dims = bench.build_inputs(wl)
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=n_it + 4, min_iter=n_it + 4)
eng = sess.eng
acc = {}


def wrap(obj, name, key):
    orig = getattr(obj, name)

    def timed(*a, **k):
        t0 = time.perf_counter()
        r = orig(*a, **k)
        d = acc.setdefault(key, [0.0, 0])
        d[0] += time.perf_counter() - t0
        d[1] += 1
        return r
    setattr(obj, name, timed)


wrap(eng, "norms", "norms")
wrap(eng, "hstep_objective", "h_device_call")
wrap(eng, "mstep_begin", "mstep_begin")
wrap(eng, "mstep_end", "mstep_end")
wrap(E, "constrain_loading", "constrain_loading")
wrap(E, "constrain_latent", "constrain_latent")
wrap(E, "estep", "estep_enqueue")
wrap(E, "hstep", "hstep")
wrap(E, "_pull_params", "pull_params")
for name in ("make_cholesky", "build_prior"):
    if hasattr(eng, name):
        wrap(eng, name, name)

rows = []
for it in range(n_it + 4):
    acc.clear()
    t0 = time.perf_counter()
    sess.em_iteration()
    tot = time.perf_counter() - t0
    if it >= 4:
        rows.append((tot, {k: tuple(v) for k, v in acc.items()}, sess.runtime["e_elapsed"][-1]))
keys = sorted({k for _, a, _ in rows for k in a})
print("workload %s: %d iterations, median em_iteration %.3f ms (E phase %.3f ms)" % (
    wl, len(rows), 1e3 * np.median([r[0] for r in rows]), 1e3 * np.median([r[2] for r in rows])))
for k in keys:
    t = np.median([a.get(k, (0, 0))[0] for _, a, _ in rows])
    n = np.median([a.get(k, (0, 0))[1] for _, a, _ in rows])
    print("  %-20s %8.3f ms  (%g calls, %.1f us each)" % (k, 1e3 * t, n, 1e6 * t / max(n, 1)))
sess.close()
