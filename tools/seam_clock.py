"""Host clock of the seam between two EM iterations at C3: from the return of the H-step's round loop to the first E-step
launch call of the next iteration (medians over the steady-state iterations, microseconds)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd import engine as E, gp
from vlgp_amd.api import FitSession

trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
n_it = int(os.environ.get("ITERS", "24"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=n_it + 6, min_iter=n_it + 6)
eng = sess.eng
marks = []


def stamp(name):
    marks.append((name, time.perf_counter()))


def wrap(obj, name, key):
    orig = getattr(obj, name)

    def timed(*a, **k):
        stamp(key + ">")
        r = orig(*a, **k)
        stamp(key + "<")
        return r
    setattr(obj, name, timed)


wrap(gp, "lockstep_minimize_own", "rounds")
wrap(eng, "build_prior", "build_prior")
wrap(eng, "mstep_end", "mstep_end")
wrap(eng, "get_params", "get_params")
wrap(eng, "norms_end", "norms_end")
wrap(eng, "apply_latent_map", "latent_map")
wrap(eng, "set_loading", "set_loading")
wrap(eng, "estep", "estep_call")
wrap(eng, "synchronize", "sync")
wrap(eng, "hstep_prepare", "hstep_prepare")
wrap(eng, "norms_begin", "norms_begin")
wrap(eng, "hstep_begin", "hstep_begin")
rows = []
for it in range(n_it + 6):
    sess.em_iteration()
names = []
seq = {}
# chain: rounds< -> build_prior> -> build_prior< -> mstep_end> ... -> estep_call> (next iteration); and sync< -> rounds>
for i in range(len(marks) - 1):
    (n0, t0), (n1, t1) = marks[i], marks[i + 1]
    key = "%s -> %s" % (n0, n1)
    if key not in seq:
        names.append(key)
    seq.setdefault(key, []).append(t1 - t0)
for key in names:
    v = np.array(seq[key][6:]) * 1e6
    if len(v):
        print("%-40s median %8.1f us   (n = %d)" % (key, np.median(v), len(v)))
sess.close()
