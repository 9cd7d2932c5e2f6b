"""BASELINE configs[4] on ONE GPU: 500 trials of 500..2000 bins (multiples of 50), 200 channels (150 Poisson + 50
Gaussian), 10 latents -- wall time of fit() with a few EM iterations, per phase."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlgp_amd import synth
from vlgp_amd.api import FitSession
n_trials = int(os.environ.get("TRIALS", "500"))
rng = np.random.default_rng(0)
lengths = (50 * rng.integers(10, 41, n_trials)).tolist()
t0 = time.perf_counter()
trials = synth.make_trials(n_trials, 2000, 200, 10, seed=0, n_gauss=50, lengths=lengths)
print("synth %.1f s, %d rows" % (time.perf_counter() - t0, sum(lengths)))
lik = ["poisson"] * 150 + ["gaussian"] * 50
np.random.seed(0)
t0 = time.perf_counter()
sess = FitSession(trials, 10, verbose=False, lik=lik, max_iter=4, min_iter=4)
t1 = time.perf_counter()
sess.run(); sess.eng.synchronize()
t2 = time.perf_counter()
rt = sess.runtime
print("init %.2f s | vem 4 it %.2f s | per it: E %s M %s H %s ms" % (t1 - t0, t2 - t1,
      np.round(1e3 * np.array(rt["e_elapsed"]), 1), np.round(1e3 * np.array(rt["m_elapsed"]), 1), np.round(1e3 * np.array(rt["h_elapsed"]), 1)))
res = sess.finish()
t3 = time.perf_counter()
print("final infer + download %.2f s; omega %s" % (t3 - t2, np.round(res["params"]["omega"], 4)))
