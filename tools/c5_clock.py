import os, sys, time
import numpy as np
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from vlgp_amd import synth, engine as E
from vlgp_amd.api import FitSession
n_trials = 120
rng = np.random.default_rng(0)
lengths = (50 * rng.integers(10, 41, n_trials)).tolist()
trials = synth.make_trials(n_trials, 2000, 200, 10, seed=0, n_gauss=50, lengths=lengths)
lik = ["poisson"] * 150 + ["gaussian"] * 50
np.random.seed(0)
sess = FitSession(trials, 10, verbose=False, lik=lik, max_iter=6, min_iter=6)
for _ in range(4):
    sess.em_iteration()
eng = sess.eng
names = ["staging", "ya+factor0", "residual pass", "mean update", "curvature pass", "factor+variance"]
eng.phase_clock(True)
t0 = time.perf_counter(); E.estep(sess.segs, sess.params, sess.config); eng.synchronize(); t1 = time.perf_counter()
clk = eng.phase_clock(False)
tot = sum(clk[:6])
print("segments", len(sess.segs), "E-step %.1f ms" % (1e3 * (t1 - t0)), "ranks", eng.get_prior(50, with_rank=True)[1].tolist())
print(" | ".join("%s %.0f%%" % (n, 100.0 * c / max(tot, 1)) for n, c in zip(names, clk)))
sess.close()
