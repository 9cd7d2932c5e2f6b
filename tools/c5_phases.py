"""Where the initialisation and the final phase of fit() go at BASELINE configs[4] (500 ragged trials, 643 k bins,
200 mixed channels, 10 latents) on one GPU: cProfile by cumulative time."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlgp_amd import synth
from vlgp_amd.api import FitSession
n_trials = int(os.environ.get("TRIALS", "500"))
rng = np.random.default_rng(0)
lengths = (50 * rng.integers(10, 41, n_trials)).tolist()
trials = synth.make_trials(n_trials, 2000, 200, 10, seed=0, n_gauss=50, lengths=lengths)
lik = ["poisson"] * 150 + ["gaussian"] * 50
np.random.seed(0)
pr = cProfile.Profile(); pr.enable()
sess = FitSession(trials, 10, verbose=False, lik=lik, max_iter=2, min_iter=2)
pr.disable()
sess.run(); sess.eng.synchronize()
pr2 = cProfile.Profile(); pr2.enable()
res = sess.finish()
pr2.disable()
print("---- FitSession.__init__")
pstats.Stats(pr).sort_stats("cumtime").print_stats(16)
print("---- finish")
pstats.Stats(pr2).sort_stats("cumtime").print_stats(14)
