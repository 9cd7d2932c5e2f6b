"""Where an EM iteration's wall time goes on the host side at C3: cProfile over 20 iterations (cumulative, top 25)."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=40, min_iter=40)
for _ in range(8):
    sess.em_iteration()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    sess.em_iteration()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
sess.close()
