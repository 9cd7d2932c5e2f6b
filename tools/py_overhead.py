"""Where the HOST spends an EM iteration (cProfile over 20 steady-state iterations of the bench workload): the Python /
NumPy glue around the C ABI calls, sorted by own time."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs(os.environ.get("WL", "C3"))
n = 30
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=n, min_iter=n)
for _ in range(8):
    sess.em_iteration()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    sess.em_iteration()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(25)
sess.close()
