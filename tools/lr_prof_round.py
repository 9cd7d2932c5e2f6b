"""One rank class of the low-rank round at C3 for the counter passes: OM=omega NE=evaluations REP=calls."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from vlgp_amd.api import FitSession
trials, a0, b0, dims = bench.build_inputs("C3")
sess = FitSession(trials, dims[3], verbose=False, a=a0.copy(), b=b0.copy(), max_iter=20, min_iter=20)
eng = sess.eng
sess.em_iteration()
sid = sess.segs.set_id
om = float(os.environ.get("OM", "6e-3")); ne = int(os.environ.get("NE", "15")); rep = int(os.environ.get("REP", "20"))
eng.hstep_begin(sid, 50)
lat = np.arange(ne, dtype=np.int32) % dims[3]
logp = np.log(np.array([[1.0, om, 1e-4]] * ne))
for _ in range(rep):
    eng.hstep_objective(sid, 50, 1.0, lat, logp)
print(eng.last_hstep_path)
eng.hstep_end(); sess.close()
