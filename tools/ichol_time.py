import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import vlgp_amd as V
om = np.array([2e-3, 3e-3, 2.5e-3, 4e-3, 1.5e-3])
sg = np.ones(5)
for env in ("", "1"):
    if env:
        os.environ["VLGP_ICHOL_BLOCK"] = "1"
    with V.Engine(100, 5, 1, 50) as eng:
        for om_scale in (1.0, 10.0):
            eng.build_prior([50], om * om_scale, sg)
            t0 = time.perf_counter()
            for _ in range(300):
                eng.build_prior([50], om * om_scale, sg)
            dt = (time.perf_counter() - t0) / 300
            print("block" if env else "wave ", "omega x%g" % om_scale, "ranks", eng.prior_ranks(50), "%.1f us per build_prior" % (dt * 1e6))
