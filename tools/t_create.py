import time, sys, numpy as np
sys.path.insert(0, "/root/repo")
import vlgp_amd
for i in range(4):
    t0 = time.perf_counter(); e = vlgp_amd.Engine(100, 5, 1, 50); t1 = time.perf_counter(); e.close(); t2 = time.perf_counter()
    print("create %.1f ms destroy %.1f ms" % (1e3*(t1-t0), 1e3*(t2-t1)))
