"""Final full-length inference (core.infer on the trials, Eniter = 10) at C3 / C5-like through the persistent long-unit
kernel (VLGP_ESTEP_LSPLIT=0) and through the task-parallel launch sequence (=1): wall time per call, results compared."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import vlgp_amd as V

wl = os.environ.get("WL", "C3")
trials, a0, b0, (n_trials, n_bins, N, L) = bench.build_inputs(wl)
rng = np.random.default_rng(0)
omega = np.array([1.3e-2, 8e-3, 4e-3, 1e-2, 1e-3, 6e-3, 2e-3, 9e-3, 3e-3, 5e-3][:L])
out = {}
for mode in ("0", "1"):
    os.environ["VLGP_ESTEP_LSPLIT"] = mode
    with V.Engine(N, L, 1, 50) as eng:
        eng.set_params(a0, b0, np.ones(N))
        eng.upload(0, [{"y": t["y"], "mu": t["mu"]} for t in trials])
        eng.build_prior([n_bins], omega, np.ones(L))
        eng.update_w(0)
        eng.update_v(0)
        eng.synchronize()
        ts = []
        for rep in range(4):
            t0 = time.perf_counter()
            eng.estep(0, 10, count=False)
            eng.synchronize()
            ts.append(time.perf_counter() - t0)
        out[mode] = eng.download(0)
        print("LSPLIT=%s path %s: %s ms per 10-sweep call" % (mode, eng.last_estep_path, np.round(1e3 * np.array(ts), 2)))
for k in ("mu", "v", "w"):
    d = np.abs(out["0"][k] - out["1"][k]).max() / np.abs(out["0"][k]).max()
    print("  %s: max rel difference between the two paths %.2e" % (k, d))
