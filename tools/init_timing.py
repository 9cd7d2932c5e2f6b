"""Wall time of the steps of FitSession.__init__ at C3 (the path vlgp_amd.fit takes: factor analysis on the host,
latent projection on the device), two repetitions in one process."""
import copy, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vlgp_amd import synth, engine as E
from vlgp_amd.preprocess import get_config, get_params, initialize, fill_params, fill_trials
from vlgp_amd.api import _segments, SET_TRIALS
import bench
n_trials, n_bins, N, L = bench.WORKLOADS[os.environ.get("WL", "C3")]
for rep in range(3):
    trials = synth.make_trials(n_trials, n_bins, N, L, seed=0)
    np.random.seed(0)
    T = [time.perf_counter()]
    def lap(name):
        T.append(time.perf_counter()); print("  %-28s %.1f ms" % (name, 1e3 * (T[-1] - T[-2])))
    config = get_config(max_iter=10, min_iter=10)
    params = get_params(trials, L, omega_bound=config["omega_bound"]); lap("config/params")
    eng = E.Engine(params["ydim"], params["zdim"], params["xdim"], params["rank"], np.asarray(params["likelihood"]) == "gaussian"); lap("engine")
    plan = initialize(trials, params, config, defer_latent=True); lap("initialize (deferred)")
    fill_trials(trials); lap("fill_trials")
    eng.upload(SET_TRIALS, trials); lap("upload")
    colsum = eng.project_latent(SET_TRIALS, plan["proj"], plan["shift"]); lap("project_latent")
    if plan["need_b"]:
        params["b"] = np.log(np.maximum(colsum[None, :] / plan["rows"], config["eps"]))
    fill_params(params)
    for key in ("a", "b", "noise", "omega", "sigma"):
        params[key] = np.array(params[key], dtype=float)
    eng.set_params(params["a"], params["b"], params["noise"]); lap("fill/set_params")
    dev = E.DeviceTrials(trials, eng, SET_TRIALS)
    E.make_cholesky(dev, params, config); lap("make_cholesky trials")
    E.update_w(dev, params, config); E.update_v(dev, params, config); eng.synchronize(); lap("update_w/v")
    segs = _segments(trials, config["window"], eng); lap("_segments")
    E.make_cholesky(segs, params, config); fill_trials(segs); lap("cholesky segs + fill")
    snapshot = {k: v for k, v in params.items() if k not in ("cholesky", "transform")}
    params["initial"] = copy.deepcopy(snapshot); lap("deepcopy")
    E._push_params(eng, params); eng.synchronize(); lap("push")
    print("rep total %.1f ms" % (1e3 * (T[-1] - T[0])))
    eng.close()
