import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import vlgp_amd as V
from oracle import vlgp_oracle as O
import test_gpu_parity as T
def rel(a,b): return float(np.abs(np.asarray(a)-np.asarray(b)).max()/max(np.abs(b).max(),1e-300))
G = lambda n: dict(np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),"tests","golden",n+".npz")))
for tag in ("pois","mixed"):
    g = G("estep_"+tag)
    for method in ("VB","MAP"):
        units = T._units(g); params = T._params(g,3,20,chol={50:g["G"]})
        V.estep(units, params, V.get_config(method=method, Eniter=25))
        print(tag, method, {k: max(rel(units[m][k], g["%s_%s_25"%(k,method)][m]) for m in range(4)) for k in ("mu","v","w","dmu")},
              "absdmu", max(np.abs(units[m]["dmu"]-g["dmu_%s_25"%method][m]).max() for m in range(4)), "dmu scale", np.abs(g["dmu_%s_25"%method]).max())
import zlib
case = dict(lengths=[50, 50], N=130, L=6, P=3, g=0)
rng = np.random.default_rng(zlib.crc32(str(sorted(case.items())).encode()))
units, params, gauss = T._random_problem(rng, case["lengths"], case["N"], case["L"], case["P"], case["g"])
want = [O.estep_unit(u["y"], u["x"], u["mu"], u["v"], u["w"], params["a"], params["b"], params["noise"], gauss, params["cholesky"][u["y"].shape[0]], 4) for u in units]
V.estep(units, params, V.get_config(Eniter=4))
for u, ref in zip(units, want):
    print("case2", {k: rel(u[k], r) for k, r in zip(("mu","v","w","dmu"), ref)})
for tag in ("p1","mixed"):
    g = G("mstep_"+tag); M = g["y"].shape[0]
    units = [{k: g[k][m].copy() for k in ("y","x","mu","v")} for m in range(M)]
    for u in units: u["w"] = np.zeros_like(u["mu"])
    params = T._params(g, 3, 20, g["b"].shape[0])
    V.mstep(units, params, V.get_config(Mniter=25))
    print("mstep", tag, {k: rel(params[k], g[k+"_H_25"]) for k in ("a","b","da","db","noise")}, "da scale", np.abs(g["da_H_25"]).max())
g = G("vem_c1")
for tag, hs in (("H0", False), ("H1", True)):
    for ich in ("device","host"):
        trials = T._c1(g); traj=[]
        def spy(tr_, p_, c_): traj.append((np.linalg.norm(np.concatenate([s["mu"] for s in tr_])), np.linalg.norm(p_["a"]), np.linalg.norm(p_["b"]), np.array(p_["omega"])))
        np.random.seed(3)
        res = V.fit(trials, 3, a=g["a0"].copy(), b=g["b0"].copy(), Hstep=hs, max_iter=6, min_iter=6, callbacks=[spy], verbose=False, ichol=ich)
        print(tag, ich, "mu", rel([t[0] for t in traj], g["norm_mu_"+tag]), "a", rel([t[1] for t in traj], g["norm_a_"+tag]), "omega", rel(np.array([t[3] for t in traj]), g["omega_"+tag]),
              "final a", rel(res["params"]["a"], g["a_"+tag]), "rt", {k: np.round(v,4).tolist() if isinstance(v, list) else v for k,v in res["config"]["runtime"].items()})
        print("   per-iter mu err", [abs(t[0]-r)/r for t, r in zip(traj, g["norm_mu_"+tag])])
