import sys, os, numpy as np, zlib
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import vlgp_amd as V
from oracle import vlgp_oracle as O
import test_gpu_parity as T
def rel(a,b): return float(np.abs(np.asarray(a)-np.asarray(b)).max()/max(np.abs(b).max(),1e-300))
for case in [dict(lengths=[50,50],N=130,L=6,P=3,g=0), dict(lengths=[50,50],N=130,L=6,P=1,g=0), dict(lengths=[50,50],N=40,L=6,P=3,g=0),
             dict(lengths=[50,50],N=130,L=3,P=3,g=0), dict(lengths=[50,50],N=130,L=5,P=1,g=0), dict(lengths=[50,50],N=100,L=6,P=1,g=0), dict(lengths=[50,50],N=64,L=6,P=1,g=0)]:
    for nit in (1, 4):
        rng = np.random.default_rng(zlib.crc32(str(sorted(case.items())).encode()))
        units, params, gauss = T._random_problem(rng, case["lengths"], case["N"], case["L"], case["P"], case["g"])
        want = [O.estep_unit(u["y"], u["x"], u["mu"], u["v"], u["w"], params["a"], params["b"], params["noise"], gauss, params["cholesky"][u["y"].shape[0]], nit) for u in units]
        V.estep(units, params, V.get_config(Eniter=nit))
        print(case, nit, {k: max(rel(u[k], r[i]) for u, r in zip(units, want)) for i, k in enumerate(("mu","v","w","dmu"))},
              "ranks", [int((np.abs(params["cholesky"][50][l]).sum(0)>0).sum()) for l in range(case["L"])], "maxrate", float(np.exp(np.minimum((units[0]["mu"]@params["a"]).max(),10))))
