"""Low-rank H-step round (hstep_lr.h) against the dense matrix-pipe round and the oracle: (ll, dll) over a sweep of
omega (every rank class), odd / short windows, large w.  Usage: python tools/lr_check.py [M]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vlgp_amd.engine as V
from oracle.vlgp_oracle import gp_objective

def run(T, M, L, omegas, wscale, seed=0, oracle=True):
    rng = np.random.default_rng(seed)
    t = np.arange(T) * 1.0
    units = []
    for m in range(M):
        mu = np.cumsum(rng.standard_normal((T, L)), 0) * 0.3
        w = rng.uniform(0.02, 1.0, (T, L)) * wscale
        units.append({"y": np.zeros((T, 2)), "mu": mu, "w": w, "v": np.zeros((T, L))})
    worst = 0.0
    with V.Engine(2, L, 1, 50) as eng:
        eng.upload(0, units)
        for om in omegas:
            lat = np.arange(L)
            logp = np.tile(np.log([0.9, om, 1e-4]), (L, 1))
            logp[:, 1] += np.linspace(0, 0.3, L)
            os.environ.pop("VLGP_HSTEP_DENSE", None)
            os.environ["VLGP_HSTEP_LOWRANK"] = "1"  # whatever the size rule says
            eng.reload_switches()  # (cached at vlgp_create)
            ll1, dll1 = eng.hstep_objective(0, T, 1.0, lat, logp)
            p1 = eng.last_hstep_path
            os.environ["VLGP_HSTEP_DENSE"] = "1"
            eng.reload_switches()
            ll2, dll2 = eng.hstep_objective(0, T, 1.0, lat, logp)
            p2 = eng.last_hstep_path
            os.environ.pop("VLGP_HSTEP_DENSE", None)
            e_ll = np.max(np.abs(ll1 - ll2) / np.abs(ll2))
            e_d = np.max(np.abs(dll1[:, 1] - dll2[:, 1]) / np.maximum(np.abs(dll2[:, 1]), 1e-3 * np.abs(ll2)))
            msg = "T %3d M %4d omega %.2e  paths %s/%s  lr vs dense: ll %.1e dll %.1e" % (T, M, om, p1, p2, e_ll, e_d)
            if oracle:
                eo = []
                for i in range(L):
                    mu_l = np.stack([u["mu"][:, i] for u in units], 1)
                    w_l = np.stack([u["w"][:, i] for u in units], 1)
                    ll0, dll0 = gp_objective(logp[i], t, mu_l, w_l)
                    eo.append((abs(ll1[i] - ll0) / abs(ll0), abs(dll1[i, 1] - dll0[1]) / max(abs(dll0[1]), 1e-3 * abs(ll0)),
                               abs(ll2[i] - ll0) / abs(ll0), abs(dll2[i, 1] - dll0[1]) / max(abs(dll0[1]), 1e-3 * abs(ll0))))
                eo = np.max(np.array(eo), 0)
                msg += " | vs oracle: lr ll %.1e dll %.1e, dense ll %.1e dll %.1e" % tuple(eo)
                worst = max(worst, eo[0], eo[1])
            worst = max(worst, e_ll, e_d)
            print(msg, flush=True)
    return worst

if __name__ == "__main__":
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    w = 0.0
    w = max(w, run(50, M, 3, [5e-4, 1e-3, 2e-3, 4e-3, 8e-3, 1.3e-2, 2e-2, 3e-2, 5e-2], 1.0))
    w = max(w, run(50, 37, 3, [1e-3, 8e-3], 30.0, seed=1))
    w = max(w, run(49, 33, 2, [1e-3, 8e-3, 2e-2], 1.0, seed=2))
    w = max(w, run(24, 17, 3, [1e-3, 2e-2, 5e-2], 1.0, seed=3))
    w = max(w, run(64, 20, 2, [1e-3, 8e-3, 1.5e-2], 1.0, seed=4))
    w = max(w, run(33, 16, 1, [5e-3], 1.0, seed=5))
    print("worst relative error %.2e" % w)
    sys.exit(0 if w < 1e-9 else 1)
