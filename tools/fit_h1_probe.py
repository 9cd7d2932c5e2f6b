"""Error of fit() with default arguments against the reference's golden fit (H-step on)."""
import numpy as np
import vlgp_amd as V
g = dict(np.load("tests/golden/fit_c1_h1.npz"))
def rel(a, b): return float(np.abs(np.asarray(a) - b).max() / max(np.abs(b).max(), 1e-300))
y = g["y"].astype(float)
trials = [{"ID": i, "y": y[i].copy(), "mu": g["mu0"][i].copy()} for i in range(y.shape[0])]
np.random.seed(3)
res = V.fit(trials, 3, a=g["a0"].copy(), b=g["b0"].copy(), max_iter=5, min_iter=5, verbose=False)
p = res["params"]
print("injected init: it", res["config"]["runtime"]["it"], int(g["it"]))
for k in ("a", "b", "noise", "omega", "sigma"): print(k, rel(p[k], g[k]))
print("G200 equal", np.array_equal(p["cholesky"][200], g["G200"]), [np.array_equal(p["cholesky"][200][l], g["G200"][l]) for l in range(3)])
print("omega bits equal", p["omega"] == g["omega"], p["omega"], g["omega"])
for k in ("mu", "v", "w", "dmu"): print(k, rel(np.stack([t[k] for t in trials]), g[k]))
trials = [{"ID": i, "y": y[i].copy()} for i in range(y.shape[0])]
np.random.seed(5)
res = V.fit(trials, 3, max_iter=8, verbose=False)
p = res["params"]
print("default init: it", res["config"]["runtime"]["it"], int(g["d_it"]))
for k in ("a", "b", "noise", "omega", "sigma"): print(k, rel(p[k], g["d_" + k]))
print("G200 equal", [np.array_equal(p["cholesky"][200][l], g["d_G200"][l]) for l in range(3)], p["omega"], g["d_omega"])
for k in ("mu", "v", "w"): print(k, rel(np.stack([t[k] for t in trials]), g["d_" + k]))
