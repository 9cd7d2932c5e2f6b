"""The H-step's lock-step L-BFGS-B driver must be scipy.optimize.minimize,
decision for decision: same iterates, same number of evaluations."""
import numpy as np
import pytest
from scipy.optimize import minimize

from vlgp_amd import gp


def _objective(k):
    c = np.array([0.3 * k - 1.0, 0.5 - 0.2 * k, 0.1 * k])
    A = np.array([[3.0, 0.5, 0.1], [0.5, 2.0 + k, 0.3], [0.1, 0.3, 1.0 + 0.5 * k]])

    def fun(x):
        d = x - c
        f = 0.5 * d @ A @ d + 0.1 * np.sum(d ** 4) + np.cos(x[1] + k)
        g = A @ d + 0.4 * d ** 3
        g[1] -= np.sin(x[1] + k)
        return f, g

    return fun


BOUNDS = np.array([(-1.0, 1.0), (-2.0, 0.5), (-0.5, 0.5)])


def _reference(n):
    out, nfev = [], []
    for k in range(n):
        res = minimize(_objective(k), np.array([0.2, -0.3, 0.1]) * (k + 1) / n, jac=True, bounds=BOUNDS)
        out.append(res.x)
        nfev.append(res.nfev)
    return out, nfev


@pytest.mark.parametrize("force_threads", [False, True])
def test_lockstep_matches_scipy_minimize(monkeypatch, force_threads):
    n = 5
    want, want_nfev = _reference(n)
    funs = [_objective(k) for k in range(n)]
    count = np.zeros(n, dtype=int)
    rounds = [0]

    def batch(keys, X):
        rounds[0] += 1
        f, G = [], []
        for k, x in zip(keys, X):
            fk, gk = funs[k](x)
            count[k] += 1
            f.append(fk)
            G.append(gk)
        return np.array(f), np.array(G)

    if force_threads:
        monkeypatch.setattr(gp, "_setulb_or_none", lambda: None)
    elif gp._setulb_or_none() is None:
        pytest.skip("this SciPy does not expose the known setulb signature")
    x0s = [np.array([0.2, -0.3, 0.1]) * (k + 1) / n for k in range(n)]
    got = gp.lockstep_minimize(batch, x0s, BOUNDS)
    for k in range(n):
        assert np.array_equal(got[k], want[k]), k
        assert count[k] == want_nfev[k], (k, count[k], want_nfev[k])
    assert rounds[0] == max(want_nfev)  # one batched call per round


def test_masked_gradient_like_the_hstep():
    # the H-step objective has a zero gradient in two of three coordinates
    def fun(x):
        return (x[1] + 5.0) ** 2 + 0.1 * x[1] ** 4, np.array([0.0, 2 * (x[1] + 5.0) + 0.4 * x[1] ** 3, 0.0])

    b = np.log(np.array([(1e-3, 1.0), (5e-4, 5e-2), (5e-5, 2e-4)]))
    x0 = np.log(np.array([1.0, 5e-2, 1e-4]))
    want = minimize(fun, x0, jac=True, bounds=b).x
    got = gp.lockstep_minimize(lambda ks, X: (np.array([fun(x)[0] for x in X]), np.array([fun(x)[1] for x in X])),
                               [x0], b)[0]
    assert np.array_equal(got, want)


def test_native_driver_matches_scipy_minimize():
    """vlgp_amd._lockstep (csrc/lockstep_ext.c): the loop around SciPy's setulb and the objective call in C.  Same calls
    to the same routine, so the iterates equal scipy.optimize.minimize's bit for bit -- also with more than sixteen runs
    (evaluations go to the objective in slices of sixteen).  The objective here is a ctypes callback with the
    signature of vlgp_hstep_objective (it returns the UN-negated value and gradient, as the C ABI does)."""
    import ctypes as C

    from scipy.optimize import minimize

    if gp._setulb_or_none() is None or gp._lockstep_ext() is None:
        pytest.skip("needs SciPy's setulb and the built extension")
    rng = np.random.default_rng(0)
    cs = [rng.normal(size=3) * 0.5 for _ in range(20)]

    def fun(k, x):
        d = x - cs[k]
        return float(np.sum(d ** 4) + 0.5 * np.sum(d * d) + np.sin(x[1])), 4 * d ** 3 + d + np.array([0, np.cos(x[1]), 0])

    proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_int),
                        C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double))
    sizes = []

    def objective(ctx, set_id, window, dt, n, lat, logp, ll, dll):
        sizes.append(n)
        for e in range(n):
            f, g = fun(lat[e], np.array([logp[3 * e], logp[3 * e + 1], logp[3 * e + 2]]))
            ll[e] = -f
            for j in range(3):
                dll[3 * e + j] = -g[j]
        return 0

    cb = proto(objective)
    bounds = np.array([(-2.0, 2.0)] * 3)
    for n in (5, 20):
        x0s = [np.zeros(3) for _ in range(n)]
        xs, status = gp.lockstep_minimize_native(C.cast(cb, C.c_void_p).value, 0, 0, 50, 1.0, range(n), x0s, bounds)
        assert status == 0
        for k in range(n):
            ref = minimize(lambda x, k=k: fun(k, x), x0s[k], jac=True, bounds=bounds).x
            assert np.array_equal(xs[k], ref), k
    assert max(sizes) == 16

    def failing(ctx, set_id, window, dt, n, lat, logp, ll, dll):
        return -3

    cb2 = proto(failing)
    _, status = gp.lockstep_minimize_native(C.cast(cb2, C.c_void_p).value, 0, 0, 50, 1.0, range(2),
                                            [np.zeros(3)] * 2, bounds)
    assert status == -3
