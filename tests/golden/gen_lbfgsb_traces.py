#!/usr/bin/env python
"""Recorded L-BFGS-B traces of the LIVE SciPy (scipy.optimize._lbfgsb.setulb, the routine behind
scipy.optimize.minimize(method="L-BFGS-B") that vlgp/gp.py:114 calls) -> tests/golden/lbfgsb_traces.npz.

    python tests/golden/gen_lbfgsb_traces.py

Per problem: bounds, start point, and for every call of the routine the task code it returned, the iterate x it left and
the (f, g) fed back to it.  Data only; tests/test_lockstep_lbfgsb.py replays the (f, g) sequence into csrc/lbfgsb.c and
compares iterates and decisions (machine-independent: the objective values come from the file, not from this host's libm).
Problems: the H-step's shape (three log-parameters, gradient masked to omega, sigma^2 starting on its upper bound:
vlgp/gp.py:81-85,100-123) with objectives of the H-step's kind, plus general box-constrained problems (every bound kind,
1 .. 8 variables, 1 .. 10 corrections) that reach the code the H-step rarely does (several breakpoints, entering / leaving
variables, the circular shift of a full history, failed line searches).
"""
import os

import numpy as np
import scipy
from scipy.optimize import _lbfgsb

HERE = os.path.dirname(os.path.abspath(__file__))


def hstep_like(seed):
    rng = np.random.default_rng(seed)
    c, a, k = rng.uniform(-7.5, -3.2), rng.uniform(0.5, 200.0), rng.uniform(0.1, 3)

    def fun(x):
        t = x[1] - c
        return a * (np.cosh(k * t) - 1.0) + 0.3 * a * np.sin(t) ** 2, np.array([0.0, a * k * np.sinh(k * t) + 0.3 * a * np.sin(2 * t), 0.0])

    lo = np.log(np.array([1e-3, 5e-4, 5e-4]))
    hi = np.log(np.array([1.0, 5e-2, 2e-3]))
    x0 = np.array([0.0, rng.uniform(lo[1], hi[1]), np.log(1e-3)])
    return fun, x0, lo, hi, np.full(3, 2, np.int32), 10, 1e7, 1e-5, 20


def general(seed):
    rng = np.random.default_rng(10_000 + seed)
    n, m = int(rng.integers(1, 9)), int(rng.choice([1, 2, 3, 5, 10]))
    A = rng.standard_normal((n, n))
    A = A @ A.T + 0.05 * np.eye(n)
    b = 3 * rng.standard_normal(n)
    kind = seed % 4

    def fun(x):
        f = 0.5 * x @ A @ x - b @ x + 0.05 * np.sum(x ** 4)
        g = A @ x - b + 0.2 * x ** 3
        if kind == 1:  # kink
            f, g = f + np.sum(np.abs(x)), g + np.sign(x)
        if kind == 2:  # inconsistent gradient: failed line searches, restarts, abnormal termination
            g = 1.7 * g - 0.3
        return f, g

    nbd = rng.integers(0, 4, n).astype(np.int32)
    lo = rng.uniform(-2, 0.5, n)
    hi = lo + rng.uniform(0.0 if seed % 7 == 0 else 0.2, 3, n)
    return (fun, rng.uniform(-3, 3, n), lo, hi, nbd, m, float(rng.choice([1e7, 10.0, 1e12])), float(rng.choice([1e-5, 1e-10])),
            int(rng.choice([20, 3])))


def record(problem):
    fun, x0, lo, hi, nbd, m, factr, pgtol, maxls = problem
    n = x0.size
    x = np.array(x0, dtype=float)
    f, g = 0.0, np.zeros(n)
    wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m)
    iwa, task, ln_task = np.zeros(3 * n, np.int32), np.zeros(2, np.int32), np.zeros(2, np.int32)
    lsave, isave, dsave = np.zeros(4, np.int32), np.zeros(44, np.int32), np.zeros(29)
    tasks, xs, fs, gs = [], [], [], []
    for _ in range(4000):
        _lbfgsb.setulb(m, x, lo, hi, nbd, f, g, factr, pgtol, wa, iwa, task, lsave, isave, dsave, maxls, ln_task)
        tasks.append(task.copy())
        xs.append(x.copy())
        if task[0] == 3:
            f, g = fun(x)
            f, g = float(f), np.array(g, dtype=float)
        elif task[0] != 1:
            fs.append(f)
            gs.append(g.copy())
            break
        fs.append(f)
        gs.append(g.copy())
    return dict(x0=x0, lo=lo, hi=hi, nbd=nbd, m=m, factr=factr, pgtol=pgtol, maxls=maxls, tasks=np.array(tasks), xs=np.array(xs),
                fs=np.array(fs), gs=np.array(gs))


def main():
    out = {"scipy_version": np.array(scipy.__version__)}
    problems = [hstep_like(s) for s in range(24)] + [general(s) for s in range(40)]
    for i, pr in enumerate(problems):
        for k, v in record(pr).items():
            out["p%02d_%s" % (i, k)] = np.asarray(v)
    out["n_problems"] = np.array(len(problems))
    np.savez_compressed(os.path.join(HERE, "lbfgsb_traces.npz"), **out)
    print("wrote", len(problems), "traces;", sum(len(out["p%02d_tasks" % i]) for i in range(len(problems))), "calls")


if __name__ == "__main__":
    main()
