/* vlgp_amd._lockstep: the lock-step L-BFGS-B driver of the H-step (vlgp_amd/gp.py, lockstep_minimize) with its inner
 * loop in C.
 *
 * The optimiser is still SciPy's: every step goes through the reverse-communication routine
 * scipy.optimize._lbfgsb.setulb, called here as the Python callable it is (the module exports no C symbol), with the
 * very argument objects gp._Lbfgsb builds -- so the iterates are those of scipy.optimize.minimize, decision for
 * decision (tests/test_lockstep_lbfgsb.py holds both drivers to array_equal).  What moves to C is everything around
 * those calls, which a round of the H-step pays once per latent on the critical path of the EM iteration: the
 * state machine of gp._Lbfgsb.advance / feed, the gathering of the pending points, and the objective call itself --
 * vlgp_hstep_objective (include/vlgp_hip.h) through its address instead of NumPy staging + ctypes.
 *
 *   run(setulb, runs, objective_address, handle_address, set_id, window, dt, maxiter, maxfun) -> status
 *
 * runs: list of (head, tail, latent) with head = (m, x, l, u, nbd) and tail = (g, factr, pgtol, wa, iwa, task,
 * lsave, isave, dsave, maxls, ln_task) exactly as gp._Lbfgsb keeps them; x, g: float64 arrays of 3, task: int32 array.
 * The objective is minimised as -ll with gradient -dll (vlgp/gp.py:107-111).  Returns the status of the first failing
 * objective call (the caller turns it into the handle's error), 0 otherwise.  Host control logic only: no arithmetic of
 * the model lives here, and gp.py falls back to its Python loop when the module is not built.
 */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <stdint.h>
#include <string.h>

#include "lbfgsb.h"

typedef int (*objective_fn)(void* ctx, int set, int window, double dt, int n_eval, const int* latent, const double* logp,
                            double* ll, double* dll);

#define MAX_RUNS 64

typedef struct {
    PyObject *head, *tail;  /* borrowed from the list (which outlives the call) */
    PyObject* args;         /* owned: (m, x, l, u, nbd, f, g, ..., ln_task), built once; only slot 5 (f) is replaced */
    Py_buffer x, g, task;
    int have_x, have_g, have_task;
    int latent, done, nit, nfev;
    double f;
} Run;

static void release_runs(Run* r, int n) {
    for (int i = 0; i < n; ++i) {
        if (r[i].have_x) PyBuffer_Release(&r[i].x);
        if (r[i].have_g) PyBuffer_Release(&r[i].g);
        if (r[i].have_task) PyBuffer_Release(&r[i].task);
        Py_XDECREF(r[i].args);
    }
}

/* gp._Lbfgsb.advance: call setulb until it asks for (f, g) or stops.  1: evaluation wanted, 0: finished, -1: error */
static int advance(PyObject* setulb, Run* r, long maxiter, long maxfun) {
    int32_t* task = (int32_t*)r->task.buf;
    while (!r->done) {
        PyObject* fobj = PyFloat_FromDouble(r->f);
        if (!fobj) return -1;
        if (!r->args) {
            r->args = PyTuple_New(17);
            if (!r->args) { Py_DECREF(fobj); return -1; }
            for (int q = 0; q < 5; ++q) {
                PyObject* o = PyTuple_GET_ITEM(r->head, q);
                Py_INCREF(o);
                PyTuple_SET_ITEM(r->args, q, o);
            }
            for (int q = 0; q < 11; ++q) {
                PyObject* o = PyTuple_GET_ITEM(r->tail, q);
                Py_INCREF(o);
                PyTuple_SET_ITEM(r->args, 6 + q, o);
            }
            PyTuple_SET_ITEM(r->args, 5, fobj);
        } else {  /* the tuple is ours alone (the callee keeps no reference to it): swap the one by-value argument */
            PyObject* old = PyTuple_GET_ITEM(r->args, 5);
            PyTuple_SET_ITEM(r->args, 5, fobj);
            Py_DECREF(old);
        }
        PyObject* res = PyObject_CallObject(setulb, r->args);
        if (!res) return -1;
        Py_DECREF(res);
        const int t0 = task[0];
        if (t0 == 3) return 1;
        if (t0 == 1) { /* new iterate accepted */
            r->nit += 1;
            if (r->nit >= maxiter) { task[0] = 5; task[1] = 504; }
            else if (r->nfev > maxfun) { task[0] = 5; task[1] = 502; }
        } else {
            r->done = 1;
        }
    }
    return 0;
}

static PyObject* lockstep_run(PyObject* self, PyObject* a) {
    PyObject *setulb, *runs;
    unsigned long long fn_addr, ctx_addr;
    int set_id, window;
    double dt;
    long maxiter, maxfun;
    if (!PyArg_ParseTuple(a, "OOKKiidll", &setulb, &runs, &fn_addr, &ctx_addr, &set_id, &window, &dt, &maxiter, &maxfun))
        return NULL;
    if (!PyList_Check(runs)) { PyErr_SetString(PyExc_TypeError, "runs must be a list"); return NULL; }
    const int n = (int)PyList_GET_SIZE(runs);
    if (n < 1 || n > MAX_RUNS) { PyErr_SetString(PyExc_ValueError, "1 .. 64 runs"); return NULL; }
    objective_fn fn = (objective_fn)(uintptr_t)fn_addr;
    void* ctx = (void*)(uintptr_t)ctx_addr;
    Run r[MAX_RUNS];
    memset(r, 0, sizeof(r));
    for (int i = 0; i < n; ++i) {
        PyObject* it = PyList_GET_ITEM(runs, i);
        if (!PyTuple_Check(it) || PyTuple_GET_SIZE(it) != 3) goto bad;
        r[i].head = PyTuple_GET_ITEM(it, 0);
        r[i].tail = PyTuple_GET_ITEM(it, 1);
        r[i].latent = (int)PyLong_AsLong(PyTuple_GET_ITEM(it, 2));
        if (!PyTuple_Check(r[i].head) || PyTuple_GET_SIZE(r[i].head) != 5 || !PyTuple_Check(r[i].tail) ||
            PyTuple_GET_SIZE(r[i].tail) != 11)
            goto bad;
        if (PyObject_GetBuffer(PyTuple_GET_ITEM(r[i].head, 1), &r[i].x, PyBUF_WRITABLE | PyBUF_FORMAT) < 0) goto fail;
        r[i].have_x = 1;
        if (PyObject_GetBuffer(PyTuple_GET_ITEM(r[i].tail, 0), &r[i].g, PyBUF_WRITABLE | PyBUF_FORMAT) < 0) goto fail;
        r[i].have_g = 1;
        if (PyObject_GetBuffer(PyTuple_GET_ITEM(r[i].tail, 5), &r[i].task, PyBUF_WRITABLE | PyBUF_FORMAT) < 0) goto fail;
        r[i].have_task = 1;
        if (r[i].x.len != 3 * (Py_ssize_t)sizeof(double) || r[i].g.len != 3 * (Py_ssize_t)sizeof(double) ||
            r[i].x.itemsize != 8 || r[i].g.itemsize != 8 || r[i].task.itemsize != 4 || r[i].task.len < 8)
            goto bad;
    }
    {
        int status = 0;
        int active[MAX_RUNS], n_act = n;
        for (int i = 0; i < n; ++i) active[i] = i;
        int lat[MAX_RUNS];
        double logp[3 * 16], ll[16], dll[3 * 16];
        for (;;) {
            int m = 0;
            for (int q = 0; q < n_act; ++q) {
                const int k = active[q];
                const int w = advance(setulb, &r[k], maxiter, maxfun);
                if (w < 0) goto fail;
                if (w) active[m++] = k;
            }
            n_act = m;
            if (n_act == 0) break;
            for (int base = 0; base < n_act && status == 0; base += 16) { /* a call takes at most 16 evaluations */
                const int cnt = n_act - base < 16 ? n_act - base : 16;
                for (int q = 0; q < cnt; ++q) {
                    const Run* rk = &r[active[base + q]];
                    const double* x = (const double*)rk->x.buf;
                    lat[q] = rk->latent;
                    logp[3 * q + 0] = x[0]; logp[3 * q + 1] = x[1]; logp[3 * q + 2] = x[2];
                }
                status = fn(ctx, set_id, window, dt, cnt, lat, logp, ll, dll);
                if (status != 0) break;
                for (int q = 0; q < cnt; ++q) {
                    Run* rk = &r[active[base + q]];
                    double* g = (double*)rk->g.buf;
                    rk->f = -ll[q];
                    g[0] = -dll[3 * q + 0]; g[1] = -dll[3 * q + 1]; g[2] = -dll[3 * q + 2];
                    rk->nfev += 1;
                }
            }
            if (status != 0) break;
        }
        release_runs(r, n);
        return PyLong_FromLong(status);
    }
bad:
    PyErr_SetString(PyExc_ValueError, "runs: list of ((m, x, l, u, nbd), (g, ..., ln_task), latent) with 3 parameters");
fail:
    release_runs(r, n);
    return NULL;
}


/* ---- round 6: the optimiser itself in C (lbfgsb.c), no SciPy object on the path ------------------------------------
 *
 *   run_own(latents, X, log_bounds, objective_address, handle_address, set_id, window, dt, maxiter, maxfun, blas) -> status
 *
 * X: (n_runs, 3) float64, C-contiguous, in / out (the start points, clipped to the bounds as SciPy's driver clips them;
 * the minimisers on return); log_bounds: (3, 2) float64; blas: None (the portable loops of lbfgsb.c) or a tuple of seven
 * addresses (ddot, daxpy, dscal, dcopy, dnrm2, dpotrf, dtrtrs: Fortran ABI) -- gp.py passes those of the OpenBLAS SciPy
 * ships, which makes every iterate bit-identical to scipy.optimize.minimize's.  The driver loop is _minimize_lbfgsb's
 * (scipy/optimize/_lbfgsb_py.py): m = 10, factr = 1e7, pgtol = 1e-5, maxls = 20, the iteration / evaluation limits checked
 * after every accepted iterate.
 */
#define LB_N 3
#define LB_M 10
#define LB_WA (2 * LB_M * LB_N + 5 * LB_N + 11 * LB_M * LB_M + 8 * LB_M)

typedef struct {
    double x[LB_N], g[LB_N], f;
    double wa[LB_WA], dsave[29];
    int iwa[3 * LB_N], task[2], ln_task[2], lsave[4], isave[44];
    int latent, done, nit, nfev;
} OwnRun;

static int parse_blas(PyObject* o, lbfgsb_blas* b, const lbfgsb_blas** out) {
    if (o == Py_None) { *out = NULL; return 0; }
    if (!PyTuple_Check(o) || PyTuple_GET_SIZE(o) != 7) {
        PyErr_SetString(PyExc_ValueError, "blas: None or 7 addresses (ddot, daxpy, dscal, dcopy, dnrm2, dpotrf, dtrtrs)");
        return -1;
    }
    void* a[7];
    for (int i = 0; i < 7; ++i) {
        a[i] = PyLong_AsVoidPtr(PyTuple_GET_ITEM(o, i));
        if (PyErr_Occurred()) return -1;
        if (!a[i]) { PyErr_SetString(PyExc_ValueError, "blas: null address"); return -1; }
    }
    memcpy(&b->ddot, &a[0], sizeof(void*));
    memcpy(&b->daxpy, &a[1], sizeof(void*));
    memcpy(&b->dscal, &a[2], sizeof(void*));
    memcpy(&b->dcopy, &a[3], sizeof(void*));
    memcpy(&b->dnrm2, &a[4], sizeof(void*));
    memcpy(&b->dpotrf, &a[5], sizeof(void*));
    memcpy(&b->dtrtrs, &a[6], sizeof(void*));
    *out = b;
    return 0;
}

/* advance one run until it wants (f, g) or stops: 1 evaluation wanted, 0 finished */
static int own_advance(OwnRun* r, const double* lo, const double* hi, const int* nbd, double factr, double pgtol, int maxls,
                       long maxiter, long maxfun, const lbfgsb_blas* blas) {
    while (!r->done) {
        lbfgsb_setulb(LB_N, LB_M, r->x, lo, hi, nbd, r->f, r->g, factr, pgtol, r->wa, r->iwa, r->task, r->lsave, r->isave,
                      r->dsave, maxls, r->ln_task, blas);
        if (r->task[0] == LB_FG) return 1;
        if (r->task[0] == LB_NEW_X) {
            r->nit += 1;
            if (r->nit >= maxiter) { r->task[0] = LB_STOP; r->task[1] = 504; }
            else if (r->nfev > maxfun) { r->task[0] = LB_STOP; r->task[1] = 502; }
        } else {
            r->done = 1;
        }
    }
    return 0;
}

static PyObject* lockstep_run_own(PyObject* self, PyObject* a) {
    PyObject *lat_o, *blas_o;
    Py_buffer X, BND;
    unsigned long long fn_addr, ctx_addr;
    int set_id, window;
    double dt;
    long maxiter, maxfun;
    if (!PyArg_ParseTuple(a, "Ow*y*KKiidllO", &lat_o, &X, &BND, &fn_addr, &ctx_addr, &set_id, &window, &dt, &maxiter, &maxfun,
                          &blas_o))
        return NULL;
    PyObject* ret = NULL;
    OwnRun* r = NULL;
    lbfgsb_blas btab;
    const lbfgsb_blas* blas = NULL;
    if (parse_blas(blas_o, &btab, &blas) < 0) goto out;
    if (!PyList_Check(lat_o)) { PyErr_SetString(PyExc_TypeError, "latents must be a list"); goto out; }
    const int n = (int)PyList_GET_SIZE(lat_o);
    if (n < 1 || n > MAX_RUNS) { PyErr_SetString(PyExc_ValueError, "1 .. 64 runs"); goto out; }
    if (X.len != (Py_ssize_t)(n * LB_N * sizeof(double)) || BND.len != (Py_ssize_t)(LB_N * 2 * sizeof(double))) {
        PyErr_SetString(PyExc_ValueError, "X: (n_runs, 3) float64; log_bounds: (3, 2) float64");
        goto out;
    }
    objective_fn fn = (objective_fn)(uintptr_t)fn_addr;
    void* ctx = (void*)(uintptr_t)ctx_addr;
    r = (OwnRun*)calloc((size_t)n, sizeof(OwnRun));
    if (!r) { PyErr_NoMemory(); goto out; }
    double lo[LB_N], hi[LB_N];
    int nbd[LB_N];
    for (int j = 0; j < LB_N; ++j) {
        lo[j] = ((const double*)BND.buf)[2 * j];
        hi[j] = ((const double*)BND.buf)[2 * j + 1];
        nbd[j] = 2;
    }
    double* Xp = (double*)X.buf;
    for (int i = 0; i < n; ++i) {
        r[i].latent = (int)PyLong_AsLong(PyList_GET_ITEM(lat_o, i));
        if (PyErr_Occurred()) goto out;
        for (int j = 0; j < LB_N; ++j) {  /* np.clip(x0, lo, hi) of SciPy's driver */
            double v = Xp[LB_N * i + j];
            v = v < lo[j] ? lo[j] : v;
            v = v > hi[j] ? hi[j] : v;
            r[i].x[j] = v;
        }
    }
    const double factr = 2.2204460492503131e-09 / 2.220446049250313e-16, pgtol = 1e-5;
    int status = 0;
    {
        int active[MAX_RUNS], n_act = n, lat[16];
        double logp[3 * 16], ll[16], dll[3 * 16];
        for (int i = 0; i < n; ++i) active[i] = i;
        Py_BEGIN_ALLOW_THREADS
        for (;;) {
            int m = 0;
            for (int q = 0; q < n_act; ++q) {
                const int k = active[q];
                if (own_advance(&r[k], lo, hi, nbd, factr, pgtol, 20, maxiter, maxfun, blas)) active[m++] = k;
            }
            n_act = m;
            if (n_act == 0) break;
            for (int base = 0; base < n_act && status == 0; base += 16) {
                const int cnt = n_act - base < 16 ? n_act - base : 16;
                for (int q = 0; q < cnt; ++q) {
                    const OwnRun* rk = &r[active[base + q]];
                    lat[q] = rk->latent;
                    logp[3 * q + 0] = rk->x[0]; logp[3 * q + 1] = rk->x[1]; logp[3 * q + 2] = rk->x[2];
                }
                status = fn(ctx, set_id, window, dt, cnt, lat, logp, ll, dll);
                if (status != 0) break;
                for (int q = 0; q < cnt; ++q) {
                    OwnRun* rk = &r[active[base + q]];
                    rk->f = -ll[q];
                    rk->g[0] = -dll[3 * q + 0]; rk->g[1] = -dll[3 * q + 1]; rk->g[2] = -dll[3 * q + 2];
                    rk->nfev += 1;
                }
            }
            if (status != 0) break;
        }
        Py_END_ALLOW_THREADS
    }
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < LB_N; ++j) Xp[LB_N * i + j] = r[i].x[j];
    ret = PyLong_FromLong(status);
out:
    free(r);
    PyBuffer_Release(&X);
    PyBuffer_Release(&BND);
    return ret;
}

/* setulb(m, x, l, u, nbd, f, g, factr, pgtol, wa, iwa, task, lsave, isave, dsave, maxls, ln_task[, blas]): ONE step of
 * lbfgsb.c's reverse communication with the argument list of scipy.optimize._lbfgsb.setulb (tests drive the two side by
 * side, call for call, and compare every array) */
static PyObject* lockstep_setulb(PyObject* self, PyObject* a) {
    int m, maxls;
    double f, factr, pgtol;
    Py_buffer x, l, u, nbd, g, wa, iwa, task, lsave, isave, dsave, ln_task;
    PyObject* blas_o = Py_None;
    if (!PyArg_ParseTuple(a, "iw*y*y*y*dw*ddw*w*w*w*w*w*iw*|O", &m, &x, &l, &u, &nbd, &f, &g, &factr, &pgtol, &wa, &iwa, &task,
                          &lsave, &isave, &dsave, &maxls, &ln_task, &blas_o))
        return NULL;
    PyObject* ret = NULL;
    lbfgsb_blas btab;
    const lbfgsb_blas* blas = NULL;
    const int n = (int)(x.len / (Py_ssize_t)sizeof(double));
    if (parse_blas(blas_o, &btab, &blas) < 0) goto out;
    if (n < 1 || m < 1 || l.len != x.len || u.len != x.len || g.len != x.len || nbd.len != n * (Py_ssize_t)sizeof(int) ||
        wa.len < (Py_ssize_t)sizeof(double) * (2 * m * n + 5 * n + 11 * m * m + 8 * m) ||
        iwa.len < (Py_ssize_t)sizeof(int) * 3 * n || task.len < 8 || ln_task.len < 8 || lsave.len < 16 || isave.len < 176 ||
        dsave.len < 232) {
        PyErr_SetString(PyExc_ValueError, "setulb: array sizes (float64 x, l, u, g, wa, dsave[29]; int32 nbd, iwa[3n], task[2], "
                                          "lsave[4], isave[44], ln_task[2])");
        goto out;
    }
    lbfgsb_setulb(n, m, (double*)x.buf, (const double*)l.buf, (const double*)u.buf, (const int*)nbd.buf, f, (double*)g.buf, factr,
                  pgtol, (double*)wa.buf, (int*)iwa.buf, (int*)task.buf, (int*)lsave.buf, (int*)isave.buf, (double*)dsave.buf,
                  maxls, (int*)ln_task.buf, blas);
    Py_INCREF(Py_None);
    ret = Py_None;
out:
    PyBuffer_Release(&x); PyBuffer_Release(&l); PyBuffer_Release(&u); PyBuffer_Release(&nbd); PyBuffer_Release(&g);
    PyBuffer_Release(&wa); PyBuffer_Release(&iwa); PyBuffer_Release(&task); PyBuffer_Release(&lsave);
    PyBuffer_Release(&isave); PyBuffer_Release(&dsave); PyBuffer_Release(&ln_task);
    return ret;
}

static PyMethodDef methods[] = {
    {"run", lockstep_run, METH_VARARGS, "lock-step L-BFGS-B over SciPy's setulb with the objective called by address"},
    {"run_own", lockstep_run_own, METH_VARARGS, "lock-step L-BFGS-B with the optimiser of lbfgsb.c, objective called by address"},
    {"setulb", lockstep_setulb, METH_VARARGS, "one reverse-communication step of lbfgsb.c (scipy.optimize._lbfgsb.setulb's arguments)"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_lockstep", NULL, -1, methods};

PyMODINIT_FUNC PyInit__lockstep(void) { return PyModule_Create(&moduledef); }
