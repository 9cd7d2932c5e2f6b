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

static PyMethodDef methods[] = {
    {"run", lockstep_run, METH_VARARGS, "lock-step L-BFGS-B over SciPy's setulb with the objective called by address"},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef moduledef = {PyModuleDef_HEAD_INIT, "_lockstep", NULL, -1, methods};

PyMODINIT_FUNC PyInit__lockstep(void) { return PyModule_Create(&moduledef); }
