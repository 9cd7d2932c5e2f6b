/* Bounded limited-memory BFGS (L-BFGS-B) as a reverse-communication routine in C -- the optimiser of the H-step.
 *
 * The reference minimises -elbo over log(sigma^2, omega, eps) with scipy.optimize.minimize(..., bounds=...)
 * (vlgp/gp.py:114), i.e. SciPy's L-BFGS-B: Byrd, Lu, Nocedal, Zhu, "A limited memory algorithm for bound constrained
 * optimization" (SIAM J. Sci. Comput. 16, 1995); Zhu, Byrd, Lu, Nocedal, Algorithm 778 (ACM TOMS 23, 1997); Morales,
 * Nocedal, "Remark on Algorithm 778" (ACM TOMS 38, 2011: version 3.0, the subspace-minimisation refinement); line search
 * of More' and Thuente (ACM TOMS 20, 1994).  This file restates that published algorithm with SciPy 1.15's calling
 * convention (integer task codes, `maxls`) so that the two can be driven call for call with the same arguments:
 * tests/test_lockstep_lbfgsb.py holds them to identical iterates.
 *
 * The dense kernels the algorithm needs (dot products, the Cholesky factorisations and triangular solves of the
 * 2m x 2m middle matrices) go through a small table of BLAS / LAPACK entry points: `lbfgsb_own_blas()` (portable loops
 * in this file, netlib operation order) or whatever the caller binds -- vlgp_amd/gp.py binds the OpenBLAS that SciPy itself
 * ships when it finds it, which makes the iterates bit-identical to scipy.optimize's on that machine; with the own loops
 * they agree to rounding.
 */
#ifndef VLGP_LBFGSB_H
#define VLGP_LBFGSB_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct lbfgsb_blas {
    double (*ddot)(const int* n, const double* x, const int* incx, const double* y, const int* incy);
    void (*daxpy)(const int* n, const double* a, const double* x, const int* incx, double* y, const int* incy);
    void (*dscal)(const int* n, const double* a, double* x, const int* incx);
    void (*dcopy)(const int* n, const double* x, const int* incx, double* y, const int* incy);
    double (*dnrm2)(const int* n, const double* x, const int* incx);
    void (*dpotrf)(const char* uplo, const int* n, double* a, const int* lda, int* info);
    void (*dtrtrs)(const char* uplo, const char* trans, const char* diag, const int* n, const int* nrhs, const double* a,
                   const int* lda, double* b, const int* ldb, int* info);
} lbfgsb_blas;

const lbfgsb_blas* lbfgsb_own_blas(void);

/* task[0]: what the caller has to do / what happened (SciPy 1.15's codes) */
enum { LB_START = 0, LB_NEW_X = 1, LB_RESTART = 2, LB_FG = 3, LB_CONVERGENCE = 4, LB_STOP = 5, LB_WARNING = 6,
       LB_ERROR = 7, LB_ABNORMAL = 8 };

/* One step of the reverse communication (same arguments as scipy.optimize._lbfgsb.setulb):
 *   n, m          problem size, number of corrections kept
 *   x, l, u, nbd  iterate (in / out), bounds, bound kinds (0 none, 1 lower, 2 both, 3 upper)
 *   f, g          objective and gradient at x when task[0] == LB_FG on entry to the NEXT call
 *   factr, pgtol  stopping rules: (f_k - f_k+1) / max(|f_k|, |f_k+1|, 1) <= factr * epsmch; max |proj g_i| <= pgtol
 *   wa            2 m n + 5 n + 11 m m + 8 m doubles;  iwa: 3 n ints
 *   task[2], lsave[4], isave[44], dsave[29], ln_task[2]: state, zero before the first call (task[0] = LB_START)
 *   maxls         function evaluations per line search
 */
void lbfgsb_setulb(int n, int m, double* x, const double* l, const double* u, const int* nbd, double f, double* g,
                   double factr, double pgtol, double* wa, int* iwa, int* task, int* lsave, int* isave, double* dsave,
                   int maxls, int* ln_task, const lbfgsb_blas* blas);

#ifdef __cplusplus
}
#endif
#endif
