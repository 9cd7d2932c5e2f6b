/* L-BFGS-B, reverse communication: see lbfgsb.h.  Host control logic of the H-step; no model arithmetic lives here.
 *
 * Notation as in the papers: S, Y the last `col` correction pairs (columns of ws, wy, a circular list starting at
 * `head`), theta the scaling of the initial matrix, sy = S'Y, ss = S'S, wt the Cholesky factor of
 * theta S'S + L D^-1 L', wn the factored 2 col x 2 col matrix of the subspace problem.  Matrices are column-major.
 *
 * Compiled with -ffp-contract=off: every floating-point expression is evaluated as written (no fused multiply-adds),
 * which is what makes the iterates reproducible against SciPy's build.
 */
#include "lbfgsb.h"

#include <float.h>
#include <math.h>
#include <string.h>

/* secondary task codes (task[1]) */
enum { T_NONE = 0, T_FG_START = 301, T_FG_LNSRCH = 302, T_CONV_PGTOL = 401, T_CONV_FACTR = 402, T_STOP_CPU = 501,
       T_STOP_NFEV = 502, T_STOP_PG = 503, T_STOP_ITER = 504, T_STOP_CALLBACK = 505, T_WARN_ROUND = 601,
       T_WARN_STPMAX = 602, T_WARN_STPMIN = 603, T_WARN_XTOL = 604, T_ERR_NOFEAS = 701, T_ERR_FACTR = 702,
       T_ERR_FTOL = 703, T_ERR_GTOL = 704, T_ERR_XTOL = 705, T_ERR_STP_LT_MIN = 706, T_ERR_STP_GT_MAX = 707,
       T_ERR_STPMIN_NEG = 708, T_ERR_STPMAX_LT_MIN = 709, T_ERR_INITIAL_G = 710, T_ERR_M = 711, T_ERR_N = 712,
       T_ERR_NBD = 713 };

static const int I1 = 1;

/* ---- portable BLAS / LAPACK subset (netlib operation order) ------------------------------------------------ */
static double own_ddot(const int* n, const double* x, const int* incx, const double* y, const int* incy) {
    double s = 0.0;
    for (int i = 0; i < *n; ++i) s += x[i * *incx] * y[i * *incy];
    return s;
}
static void own_daxpy(const int* n, const double* a, const double* x, const int* incx, double* y, const int* incy) {
    if (*a == 0.0) return;
    for (int i = 0; i < *n; ++i) y[i * *incy] += *a * x[i * *incx];
}
static void own_dscal(const int* n, const double* a, double* x, const int* incx) {
    for (int i = 0; i < *n; ++i) x[i * *incx] *= *a;
}
static void own_dcopy(const int* n, const double* x, const int* incx, double* y, const int* incy) {
    for (int i = 0; i < *n; ++i) y[i * *incy] = x[i * *incx];
}
static double own_dnrm2(const int* n, const double* x, const int* incx) {
    double scale = 0.0, ssq = 1.0;
    for (int i = 0; i < *n; ++i) {
        const double v = x[i * *incx];
        if (v != 0.0) {
            const double a = fabs(v);
            if (scale < a) {
                ssq = 1.0 + ssq * (scale / a) * (scale / a);
                scale = a;
            } else {
                ssq += (a / scale) * (a / scale);
            }
        }
    }
    return scale * sqrt(ssq);
}
/* upper Cholesky A = U'U in place (unblocked, column by column) */
static void own_dpotrf(const char* uplo, const int* n, double* a, const int* lda, int* info) {
    (void)uplo;
    const int N = *n, ld = *lda;
    *info = 0;
    for (int j = 0; j < N; ++j) {
        double ajj = a[j + j * ld];
        for (int k = 0; k < j; ++k) ajj -= a[k + j * ld] * a[k + j * ld];
        if (!(ajj > 0.0)) {
            a[j + j * ld] = ajj;
            *info = j + 1;
            return;
        }
        ajj = sqrt(ajj);
        a[j + j * ld] = ajj;
        for (int c = j + 1; c < N; ++c) {
            double s = a[j + c * ld];
            for (int k = 0; k < j; ++k) s -= a[k + j * ld] * a[k + c * ld];
            a[j + c * ld] = s / ajj;
        }
    }
}
/* upper triangular solves, non-unit diagonal: trans 'N': U x = b, 'T': U'x = b */
static void own_dtrtrs(const char* uplo, const char* trans, const char* diag, const int* n, const int* nrhs, const double* a,
                       const int* lda, double* b, const int* ldb, int* info) {
    (void)uplo; (void)diag;
    const int N = *n, ld = *lda;
    *info = 0;
    for (int i = 0; i < N; ++i)
        if (a[i + i * ld] == 0.0) { *info = i + 1; return; }
    for (int c = 0; c < *nrhs; ++c) {
        double* x = b + (long)c * *ldb;
        if (*trans == 'N' || *trans == 'n') {
            for (int j = N - 1; j >= 0; --j) {
                if (x[j] != 0.0) {
                    x[j] /= a[j + j * ld];
                    for (int i = 0; i < j; ++i) x[i] -= x[j] * a[i + j * ld];
                }
            }
        } else {
            for (int j = 0; j < N; ++j) {
                double s = x[j];
                for (int i = 0; i < j; ++i) s -= a[i + j * ld] * x[i];
                x[j] = s / a[j + j * ld];
            }
        }
    }
}
static const lbfgsb_blas OWN = {own_ddot, own_daxpy, own_dscal, own_dcopy, own_dnrm2, own_dpotrf, own_dtrtrs};
const lbfgsb_blas* lbfgsb_own_blas(void) { return &OWN; }

/* ---- the pieces ------------------------------------------------------------------------------------------- */
typedef struct {
    int n, m;
    double *ws, *wy, *sy, *ss, *wt, *wn, *snd, *z, *r, *d, *t, *xp, *wa;
    int *index, *iwhere, *indx2;
    const lbfgsb_blas* B;
} Work;

#define WS(i, j) W->ws[(i) + (long)(j) * W->n]
#define WY(i, j) W->wy[(i) + (long)(j) * W->n]
#define SY(i, j) W->sy[(i) + (long)(j) * W->m]
#define SS(i, j) W->ss[(i) + (long)(j) * W->m]
#define WT(i, j) W->wt[(i) + (long)(j) * W->m]
#define WN(i, j) W->wn[(i) + (long)(j) * 2 * W->m]
#define WN1(i, j) W->snd[(i) + (long)(j) * 2 * W->m]

/* project x onto the box, classify the variables */
static void active(const Work* W, const double* l, const double* u, const int* nbd, double* x, int* prjctd, int* cnstnd,
                   int* boxed) {
    const int n = W->n;
    *prjctd = 0; *cnstnd = 0; *boxed = 1;
    for (int i = 0; i < n; ++i) {
        if (nbd[i] > 0) {
            if (nbd[i] <= 2 && x[i] <= l[i]) {
                if (x[i] < l[i]) { *prjctd = 1; x[i] = l[i]; }
            } else if (nbd[i] >= 2 && x[i] >= u[i]) {
                if (x[i] > u[i]) { *prjctd = 1; x[i] = u[i]; }
            }
        }
    }
    for (int i = 0; i < n; ++i) {
        if (nbd[i] != 2) *boxed = 0;
        if (nbd[i] == 0) {
            W->iwhere[i] = -1;
        } else {
            *cnstnd = 1;
            W->iwhere[i] = (nbd[i] == 2 && u[i] - l[i] <= 0.0) ? 3 : 0;
        }
    }
}

/* infinity norm of the projected gradient */
static double projgr(int n, const double* l, const double* u, const int* nbd, const double* x, const double* g) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
        double gi = g[i];
        if (nbd[i] != 0) {
            if (gi < 0.0) {
                if (nbd[i] >= 2) gi = fmax(x[i] - u[i], gi);
            } else {
                if (nbd[i] <= 2) gi = fmin(x[i] - l[i], gi);
            }
        }
        s = fmax(s, fabs(gi));
    }
    return s;
}

/* p = M v with M the 2 col x 2 col middle matrix of the compact form (through sy and the factor wt) */
static int bmv(const Work* W, int col, const double* v, double* p) {
    const int m = W->m;
    int info = 0;
    if (col == 0) return 0;
    p[col] = v[col];
    for (int i = 1; i < col; ++i) {
        double sum = 0.0;
        for (int k = 0; k < i; ++k) sum += SY(i, k) * v[k] / SY(k, k);
        p[col + i] = v[col + i] + sum;
    }
    W->B->dtrtrs("U", "T", "N", &col, &I1, W->wt, &m, p + col, &col, &info);
    if (info != 0) return info;
    for (int i = 0; i < col; ++i) p[i] = v[i] / sqrt(SY(i, i));
    W->B->dtrtrs("U", "N", "N", &col, &I1, W->wt, &m, p + col, &col, &info);
    if (info != 0) return info;
    for (int i = 0; i < col; ++i) p[i] = -p[i] / sqrt(SY(i, i));
    for (int i = 0; i < col; ++i) {
        double sum = 0.0;
        for (int k = i + 1; k < col; ++k) sum += SY(k, i) * p[col + k] / SY(i, i);
        p[i] += sum;
    }
    return 0;
}

/* heap of breakpoints: t[0 .. n) with their variable indices; on return the least one sits in t[n - 1] */
static void hpsolb(int n, double* t, int* iorder, int iheap) {
    if (iheap == 0) {
        for (int k = 2; k <= n; ++k) {
            const double ddum = t[k - 1];
            const int indxin = iorder[k - 1];
            int i = k;
            while (i > 1) {
                const int j = i / 2;
                if (ddum < t[j - 1]) {
                    t[i - 1] = t[j - 1];
                    iorder[i - 1] = iorder[j - 1];
                    i = j;
                } else {
                    break;
                }
            }
            t[i - 1] = ddum;
            iorder[i - 1] = indxin;
        }
    }
    if (n > 1) {
        int i = 1;
        const double out = t[0];
        const int indxou = iorder[0];
        const double ddum = t[n - 1];
        const int indxin = iorder[n - 1];
        for (;;) {
            int j = i + i;
            if (j <= n - 1) {
                if (t[j] < t[j - 1]) j = j + 1;
                if (t[j - 1] < ddum) {
                    t[i - 1] = t[j - 1];
                    iorder[i - 1] = iorder[j - 1];
                    i = j;
                    continue;
                }
            }
            break;
        }
        t[i - 1] = ddum;
        iorder[i - 1] = indxin;
        t[n - 1] = out;
        iorder[n - 1] = indxou;
    }
}

/* generalised Cauchy point along the projected steepest-descent path; xcp = W->z, c = W'(xcp - x) in wa[2m ..] */
static int cauchy(const Work* W, const double* x, const double* l, const double* u, const int* nbd, const double* g,
                  double theta, int col, int head, int* nseg, double sbgnrm, double epsmch) {
    const int n = W->n, m = W->m;
    double* p = W->wa;
    double* c = W->wa + 2 * m;
    double* wbp = W->wa + 4 * m;
    double* v = W->wa + 6 * m;
    double* xcp = W->z;
    double* d = W->d;
    double* t = W->t;
    int* iorder = W->indx2;
    int* iwhere = W->iwhere;
    const lbfgsb_blas* B = W->B;
    if (sbgnrm <= 0.0) {
        B->dcopy(&n, x, &I1, xcp, &I1);
        return 0;
    }
    int bnded = 1, nfree = n + 1, nbreak = 0, ibkmin = 0;
    double bkmin = 0.0, f1 = 0.0, tl = 0.0, tu = 0.0;
    const int col2 = 2 * col;
    for (int i = 0; i < col2; ++i) p[i] = 0.0;
    for (int i = 0; i < n; ++i) {
        const double neggi = -g[i];
        if (iwhere[i] != 3 && iwhere[i] != -1) {
            if (nbd[i] <= 2) tl = x[i] - l[i];
            if (nbd[i] >= 2) tu = u[i] - x[i];
            const int xlower = nbd[i] <= 2 && tl <= 0.0;
            const int xupper = nbd[i] >= 2 && tu <= 0.0;
            iwhere[i] = 0;
            if (xlower) {
                if (neggi <= 0.0) iwhere[i] = 1;
            } else if (xupper) {
                if (neggi >= 0.0) iwhere[i] = 2;
            } else {
                if (fabs(neggi) <= 0.0) iwhere[i] = -3;
            }
        }
        int pointr = head;
        if (iwhere[i] != 0 && iwhere[i] != -1) {
            d[i] = 0.0;
        } else {
            d[i] = neggi;
            f1 -= neggi * neggi;
            for (int j = 0; j < col; ++j) {
                p[j] += WY(i, pointr) * neggi;
                p[col + j] += WS(i, pointr) * neggi;
                pointr = (pointr + 1) % m;
            }
            if (nbd[i] <= 2 && nbd[i] != 0 && neggi < 0.0) {
                iorder[nbreak] = i;
                t[nbreak] = tl / (-neggi);
                if (nbreak == 0 || t[nbreak] < bkmin) { bkmin = t[nbreak]; ibkmin = nbreak; }
                ++nbreak;
            } else if (nbd[i] >= 2 && neggi > 0.0) {
                iorder[nbreak] = i;
                t[nbreak] = tu / neggi;
                if (nbreak == 0 || t[nbreak] < bkmin) { bkmin = t[nbreak]; ibkmin = nbreak; }
                ++nbreak;
            } else {
                --nfree;
                iorder[nfree - 1] = i;
                if (fabs(neggi) > 0.0) bnded = 0;
            }
        }
    }
    if (theta != 1.0) B->dscal(&col, &theta, p + col, &I1);
    B->dcopy(&n, x, &I1, xcp, &I1);
    if (nbreak == 0 && nfree == n + 1) return 0;
    for (int j = 0; j < col2; ++j) c[j] = 0.0;
    double f2 = -theta * f1;
    const double f2_org = f2;
    if (col > 0) {
        const int info = bmv(W, col, p, v);
        if (info != 0) return info;
        f2 -= B->ddot(&col2, v, &I1, p, &I1);
    }
    double dtm = -f1 / f2;
    double tsum = 0.0;
    *nseg = 1;
    int all_fixed = 0;
    if (nbreak > 0) {
        int nleft = nbreak, iter = 1, ibp;
        double tj = 0.0;
        for (;;) {
            const double tj0 = tj;
            if (iter == 1) {
                tj = bkmin;
                ibp = iorder[ibkmin];
            } else {
                if (iter == 2) {
                    if (ibkmin != nbreak - 1) {
                        t[ibkmin] = t[nbreak - 1];
                        iorder[ibkmin] = iorder[nbreak - 1];
                    }
                }
                hpsolb(nleft, t, iorder, iter - 2);
                tj = t[nleft - 1];
                ibp = iorder[nleft - 1];
            }
            const double dt = tj - tj0;
            if (dtm < dt) break;
            tsum += dt;
            --nleft;
            ++iter;
            const double dibp = d[ibp];
            d[ibp] = 0.0;
            double zibp;
            if (dibp > 0.0) {
                zibp = u[ibp] - x[ibp];
                xcp[ibp] = u[ibp];
                iwhere[ibp] = 2;
            } else {
                zibp = l[ibp] - x[ibp];
                xcp[ibp] = l[ibp];
                iwhere[ibp] = 1;
            }
            if (nleft == 0 && nbreak == n) {
                dtm = dt;
                all_fixed = 1;
                break;
            }
            ++*nseg;
            const double dibp2 = dibp * dibp;
            f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;
            f2 = f2 - theta * dibp2;
            if (col > 0) {
                B->daxpy(&col2, &dt, p, &I1, c, &I1);
                int pointr = head;
                for (int j = 0; j < col; ++j) {
                    wbp[j] = WY(ibp, pointr);
                    wbp[col + j] = theta * WS(ibp, pointr);
                    pointr = (pointr + 1) % m;
                }
                const int info = bmv(W, col, wbp, v);
                if (info != 0) return info;
                const double wmc = B->ddot(&col2, c, &I1, v, &I1);
                const double wmp = B->ddot(&col2, p, &I1, v, &I1);
                const double wmw = B->ddot(&col2, wbp, &I1, v, &I1);
                const double mdibp = -dibp;
                B->daxpy(&col2, &mdibp, wbp, &I1, p, &I1);
                f1 = f1 + dibp * wmc;
                f2 = f2 + 2.0 * dibp * wmp - dibp2 * wmw;
            }
            f2 = fmax(epsmch * f2_org, f2);
            if (nleft > 0) {
                dtm = -f1 / f2;
                continue;
            } else if (bnded) {
                f1 = 0.0;
                f2 = 0.0;
                dtm = 0.0;
            } else {
                dtm = -f1 / f2;
            }
            break;
        }
    }
    if (!all_fixed) {
        if (dtm <= 0.0) dtm = 0.0;
        tsum += dtm;
        B->daxpy(&n, &tsum, d, &I1, xcp, &I1);
    }
    if (col > 0) B->daxpy(&col2, &dtm, p, &I1, c, &I1);
    return 0;
}

/* free and active sets at the Cauchy point, entering and leaving variables */
static void freev(const Work* W, int* nfree, int* nenter, int* ileave, int* wrk, int updatd, int cnstnd, int iter) {
    const int n = W->n;
    int* index = W->index;
    int* indx2 = W->indx2;
    const int* iwhere = W->iwhere;
    *nenter = 0;
    *ileave = n + 1;
    if (iter > 0 && cnstnd) {
        for (int i = 0; i < *nfree; ++i) {
            const int k = index[i];
            if (iwhere[k] > 0) {
                --*ileave;
                indx2[*ileave - 1] = k;
            }
        }
        for (int i = *nfree; i < n; ++i) {
            const int k = index[i];
            if (iwhere[k] <= 0) {
                ++*nenter;
                indx2[*nenter - 1] = k;
            }
        }
    }
    *wrk = (*ileave < n + 1) || (*nenter > 0) || updatd;
    *nfree = 0;
    int iact = n + 1;
    for (int i = 0; i < n; ++i) {
        if (iwhere[i] <= 0) {
            ++*nfree;
            index[*nfree - 1] = i;
        } else {
            --iact;
            index[iact - 1] = i;
        }
    }
}

/* the factored matrix of the subspace problem */
static int formk(const Work* W, int nsub, int nenter, int ileave, int iupdat, int updatd, double theta, int col, int head) {
    const int n = W->n, m = W->m, m2 = 2 * m;
    const int* ind = W->index;
    const int* indx2 = W->indx2;
    const lbfgsb_blas* B = W->B;
    int upcl;
    if (updatd) {
        if (iupdat > m) {
            for (int jy = 0; jy < m - 1; ++jy) {
                const int js = m + jy;
                int len = m - 1 - jy;
                B->dcopy(&len, &WN1(jy + 1, jy + 1), &I1, &WN1(jy, jy), &I1);
                B->dcopy(&len, &WN1(js + 1, js + 1), &I1, &WN1(js, js), &I1);
                len = m - 1;
                B->dcopy(&len, &WN1(m + 1, jy + 1), &I1, &WN1(m, jy), &I1);
            }
        }
        int ipntr = head + col - 1;
        if (ipntr >= m) ipntr -= m;
        const int iy = col - 1, is = m + col - 1;
        int jpntr = head;
        for (int jy = 0; jy < col; ++jy) {
            const int js = m + jy;
            double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0;
            for (int k = 0; k < nsub; ++k) {
                const int k1 = ind[k];
                temp1 += WY(k1, ipntr) * WY(k1, jpntr);
            }
            for (int k = nsub; k < n; ++k) {
                const int k1 = ind[k];
                temp2 += WS(k1, ipntr) * WS(k1, jpntr);
                temp3 += WS(k1, ipntr) * WY(k1, jpntr);
            }
            WN1(iy, jy) = temp1;
            WN1(is, js) = temp2;
            WN1(is, jy) = temp3;
            jpntr = (jpntr + 1) % m;
        }
        const int jy = col - 1;
        jpntr = head + col - 1;
        if (jpntr >= m) jpntr -= m;
        ipntr = head;
        for (int i = 0; i < col; ++i) {
            const int is2 = m + i;
            double temp3 = 0.0;
            for (int k = 0; k < nsub; ++k) {
                const int k1 = ind[k];
                temp3 += WS(k1, ipntr) * WY(k1, jpntr);
            }
            ipntr = (ipntr + 1) % m;
            WN1(is2, jy) = temp3;
        }
        upcl = col - 1;
    } else {
        upcl = col;
    }
    int ipntr = head;
    for (int iy = 0; iy < upcl; ++iy) {
        const int is = m + iy;
        int jpntr = head;
        for (int jy = 0; jy <= iy; ++jy) {
            const int js = m + jy;
            double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0, temp4 = 0.0;
            for (int k = 0; k < nenter; ++k) {
                const int k1 = indx2[k];
                temp1 += WY(k1, ipntr) * WY(k1, jpntr);
                temp2 += WS(k1, ipntr) * WS(k1, jpntr);
            }
            for (int k = ileave - 1; k < n; ++k) {
                const int k1 = indx2[k];
                temp3 += WY(k1, ipntr) * WY(k1, jpntr);
                temp4 += WS(k1, ipntr) * WS(k1, jpntr);
            }
            WN1(iy, jy) = WN1(iy, jy) + temp1 - temp3;
            WN1(is, js) = WN1(is, js) - temp2 + temp4;
            jpntr = (jpntr + 1) % m;
        }
        ipntr = (ipntr + 1) % m;
    }
    ipntr = head;
    for (int is = m; is < m + upcl; ++is) {
        int jpntr = head;
        for (int jy = 0; jy < upcl; ++jy) {
            double temp1 = 0.0, temp3 = 0.0;
            for (int k = 0; k < nenter; ++k) {
                const int k1 = indx2[k];
                temp1 += WS(k1, ipntr) * WY(k1, jpntr);
            }
            for (int k = ileave - 1; k < n; ++k) {
                const int k1 = indx2[k];
                temp3 += WS(k1, ipntr) * WY(k1, jpntr);
            }
            if (is <= jy + m) WN1(is, jy) = WN1(is, jy) + temp1 - temp3;
            else WN1(is, jy) = WN1(is, jy) - temp1 + temp3;
            jpntr = (jpntr + 1) % m;
        }
        ipntr = (ipntr + 1) % m;
    }
    for (int iy = 0; iy < col; ++iy) {
        const int is = col + iy, is1 = m + iy;
        for (int jy = 0; jy <= iy; ++jy) {
            const int js = col + jy, js1 = m + jy;
            WN(jy, iy) = WN1(iy, jy) / theta;
            WN(js, is) = WN1(is1, js1) * theta;
        }
        for (int jy = 0; jy < iy; ++jy) WN(jy, is) = -WN1(is1, jy);
        for (int jy = iy; jy < col; ++jy) WN(jy, is) = WN1(is1, jy);
        WN(iy, iy) += SY(iy, iy);
    }
    int info = 0;
    B->dpotrf("U", &col, W->wn, &m2, &info);
    if (info != 0) return -1;
    B->dtrtrs("U", "T", "N", &col, &col, W->wn, &m2, &WN(0, col), &m2, &info);
    for (int is = col; is < 2 * col; ++is)
        for (int js = is; js < 2 * col; ++js) WN(is, js) += B->ddot(&col, &WN(0, is), &I1, &WN(0, js), &I1);
    B->dpotrf("U", &col, &WN(col, col), &m2, &info);
    if (info != 0) return -2;
    return 0;
}

/* r = -Z'(B (xcp - x) + g) */
static int cmprlb(const Work* W, const double* x, const double* g, double theta, int col, int head, int nfree, int cnstnd) {
    const int n = W->n, m = W->m;
    double* r = W->r;
    const double* z = W->z;
    const int* index = W->index;
    if (!cnstnd && col > 0) {
        for (int i = 0; i < n; ++i) r[i] = -g[i];
        return 0;
    }
    for (int i = 0; i < nfree; ++i) {
        const int k = index[i];
        r[i] = -theta * (z[k] - x[k]) - g[k];
    }
    const int info = bmv(W, col, W->wa + 2 * m, W->wa);
    if (info != 0) return -8;
    int pointr = head;
    for (int j = 0; j < col; ++j) {
        const double a1 = W->wa[j], a2 = theta * W->wa[col + j];
        for (int i = 0; i < nfree; ++i) {
            const int k = index[i];
            r[i] = r[i] + WY(k, pointr) * a1 + WS(k, pointr) * a2;  /* (left to right) */
        }
        pointr = (pointr + 1) % m;
    }
    return 0;
}

/* subspace minimisation over the free variables, then the projected refinement of version 3.0 */
static int subsm(const Work* W, int nsub, const double* l, const double* u, const int* nbd, double theta, const double* xx,
                 const double* gg, int col, int head, int* iword) {
    const int n = W->n, m = W->m, m2 = 2 * m, col2 = 2 * col;
    const int* ind = W->index;
    double* x = W->z;
    double* d = W->r;
    double* xp = W->xp;
    double* wv = W->wa;
    const lbfgsb_blas* B = W->B;
    int info = 0;
    if (nsub <= 0) return 0;
    int pointr = head;
    for (int i = 0; i < col; ++i) {
        double temp1 = 0.0, temp2 = 0.0;
        for (int j = 0; j < nsub; ++j) {
            const int k = ind[j];
            temp1 += WY(k, pointr) * d[j];
            temp2 += WS(k, pointr) * d[j];
        }
        wv[i] = temp1;
        wv[col + i] = theta * temp2;
        pointr = (pointr + 1) % m;
    }
    B->dtrtrs("U", "T", "N", &col2, &I1, W->wn, &m2, wv, &col2, &info);
    if (info != 0) return info;
    for (int i = 0; i < col; ++i) wv[i] = -wv[i];
    B->dtrtrs("U", "N", "N", &col2, &I1, W->wn, &m2, wv, &col2, &info);
    if (info != 0) return info;
    pointr = head;
    for (int jy = 0; jy < col; ++jy) {
        const int js = col + jy;
        for (int i = 0; i < nsub; ++i) {
            const int k = ind[i];
            d[i] = d[i] + WY(k, pointr) * wv[jy] / theta + WS(k, pointr) * wv[js];
        }
        pointr = (pointr + 1) % m;
    }
    const double rtheta = 1.0 / theta;
    B->dscal(&nsub, &rtheta, d, &I1);
    *iword = 0;
    B->dcopy(&n, x, &I1, xp, &I1);
    for (int i = 0; i < nsub; ++i) {
        const int k = ind[i];
        const double dk = d[i];
        double xk = x[k];
        if (nbd[k] != 0) {
            if (nbd[k] == 1) {
                x[k] = fmax(l[k], xk + dk);
                if (x[k] == l[k]) *iword = 1;
            } else if (nbd[k] == 2) {
                xk = fmax(l[k], xk + dk);
                x[k] = fmin(u[k], xk);
                if (x[k] == l[k] || x[k] == u[k]) *iword = 1;
            } else if (nbd[k] == 3) {
                x[k] = fmin(u[k], xk + dk);
                if (x[k] == u[k]) *iword = 1;
            }
        } else {
            x[k] = xk + dk;
        }
    }
    if (*iword == 0) return 0;
    double dd_p = 0.0;
    for (int i = 0; i < n; ++i) dd_p += (x[i] - xx[i]) * gg[i];
    if (dd_p > 0.0) {
        B->dcopy(&n, xp, &I1, x, &I1);
        double alpha = 1.0, temp1 = alpha;
        int ibd = 0;
        for (int i = 0; i < nsub; ++i) {
            const int k = ind[i];
            const double dk = d[i];
            if (nbd[k] != 0) {
                if (dk < 0.0 && nbd[k] <= 2) {
                    const double temp2 = l[k] - x[k];
                    if (temp2 >= 0.0) temp1 = 0.0;
                    else if (dk * alpha < temp2) temp1 = temp2 / dk;
                } else if (dk > 0.0 && nbd[k] >= 2) {
                    const double temp2 = u[k] - x[k];
                    if (temp2 <= 0.0) temp1 = 0.0;
                    else if (dk * alpha > temp2) temp1 = temp2 / dk;
                }
                if (temp1 < alpha) {
                    alpha = temp1;
                    ibd = i;
                }
            }
        }
        if (alpha < 1.0) {
            const double dk = d[ibd];
            const int k = ind[ibd];
            if (dk > 0.0) {
                x[k] = u[k];
                d[ibd] = 0.0;
            } else if (dk < 0.0) {
                x[k] = l[k];
                d[ibd] = 0.0;
            }
        }
        for (int i = 0; i < nsub; ++i) {
            const int k = ind[i];
            x[k] += alpha * d[i];
        }
    }
    return 0;
}

/* safeguarded cubic / quadratic step of the More'-Thuente search */
static void dcstep(double* stx, double* fx, double* dx, double* sty, double* fy, double* dy, double* stp, double fp, double dp,
                   int* brackt, double stpmin, double stpmax) {
    double gamma, p, q, r, s, stpc, stpf, stpq, theta;
    const double sgnd = dp * (*dx / fabs(*dx));
    if (fp > *fx) {
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = fmax(fmax(fabs(theta), fabs(*dx)), fabs(dp));
        gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp < *stx) gamma = -gamma;
        p = (gamma - *dx) + theta;
        q = ((gamma - *dx) + gamma) + dp;
        r = p / q;
        stpc = *stx + r * (*stp - *stx);
        stpq = *stx + ((*dx / ((*fx - fp) / (*stp - *stx) + *dx)) / 2.0) * (*stp - *stx);
        if (fabs(stpc - *stx) < fabs(stpq - *stx)) stpf = stpc;
        else stpf = stpc + (stpq - stpc) / 2.0;
        *brackt = 1;
    } else if (sgnd < 0.0) {
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = fmax(fmax(fabs(theta), fabs(*dx)), fabs(dp));
        gamma = s * sqrt((theta / s) * (theta / s) - (*dx / s) * (dp / s));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = ((gamma - dp) + gamma) + *dx;
        r = p / q;
        stpc = *stp + r * (*stx - *stp);
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc;
        else stpf = stpq;
        *brackt = 1;
    } else if (fabs(dp) < fabs(*dx)) {
        theta = 3.0 * (*fx - fp) / (*stp - *stx) + *dx + dp;
        s = fmax(fmax(fabs(theta), fabs(*dx)), fabs(dp));
        gamma = s * sqrt(fmax(0.0, (theta / s) * (theta / s) - (*dx / s) * (dp / s)));
        if (*stp > *stx) gamma = -gamma;
        p = (gamma - dp) + theta;
        q = (gamma + (*dx - dp)) + gamma;
        r = p / q;
        if (r < 0.0 && gamma != 0.0) stpc = *stp + r * (*stx - *stp);
        else if (*stp > *stx) stpc = stpmax;
        else stpc = stpmin;
        stpq = *stp + (dp / (dp - *dx)) * (*stx - *stp);
        if (*brackt) {
            if (fabs(stpc - *stp) < fabs(stpq - *stp)) stpf = stpc;
            else stpf = stpq;
            if (*stp > *stx) stpf = fmin(*stp + 0.66 * (*sty - *stp), stpf);
            else stpf = fmax(*stp + 0.66 * (*sty - *stp), stpf);
        } else {
            if (fabs(stpc - *stp) > fabs(stpq - *stp)) stpf = stpc;
            else stpf = stpq;
            stpf = fmin(stpmax, stpf);
            stpf = fmax(stpmin, stpf);
        }
    } else {
        if (*brackt) {
            theta = 3.0 * (fp - *fy) / (*sty - *stp) + *dy + dp;
            s = fmax(fmax(fabs(theta), fabs(*dy)), fabs(dp));
            gamma = s * sqrt((theta / s) * (theta / s) - (*dy / s) * (dp / s));
            if (*stp > *sty) gamma = -gamma;
            p = (gamma - dp) + theta;
            q = ((gamma - dp) + gamma) + *dy;
            r = p / q;
            stpc = *stp + r * (*sty - *stp);
            stpf = stpc;
        } else if (*stp > *stx) {
            stpf = stpmax;
        } else {
            stpf = stpmin;
        }
    }
    if (fp > *fx) {
        *sty = *stp;
        *fy = fp;
        *dy = dp;
    } else {
        if (sgnd < 0.0) {
            *sty = *stx;
            *fy = *fx;
            *dy = *dx;
        }
        *stx = *stp;
        *fx = fp;
        *dx = dp;
    }
    *stp = stpf;
}

/* More'-Thuente line search, reverse communication; state in isave[0 .. 2), dsave[0 .. 13) */
static void dcsrch(double f, double g, double* stp, double ftol, double gtol, double xtol, double stpmin, double stpmax,
                   int* task, int* isave, double* dsave) {
    const double p5 = 0.5, p66 = 0.66, xtrapl = 1.1, xtrapu = 4.0;
    int brackt, stage;
    double finit, ftest, fm, fx, fxm, fy, fym, ginit, gtest, gm, gx, gxm, gy, gym, stx, sty, stmin, stmax, width, width1;
    if (task[0] == LB_START) {
        if (*stp < stpmin) { task[0] = LB_ERROR; task[1] = T_ERR_STP_LT_MIN; }
        if (*stp > stpmax) { task[0] = LB_ERROR; task[1] = T_ERR_STP_GT_MAX; }
        if (g >= 0.0) { task[0] = LB_ERROR; task[1] = T_ERR_INITIAL_G; }
        if (ftol < 0.0) { task[0] = LB_ERROR; task[1] = T_ERR_FTOL; }
        if (gtol < 0.0) { task[0] = LB_ERROR; task[1] = T_ERR_GTOL; }
        if (xtol < 0.0) { task[0] = LB_ERROR; task[1] = T_ERR_XTOL; }
        if (stpmin < 0.0) { task[0] = LB_ERROR; task[1] = T_ERR_STPMIN_NEG; }
        if (stpmax < stpmin) { task[0] = LB_ERROR; task[1] = T_ERR_STPMAX_LT_MIN; }
        if (task[0] == LB_ERROR) return;
        brackt = 0;
        stage = 1;
        finit = f;
        ginit = g;
        gtest = ftol * ginit;
        width = stpmax - stpmin;
        width1 = width / p5;
        stx = 0.0; fx = finit; gx = ginit;
        sty = 0.0; fy = finit; gy = ginit;
        stmin = 0.0;
        stmax = *stp + xtrapu * *stp;
        task[0] = LB_FG;
        task[1] = T_NONE;
        goto save;
    }
    brackt = isave[0];
    stage = isave[1];
    ginit = dsave[0]; gtest = dsave[1]; gx = dsave[2]; gy = dsave[3]; finit = dsave[4]; fx = dsave[5]; fy = dsave[6];
    stx = dsave[7]; sty = dsave[8]; stmin = dsave[9]; stmax = dsave[10]; width = dsave[11]; width1 = dsave[12];

    ftest = finit + *stp * gtest;
    if (stage == 1 && f <= ftest && g >= 0.0) stage = 2;
    if (brackt && (*stp <= stmin || *stp >= stmax)) { task[0] = LB_WARNING; task[1] = T_WARN_ROUND; }
    if (brackt && stmax - stmin <= xtol * stmax) { task[0] = LB_WARNING; task[1] = T_WARN_XTOL; }
    if (*stp == stpmax && f <= ftest && g <= gtest) { task[0] = LB_WARNING; task[1] = T_WARN_STPMAX; }
    if (*stp == stpmin && (f > ftest || g >= gtest)) { task[0] = LB_WARNING; task[1] = T_WARN_STPMIN; }
    if (f <= ftest && fabs(g) <= gtol * (-ginit)) { task[0] = LB_CONVERGENCE; task[1] = T_NONE; }
    if (task[0] == LB_WARNING || task[0] == LB_CONVERGENCE) goto save;

    if (stage == 1 && f <= fx && f > ftest) {
        fm = f - *stp * gtest;
        fxm = fx - stx * gtest;
        fym = fy - sty * gtest;
        gm = g - gtest;
        gxm = gx - gtest;
        gym = gy - gtest;
        dcstep(&stx, &fxm, &gxm, &sty, &fym, &gym, stp, fm, gm, &brackt, stmin, stmax);
        fx = fxm + stx * gtest;
        fy = fym + sty * gtest;
        gx = gxm + gtest;
        gy = gym + gtest;
    } else {
        dcstep(&stx, &fx, &gx, &sty, &fy, &gy, stp, f, g, &brackt, stmin, stmax);
    }
    if (brackt) {
        if (fabs(sty - stx) >= p66 * width1) *stp = stx + p5 * (sty - stx);
        width1 = width;
        width = fabs(sty - stx);
    }
    if (brackt) {
        stmin = fmin(stx, sty);
        stmax = fmax(stx, sty);
    } else {
        stmin = *stp + xtrapl * (*stp - stx);
        stmax = *stp + xtrapu * (*stp - stx);
    }
    *stp = fmax(*stp, stpmin);
    *stp = fmin(*stp, stpmax);
    if ((brackt && (*stp <= stmin || *stp >= stmax)) || (brackt && stmax - stmin <= xtol * stmax)) *stp = stx;
    task[0] = LB_FG;
    task[1] = T_NONE;
save:
    isave[0] = brackt;
    isave[1] = stage;
    dsave[0] = ginit; dsave[1] = gtest; dsave[2] = gx; dsave[3] = gy; dsave[4] = finit; dsave[5] = fx; dsave[6] = fy;
    dsave[7] = stx; dsave[8] = sty; dsave[9] = stmin; dsave[10] = stmax; dsave[11] = width; dsave[12] = width1;
}

/* one correction pair into S, Y, S'Y, S'S */
static void matupd(const Work* W, int* itail, int iupdat, int* col, int* head, double* theta, double rr, double dr, double stp,
                   double dtd) {
    const int n = W->n, m = W->m;
    const lbfgsb_blas* B = W->B;
    if (iupdat <= m) {
        *col = iupdat;
        *itail = (*head + iupdat - 1) % m;
    } else {
        *itail = (*itail + 1) % m;
        *head = (*head + 1) % m;
    }
    B->dcopy(&n, W->d, &I1, &WS(0, *itail), &I1);
    B->dcopy(&n, W->r, &I1, &WY(0, *itail), &I1);
    *theta = rr / dr;
    if (iupdat > m) {
        for (int j = 0; j < *col - 1; ++j) {
            int len = j + 1;
            B->dcopy(&len, &SS(1, j + 1), &I1, &SS(0, j), &I1);
            len = *col - (j + 1);
            B->dcopy(&len, &SY(j + 1, j + 1), &I1, &SY(j, j), &I1);
        }
    }
    int pointr = *head;
    for (int j = 0; j < *col - 1; ++j) {
        SY(*col - 1, j) = B->ddot(&n, W->d, &I1, &WY(0, pointr), &I1);
        SS(j, *col - 1) = B->ddot(&n, &WS(0, pointr), &I1, W->d, &I1);
        pointr = (pointr + 1) % m;
    }
    if (stp == 1.0) SS(*col - 1, *col - 1) = dtd;
    else SS(*col - 1, *col - 1) = stp * stp * dtd;
    SY(*col - 1, *col - 1) = dr;
}

/* T = theta S'S + L D^-1 L', factored */
static int formt(const Work* W, int col, double theta) {
    const int m = W->m;
    for (int j = 0; j < col; ++j) WT(0, j) = theta * SS(0, j);
    for (int i = 1; i < col; ++i) {
        for (int j = i; j < col; ++j) {
            const int k1 = (i < j ? i : j);
            double ddum = 0.0;
            for (int k = 0; k < k1; ++k) ddum += SY(i, k) * SY(j, k) / SY(k, k);
            WT(i, j) = ddum + theta * SS(i, j);
        }
    }
    int info = 0;
    W->B->dpotrf("U", &col, W->wt, &m, &info);
    return info != 0 ? -3 : 0;
}

/* ---- the driver -------------------------------------------------------------------------------------------- */
enum { IS_NINTOL = 22, IS_IBACK = 24, IS_NSKIP, IS_HEAD, IS_COL, IS_ITAIL, IS_ITER, IS_IUPDAT, IS_NSEG, IS_NFGV, IS_INFO,
       IS_IFUN, IS_IWORD, IS_NFREE, IS_NACT, IS_ILEAVE, IS_NENTER, IS_LS0 = 40 /* dcsrch: brackt, stage */ };
enum { DS_THETA = 0, DS_FOLD, DS_TOL, DS_DNORM, DS_EPSMCH, DS_GD = 10, DS_STPMX, DS_SBGNRM, DS_STP, DS_GDOLD, DS_DTD,
       DS_LS0 = 16 /* dcsrch: 13 doubles */ };

void lbfgsb_setulb(int n, int m, double* x, const double* l, const double* u, const int* nbd, double f, double* g,
                   double factr, double pgtol, double* wa, int* iwa, int* task, int* lsave, int* isave, double* dsave,
                   int maxls, int* ln_task, const lbfgsb_blas* blas) {
    Work Wk;
    Work* W = &Wk;
    W->n = n; W->m = m; W->B = blas ? blas : &OWN;
    W->ws = wa;
    W->wy = W->ws + (long)m * n;
    W->sy = W->wy + (long)m * n;
    W->ss = W->sy + m * m;
    W->wt = W->ss + m * m;
    W->wn = W->wt + m * m;
    W->snd = W->wn + 4 * m * m;
    W->z = W->snd + 4 * m * m;
    W->r = W->z + n;
    W->d = W->r + n;
    W->t = W->d + n;
    W->xp = W->t + n;
    W->wa = W->xp + n;
    W->index = iwa;
    W->iwhere = iwa + n;
    W->indx2 = iwa + 2 * n;
    const lbfgsb_blas* B = W->B;

    int prjctd, cnstnd, boxed, updatd;
    int nintol, iback, nskip, head, col, itail, iter, iupdat, nseg, nfgv, info, ifun, iword, nfree, nact, ileave, nenter;
    double theta, fold, tol, dnorm, epsmch, gd, stpmx, sbgnrm, stp, gdold, dtd;
    int wrk = 0;

    if (task[0] == LB_START) {
        epsmch = DBL_EPSILON;
        col = 0; head = 0; theta = 1.0; iupdat = 0; updatd = 0;
        iback = 0; itail = 0; iword = 0; nact = 0; ileave = 0; nenter = 0;
        fold = 0.0; dnorm = 0.0; gd = 0.0; stpmx = 0.0; sbgnrm = 0.0; stp = 0.0; gdold = 0.0; dtd = 0.0;
        iter = 0; nfgv = 0; nseg = 0; nintol = 0; nskip = 0; nfree = n; ifun = 0;
        tol = factr * epsmch;
        info = 0;
        /* argument check */
        if (n <= 0) { task[0] = LB_ERROR; task[1] = T_ERR_N; }
        if (m <= 0) { task[0] = LB_ERROR; task[1] = T_ERR_M; }
        if (factr < 0.0) { task[0] = LB_ERROR; task[1] = T_ERR_FACTR; }
        for (int i = 0; i < n && task[0] != LB_ERROR; ++i) {
            if (nbd[i] < 0 || nbd[i] > 3) { task[0] = LB_ERROR; task[1] = T_ERR_NBD; }
            else if (nbd[i] == 2 && l[i] > u[i]) { task[0] = LB_ERROR; task[1] = T_ERR_NOFEAS; }
        }
        if (task[0] == LB_ERROR) return;
        active(W, l, u, nbd, x, &prjctd, &cnstnd, &boxed);
        task[0] = LB_FG;
        task[1] = T_FG_START;
        goto save;
    }
    prjctd = lsave[0]; cnstnd = lsave[1]; boxed = lsave[2]; updatd = lsave[3];
    nintol = isave[IS_NINTOL]; iback = isave[IS_IBACK]; nskip = isave[IS_NSKIP]; head = isave[IS_HEAD]; col = isave[IS_COL];
    itail = isave[IS_ITAIL]; iter = isave[IS_ITER]; iupdat = isave[IS_IUPDAT]; nseg = isave[IS_NSEG]; nfgv = isave[IS_NFGV];
    info = isave[IS_INFO]; ifun = isave[IS_IFUN]; iword = isave[IS_IWORD]; nfree = isave[IS_NFREE]; nact = isave[IS_NACT];
    ileave = isave[IS_ILEAVE]; nenter = isave[IS_NENTER];
    theta = dsave[DS_THETA]; fold = dsave[DS_FOLD]; tol = dsave[DS_TOL]; dnorm = dsave[DS_DNORM]; epsmch = dsave[DS_EPSMCH];
    gd = dsave[DS_GD]; stpmx = dsave[DS_STPMX]; sbgnrm = dsave[DS_SBGNRM]; stp = dsave[DS_STP]; gdold = dsave[DS_GDOLD];
    dtd = dsave[DS_DTD];

    if (task[0] == LB_FG && task[1] == T_FG_LNSRCH) goto line_search;
    if (task[0] == LB_NEW_X) goto new_x;
    if (task[0] == LB_FG && task[1] == T_FG_START) goto first_fg;
    if (task[0] == LB_STOP) {
        if (task[1] == T_STOP_CPU) B->dcopy(&n, W->t, &I1, x, &I1);
        goto save;
    }
    goto save; /* nothing to do for any other code */

first_fg:
    nfgv = 1;
    sbgnrm = projgr(n, l, u, nbd, x, g);
    if (sbgnrm <= pgtol) {
        task[0] = LB_CONVERGENCE;
        task[1] = T_CONV_PGTOL;
        goto save;
    }

iteration:
    iword = -1;
    if (!cnstnd && col > 0) {
        B->dcopy(&n, x, &I1, W->z, &I1);
        wrk = updatd;
        nseg = 0;
    } else {
        info = cauchy(W, x, l, u, nbd, g, theta, col, head, &nseg, sbgnrm, epsmch);
        if (info != 0) {  /* singular triangular system: drop the corrections and start over */
            info = 0; col = 0; head = 0; theta = 1.0; iupdat = 0; updatd = 0;
            goto iteration;
        }
        nintol += nseg;
        freev(W, &nfree, &nenter, &ileave, &wrk, updatd, cnstnd, iter);
        nact = n - nfree;
    }
    if (nfree != 0 && col != 0) {
        if (wrk) info = formk(W, nfree, nenter, ileave, iupdat, updatd, theta, col, head);
        if (info == 0) info = cmprlb(W, x, g, theta, col, head, nfree, cnstnd);
        if (info == 0) info = subsm(W, nfree, l, u, nbd, theta, x, g, col, head, &iword);
        if (info != 0) {
            info = 0; col = 0; head = 0; theta = 1.0; iupdat = 0; updatd = 0;
            goto iteration;
        }
    }
    for (int i = 0; i < n; ++i) W->d[i] = W->z[i] - x[i];
    ln_task[0] = LB_START;
    ln_task[1] = T_NONE;

line_search:
    {
        const double big = 1e10, ftol = 1e-3, gtol = 0.9, xtol = 0.1;
        int skip_setup = (task[0] == LB_FG && task[1] == T_FG_LNSRCH);
        info = 0;
        if (!skip_setup) {
            dnorm = B->dnrm2(&n, W->d, &I1);
            dtd = dnorm * dnorm;
            stpmx = big;
            if (cnstnd) {
                if (iter == 0) {
                    stpmx = 1.0;
                } else {
                    for (int i = 0; i < n; ++i) {
                        const double a1 = W->d[i];
                        if (nbd[i] != 0) {
                            if (a1 < 0.0 && nbd[i] <= 2) {
                                const double a2 = l[i] - x[i];
                                if (a2 >= 0.0) stpmx = 0.0;
                                else if (a1 * stpmx < a2) stpmx = a2 / a1;
                            } else if (a1 > 0.0 && nbd[i] >= 2) {
                                const double a2 = u[i] - x[i];
                                if (a2 <= 0.0) stpmx = 0.0;
                                else if (a1 * stpmx > a2) stpmx = a2 / a1;
                            }
                        }
                    }
                }
            }
            if (iter == 0 && !boxed) stp = fmin(1.0 / dnorm, stpmx);
            else stp = 1.0;
            B->dcopy(&n, x, &I1, W->t, &I1);
            B->dcopy(&n, g, &I1, W->r, &I1);
            fold = f;
            ifun = 0;
            iback = 0;
            ln_task[0] = LB_START;
            ln_task[1] = T_NONE;
        }
        gd = B->ddot(&n, g, &I1, W->d, &I1);
        if (ifun == 0) {
            gdold = gd;
            if (gd >= 0.0) info = -4;  /* not a descent direction: line search impossible */
        }
        if (info == 0) {
            dcsrch(f, gd, &stp, ftol, gtol, xtol, 0.0, stpmx, ln_task, isave + IS_LS0, dsave + DS_LS0);
            if (ln_task[0] != LB_CONVERGENCE && ln_task[0] != LB_WARNING) {
                task[0] = LB_FG;
                task[1] = T_FG_LNSRCH;
                ++ifun;
                ++nfgv;
                iback = ifun - 1;
                if (stp == 1.0) {
                    B->dcopy(&n, W->z, &I1, x, &I1);
                } else {
                    for (int i = 0; i < n; ++i) x[i] = stp * W->d[i] + W->t[i];
                }
            } else {
                task[0] = LB_NEW_X;
                task[1] = T_NONE;
            }
        }
    }
    if (info != 0 || iback >= maxls) {
        /* back to the previous iterate */
        B->dcopy(&n, W->t, &I1, x, &I1);
        B->dcopy(&n, W->r, &I1, g, &I1);
        f = fold;
        if (col == 0) {
            if (info == 0) {
                info = -9;
                --nfgv; --ifun; --iback;
            }
            task[0] = LB_ABNORMAL;
            task[1] = T_NONE;
            ++iter;
            goto save;
        }
        if (info == 0) --nfgv;
        info = 0; col = 0; head = 0; theta = 1.0; iupdat = 0; updatd = 0;
        task[0] = LB_RESTART;
        task[1] = T_NONE;
        goto iteration;
    }
    if (task[0] == LB_FG) goto save;  /* the caller evaluates f, g at x */
    /* new iterate accepted */
    ++iter;
    sbgnrm = projgr(n, l, u, nbd, x, g);
    goto save;  /* task = NEW_X: the caller may look, then calls again */

new_x:
    if (sbgnrm <= pgtol) {
        task[0] = LB_CONVERGENCE;
        task[1] = T_CONV_PGTOL;
        goto save;
    }
    {
        const double ddum = fmax(fmax(fabs(fold), fabs(f)), 1.0);
        if (fold - f <= tol * ddum) {
            task[0] = LB_CONVERGENCE;
            task[1] = T_CONV_FACTR;
            if (iback >= 10) info = -5;
            goto save;
        }
    }
    {
        for (int i = 0; i < n; ++i) W->r[i] = g[i] - W->r[i];
        const double rnrm = B->dnrm2(&n, W->r, &I1);
        const double rr = rnrm * rnrm;
        double dr, ddum;
        if (stp == 1.0) {
            dr = gd - gdold;
            ddum = -gdold;
        } else {
            dr = (gd - gdold) * stp;
            B->dscal(&n, &stp, W->d, &I1);
            ddum = -gdold * stp;
        }
        if (dr <= epsmch * ddum) {
            ++nskip;
            updatd = 0;
        } else {
            updatd = 1;
            ++iupdat;
            matupd(W, &itail, iupdat, &col, &head, &theta, rr, dr, stp, dtd);
            info = formt(W, col, theta);
            if (info != 0) {
                info = 0; col = 0; head = 0; theta = 1.0; iupdat = 0; updatd = 0;
            }
        }
    }
    goto iteration;

save:
    lsave[0] = prjctd; lsave[1] = cnstnd; lsave[2] = boxed; lsave[3] = updatd;
    isave[IS_NINTOL] = nintol; isave[IS_IBACK] = iback; isave[IS_NSKIP] = nskip; isave[IS_HEAD] = head; isave[IS_COL] = col;
    isave[IS_ITAIL] = itail; isave[IS_ITER] = iter; isave[IS_IUPDAT] = iupdat; isave[IS_NSEG] = nseg; isave[IS_NFGV] = nfgv;
    isave[IS_INFO] = info; isave[IS_IFUN] = ifun; isave[IS_IWORD] = iword; isave[IS_NFREE] = nfree; isave[IS_NACT] = nact;
    isave[IS_ILEAVE] = ileave; isave[IS_NENTER] = nenter;
    dsave[DS_THETA] = theta; dsave[DS_FOLD] = fold; dsave[DS_TOL] = tol; dsave[DS_DNORM] = dnorm; dsave[DS_EPSMCH] = epsmch;
    dsave[DS_GD] = gd; dsave[DS_STPMX] = stpmx; dsave[DS_SBGNRM] = sbgnrm; dsave[DS_STP] = stp; dsave[DS_GDOLD] = gdold;
    dsave[DS_DTD] = dtd;
}
