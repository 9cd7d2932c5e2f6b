// CPU build of vlgp_amd/csrc/np_exact.h for the tests (g++ -O2 -mfma -ffp-contract=off): the same
// functions the device kernel is made of, exposed with C linkage so that pytest can compare them
// with the live NumPy / OpenBLAS and with the reference's golden factors.  Test infrastructure only.
#include <vector>

#include "../../vlgp_amd/csrc/np_exact.h"

extern "C" {
void npx_exp_array(const double* x, double* y, long n) {
    for (long i = 0; i < n; ++i) y[i] = npx_exp(x[i]);
}
double npx_sum_array(const double* a, int n) { return npx_sum(a, n); }
// A: (mo, k) row-major with leading dimension lda
void npx_dot_array(const double* A, int lda, const double* x, int mo, int k, double* y) {
    for (int j = 0; j < mo; ++j) {
        const double* row = A + (long)j * lda;
        y[j] = npx_dot_row([&](int l) { return row[l]; }, [&](int l) { return x[l]; }, k, j, mo);
    }
}
// math.ichol_gauss (vlgp/math.py:101-126), serial, rows kept in original order
// G: (n, r) zero-filled on entry; piv: (n) the pivot permutation; returns the number of columns built
int npx_ichol_gauss(int n, double omega, int r, double* G, int* piv) {
    std::vector<double> d(n, 1.0);
    for (int j = 0; j < n; ++j) piv[j] = j;
    const double tol_n = 1e-6 * (double)n;
    int i = 0;
    while (i < r && npx_sum(d.data() + i, n - i) > tol_n) {
        int jast = 0;
        if (i > 0) {
            jast = i;
            for (int j = i + 1; j < n; ++j)
                if (d[j] > d[jast]) jast = j;  // first maximum, like numpy.argmax
            const int t = piv[i]; piv[i] = piv[jast]; piv[jast] = t;
        }
        const double pivot = sqrt(d[jast]);
        double* prow = G + (long)piv[i] * r;
        prow[i] = pivot;
        const int mo = n - i - 1;
        for (int jj = 0; jj < mo; ++jj) {
            const int row = piv[i + 1 + jj];
            const double dx = (double)row - (double)piv[i];
            const double kv = npx_exp(-omega * (dx * dx));
            d[i + 1 + jj] = npx_ichol_row(G + (long)row * r, prow, i, jj, mo, kv, pivot);
        }
        ++i;
    }
    return i;
}
}
