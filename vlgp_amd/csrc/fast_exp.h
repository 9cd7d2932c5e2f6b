// Shared device helper.
#pragma once
#include <hip/hip_runtime.h>

// min(x, 10) as np.minimum computes it (vlgp/math.py:24-38): a NaN stays a NaN (fmin would return 10)
__device__ __forceinline__ double clamp10(double x) { return x > 10.0 ? 10.0 : x; }

// exp(x) for x <= 10 (callers clamp): Cody-Waite reduction by ln 2, degree-13
// Taylor polynomial on |r| <= ln2/2 (truncation 4e-18), one ldexp.  < 2 ulp.
__device__ __forceinline__ double fast_exp(double x) {
    x = x < -745.0 ? -745.0 : x;  // keeps NaN (comparison false), avoids int overflow below
    const double k = rint(x * 1.4426950408889634074);
    double r = fma(k, -6.93147180369123816490e-01, x);
    r = fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;
    p = fma(p, r, 2.0876756987868100e-09);
    p = fma(p, r, 2.5052108385441720e-08);
    p = fma(p, r, 2.7557319223985888e-07);
    p = fma(p, r, 2.7557319223985893e-06);
    p = fma(p, r, 2.4801587301587302e-05);
    p = fma(p, r, 1.9841269841269841e-04);
    p = fma(p, r, 1.3888888888888889e-03);
    p = fma(p, r, 8.3333333333333332e-03);
    p = fma(p, r, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// 2^(j/64), j = 0 .. 63, correctly rounded (generated with 60-digit decimal arithmetic)
static __device__ const double vlgp_exp2_tab64[64] = {
    0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
    0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
    0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
    0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
    0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
    0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
    0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
    0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
    0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
    0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
    0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
    0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
    0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
    0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
    0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
    0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};

// the table into LDS (first 64 threads of the workgroup; the caller synchronises)
__device__ __forceinline__ void fast_exp_tab_init(double* tab, int tid) {
    if (tid < 64) tab[tid] = vlgp_exp2_tab64[tid];
}

// exp(x) for x <= 10 through the 64-entry table (in LDS): x = (64 e + j) ln2/64 + r, |r| <= ln2/128,
// exp(x) = 2^e 2^(j/64) (1 + expm1(r)), expm1 by its degree-5 Taylor polynomial (truncation 3.4e-17).
// Twelve double-precision operations instead of twenty-one; < 1.5 ulp.
__device__ __forceinline__ double fast_exp_tab(double x, const double* tab) {
    x = x < -745.0 ? -745.0 : x;
    const double k = rint(x * 0x1.71547652b82fep+6);  // 64 / ln 2
    double r = fma(k, -0x1.62e42fee00000p-7, x);  // ln2_hi / 64 (low 32 bits zero: k * hi is exact)
    r = fma(k, -0x1.a39ef35793c76p-39, r);  // ln2_lo / 64
    const int ki = (int)k;
    const double t = tab[ki & 63];
    double p = fma(r, 8.3333333333333332e-03, 4.1666666666666664e-02);
    p = fma(p, r, 1.6666666666666666e-01);
    p = fma(p, r, 0.5);
    const double q = fma(p, r * r, r);
    return ldexp(fma(t, q, t), ki >> 6);
}
