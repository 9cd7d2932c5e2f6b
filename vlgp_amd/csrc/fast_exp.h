// Shared device helper.
#pragma once
#include <hip/hip_runtime.h>

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

