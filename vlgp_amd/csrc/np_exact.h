// Bit-exact restatements of the three NumPy primitives math.ichol_gauss (vlgp/math.py:101-126)
// is built from, so that the device factorisation reproduces the reference's pivot sequence and
// its G bit for bit (the pivots are an arg-max over residuals that tie up to rounding noise:
// anything short of identical arithmetic picks other pivots).
//
// The reference's arithmetic is not IEEE-defined: it is whatever NumPy 2.2 / OpenBLAS 0.3.29 do
// on an AVX-512 host (third-party dependencies of the reference, not under /root/reference;
// requirements.txt:1-4 pins no versions).  Restated here from their published algorithms and
// checked against the live libraries (tests/test_np_exact.py) and the reference's golden factors:
//
//   np.exp   (math.py:115)  Intel SVML __svml_exp8_ha as bundled with NumPy (numpy/_core/src/umath/svml):
//            table-driven 2^(j/16) * (1 + p(r)), k = trunc-rounded x*log2(e) in units of 1/16, degree-5
//            polynomial in FMA form; |x| >= 707.7 goes through the scalar "rare" path (64-entry table,
//            unfused Horner, two-product split for subnormal results).
//   np.dot   (math.py:117)  cblas_dgemv (row major, no transpose) -> OpenBLAS dgemv_t, Haswell/SkylakeX
//            micro-kernels: outputs in groups of four use a 4-lane FMA accumulator over the first
//            4*floor(k/4) terms; a trailing pair of outputs a 2-lane multiply-then-add accumulator; a
//            trailing single output two 2-lane accumulators; the k mod 4 tail is scalar (compiler-fused);
//            a single output row is cblas_ddot (16/32-wide FMA blocks, then a scalar FMA tail).
//   np.sum   (math.py:105,119)  NumPy's pairwise summation (8 accumulators below 128 elements,
//            recursive halving above).
//
// Every function is a sequence of single IEEE-754 binary64 operations; compile with contraction off
// (#pragma below) so that only the explicit fma() calls fuse.
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define NPX_FN __host__ __device__ static inline
#define NPX_CONST static __device__ const
#else
#define NPX_FN static inline
#define NPX_CONST static const
#endif

#if defined(__clang__)
#pragma clang fp contract(off)
#endif  // g++: compile with -ffp-contract=off

NPX_FN double npx_u2d(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __longlong_as_double((long long)u);
#else
    double d;
    memcpy(&d, &u, 8);
    return d;
#endif
}
NPX_FN uint64_t npx_d2u(double d) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint64_t)__double_as_longlong(d);
#else
    uint64_t u;
    memcpy(&u, &d, 8);
    return u;
#endif
}

// ---- np.exp -----------------------------------------------------------------------------------
// 2^(j/16), leading and trailing parts
NPX_CONST uint64_t NPX_EXP_TH[16] = {
    0x3ff0000000000000ULL, 0x3ff0b5586cf9890fULL, 0x3ff172b83c7d517bULL, 0x3ff2387a6e756238ULL,
    0x3ff306fe0a31b715ULL, 0x3ff3dea64c123422ULL, 0x3ff4bfdad5362a27ULL, 0x3ff5ab07dd485429ULL,
    0x3ff6a09e667f3bcdULL, 0x3ff7a11473eb0187ULL, 0x3ff8ace5422aa0dbULL, 0x3ff9c49182a3f090ULL,
    0x3ffae89f995ad3adULL, 0x3ffc199bdd85529cULL, 0x3ffd5818dcfba487ULL, 0x3ffea4afa2a490daULL};
NPX_CONST uint64_t NPX_EXP_TL[16] = {
    0x0000000000000000ULL, 0x3c979aa65d837b6dULL, 0xbc801b15eaa59348ULL, 0x3c968efde3a8a894ULL,
    0x3c834d754db0abb6ULL, 0x3c859f48a72a4c6dULL, 0x3c7690cebb7aafb0ULL, 0x3c9063e1e21c5409ULL,
    0xbc93b3efbf5e2228ULL, 0xbc7b32dcb94da51dULL, 0x3c8db72fc1f0eab4ULL, 0x3c71affc2b91ce27ULL,
    0x3c8c1a7792cb3387ULL, 0x3c736eae30af0cb3ULL, 0x3c74a385a63d07a7ULL, 0xbc8ff7128fd391f0ULL};
// 2^(j/64), (leading, trailing) pairs: the scalar path of the large-|x| lanes
NPX_CONST uint64_t NPX_EXP_T64[128] = {
    0x3ff0000000000000ULL, 0x0000000000000000ULL, 0x3ff02c9a3e778061ULL, 0xbc7160139cd8dc5dULL,
    0x3ff059b0d3158574ULL, 0x3c8cd2523567f613ULL, 0x3ff0874518759bc8ULL, 0x3c60f74e61e6c861ULL,
    0x3ff0b5586cf9890fULL, 0x3c979aa65d837b6dULL, 0x3ff0e3ec32d3d1a2ULL, 0x3c3ebe3d702f9cd1ULL,
    0x3ff11301d0125b51ULL, 0xbc9556522a2fbd0eULL, 0x3ff1429aaea92de0ULL, 0xbc91c923b9d5f416ULL,
    0x3ff172b83c7d517bULL, 0xbc801b15eaa59348ULL, 0x3ff1a35beb6fcb75ULL, 0x3c8b898c3f1353bfULL,
    0x3ff1d4873168b9aaULL, 0x3c9aecf73e3a2f60ULL, 0x3ff2063b88628cd6ULL, 0x3c8a6f4144a6c38dULL,
    0x3ff2387a6e756238ULL, 0x3c968efde3a8a894ULL, 0x3ff26b4565e27cddULL, 0x3c80472b981fe7f2ULL,
    0x3ff29e9df51fdee1ULL, 0x3c82f7e16d09ab31ULL, 0x3ff2d285a6e4030bULL, 0x3c8b3782720c0ab4ULL,
    0x3ff306fe0a31b715ULL, 0x3c834d754db0abb6ULL, 0x3ff33c08b26416ffULL, 0x3c8fdd395dd3f84aULL,
    0x3ff371a7373aa9cbULL, 0xbc924aedcc4b5068ULL, 0x3ff3a7db34e59ff7ULL, 0xbc71d1e83e9436d2ULL,
    0x3ff3dea64c123422ULL, 0x3c859f48a72a4c6dULL, 0x3ff4160a21f72e2aULL, 0xbc58a78f4817895bULL,
    0x3ff44e086061892dULL, 0x3c4363ed60c2ac11ULL, 0x3ff486a2b5c13cd0ULL, 0x3c6ecce1daa10379ULL,
    0x3ff4bfdad5362a27ULL, 0x3c7690cebb7aafb0ULL, 0x3ff4f9b2769d2ca7ULL, 0xbc8f94340071a38eULL,
    0x3ff5342b569d4f82ULL, 0xbc78dec6bd0f385fULL, 0x3ff56f4736b527daULL, 0x3c93350518fdd78eULL,
    0x3ff5ab07dd485429ULL, 0x3c9063e1e21c5409ULL, 0x3ff5e76f15ad2148ULL, 0x3c9432e62b64c035ULL,
    0x3ff6247eb03a5585ULL, 0xbc8c33c53bef4da8ULL, 0x3ff6623882552225ULL, 0xbc93cedd78565858ULL,
    0x3ff6a09e667f3bcdULL, 0xbc93b3efbf5e2228ULL, 0x3ff6dfb23c651a2fULL, 0xbc6367efb86da9eeULL,
    0x3ff71f75e8ec5f74ULL, 0xbc781f647e5a3ecfULL, 0x3ff75feb564267c9ULL, 0xbc8619321e55e68aULL,
    0x3ff7a11473eb0187ULL, 0xbc7b32dcb94da51dULL, 0x3ff7e2f336cf4e62ULL, 0x3c65ebe1abd66c55ULL,
    0x3ff82589994cce13ULL, 0xbc9369b6f13b3734ULL, 0x3ff868d99b4492edULL, 0xbc94d450d872576eULL,
    0x3ff8ace5422aa0dbULL, 0x3c8db72fc1f0eab4ULL, 0x3ff8f1ae99157736ULL, 0x3c7bf68359f35f44ULL,
    0x3ff93737b0cdc5e5ULL, 0xbc5da9b88b6c1e29ULL, 0x3ff97d829fde4e50ULL, 0xbc92434322f4f9aaULL,
    0x3ff9c49182a3f090ULL, 0x3c71affc2b91ce27ULL, 0x3ffa0c667b5de565ULL, 0xbc87c50422622263ULL,
    0x3ffa5503b23e255dULL, 0xbc91bbd1d3bcbb15ULL, 0x3ffa9e6b5579fdbfULL, 0x3c8469846e735ab3ULL,
    0x3ffae89f995ad3adULL, 0x3c8c1a7792cb3387ULL, 0x3ffb33a2b84f15fbULL, 0xbc55c3d956dcaebaULL,
    0x3ffb7f76f2fb5e47ULL, 0xbc68d6f438ad9334ULL, 0x3ffbcc1e904bc1d2ULL, 0x3c74ffd70a5fddcdULL,
    0x3ffc199bdd85529cULL, 0x3c736eae30af0cb3ULL, 0x3ffc67f12e57d14bULL, 0x3c84e08fd10959acULL,
    0x3ffcb720dcef9069ULL, 0x3c676b2c6c921968ULL, 0x3ffd072d4a07897cULL, 0xbc8fad5d3ffffa6fULL,
    0x3ffd5818dcfba487ULL, 0x3c74a385a63d07a7ULL, 0x3ffda9e603db3285ULL, 0x3c8e5a50d5c192acULL,
    0x3ffdfc97337b9b5fULL, 0xbc82d52107b43e1fULL, 0x3ffe502ee78b3ff6ULL, 0x3c74b604603a88d3ULL,
    0x3ffea4afa2a490daULL, 0xbc8ff7128fd391f0ULL, 0x3ffefa1bee615a27ULL, 0x3c8ec3bc41aa2008ULL,
    0x3fff50765b6e4540ULL, 0x3c8a64a931d185eeULL, 0x3fffa7c1819e90d8ULL, 0x3c77893b4d91cd9dULL};

// lanes with |x| >= 707.7, |x| < 2^-53, infinities and NaN
NPX_FN double npx_exp_rare(double x) {
    const uint64_t ux = npx_d2u(x);
    const unsigned ex = (unsigned)((ux >> 52) & 0x7ff);
    if (ex == 0x7ff) {
        if ((ux >> 63) && !(ux & 0xfffffffffffffULL)) return 0.0;  // exp(-inf)
        return x * x;                                              // +inf, NaN
    }
    if (ex <= 0x3ca) return 1.0 + x;
    if (x > npx_u2d(0x40862e42fefa39efULL)) {  // overflow
        const double h = npx_u2d(0x7fefffffffffffffULL);
        return h * h;
    }
    if (x < npx_u2d(0xc0874910d52d3051ULL)) {  // below the smallest subnormal
        const double t = npx_u2d(0x0010000000000001ULL);
        return t * t;
    }
    const double SH = npx_u2d(0x4338000000000000ULL);
    const double t1 = x * npx_u2d(0x40571547652b82feULL) + SH;  // round(64 x / ln 2) in the low mantissa bits
    const uint32_t n = (uint32_t)npx_d2u(t1);
    const unsigned j = n & 63u;
    const uint32_t m = n >> 6;
    const double Nf = t1 - SH;
    double r = x - Nf * npx_u2d(0x3f862e42fefa0000ULL);
    r = r - Nf * npx_u2d(0x3d1cf79abc9e3b3aULL);
    double p = npx_u2d(0x3f56c16a1c2a3ffdULL) * r;
    p = p + npx_u2d(0x3f8111123aaf20d3ULL);
    p = p * r;
    p = p + npx_u2d(0x3fa5555555558fccULL);
    p = p * r;
    p = p + npx_u2d(0x3fc55555555548f8ULL);
    p = p * r;
    p = p + 0.5;
    p = p * r;
    p = p * r;
    p = p + r;
    p = p + npx_u2d(NPX_EXP_T64[2 * j + 1]);
    const double Th = npx_u2d(NPX_EXP_T64[2 * j]);
    p = p * Th;
    if (!(x < npx_u2d(0xc086232bdd7abcd2ULL))) {  // normal result
        unsigned e = (m + 0x3ffu) & 0x7ffu;
        p = p + Th;
        if (e <= 0x7fe) return p * npx_u2d((uint64_t)e << 52);
        e = (e - 1) & 0x7ff;
        return (p * npx_u2d((uint64_t)e << 52)) * 2.0;
    }
    // subnormal result: build it 2^60 too large, round once on the way down
    const unsigned e = (m + 0x43bu) & 0x7ffu;
    const double sc = npx_u2d((uint64_t)e << 52);
    const double x2 = p * sc, x1 = sc * Th, s = x1 + x2;
    const double TWOM60 = npx_u2d(0x3c30000000000000ULL);
    if (e <= 0x32) return s * TWOM60;
    const double er = (x1 - s) + x2;
    const double tt = s * npx_u2d(0x41f8000000000000ULL);
    const double a = s + tt;
    double hi = a - tt;
    double lo = s - hi;
    lo = er + lo;
    hi = hi * TWOM60;
    lo = lo * TWOM60;
    return hi + lo;
}

NPX_FN double npx_exp(double x) {
    const double SH = npx_u2d(0x42f8000000003ff0ULL);
    const double L2E = npx_u2d(0x3ff71547652b82feULL);
    // z = x*log2(e) + SH rounded TOWARD ZERO (z > 0, so downwards): round to nearest, then step back
    // one ulp when the exact value lies below (the sign of the exactly rounded residual decides)
    double z = fma(x, L2E, SH);
    if (fma(x, L2E, SH - z) < 0.0) z = npx_u2d(npx_d2u(z) - 1);
    if (!(fabs(x) < npx_u2d(0x40861da04cbafe44ULL))) return npx_exp_rare(x);
    const double N = z - SH;  // x*log2(e) in units of 1/16
    const unsigned j = (unsigned)(npx_d2u(z) & 15u);
    double r = fma(-N, npx_u2d(0x3fe62e42fefa39efULL), x);
    r = fma(-npx_u2d(0x3c7abc9e3b39803fULL), N, r);
    const double R2 = r * r;
    double p = fma(npx_u2d(0x3f57411836940c04ULL), r, npx_u2d(0x3f81101cbbc265c0ULL));
    const double p1 = fma(npx_u2d(0x3fa55557242d68feULL), r, npx_u2d(0x3fc5555553939732ULL));
    const double p0 = fma(npx_u2d(0x3fe000000000d008ULL), r, npx_u2d(0x3fefffffffffff70ULL));
    p = fma(R2, p, p1);
    p = fma(R2, p, p0);
    double q = fma(p, r, npx_u2d(NPX_EXP_TL[j]));
    const double Th = npx_u2d(NPX_EXP_TH[j]);
    q = fma(Th, q, Th);
    // * 2^floor(N): N > -1022 here, the scaling is exact
    const int e = (int)floor(N);
    return q * npx_u2d((uint64_t)(e + 1023) << 52);
}

// ---- np.sum -----------------------------------------------------------------------------------
// NumPy's pairwise sum of n <= 128 elements a[0], a[s], ... (the unrolled leaf of the recursion)
template <class F>
NPX_FN double npx_pairwise_leaf(F at, int n) {
    if (n < 8) {
        double res = -0.0;
        for (int i = 0; i < n; ++i) res = res + at(i);
        return res;
    }
    double r0 = at(0), r1 = at(1), r2 = at(2), r3 = at(3), r4 = at(4), r5 = at(5), r6 = at(6), r7 = at(7);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 = r0 + at(i); r1 = r1 + at(i + 1); r2 = r2 + at(i + 2); r3 = r3 + at(i + 3);
        r4 = r4 + at(i + 4); r5 = r5 + at(i + 5); r6 = r6 + at(i + 6); r7 = r7 + at(i + 7);
    }
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res = res + at(i);
    return res;
}
// NumPy's pairwise sum of any length over an accessor (recursive halving down to leaves of <= 128)
template <class F>
NPX_FN double npx_pairwise(F at, int n) {
    if (n <= 128) return npx_pairwise_leaf(at, n);
    // explicit stack instead of recursion: (start, len, state); depth <= log2(n / 128) + 1
    int st_start[32], st_len[32], st_state[32];
    double st_left[32];
    int sp = 0;
    st_start[0] = 0; st_len[0] = n; st_state[0] = 0;
    double ret = 0.0;
    while (sp >= 0) {
        const int s = st_start[sp], len = st_len[sp];
        if (len <= 128) {
            ret = npx_pairwise_leaf([&](int i) { return at(s + i); }, len);
            --sp;
            continue;
        }
        int n2 = len / 2;
        n2 -= n2 % 8;
        if (st_state[sp] == 0) {          // descend left
            st_state[sp] = 1;
            ++sp; st_start[sp] = s; st_len[sp] = n2; st_state[sp] = 0;
        } else if (st_state[sp] == 1) {   // left done -> descend right
            st_left[sp] = ret;
            st_state[sp] = 2;
            ++sp; st_start[sp] = s + n2; st_len[sp] = len - n2; st_state[sp] = 0;
        } else {                          // both done
            ret = st_left[sp] + ret;
            --sp;
        }
    }
    return ret;
}
// np.sum over a contiguous 1-D array (add.reduce: identity 0, then the pairwise sum)
NPX_FN double npx_sum(const double* a, int n) {
    return 0.0 + npx_pairwise([&](int i) { return a[i]; }, n);
}

// ---- np.dot (one output of A @ x) ---------------------------------------------------------------
// jj: index of this output among the `mo` outputs of the call, k: length of the dot product.
// a(l), x(l): the l-th factor of the row and of the vector.
template <class FA, class FX>
NPX_FN double npx_dot_row(FA a, FX x, int k, int jj, int mo) {
    if (k == 0) return 0.0;
    if (mo == 1) {  // cblas_ddot
        const int n1 = k & ~15, n32 = n1 & ~31;
        double dot = 0.0;
        if (n1) {
            double acc[4][4];
            for (int q = 0; q < 4; ++q)
                for (int i = 0; i < 4; ++i) acc[q][i] = 0.0;
            if (n32) {
                double z[4][8];
                for (int q = 0; q < 4; ++q)
                    for (int i = 0; i < 8; ++i) z[q][i] = 0.0;
                for (int l = 0; l < n32; ++l) z[(l & 31) >> 3][l & 7] = fma(a(l), x(l), z[(l & 31) >> 3][l & 7]);
                for (int q = 0; q < 4; ++q)
                    for (int i = 0; i < 4; ++i) acc[q][i] = z[q][i] + z[q][i + 4];
            }
            for (int l = n32; l < n1; ++l) acc[((l - n32) & 15) >> 2][l & 3] = fma(a(l), x(l), acc[((l - n32) & 15) >> 2][l & 3]);
            double s[4];
            for (int i = 0; i < 4; ++i) s[i] = ((acc[0][i] + acc[1][i]) + acc[2][i]) + acc[3][i];
            dot = (s[0] + s[2]) + (s[1] + s[3]);
        }
        for (int l = n1; l < k; ++l) dot = fma(a(l), x(l), dot);
        return dot;
    }
    const int k4 = k & ~3, n4 = (mo >> 2) << 2;
    double t = 0.0;
    if (k4) {
        if (jj < n4) {  // four outputs at a time: 4-lane FMA accumulator
            double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
            for (int l = 0; l < k4; l += 4) {
                c0 = fma(a(l), x(l), c0);
                c1 = fma(a(l + 1), x(l + 1), c1);
                c2 = fma(a(l + 2), x(l + 2), c2);
                c3 = fma(a(l + 3), x(l + 3), c3);
            }
            t = (c0 + c2) + (c1 + c3);
        } else if ((mo & 2) && jj < n4 + 2) {  // trailing pair: 2-lane multiply, add
            double c0 = 0.0, c1 = 0.0;
            for (int l = 0; l < k4; l += 2) {
                c0 = c0 + a(l) * x(l);
                c1 = c1 + a(l + 1) * x(l + 1);
            }
            t = c0 + c1;
        } else {  // trailing single output: two 2-lane accumulators
            double a0 = 0.0, a1 = 0.0, b0 = 0.0, b1 = 0.0;
            for (int l = 0; l < k4; l += 4) {
                a0 = a0 + a(l) * x(l);
                a1 = a1 + a(l + 1) * x(l + 1);
                b0 = b0 + a(l + 2) * x(l + 2);
                b1 = b1 + a(l + 3) * x(l + 3);
            }
            t = (a0 + b0) + (a1 + b1);
        }
    }
    double y = 0.0 + t;
    const int r = k - k4;
    if (r == 1) y = fma(a(k4), x(k4), y);
    else if (r == 2) y = y + fma(a(k4), x(k4), a(k4 + 1) * x(k4 + 1));
    else if (r == 3) y = y + fma(a(k4 + 2), x(k4 + 2), fma(a(k4), x(k4), a(k4 + 1) * x(k4 + 1)));
    return y;
}

// ---- one row of one ichol_gauss step (math.py:114-119) ------------------------------------------
// row: G[row, 0..R) of the row at permuted position i+1+jj; prow: the pivot row (position i);
// kv = exp(-omega (x_row - x_piv)^2); piv = G[i, i].  Writes row[i], returns the new residual d.
NPX_FN double npx_ichol_row(double* row, const double* prow, int i, int jj, int mo, double kv, double piv) {
    const double dot = npx_dot_row([&](int l) { return row[l]; }, [&](int l) { return prow[l]; }, i, jj, mo);
    const double g = (kv - dot) / piv;
    row[i] = g;
    const double ss = 0.0 + npx_pairwise([&](int l) { const double v = row[l]; return v * v; }, i + 1);
    return 1.0 - ss;
}
