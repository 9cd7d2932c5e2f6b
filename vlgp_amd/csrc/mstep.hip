// M-step of vLGP on gfx950 (core.mstep, vlgp/core.py:129-249).
//
// All units are treated as one (rows x N) matrix.  Per Newton iteration the
// reference recomputes the rate and then, per channel n, forms
//     grad_a = mu' y_n - (mu + v a_n)' r_n
//     nhess_a = (mu + v a_n)' diag(r_n) (mu + v a_n) + diag(r_n' v)
//     grad_b = x_n' (y_n - r_n),  nhess_b = x_n' diag(r_n) x_n
// (core.py:181-220).  The y-dependent halves (mu'y, x'y) do not change inside
// the M-step, so they are accumulated once (PREP pass, the only full read of y
// besides the noise pass); each Newton iteration then streams mu, v (and x if
// it is not identically 1) and accumulates the rate-dependent sums.  Gaussian
// channels use the closed-form alternating least squares of core.py:224-235,
// which only needs moments gathered in the PREP pass.
//
// Mapping: lane <-> channel (coalesced rows of y/x), several row slices per
// workgroup, accumulators in registers, mu/v row tiles broadcast from LDS.
// Reduction is deterministic: per-workgroup partials, then a fixed-order sum,
// then (multi-GPU) one RCCL all-reduce of the fused statistics buffer.
#include <type_traits>

#include "ctx.h"
#include "fast_exp.h"

// rows of (mu | v) staged in LDS per step: 20 KB at five latents.  (Round 4: 512 / 800 rows make the M-step ALONE faster --
// 2.23 -> 2.17 / 2.14 ms, fewer barriers -- and the EM iteration slower, 139 -> 135 / 132 EM it/s on the same box: the LDS
// they take is room the H-step's round workgroups lose on every CU while the two lanes run side by side.)
#ifndef M_TILE
#define M_TILE 256
#endif

enum { K_PREP = 0, K_NEWTON = 1, K_NOISE1 = 2, K_NOISE2 = 3 };

struct MArgs {
    int N, L, P;
    int64_t rows;
    int rows_per_wg;
    int CT;          // channels per tile (grid.y tiles)
    int S;           // row slices per workgroup
    const double* y;
    const double* x;  // null = ones
    const double* mu;
    const double* v;
    const double* a;
    const double* b;
    const int* gauss;
    const double* mean;  // NOISE2: per-channel mean of (y - eta)
    double* partial;     // (G, K, N)
};

__host__ __device__ constexpr int tri(int n) { return n * (n + 1) / 2; }
template <int LT, int PT, int KIND>
__host__ __device__ constexpr int nacc() {
    return KIND == K_PREP ? LT + PT + PT * LT + tri(PT) + (PT <= 2 ? 2 + PT : 0)
         : KIND == K_NEWTON ? 2 * LT + tri(LT) + PT + tri(PT)
         : 1;
}
// runtime (exact L, P) number of statistics per channel
static inline int nstat_rt(int L, int P, int kind) {
    // (K_PREP, at most two regressors: y'y | 1'y | 1'x ride along for noise_stats_kernel; with more the padded
    // accumulators of the compiled kernels are out of registers and the noise keeps its two passes)
    return kind == K_PREP ? L + P + P * L + tri(P) + (P <= 2 ? 2 + P : 0) : kind == K_NEWTON ? 2 * L + tri(L) + P + tri(P) : 1;
}

// EXACT: L == LT, P == PT and x == 1 known at compile time -- the common case (3, 5, 8, 10 latents, no regressors):
// no per-latent predicates or branches in the row loop, and the scalar registers they held go to the exp constants
//
// The slices' partial sums meet in LDS MS_GS statistics at a time (the first version spent two barriers per statistic:
// 54 per launch at five latents, at the end of every workgroup's life with nothing else left to run on the CU).
// (Measured dead ends: the next row tile prefetched into registers across the row loop -- twelve more registers, 27
// spilled at the 128-register budget of four waves per SIMD; at ten latents, the 77 accumulators of a thread split over
// two launches per Newton iteration (gradient sums + Hessian rows < 7 / rows >= 7 + r'v sums, both recomputing the
// rate) to get from two to four waves per SIMD: a, a^2, mu, v, mt and q of ten latents are 120 registers before the
// first accumulator, 130 registers spilled, C5's M-step 46 -> 270 ms; round 4: at ten latents one row at a time with only
// a and mu + v a live across the Hessian update (a^2, q, v recomputed / re-read from LDS, bit-identical sums): no spill at
// 256 registers but no second row in flight either, 36 -> 43 ms.  A 512-thread workgroup is two waves per SIMD whatever
// the register count up to 256; more occupancy needs <= 128 registers, i.e. the accumulators of a channel split over lanes.)
#define MS_GS 8
template <int LT, int PT, int KIND, bool EXACT>
// four waves per SIMD (128 VGPRs) up to five latents; the 2 L + L (L + 1) / 2 accumulators of more latents
// need the registers more than the occupancy
__global__ void __launch_bounds__(512, (LT <= 5 ? 4 : (LT <= 8 ? 2 : 1))) mstep_accum(MArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NA = nacc<LT, PT, KIND>();
    constexpr int NBUF = 1;
    const int N = A.N, CT = A.CT, S = A.S;
    const int L = EXACT ? LT : A.L, P = EXACT ? PT : A.P;
    const int tid = threadIdx.x;
    const int nl = tid % CT, s = tid / CT;
    const int n = blockIdx.y * CT + nl;
    const bool active = s < S && n < N;
    const int tile_d = 2 * M_TILE * L;        // one buffer: mu tile | v tile
    double* red = smem;                       // aliases the tiles after the row loop: MS_GS x (S CT)
    const int red_d = MS_GS * S * CT;
    double* etab = smem + (NBUF * tile_d > red_d ? NBUF * tile_d : red_d);  // NEWTON: 2^(j/64) for fast_exp_tab
    // (the 64-entry table, not the E-step passes' 256-entry one: measured on one box, same build otherwise, the extra
    // 1.5 KB of LDS per workgroup make this kernel 3 % faster and the EM iteration 4 % SLOWER -- 143-144 against 147-150 EM
    // it/s at C3 -- because the H-step's round workgroups lose room on every CU while the two lanes run side by side)
    if constexpr (KIND == K_NEWTON) fast_exp_tab_init(etab, tid);

    double al[LT], al2[LT], bl[PT], acc[NA];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        al[l] = (active && l < L) ? A.a[l * N + n] : 0.0;
        al2[l] = 0.5 * al[l] * al[l];  // half squares: the rate's exponent is one chain x.b + mu.a + v.(a^2 / 2)
    }
#pragma unroll
    for (int j = 0; j < PT; ++j) bl[j] = (active && j < P) ? A.b[j * N + n] : 0.0;
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[k] = 0.0;
    const bool gch = active ? A.gauss[n] != 0 : false;
    const double shift = (KIND == K_NOISE2 && active) ? A.mean[n] : 0.0;

    const int64_t c0 = (int64_t)blockIdx.x * A.rows_per_wg;
    int64_t c1 = c0 + A.rows_per_wg;
    if (c1 > A.rows) c1 = A.rows;
    auto copy_tile = [&](double* buf, int64_t t0) {
        const int cnt = (int)((c1 - t0) < M_TILE ? (c1 - t0) : M_TILE) * L;
        for (int i = tid; i < cnt; i += blockDim.x) {
            buf[i] = A.mu[t0 * L + i];
            buf[M_TILE * L + i] = A.v[t0 * L + i];
        }
    };
    if (c0 < c1) copy_tile(smem, c0);
    __syncthreads();
    for (int64_t t0 = c0; t0 < c1; t0 += M_TILE) {
        const int nr = (int)((c1 - t0) < M_TILE ? (c1 - t0) : M_TILE);
        const bool more = t0 + M_TILE < c1;
        const double* mu_t = smem;
        const double* v_t = mu_t + M_TILE * L;
        if (active && !(KIND == K_NEWTON && gch)) {  // Gaussian channels need no rate statistics
        // two rows in flight: independent exp / FMA chains (the PREP pass with eight of its loads of y in flight:
        // 99 against 95 us per launch, measured in round 6 -- not kept)
#pragma unroll 2
        for (int rr = s; rr < nr; rr += S) {
            const int64_t row = t0 + rr;
            double xv[PT];
#pragma unroll
            for (int j = 0; j < PT; ++j)
                xv[j] = EXACT ? 1.0 : ((j < P) ? (A.x ? A.x[(row * P + j) * N + n] : 1.0) : 0.0);
            double mr[LT], vr[LT];
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                mr[l] = (EXACT || l < L) ? mu_t[rr * L + l] : 0.0;
                vr[l] = (EXACT || l < L) ? v_t[rr * L + l] : 0.0;
            }
            if constexpr (KIND == K_PREP) {
                const double yv = A.y[row * N + n];
                int k = 0;
#pragma unroll
                for (int l = 0; l < LT; ++l) { acc[k] = fma(mr[l], yv, acc[k]); ++k; }
#pragma unroll
                for (int j = 0; j < PT; ++j) { acc[k] = fma(xv[j], yv, acc[k]); ++k; }
#pragma unroll
                for (int j = 0; j < PT; ++j)
#pragma unroll
                    for (int l = 0; l < LT; ++l) { acc[k] = fma(xv[j], mr[l], acc[k]); ++k; }
#pragma unroll
                for (int i = 0; i < PT; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j) { acc[k] = fma(xv[i], xv[j], acc[k]); ++k; }
                // y'y | 1'y | 1'x: with the moments of mu they give var(y - eta) without another pass over y
                // (noise_stats_kernel below)
                if constexpr (PT <= 2) {
                    acc[k] = fma(yv, yv, acc[k]); ++k;
                    acc[k] += yv; ++k;
#pragma unroll
                    for (int j = 0; j < PT; ++j) { acc[k] += xv[j]; ++k; }
                }
            } else {
                double eta = 0.0;
#pragma unroll
                for (int j = 0; j < PT; ++j) eta = fma(xv[j], bl[j], eta);
#pragma unroll
                for (int l = 0; l < LT; ++l) eta = fma(mr[l], al[l], eta);
                if constexpr (KIND == K_NEWTON) {
                    double ex = eta;
#pragma unroll
                    for (int l = 0; l < LT; ++l) ex = fma(vr[l], al2[l], ex);
                    const double rate = trunc_exp_tab64(ex, etab);
                    double mt[LT], q[LT];
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        mt[l] = fma(vr[l], al[l], mr[l]);
                        q[l] = mt[l] * rate;
                    }
                    int k = 0;
#pragma unroll
                    for (int l = 0; l < LT; ++l) { acc[k] += q[l]; ++k; }
#pragma unroll
                    for (int i = 0; i < LT; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) { acc[k] = fma(q[i], mt[j], acc[k]); ++k; }
#pragma unroll
                    for (int l = 0; l < LT; ++l) { acc[k] = fma(rate, vr[l], acc[k]); ++k; }
#pragma unroll
                    for (int j = 0; j < PT; ++j) { acc[k] = fma(xv[j], rate, acc[k]); ++k; }
#pragma unroll
                    for (int i = 0; i < PT; ++i)
#pragma unroll
                        for (int j = 0; j <= i; ++j) { acc[k] = fma(xv[i] * xv[j], rate, acc[k]); ++k; }
                } else {
                    const double dres = A.y[row * N + n] - eta - shift;
                    if constexpr (KIND == K_NOISE1) acc[0] += dres;
                    else acc[0] = fma(dres, dres, acc[0]);
                }
            }
        }
        }
        if (more) {
            __syncthreads();  // everybody is done with the tile
            copy_tile(smem, t0 + M_TILE);
        }
        __syncthreads();
    }
    // slices -> one partial per workgroup, MS_GS statistics per round through LDS.  kex[k]: where the template's
    // (padded LT / PT) accumulator k sits in the exact (L, P) layout, -1 for padding.
    int kex[NA];
    {
        int k = 0, ke = 0;
        if constexpr (KIND == K_PREP) {
#pragma unroll
            for (int l = 0; l < LT; ++l, ++k) kex[k] = l < L ? ke++ : -1;
#pragma unroll
            for (int j = 0; j < PT; ++j, ++k) kex[k] = j < P ? ke++ : -1;
#pragma unroll
            for (int j = 0; j < PT; ++j)
#pragma unroll
                for (int l = 0; l < LT; ++l, ++k) kex[k] = (j < P && l < L) ? ke++ : -1;
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j, ++k) kex[k] = i < P ? ke++ : -1;
            if constexpr (PT <= 2) {
                kex[k++] = ke++;
                kex[k++] = ke++;
#pragma unroll
                for (int j = 0; j < PT; ++j, ++k) kex[k] = j < P ? ke++ : -1;
            }
        } else if constexpr (KIND == K_NEWTON) {
#pragma unroll
            for (int l = 0; l < LT; ++l, ++k) kex[k] = l < L ? ke++ : -1;
#pragma unroll
            for (int i = 0; i < LT; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j, ++k) kex[k] = i < L ? ke++ : -1;
#pragma unroll
            for (int l = 0; l < LT; ++l, ++k) kex[k] = l < L ? ke++ : -1;
#pragma unroll
            for (int j = 0; j < PT; ++j, ++k) kex[k] = j < P ? ke++ : -1;
#pragma unroll
            for (int i = 0; i < PT; ++i)
#pragma unroll
                for (int j = 0; j <= i; ++j, ++k) kex[k] = i < P ? ke++ : -1;
        } else {
            kex[0] = 0;
        }
    }
    const int K = KIND == K_PREP ? L + P + P * L + tri(P) + (P <= 2 ? 2 + P : 0) : KIND == K_NEWTON ? 2 * L + tri(L) + P + tri(P) : 1;
    const int SC = S * CT;
#pragma unroll
    for (int g0 = 0; g0 < NA; g0 += MS_GS) {
        if (s < S) {
#pragma unroll
            for (int j = 0; j < MS_GS; ++j)
                if (g0 + j < NA) red[j * SC + s * CT + nl] = active ? acc[g0 + j] : 0.0;
        }
        __syncthreads();
        // (statistic j, channel nl) pairs of this group, spread over the workgroup; slices added in slice order
        for (int o = tid; o < MS_GS * CT; o += blockDim.x) {
            const int j = o / CT, c = o - j * CT;
            int ke = -1;
#pragma unroll
            for (int jj = 0; jj < MS_GS; ++jj)
                if (g0 + jj < NA && jj == j) ke = kex[g0 + jj];
            const int nn = blockIdx.y * CT + c;
            if (ke >= 0 && nn < N) {
                double t = 0.0;
                for (int q = 0; q < S; ++q) t += red[j * SC + q * CT + c];
                A.partial[((int64_t)blockIdx.x * K + ke) * N + nn] = t;
            }
        }
        if (g0 + MS_GS < NA) __syncthreads();
    }
}

// out[i] = sum_g partial[g][i] in a fixed order: 8 strided slices of g per output
// (coalesced over i), then a fixed 8-way combine through LDS.
__global__ void __launch_bounds__(512) sum_partials_kernel(const double* partial, int G, int64_t K, double* out) {
    __shared__ double red[8][64];
    const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 64 + o;
    double s0 = 0.0, s1 = 0.0;
    if (i < K) {
        int g = sl;
        for (; g + 8 < G; g += 16) {
            s0 += partial[(int64_t)g * K + i];
            s1 += partial[(int64_t)(g + 8) * K + i];
        }
        if (g < G) s0 += partial[(int64_t)g * K + i];
    }
    red[sl][o] = s0 + s1;
    __syncthreads();
    if (sl == 0 && i < K) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += red[q][o];
        out[i] = t;
    }
}

// latent-only moments: Gram of mu (lower), column sums of mu and v, sums of squares
// out layout: [tri(L) gram | L sum_mu | L sum_v | L sum_mu^2 | 1 |dmu|^2 ]
template <int LT>
__global__ void __launch_bounds__(256)
latent_moments_kernel(int L, int64_t rows, const double* mu, const double* v, const double* dmu,
                      double* partial) {
    __shared__ double red[256];
    constexpr int NA = tri(LT) + 3 * LT + 1;
    double acc[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) acc[k] = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < rows; t += (int64_t)gridDim.x * 256) {
        double mr[LT], vr[LT], dr[LT];
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            mr[l] = l < L ? mu[t * L + l] : 0.0;
            vr[l] = (v && l < L) ? v[t * L + l] : 0.0;
            dr[l] = (dmu && l < L) ? dmu[t * L + l] : 0.0;
        }
        int k = 0;
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j) { acc[k] = fma(mr[i], mr[j], acc[k]); ++k; }
#pragma unroll
        for (int l = 0; l < LT; ++l) { acc[k] += mr[l]; ++k; }
#pragma unroll
        for (int l = 0; l < LT; ++l) { acc[k] += vr[l]; ++k; }
#pragma unroll
        for (int l = 0; l < LT; ++l) { acc[k] = fma(mr[l], mr[l], acc[k]); ++k; }
#pragma unroll
        for (int l = 0; l < LT; ++l) acc[k] = fma(dr[l], dr[l], acc[k]);
    }
    const int K = tri(L) + 3 * L + 1;
    auto emit = [&](int ke, double val) {
        __syncthreads();
        red[threadIdx.x] = val;
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * K + ke] = red[0];
    };
    int k = 0, ke = 0;
#pragma unroll
    for (int i = 0; i < LT; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j, ++k) if (i < L) emit(ke++, acc[k]);
#pragma unroll
    for (int l = 0; l < LT; ++l, ++k) if (l < L) emit(ke++, acc[k]);
#pragma unroll
    for (int l = 0; l < LT; ++l, ++k) if (l < L) emit(ke++, acc[k]);
#pragma unroll
    for (int l = 0; l < LT; ++l, ++k) if (l < L) emit(ke++, acc[k]);
    emit(ke, acc[k]);
}


// ---- loop-based fallbacks: more than 16 latents or more than 8 regressors --------------------------------------
// The kernels above keep a channel's 2 L + L (L + 1) / 2 + ... sums in registers, unrolled over the compiled (LT, PT):
// fine up to sixteen latents / eight regressors, hopeless at sixty-four (2 200 sums).  The reference has no bound on
// either (vlgp/core.py:181-220 works on whole arrays), so beyond the compiled sizes the same sums are formed
// eight at a time: a first pass leaves the quantity every Newton / noise sum shares -- the rate exp(eta + v a^2 / 2),
// or the residual y - eta -- in a (rows x N) cache, then each (row block, channel tile, group of eight statistics)
// workgroup walks its rows once.  A statistic is a product of up to three per-(row, channel) factors, named by a code.
// Same partial-sum layout as mstep_accum, so the fixed-order reduction and the solve kernels follow unchanged.
enum { T_ONE = 0, T_MU, T_V, T_MT, T_X, T_Y, T_C };
#define MG_CH 8

// k-th lower-triangle pair in row-major order: k = i (i + 1) / 2 + j, j <= i
__device__ __forceinline__ void tri_ij(int k, int& i, int& j) {
    int r = 0;
    while ((r + 1) * (r + 2) / 2 <= k) ++r;
    i = r;
    j = k - r * (r + 1) / 2;
}

__device__ void mg_stat_code(int kind, int L, int P, int k, int& t1, int& i1, int& t2, int& i2, int& t3) {
    t1 = T_ONE; i1 = 0; t2 = T_ONE; i2 = 0; t3 = T_ONE;
    if (kind == K_PREP) {  // mu'y | x'y | x'mu | x'x | y'y | 1'y | 1'x
        if (k < L) { t1 = T_MU; i1 = k; t2 = T_Y; return; }
        k -= L;
        if (k < P) { t1 = T_X; i1 = k; t2 = T_Y; return; }
        k -= P;
        if (k < P * L) { t1 = T_X; i1 = k / L; t2 = T_MU; i2 = k % L; return; }
        k -= P * L;
        if (k < tri(P)) { t1 = T_X; t2 = T_X; tri_ij(k, i1, i2); return; }
        k -= tri(P);
        if (k == 0) { t1 = T_Y; t2 = T_Y; return; }  // y'y | 1'y | 1'x
        if (k == 1) { t1 = T_Y; return; }
        t1 = T_X; i1 = k - 2;
    } else if (kind == K_NEWTON) {  // (mu + v a)'r | (mu + v a)' diag(r) (mu + v a) | v'r | x'r | x' diag(r) x
        t3 = T_C;
        if (k < L) { t1 = T_MT; i1 = k; return; }
        k -= L;
        if (k < tri(L)) { t1 = T_MT; t2 = T_MT; tri_ij(k, i1, i2); return; }
        k -= tri(L);
        if (k < L) { t1 = T_V; i1 = k; return; }
        k -= L;
        if (k < P) { t1 = T_X; i1 = k; return; }
        k -= P;
        t1 = T_X; t2 = T_X;
        tri_ij(k, i1, i2);
    } else if (kind == K_NOISE1) {
        t1 = T_C;
    } else {
        t1 = T_C; t2 = T_C;
    }
}

// cache[row, n] = rate (K_NEWTON; core.py:183) or y - eta (- mean[n]) (K_NOISE1 / K_NOISE2; core.py:177)
__global__ void __launch_bounds__(256) mstep_cache_gen(MArgs A, int kind, double* cache) {
    const int N = A.N, L = A.L, P = A.P;
    const int64_t total = A.rows * N;
    for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
        const int64_t row = e / N;
        const int n = (int)(e - row * N);
        double eta = 0.0, lin = 0.0;
        for (int j = 0; j < P; ++j) eta = fma(A.x ? A.x[(row * P + j) * N + n] : 1.0, A.b[(int64_t)j * N + n], eta);
        for (int l = 0; l < L; ++l) {
            const double al = A.a[(int64_t)l * N + n];
            eta = fma(A.mu[row * L + l], al, eta);
            lin = fma(A.v[row * L + l], al * al, lin);
        }
        double out;
        if (kind == K_NEWTON) out = A.gauss[n] ? 0.0 : fast_exp(clamp10(fma(0.5, lin, eta)));
        else out = A.y[e] - eta - (kind == K_NOISE2 ? A.mean[n] : 0.0);
        cache[e] = out;
    }
}

// grid (row blocks, channel tiles of 256, groups of MG_CH statistics); thread <-> channel
__global__ void __launch_bounds__(256) mstep_accum_gen(MArgs A, int kind, int K, const double* cache) {
    __shared__ int code[MG_CH][5];
    const int N = A.N, L = A.L, P = A.P;
    const int n = blockIdx.y * 256 + threadIdx.x;
    const int k0 = blockIdx.z * MG_CH;
    if (threadIdx.x < MG_CH) {
        int t1, i1, t2, i2, t3;
        const int k = k0 + threadIdx.x;
        mg_stat_code(kind, L, P, k < K ? k : 0, t1, i1, t2, i2, t3);
        int* c = code[threadIdx.x];
        c[0] = t1; c[1] = i1; c[2] = t2; c[3] = i2; c[4] = t3;
    }
    __syncthreads();
    const int64_t c0 = (int64_t)blockIdx.x * A.rows_per_wg;
    int64_t c1 = c0 + A.rows_per_wg;
    if (c1 > A.rows) c1 = A.rows;
    if (n >= N) return;
    double acc[MG_CH];
#pragma unroll
    for (int c = 0; c < MG_CH; ++c) acc[c] = 0.0;
    const bool skip = kind == K_NEWTON && A.gauss[n] != 0;  // Gaussian channels need no rate statistics
    auto fetch = [&](int t, int i, int64_t row) -> double {
        switch (t) {
            case T_MU: return A.mu[row * L + i];
            case T_V: return A.v[row * L + i];
            case T_MT: return fma(A.v[row * L + i], A.a[(int64_t)i * N + n], A.mu[row * L + i]);
            case T_X: return A.x ? A.x[(row * P + i) * N + n] : 1.0;
            case T_Y: return A.y[row * N + n];
            case T_C: return cache[row * N + n];
            default: return 1.0;
        }
    };
    if (!skip) {
        for (int64_t row = c0; row < c1; ++row) {
#pragma unroll
            for (int c = 0; c < MG_CH; ++c) {
                const int* cd = code[c];
                const double f1 = fetch(cd[0], cd[1], row), f2 = fetch(cd[2], cd[3], row), f3 = fetch(cd[4], 0, row);
                acc[c] = fma(f1 * f3, f2, acc[c]);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < MG_CH; ++c)
        if (k0 + c < K) A.partial[((int64_t)blockIdx.x * K + k0 + c) * N + n] = acc[c];
}

// latent_moments_kernel for any L: grid (G, groups of MG_CH statistics), same row striding and partial layout
__global__ void __launch_bounds__(256)
latent_moments_gen(int L, int64_t rows, const double* mu, const double* v, const double* dmu, double* partial) {
    __shared__ double red[256];
    const int K = tri(L) + 3 * L + 1;
    const int k0 = blockIdx.y * MG_CH;
    double acc[MG_CH];
#pragma unroll
    for (int c = 0; c < MG_CH; ++c) acc[c] = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < rows; t += (int64_t)gridDim.x * 256) {
#pragma unroll
        for (int c = 0; c < MG_CH; ++c) {
            int k = k0 + c;
            if (k >= K) continue;
            double val;
            if (k < tri(L)) {
                int i, j;
                tri_ij(k, i, j);
                val = mu[t * L + i] * mu[t * L + j];
            } else if ((k -= tri(L)) < L) {
                val = mu[t * L + k];
            } else if ((k -= L) < L) {
                val = v ? v[t * L + k] : 0.0;
            } else if ((k -= L) < L) {
                val = mu[t * L + k] * mu[t * L + k];
            } else {
                val = 0.0;
                if (dmu)
                    for (int l = 0; l < L; ++l) val = fma(dmu[t * L + l], dmu[t * L + l], val);
            }
            acc[c] += val;
        }
    }
    for (int c = 0; c < MG_CH; ++c) {
        if (k0 + c >= K) break;
        __syncthreads();
        red[threadIdx.x] = acc[c];
        __syncthreads();
        for (int o = 128; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * K + k0 + c] = red[0];
    }
}

// ---- per-channel Newton / least-squares update -----------------------------
#define SOLVE_MAXD 16
// in-place Cholesky solve of the packed-lower SPD system H x = g (dimension d);
// returns false on a non-positive pivot.
__device__ bool chol_solve_small(int d, double* H /* d*d full, lower used */, double* g) {
    for (int k = 0; k < d; ++k) {
        double s = H[k * d + k];
        for (int i = 0; i < k; ++i) s -= H[k * d + i] * H[k * d + i];
        if (!(s > 0.0) || !(s < 1e300)) return false;
        const double sd = sqrt(s);
        H[k * d + k] = sd;
        for (int j = k + 1; j < d; ++j) {
            double t = H[j * d + k];
            for (int i = 0; i < k; ++i) t -= H[j * d + i] * H[k * d + i];
            H[j * d + k] = t / sd;
        }
    }
    for (int i = 0; i < d; ++i) {
        double t = g[i];
        for (int j = 0; j < i; ++j) t -= H[i * d + j] * g[j];
        g[i] = t / H[i * d + i];
    }
    for (int i = d - 1; i >= 0; --i) {
        double t = g[i];
        for (int j = i + 1; j < d; ++j) t -= H[j * d + i] * g[j];
        g[i] = t / H[i * d + i];
    }
    return true;
}

struct SolveArgs {
    int N, L, P;
    int use_hessian;
    double eps, lr, da_bound, db_bound;
    const double* prep;   // (Kprep, N): MtY, XtY, XtM, XtX
    const double* stats;  // (Knewton, N)
    const double* lat;    // tri(L) gram | L sum_mu | L sum_v | ...
    const int* gauss;
    double *a, *b, *da, *db;
    int* fail;
    double* hws;          // systems beyond SOLVE_MAXD: per channel D (D + 3) doubles of global workspace, else null
};

__device__ void mstep_solve_channel(const SolveArgs& A, int n) {
    const int N = A.N, L = A.L, P = A.P;
    double Hl[SOLVE_MAXD * SOLVE_MAXD], gl[SOLVE_MAXD], gsl[SOLVE_MAXD], anl[SOLVE_MAXD];
    double *H = Hl, *g = gl, *gs = gsl, *an = anl;
    if (A.hws) {  // (L or P beyond the register-sized arrays)
        const int D = L > P ? L : P;
        H = A.hws + (int64_t)n * D * (D + 3);
        g = H + (int64_t)D * D;
        gs = g + D;
        an = gs + D;
    }
    const double* MtY = A.prep;                       // L rows
    const double* XtY = A.prep + (int64_t)L * N;      // P rows
    const double* XtM = XtY + (int64_t)P * N;         // P*L rows: [j*L + l]
    const double* XtX = XtM + (int64_t)P * L * N;     // tri(P) rows
    if (!A.gauss[n]) {
        const double* g1 = A.stats;
        const double* Hs = g1 + (int64_t)L * N;
        const double* rv = Hs + (int64_t)tri(L) * N;
        const double* gb = rv + (int64_t)L * N;
        const double* Hb = gb + (int64_t)P * N;
        // loading
        for (int l = 0; l < L; ++l) g[l] = MtY[(int64_t)l * N + n] - g1[(int64_t)l * N + n];
        bool newton = A.use_hessian != 0;
        if (newton) {
            int k = 0;
            for (int i = 0; i < L; ++i)
                for (int j = 0; j <= i; ++j, ++k) H[i * L + j] = Hs[(int64_t)k * N + n];
            for (int l = 0; l < L; ++l) H[l * L + l] += rv[(int64_t)l * N + n] + A.eps;
            for (int l = 0; l < L; ++l) gs[l] = g[l];
            if (chol_solve_small(L, H, gs)) {
                for (int l = 0; l < L; ++l) g[l] = gs[l];
            } else {
                newton = false;
                atomicAdd(A.fail, 1);
            }
        }
        for (int l = 0; l < L; ++l) {
            double st = newton ? g[l] : A.lr * g[l];
            st = fmin(fmax(st, -A.da_bound), A.da_bound);
            A.da[(int64_t)l * N + n] = st;
            A.a[(int64_t)l * N + n] += st;
        }
        // bias / regression (uses the same, un-refreshed rate: core.py:205)
        for (int j = 0; j < P; ++j) g[j] = XtY[(int64_t)j * N + n] - gb[(int64_t)j * N + n];
        newton = A.use_hessian != 0;
        if (newton) {
            int k = 0;
            for (int i = 0; i < P; ++i)
                for (int j = 0; j <= i; ++j, ++k) H[i * P + j] = Hb[(int64_t)k * N + n];
            for (int j = 0; j < P; ++j) H[j * P + j] += A.eps;
            for (int j = 0; j < P; ++j) gs[j] = g[j];
            if (chol_solve_small(P, H, gs)) {
                for (int j = 0; j < P; ++j) g[j] = gs[j];
            } else {
                newton = false;
                atomicAdd(A.fail, 1);
            }
        }
        for (int j = 0; j < P; ++j) {
            double st = newton ? g[j] : A.lr * g[j];
            st = fmin(fmax(st, -A.db_bound), A.db_bound);
            A.db[(int64_t)j * N + n] = st;
            A.b[(int64_t)j * N + n] += st;
        }
    } else {
        // Gaussian channel: a_n = (M'M + diag(sum v))^-1 M'(y_n - X_n b_n), then
        // b_n = (X_n'X_n)^-1 X_n'(y_n - M a_n), b_n[1:] = 0   (core.py:224-235)
        const double* gram = A.lat;
        const double* sumv = A.lat + tri(L) + L;
        int k = 0;
        for (int i = 0; i < L; ++i)
            for (int j = 0; j <= i; ++j, ++k) H[i * L + j] = gram[k];
        for (int l = 0; l < L; ++l) H[l * L + l] += sumv[l];
        for (int l = 0; l < L; ++l) {
            double t = MtY[(int64_t)l * N + n];
            for (int j = 0; j < P; ++j) t -= XtM[((int64_t)j * L + l) * N + n] * A.b[(int64_t)j * N + n];
            g[l] = t;
        }
        if (!chol_solve_small(L, H, g)) {
            atomicAdd(A.fail, 1);
            return;
        }
        for (int l = 0; l < L; ++l) A.a[(int64_t)l * N + n] = g[l];
        for (int l = 0; l < L; ++l) an[l] = g[l];
        k = 0;
        for (int i = 0; i < P; ++i)
            for (int j = 0; j <= i; ++j, ++k) H[i * P + j] = XtX[(int64_t)k * N + n];
        for (int j = 0; j < P; ++j) {
            double t = XtY[(int64_t)j * N + n];
            for (int l = 0; l < L; ++l) t -= XtM[((int64_t)j * L + l) * N + n] * an[l];
            g[j] = t;
        }
        if (!chol_solve_small(P, H, g)) {
            atomicAdd(A.fail, 1);
            return;
        }
        A.b[(int64_t)0 * N + n] = g[0];
        for (int j = 1; j < P; ++j) A.b[(int64_t)j * N + n] = 0.0;
    }
}

__global__ void __launch_bounds__(64) mstep_solve_kernel(SolveArgs A) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n < A.N) mstep_solve_channel(A, n);
}

// Newton update of one Poisson channel with the dimensions known at compile time and no regressors besides the
// constant one (P == 1): H, g live in registers (the generic routine's dynamically indexed arrays live in scratch
// memory: ~100 dependent scratch round trips per 5 x 5 system).  Same arithmetic, same order as
// mstep_solve_channel / chol_solve_small.
template <int LT>
__device__ void mstep_solve_poisson_fixed(const SolveArgs& A, int n) {
    const int N = A.N;
    const double* MtY = A.prep;
    const double* XtY = A.prep + (int64_t)LT * N;
    const double* g1 = A.stats;
    const double* Hs = g1 + (int64_t)LT * N;
    const double* rv = Hs + (int64_t)tri(LT) * N;
    const double* gb = rv + (int64_t)LT * N;
    const double* Hb = gb + (int64_t)N;
    double H[LT][LT], g[LT], gs[LT];
#pragma unroll
    for (int l = 0; l < LT; ++l) g[l] = MtY[(int64_t)l * N + n] - g1[(int64_t)l * N + n];
    bool newton = A.use_hessian != 0;
    if (newton) {
        int k = 0;
#pragma unroll
        for (int i = 0; i < LT; ++i)
#pragma unroll
            for (int j = 0; j <= i; ++j, ++k) H[i][j] = Hs[(int64_t)k * N + n];
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            H[l][l] += rv[(int64_t)l * N + n] + A.eps;
            gs[l] = g[l];
        }
        bool ok = true;
#pragma unroll
        for (int k2 = 0; k2 < LT; ++k2) {
            double sd = H[k2][k2];
#pragma unroll
            for (int i = 0; i < k2; ++i) sd -= H[k2][i] * H[k2][i];
            ok = ok && (sd > 0.0) && (sd < 1e300);
            const double r = sqrt(sd);
            H[k2][k2] = r;
#pragma unroll
            for (int j = k2 + 1; j < LT; ++j) {
                double t = H[j][k2];
#pragma unroll
                for (int i = 0; i < k2; ++i) t -= H[j][i] * H[k2][i];
                H[j][k2] = t / r;
            }
        }
        if (ok) {
#pragma unroll
            for (int i = 0; i < LT; ++i) {
                double t = gs[i];
#pragma unroll
                for (int j = 0; j < i; ++j) t -= H[i][j] * gs[j];
                gs[i] = t / H[i][i];
            }
#pragma unroll
            for (int i = LT - 1; i >= 0; --i) {
                double t = gs[i];
#pragma unroll
                for (int j = i + 1; j < LT; ++j) t -= H[j][i] * gs[j];
                gs[i] = t / H[i][i];
            }
#pragma unroll
            for (int l = 0; l < LT; ++l) g[l] = gs[l];
        } else {
            newton = false;
            atomicAdd(A.fail, 1);
        }
    }
#pragma unroll
    for (int l = 0; l < LT; ++l) {
        double st = newton ? g[l] : A.lr * g[l];
        st = fmin(fmax(st, -A.da_bound), A.da_bound);
        A.da[(int64_t)l * N + n] = st;
        A.a[(int64_t)l * N + n] += st;
    }
    // bias (uses the same, un-refreshed rate: core.py:205); 1 x 1 system
    double gbv = XtY[n] - gb[n];
    newton = A.use_hessian != 0;
    if (newton) {
        const double hb = Hb[n] + A.eps;
        if ((hb > 0.0) && (hb < 1e300)) {
            const double r = sqrt(hb);
            gbv = (gbv / r) / r;
        } else {
            newton = false;
            atomicAdd(A.fail, 1);
        }
    }
    double st = newton ? gbv : A.lr * gbv;
    st = fmin(fmax(st, -A.db_bound), A.db_bound);
    A.db[n] = st;
    A.b[n] += st;
}

// Single rank: the fixed-order sum of the partials and the per-channel solves in ONE launch -- every block sums its
// 32 outputs, the block that draws the last ticket then runs the solves (one launch boundary and the latency of a
// 2-wave launch less per Newton iteration).  Sixteen row slices per output, eight independent loads in flight per
// thread (the first version walked its 64 rows two at a time: 32 dependent trips to L2 / MALL per launch, 42 us for
// 11 MB of partials at C3); FIXED is the number of latents when the register-resident solve applies, 0 otherwise.
#define MS_OUT 32
#define MS_SL 16
// ANYG: the set has Gaussian channels (or FIXED == 0): the general solve is compiled in -- its SOLVE_MAXD-sized local arrays
// are 2.3 KB of scratch per lane that every launch of an all-Poisson fit (C1 .. C4) used to reserve without touching it
template <int FIXED, bool ANYG = true>
__global__ void __launch_bounds__(512) mstep_sum_solve_kernel(const double* __restrict__ partial, int G, int64_t K,
                                                              double* __restrict__ out, unsigned* ticket, SolveArgs A) {
    __shared__ double red[MS_SL][MS_OUT];
    __shared__ int s_last;
    const int o = threadIdx.x & (MS_OUT - 1), sl = threadIdx.x / MS_OUT;
    const int64_t i = (int64_t)blockIdx.x * MS_OUT + o;
    double acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.0;
    if (i < K) {
        int g = sl;
        for (; g + 7 * MS_SL < G; g += 8 * MS_SL) {
            double ld[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) ld[q] = partial[(int64_t)(g + q * MS_SL) * K + i];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += ld[q];
        }
        for (int q = 0; g < G; g += MS_SL, ++q) acc[q] += partial[(int64_t)g * K + i];
    }
    red[sl][o] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
    __syncthreads();
    if (sl == 0 && i < K) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < MS_SL; ++q) t += red[q][o];
        out[i] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned tk = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = tk == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        *ticket = 0;  // the next launch is stream-ordered after this one
    }
    __syncthreads();
    for (int n = threadIdx.x; n < A.N; n += 512) {
        if constexpr (FIXED > 0 && !ANYG) {
            mstep_solve_poisson_fixed<FIXED>(A, n);
        } else {
            if constexpr (FIXED > 0) {
                if (!A.gauss[n]) {
                    mstep_solve_poisson_fixed<FIXED>(A, n);
                    continue;
                }
            }
            mstep_solve_channel(A, n);
        }
    }
}

__global__ void noise_mean_kernel(int N, double count, const double* s1, double* mean) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < N) mean[n] = s1[n] / count;
}
__global__ void noise_final_kernel(int N, double count, const double* s2, double* noise) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n < N) noise[n] = s2[n] / count;
}
// noise = var(y - eta) per channel (core.py:177) from sums the M-step holds anyway -- no pass over y:
//   sum (y - eta) = 1'y - a'(1'mu) - b'(1'x),   sum (y - eta)^2 = y'y - 2 (a'(mu'y) + b'(x'y)) + a'(mu'mu) a + 2 b'(x'mu) a + b'(x'x) b.
// The differences cancel where the fit is close (a Gaussian channel at high signal-to-noise): taken only for sets without
// Gaussian channels, whose noise is a by-product (y - log rate: residual and mean both O(1)); sets with Gaussian
// channels keep the two passes (mean first, then the centred squares, like np.var).
__global__ void noise_stats_kernel(int N, int L, int P, double count, const double* prep, const double* lat,
                                   const double* a, const double* b, double* noise) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const double* MtY = prep;
    const double* XtY = prep + (int64_t)L * N;
    const double* XtM = XtY + (int64_t)P * N;
    const double* XtX = XtM + (int64_t)P * L * N;
    const double* E = XtX + (int64_t)tri(P) * N;  // y'y | 1'y | 1'x
    const double* gram = lat;
    const double* smu = lat + tri(L);
    double s_eta = 0.0, s_yeta = 0.0, s_eta2 = 0.0;
    for (int l = 0; l < L; ++l) {
        const double al = a[(int64_t)l * N + n];
        s_eta = fma(al, smu[l], s_eta);
        s_yeta = fma(al, MtY[(int64_t)l * N + n], s_yeta);
        for (int m = 0; m < L; ++m) {
            const int i = l > m ? l : m, j = l > m ? m : l;
            s_eta2 = fma(al * a[(int64_t)m * N + n], gram[i * (i + 1) / 2 + j], s_eta2);
        }
    }
    for (int j = 0; j < P; ++j) {
        const double bj = b[(int64_t)j * N + n];
        s_eta = fma(bj, E[(int64_t)(2 + j) * N + n], s_eta);
        s_yeta = fma(bj, XtY[(int64_t)j * N + n], s_yeta);
        for (int l = 0; l < L; ++l)
            s_eta2 = fma(2.0 * bj * a[(int64_t)l * N + n], XtM[((int64_t)j * L + l) * N + n], s_eta2);
        for (int q = 0; q < P; ++q) {
            const int i = j > q ? j : q, k = j > q ? q : j;
            s_eta2 = fma(bj * b[(int64_t)q * N + n], XtX[((int64_t)i * (i + 1) / 2 + k) * N + n], s_eta2);
        }
    }
    const double s1 = E[(int64_t)N + n] - s_eta;
    const double s2 = (E[n] - 2.0 * s_yeta) + s_eta2;
    const double mean = s1 / count;
    noise[n] = s2 / count - mean * mean;
}
__global__ void zero_kernel(int64_t n, double* p) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

// ---------------------------------------------------------------------------
struct Geometry {
    int G, rows_per_wg, CT, S, nthr, tiles;
    size_t lds;
};

static Geometry plan(vlgp_ctx* ctx, int64_t rows) {
    Geometry g;
    const int N = ctx->N, L = ctx->L;
    if (N <= 512) {
        g.CT = N;
        double best = -1.0;
        g.nthr = 512;
        for (int k = 2; k <= 8; ++k) {
            const int nt = 64 * k;
            if (nt < N) continue;
            const double u = (double)((nt / N) * N) / nt;
            if (u > best + 1e-9) { best = u; g.nthr = nt; }
        }
        g.S = g.nthr / N;
        g.tiles = 1;
    } else {
        g.CT = 512;
        g.nthr = 512;
        g.S = 1;
        g.tiles = (N + 511) / 512;
    }
    int64_t G = (rows + 63) / 64;
    // ONE workgroup per CU (round 4; two until then).  Alone the accumulation is 3 % slower that way (2.22 against
    // 2.15 ms per M-step at C3), but the M-step runs BESIDE the H-step's rounds, which are the critical path of the phase:
    // two resident workgroups (14 waves, 256 of a SIMD's 512 registers) leave a CU room for one workgroup of the low-rank
    // round instead of three.  C3: H-step 4.3 -> 4.0 ms, 133-134 -> 137-141 EM it/s on the same boxes; 0.75 or 0.5 per CU
    // make the M-step itself the longer lane (125-131), and so do smaller workgroups (320 / 256 threads instead of 448:
    // 119 / 127 against 132-133 on the same box).  VLGP_MSTEP_WG_PER_CU overrides.
    static const double wg_per_cu = getenv("VLGP_MSTEP_WG_PER_CU") ? atof(getenv("VLGP_MSTEP_WG_PER_CU")) : 1.0;
    const int64_t cap = (int64_t)((wg_per_cu > 0 ? wg_per_cu : 1.0) * ctx->n_cu);
    if (G > cap) G = cap;
    if (G < 1) G = 1;
    g.rows_per_wg = (int)((rows + G - 1) / G);
    g.rows_per_wg = ((g.rows_per_wg + 7) / 8) * 8;  // an even split over the workgroups: every CU gets the same share
    g.G = (int)((rows + g.rows_per_wg - 1) / g.rows_per_wg);
    {   // the row tile (mu | v) aliased by the MS_GS-statistic reduction buffer, + the exp table
        const size_t tiles = (size_t)2 * M_TILE * L, redb = (size_t)MS_GS * g.S * g.CT;
        g.lds = ((tiles > redb ? tiles : redb) + 64) * 8;
    }
    return g;
}

template <int LT, int PT>
static void launch_accum_k(hipStream_t st, int kind, const Geometry& g, const MArgs& A) {
    dim3 grid(g.G, g.tiles), blk(g.nthr);
    if (kind == K_NEWTON && A.L == LT && A.P == PT && PT == 1 && A.x == nullptr) {  // the hot launch, specialised
        hipLaunchKernelGGL((mstep_accum<LT, PT, K_NEWTON, true>), grid, blk, g.lds, st, A);
        return;
    }
    switch (kind) {
        case K_PREP: hipLaunchKernelGGL((mstep_accum<LT, PT, K_PREP, false>), grid, blk, g.lds, st, A); break;
        case K_NEWTON: hipLaunchKernelGGL((mstep_accum<LT, PT, K_NEWTON, false>), grid, blk, g.lds, st, A); break;
        case K_NOISE1: hipLaunchKernelGGL((mstep_accum<LT, PT, K_NOISE1, false>), grid, blk, g.lds, st, A); break;
        default: hipLaunchKernelGGL((mstep_accum<LT, PT, K_NOISE2, false>), grid, blk, g.lds, st, A); break;
    }
}
template <int LT>
static int launch_accum_p(vlgp_ctx* ctx, int kind, const Geometry& g, const MArgs& A) {
    const int P = A.P;
    hipStream_t mst = ctx->mstream;
    if (P <= 1) launch_accum_k<LT, 1>(mst, kind, g, A);
    else if (P <= 2) launch_accum_k<LT, 2>(mst, kind, g, A);
    else if (P <= 4) launch_accum_k<LT, 4>(mst, kind, g, A);
    else if (P <= 8) launch_accum_k<LT, 8>(mst, kind, g, A);
    else return vlgp_fail(ctx, VLGP_ERR_ARG, "M-step kernel supports xdim <= 8, got %d", P);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}
// the loop-based fallback (mstep_cache_gen + mstep_accum_gen): cache = (rows x N) doubles of the M-step workspace
// (VLGP_MSTEP_GENERIC=1 takes it at any size: the tests compare it with the specialised kernels)
static bool mstep_generic(int L, int P) { return L > 16 || P > 8 || getenv("VLGP_MSTEP_GENERIC") != nullptr; }
static int launch_accum_gen(vlgp_ctx* ctx, int kind, const Geometry& g, const MArgs& A, double* cache) {
    hipStream_t st = ctx->mstream;
    if (kind != K_PREP) {
        int64_t blocks = (A.rows * A.N + 255) / 256;
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(mstep_cache_gen, dim3((unsigned)blocks), dim3(256), 0, st, A, kind, cache);
    }
    const int K = nstat_rt(A.L, A.P, kind);
    hipLaunchKernelGGL(mstep_accum_gen, dim3(g.G, (A.N + 255) / 256, (K + MG_CH - 1) / MG_CH), dim3(256), 0, st, A, kind, K,
                       cache);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

static int launch_accum(vlgp_ctx* ctx, int kind, const Geometry& g, const MArgs& A, double* cache = nullptr) {
    const int L = A.L;
    if (mstep_generic(A.L, A.P)) return launch_accum_gen(ctx, kind, g, A, cache);
    if (L <= 2) return launch_accum_p<2>(ctx, kind, g, A);
    if (L <= 3) return launch_accum_p<3>(ctx, kind, g, A);
    if (L <= 5) return launch_accum_p<5>(ctx, kind, g, A);
    if (L <= 8) return launch_accum_p<8>(ctx, kind, g, A);
    if (L <= 10) return launch_accum_p<10>(ctx, kind, g, A);
    if (L <= 16) return launch_accum_p<16>(ctx, kind, g, A);
    return vlgp_fail(ctx, VLGP_ERR_ARG, "M-step kernel supports at most 16 latents, got %d", L);
}

template <int LT>
static void launch_lat_t(hipStream_t st, int G, int L, int64_t rows, const double* mu, const double* v,
                         const double* dmu, double* partial) {
    hipLaunchKernelGGL((latent_moments_kernel<LT>), dim3(G), dim3(256), 0, st, L, rows, mu, v, dmu, partial);
}

// d_out: tri(L) gram | L sum_mu | L sum_v | L sum_mu^2 | 1 |dmu|^2 ; all-reduced over ranks
static int latent_moments_st(vlgp_ctx* ctx, UnitSet& us, double* d_partial, double* d_out, hipStream_t st) {
    const int L = ctx->L;
    int G = (int)((us.rows + 255) / 256);
    if (G > 256) G = 256;
    if (G < 1) G = 1;
    const int K = tri(L) + 3 * L + 1;
    if (L > 16 || getenv("VLGP_MSTEP_GENERIC"))
        hipLaunchKernelGGL(latent_moments_gen, dim3(G, (K + MG_CH - 1) / MG_CH), dim3(256), 0, st, L, us.rows, us.mu, us.v,
                           us.dmu, d_partial);
    else if (L <= 2) launch_lat_t<2>(st, G, L, us.rows, us.mu, us.v, us.dmu, d_partial);
    else if (L <= 3) launch_lat_t<3>(st, G, L, us.rows, us.mu, us.v, us.dmu, d_partial);
    else if (L <= 5) launch_lat_t<5>(st, G, L, us.rows, us.mu, us.v, us.dmu, d_partial);
    else if (L <= 8) launch_lat_t<8>(st, G, L, us.rows, us.mu, us.v, us.dmu, d_partial);
    else if (L <= 10) launch_lat_t<10>(st, G, L, us.rows, us.mu, us.v, us.dmu, d_partial);
    else launch_lat_t<16>(st, G, L, us.rows, us.mu, us.v, us.dmu, d_partial);
    HIPCHK(ctx, hipGetLastError());
    hipLaunchKernelGGL(sum_partials_kernel, dim3((K + 63) / 64), dim3(512), 0, st, d_partial, G,
                       (int64_t)K, d_out);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}
static int latent_moments(vlgp_ctx* ctx, UnitSet& us, double* d_partial, double* d_out, bool mlane) {
    CHK(latent_moments_st(ctx, us, d_partial, d_out, mlane ? ctx->mstream : ctx->stream));
    const int K = tri(ctx->L) + 3 * ctx->L + 1;
    return mlane ? vlgp_allreduce_m(ctx, d_out, K) : vlgp_allreduce(ctx, d_out, K);
}

// result (K doubles, all-reduced) is left at the head of ctx->d_work; the partials use its tail.  The
// workspace may be reallocated in here: take ctx->d_work AFTER the call, never before.
int launch_moments(vlgp_ctx* ctx, UnitSet& us) {
    const int L = ctx->L;
    const int K = tri(L) + 3 * L + 1;
    CHK(vlgp_ensure_work(ctx, 256LL * K + K + 64));
    double* d_partial = ctx->d_work + K + 64;
    return latent_moments(ctx, us, d_partial, ctx->d_work, false);
}

int launch_mstep(vlgp_ctx* ctx, UnitSet& us, int n_iter, int use_hessian, double eps, double lr,
                 double da_bound, double db_bound) {
    const int N = ctx->N, L = ctx->L, P = ctx->P;
    const bool gen = mstep_generic(L, P);  // beyond the compiled sizes: loop-based kernels, solves in global memory
    const Geometry g = plan(ctx, us.rows);
    const int Kp = nstat_rt(L, P, K_PREP), Kn = nstat_rt(L, P, K_NEWTON);
    const int Kl = tri(L) + 3 * L + 1;
    const int Kmax = Kp > Kn ? Kp : Kn;
    // workspace: prep | stats | lat | noise1 | mean | partial
    const int64_t o_prep = 0, o_stats = o_prep + (int64_t)Kp * N, o_lat = o_stats + (int64_t)Kn * N;
    const int64_t o_s1 = o_lat + Kl + 8, o_mean = o_s1 + N, o_tick = o_mean + N, o_part = o_tick + 2;
    int64_t part_len = (int64_t)g.G * Kmax * N;
    if (part_len < 256LL * Kl) part_len = 256LL * Kl;
    const int Dg = L > P ? L : P;
    const int64_t o_cache = o_part + part_len, o_hws = o_cache + (gen ? us.rows * N : 0);
    CHK(vlgp_ensure_work_m(ctx, o_hws + (gen ? (int64_t)N * Dg * (Dg + 3) : 0)));
    double* W = ctx->d_work_m;
    hipStream_t st = ctx->mstream;
    double *d_prep = W + o_prep, *d_stats = W + o_stats, *d_lat = W + o_lat, *d_s1 = W + o_s1,
           *d_mean = W + o_mean, *d_part = W + o_part;
    double* d_cache = gen ? W + o_cache : nullptr;
    unsigned* d_ticket = reinterpret_cast<unsigned*>(W + o_tick);

    // Single rank, no per-launch timing: record the whole sequence once and replay it (see ctx.h).
    static const bool no_graph = getenv("VLGP_NO_MGRAPH") != nullptr;
    const bool use_graph = ctx->world == 1 && !ctx->prof_on && !no_graph;
    const bool noise_passes = ctx->n_gauss > 0 || P > 2 || getenv("VLGP_NOISE_PASSES") != nullptr;  // (see noise_stats_kernel)
    std::vector<double> key;
    if (use_graph) {
        auto pk = [&](const void* p_) { key.push_back((double)(uintptr_t)p_); };
        pk(us.y); pk(us.x_ones ? nullptr : us.x); pk(us.mu); pk(us.v); pk(us.w); pk(us.dmu); pk(W); pk(ctx->d_a); pk(ctx->d_b);
        pk(ctx->d_noise); pk(ctx->d_da); pk(ctx->d_db); pk(ctx->d_fail_m); pk(ctx->d_gauss);
        for (double v_ : {(double)us.rows, (double)N, (double)L, (double)P, (double)ctx->n_gauss, (double)n_iter,
                          (double)use_hessian, eps, lr, da_bound, db_bound, (double)noise_passes})
            key.push_back(v_);
        if (ctx->m_graph_exec && key == ctx->m_graph_key) {
            HIPCHK(ctx, hipGraphLaunch(static_cast<hipGraphExec_t>(ctx->m_graph_exec), st));
            return VLGP_OK;
        }
        if (ctx->m_graph_exec) {
            (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(ctx->m_graph_exec));
            ctx->m_graph_exec = nullptr;
        }
        HIPCHK(ctx, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    }
    auto enqueue = [&]() -> int {
    HIPCHK(ctx, hipMemsetAsync(d_ticket, 0, sizeof(double) * 2, st));

    MArgs A;
    A.N = N; A.L = L; A.P = P; A.rows = us.rows; A.rows_per_wg = g.rows_per_wg; A.CT = g.CT; A.S = g.S;
    A.y = us.y; A.x = us.x_ones ? nullptr : us.x; A.mu = us.mu; A.v = us.v;
    A.a = ctx->d_a; A.b = ctx->d_b; A.gauss = ctx->d_gauss; A.mean = d_mean; A.partial = d_part;

    auto reduce_to = [&](int K, double* dst) -> int {
        const int64_t n = (int64_t)K * N;
        hipLaunchKernelGGL(sum_partials_kernel, dim3((unsigned)((n + 63) / 64)), dim3(512), 0, st,
                           d_part, g.G, n, dst);
        HIPCHK(ctx, hipGetLastError());
        return vlgp_allreduce_m(ctx, dst, n);
    };

    // sweep-invariant moments
    CHK(launch_accum(ctx, K_PREP, g, A, d_cache));
    CHK(reduce_to(Kp, d_prep));
    CHK(latent_moments(ctx, us, d_part, d_lat, true));
    double total_rows = (double)us.rows;
    if (ctx->world > 1 && us.rows_all_ranks > 0.0) {
        total_rows = us.rows_all_ranks;  // fixed at upload: exchanged once per set, not once per M-step
    } else if (ctx->world > 1) {
        // total row count over ranks rides along in the workspace
        double h = total_rows;
        HIPCHK(ctx, hipMemcpyAsync(d_lat + Kl, &h, sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        CHK(vlgp_allreduce_m(ctx, d_lat + Kl, 1));
        HIPCHK(ctx, hipMemcpyAsync(&h, d_lat + Kl, sizeof(double), hipMemcpyDeviceToHost, st));
        HIPCHK(ctx, hipStreamSynchronize(st));
        total_rows = h;
        us.rows_all_ranks = h;
    }

    SolveArgs S;
    S.N = N; S.L = L; S.P = P; S.use_hessian = use_hessian; S.eps = eps; S.lr = lr;
    S.da_bound = da_bound; S.db_bound = db_bound;
    S.prep = d_prep; S.stats = d_stats; S.lat = d_lat; S.gauss = ctx->d_gauss;
    S.a = ctx->d_a; S.b = ctx->d_b; S.da = ctx->d_da; S.db = ctx->d_db; S.fail = ctx->d_fail_m;
    S.hws = gen ? W + o_hws : nullptr;

    const bool any_poisson = ctx->n_gauss < N;
    for (int it = 0; it < n_iter; ++it) {
        if (it == n_iter - 1) {
            // noise = var(y - eta) with the parameters entering the last iteration (core.py:177)
            if (!noise_passes) {
                hipLaunchKernelGGL(noise_stats_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N, L, P, total_rows,
                                   d_prep, d_lat, ctx->d_a, ctx->d_b, ctx->d_noise);
                HIPCHK(ctx, hipGetLastError());
            } else {
            CHK(launch_accum(ctx, K_NOISE1, g, A, d_cache));
            CHK(reduce_to(1, d_s1));
            hipLaunchKernelGGL(noise_mean_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N,
                               total_rows, d_s1, d_mean);
            CHK(launch_accum(ctx, K_NOISE2, g, A, d_cache));
            CHK(reduce_to(1, d_s1));
            hipLaunchKernelGGL(noise_final_kernel, dim3((N + 255) / 256), dim3(256), 0, st, N,
                               total_rows, d_s1, ctx->d_noise);
            HIPCHK(ctx, hipGetLastError());
            }
        }
        if (any_poisson) {
            vlgp_prof_begin(ctx, VLGP_PROF_MSTEP, st);
            int rc = launch_accum(ctx, K_NEWTON, g, A, d_cache);
            vlgp_prof_end(ctx, VLGP_PROF_MSTEP, (double)us.rows, st);
            CHK(rc);
            if (ctx->world == 1) {  // no all-reduce between the sum and the solves: one launch for both
                const int64_t n = (int64_t)Kn * N;
                const dim3 sg((unsigned)((n + MS_OUT - 1) / MS_OUT));
                const int fixed = P == 1 ? L : 0;  // register-resident solve for the usual latent counts, no regressors
                const bool anyg = ctx->n_gauss > 0;
#define MS_SUM_SOLVE(F)                                                                                                  \
    do {                                                                                                                  \
        if (anyg) hipLaunchKernelGGL((mstep_sum_solve_kernel<F, true>), sg, dim3(512), 0, st, d_part, g.G, n, d_stats, d_ticket, S);  \
        else hipLaunchKernelGGL((mstep_sum_solve_kernel<F, false>), sg, dim3(512), 0, st, d_part, g.G, n, d_stats, d_ticket, S);      \
    } while (0)
                if (fixed == 3) MS_SUM_SOLVE(3);
                else if (fixed == 5) MS_SUM_SOLVE(5);
                else if (fixed == 8) MS_SUM_SOLVE(8);
                else if (fixed == 10) MS_SUM_SOLVE(10);
#undef MS_SUM_SOLVE
                else hipLaunchKernelGGL(mstep_sum_solve_kernel<0>, sg, dim3(512), 0, st, d_part, g.G, n, d_stats, d_ticket, S);
                HIPCHK(ctx, hipGetLastError());
                continue;
            }
            CHK(reduce_to(Kn, d_stats));
        }
        hipLaunchKernelGGL(mstep_solve_kernel, dim3((N + 63) / 64), dim3(64), 0, st, S);
        HIPCHK(ctx, hipGetLastError());
    }
    return VLGP_OK;
    };
    const int rc = enqueue();
    if (!use_graph) return rc;
    hipGraph_t graph = nullptr;
    const hipError_t ec = hipStreamEndCapture(st, &graph);  // always leave capture mode, whatever enqueue() said
    if (rc != VLGP_OK) {
        if (graph) (void)hipGraphDestroy(graph);
        return rc;
    }
    HIPCHK(ctx, ec);
    hipGraphExec_t exec = nullptr;
    const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    (void)hipGraphDestroy(graph);
    HIPCHK(ctx, ei);
    ctx->m_graph_exec = exec;
    ctx->m_graph_key = key;
    HIPCHK(ctx, hipGraphLaunch(exec, st));
    return VLGP_OK;
}
