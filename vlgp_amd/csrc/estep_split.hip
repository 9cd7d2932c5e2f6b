// E-step for many window-sized units (T <= 64, effective rank <= 32) as a SEQUENCE of chip-wide launches
// instead of one persistent workgroup per unit (estep_fast.hip).
//
// Same mathematics, phase order and per-(unit, latent) arithmetic as estep_fast.hip / estep.hip (reference
// vlgp/core.py:22-120): per sweep  residual pass -> mean update -> curvature pass -> factor (+ variance).
// Why split: inside one workgroup the four phases have opposite shapes -- the (T x N) passes want every lane on
// a time bin and no synchronisation, the per-latent phases are chains of dependent 16-wide steps that only
// many independent waves per SIMD can hide -- and the persistent form ties them together with barriers
// (measured: 40 k CU-cycles per unit-sweep against ~12 k of issued work, VALU 45 % busy).  Between launches the
// unit state (mu, v, w, residual projections: 8 MB each at 200 x 1000 x 5) makes a round trip through L2 /
// Infinity Cache, which costs less than the barriers did; a dependent launch boundary is ~1.5 us.
//
//   esplit_pass<KIND>   lane <-> ROW of the packed unit set (any unit: rows are independent in the passes), all
//                       N channels in sequence per lane: no cross-lane or cross-wave reduction, no LDS, 64 of 64
//                       lanes busy whatever the unit length.  Channel records (a_l, a_l^2, b, 1/noise, id),
//                       Poisson channels first, are wave-uniform scalar loads, the next record in flight while
//                       the current one is consumed.
//   esplit_factor       one wave per (unit, latent): I + G'WG on the matrix pipe, factor + inverse in registers
//                       (rank <= 16) or through LDS (rank <= 32), variance update; X goes to global memory.
//   esplit_mean         one wave per (unit, latent): Newton step on the posterior mean through X.
//                       Each wave takes the code path of ITS latent's rank: one latent above 16 no longer moves
//                       the whole launch to the slow instantiation.
#include <stdlib.h>

#include <algorithm>

#include <type_traits>

#include "estep_args.h"
#include "wave_tri.h"
#include "fast_exp.h"

namespace {

enum { SP_YA = 0, SP_RES = 1, SP_W = 2 };
typedef double double4_t __attribute__((ext_vector_type(4)));

struct SplitArgs {
    int N, L, M;
    int64_t rows;
    const int64_t* off;
    const int* unit_prior;
    const double* const* prior_base;
    const int* prior_rl;
    const int64_t* prior_goff;
    const double* y;
    const double* xb;
    double *mu, *v, *w;  // LATENT-MAJOR working copies (L, ld) of the unit-set arrays: element (row, l) at [l ld + row]
    double* dmu;         // the unit-set array itself, (rows, L)
    int64_t ld;          // = rows
    double *ra, *ya;     // (L, ld) as well
    double *sv, *dl;     // (L, ld): s = ra + w mu (written by the residual pass for the lane-per-task mean launch), and that
                         // launch's raw step target G (I + H)^-1 G's: the curvature pass that follows applies
                         // mu += clip(dl - mu) for the latents in `dmask` (estep_lane.h)
    unsigned dmask;
    double* xg;        // (M L, pkg): packed X = chol(I + G'WG)^-1 per (unit, latent)
    unsigned long long* clk;  // debug: per-phase cycle counters of the first wave of the lane-per-task launches, or null
    int prio, clk_kind;    // lane-per-task launches: s_setprio level of their waves; warm the scalar cache with G first
    double* xl;        // lane-per-task launches (estep_lane.h): X entry-major, 64 LANE_EMAX doubles per (group of 64 units, latent)
    int pkg;           // stride of xg
    int pkl;           // doubles of LDS per wave for the packed X of this launch's rank class
    int* failg;        // (M L): 1 = the factor of this (unit, latent) failed
    int* fail;
    const double* wconst;
    double dmu_bound;
    int np, ntot;      // Poisson channels, all channels
    int lds_g;         // doubles of LDS per wave for the G tile
    int do_v, last;
    int n_lat;         // latents covered by this launch (tasks = M n_lat)
    int shg;           // 1: every unit has the same prior; a workgroup = one latent x four units, G staged ONCE per
                       //    workgroup at the start of the LDS (shg_cap doubles) instead of once per wave
    int shg_cap, shg_T;
    int shg_rk[16];            // shared-G launches: rank and compact factor of lat[i] (no table lookups per wave)
    const double* shg_gl[16];
    int lat[16];       // their indices
};

// ---------------------------------------------------------------------------------------------------------
// channel records, Poisson channels first: a[LT] | a^2 / 2 [LT] | b | c (1/noise or 1) | id (integer bits) | pad
template <int LT>
constexpr int rec_len() { return (2 * LT + 3 + 1) & ~1; }

__global__ void __launch_bounds__(256)
esplit_cols_kernel(int N, int L, int LT, int REC, const double* __restrict__ a, const double* __restrict__ b,
                   const double* __restrict__ noise, const int* __restrict__ gauss, double* __restrict__ cols,
                   double* __restrict__ wconst, double* __restrict__ ycoef) {
    // (one workgroup in front of every E-step call: every load of a thread is issued before its first store -- with
    // pointers that might alias each load waited for the store before it, 21 us for a hundred channels)
    __shared__ int order[1024];
    __shared__ int gs[1024];
    __shared__ double cinv[1024];  // 1 / noise on Gaussian channels, else 1
    for (int n = threadIdx.x; n < N; n += 256) {
        const int g = gauss[n];
        gs[n] = g;
        cinv[n] = g ? 1.0 / noise[n] : 1.0;
    }
    __syncthreads();
    // Poisson channels first, both groups in channel order: every channel counts the Gaussian ones before it (the loads
    // of a thread are independent; one thread walking the list was a chain of 2 N dependent LDS accesses, 17 us at N = 100)
    {
        int total = 0;
        for (int m = 0; m < N; ++m) total += gs[m];
        for (int n = threadIdx.x; n < N; n += 256) {
            int c = 0;
            for (int m = 0; m < n; ++m) c += gs[m];
            order[gs[n] ? (N - total) + c : n - c] = n;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 256) {
        const int n = order[i];
        double av[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) av[l] = (l < L && l < LT) ? a[l * N + n] : 0.0;
        const double bn = b[n];
        double* rec = cols + (int64_t)i * REC;
#pragma unroll
        for (int l = 0; l < 16; ++l) {
            if (l < LT) {
                rec[l] = av[l];
                rec[LT + l] = 0.5 * av[l] * av[l];  // half squares: the rate's exponent is one chain b + mu.a + v.(a^2 / 2)
            }
        }
        rec[2 * LT] = bn;
        rec[2 * LT + 1] = cinv[n];
        rec[2 * LT + 2] = __longlong_as_double((long long)n);  // channel id, read back as an integer
    }
    // channel-major coefficients of the y pass, ORIGINAL channel order: ycoef[n][l] = a_ln (1/noise_n or 1)
    for (int n = threadIdx.x; n < N; n += 256) {
        double av[16];
#pragma unroll
        for (int l = 0; l < 16; ++l) av[l] = (l < L && l < LT) ? a[l * N + n] : 0.0;
        const double c = cinv[n];
#pragma unroll
        for (int l = 0; l < 16; ++l)
            if (l < LT) ycoef[n * LT + l] = l < L ? av[l] * c : 0.0;
    }
    if ((int)threadIdx.x < L) {  // w = U (a')^2 with U = 1/noise on Gaussian channels (core.py:103-104)
        double s = 0.0;
        for (int n = 0; n < N; ++n)
            if (gs[n]) s = fma(a[threadIdx.x * N + n] * a[threadIdx.x * N + n], cinv[n], s);
        wconst[threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------
// (rows, L) <-> (L, rows): the split kernels work on latent-major copies -- a wave whose lanes are rows (passes) or time
// bins of one latent (latent kernels) then touches 8 contiguous bytes per lane instead of 8 of every 8 L (PMC: ~500
// L1 line accesses per wave of the mean launch with the interleaved layout, most of them for 3 x 50 doubles).
__global__ void __launch_bounds__(256)
esplit_to_lm(int L, int64_t rows, const double* __restrict__ a0, const double* __restrict__ a1,
             const double* __restrict__ a2, double* __restrict__ b0, double* __restrict__ b1, double* __restrict__ b2) {
    // a lane takes a row: a wave reads 64 L contiguous doubles per array and writes L runs of 64 (the first version went
    // by latent-major index: 8 of every 8 L bytes per load, 30 us against the 9 us of the way back)
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    for (int l = 0; l < L; ++l) {
        const double x0 = a0[row * L + l], x1 = a1[row * L + l], x2 = a2[row * L + l];
        b0[l * rows + row] = x0;
        b1[l * rows + row] = x1;
        b2[l * rows + row] = x2;
    }
}
__global__ void __launch_bounds__(256)
esplit_from_lm(int L, int64_t rows, const double* __restrict__ b0, const double* __restrict__ b1,
               const double* __restrict__ b2, double* a0, double* a1, double* a2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // over rows * L, row-major index
    if (i >= rows * L) return;
    const int64_t row = i / L, l = i - row * L;
    const int64_t src = l * rows + row;
    a0[i] = b0[src];
    a1[i] = b1[src];
    a2[i] = b2[src];
}

// ---------------------------------------------------------------------------------------------------------
// ya[row][l] = sum_n y[row][n] c_n a_ln, once per E-step call.  Sixteen lanes share a row and walk its channels 16 at a
// time (128 contiguous bytes per row and load), four rows per wave and step; the per-channel coefficients of a lane's
// channels stay in registers (N <= 16 NJ).  The lane-per-row form (esplit_pass<SP_YA>) reads 8 bytes of every 8 N:
// 466 MB of HBM traffic for 160 MB of y at C3 (PMC), 108 us.
template <int LT, int NJ>
__global__ void __launch_bounds__(256)
esplit_ya(int N, int L, int64_t rows, int64_t ld, const double* __restrict__ y, const double* __restrict__ ycoef,
          double* __restrict__ ya, int rows_per_wave) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int r4 = lane >> 4, c16 = lane & 15;
    double cf[NJ][LT];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = c16 + 16 * j;
#pragma unroll
        for (int l = 0; l < LT; ++l) cf[j][l] = n < N ? ycoef[n * LT + l] : 0.0;
    }
    const int64_t w0 = ((int64_t)blockIdx.x * 4 + wid) * rows_per_wave;
    for (int it = 0; it < rows_per_wave; it += 4) {
        const int64_t row = w0 + it + r4;
        const bool in = row < rows;
        const double* yr = y + (in ? row : 0) * N;
        double acc[LT];
#pragma unroll
        for (int l = 0; l < LT; ++l) acc[l] = 0.0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int n = c16 + 16 * j;
            const double yv = (in && n < N) ? yr[n] : 0.0;
#pragma unroll
            for (int l = 0; l < LT; ++l) acc[l] = fma(yv, cf[j][l], acc[l]);
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1)
#pragma unroll
            for (int l = 0; l < LT; ++l) acc[l] += __shfl_xor(acc[l], o, 64);
        if (in) {
#pragma unroll
            for (int l = 0; l < LT; ++l)
                if (c16 == l && l < L) ya[(int64_t)l * ld + row] = acc[l];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// CS: the channels of one row group are split over CS waves of the workgroup (a lone wave walking all N channels
// is a chain of N dependent record loads + exponentials: measured 35 us at two waves per SIMD and 41 us at four --
// latency, not throughput); the partial sums meet in LDS and are added in wave order (deterministic).
// RPL: rows per lane (rows row0 + 64 q): one channel record serves RPL rows -- fewer scalar loads per (row, channel)
// and RPL independent exp chains per record.
template <int LT, int KIND, bool HASXB, int CS, int RPL>
__global__ void __launch_bounds__(256)
esplit_pass(SplitArgs A, const double* __restrict__ cols) {
    constexpr int REC = rec_len<LT>();
    constexpr int RPB = (256 / CS) * RPL;  // rows per workgroup
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int part = CS == 1 ? 0 : wid % CS;
    const int64_t row0 = (int64_t)blockIdx.x * RPB + (wid / CS) * (64 * RPL) + lane;
    __shared__ double etab[256];  // 2^(j/256) for fast_exp_tab256
    if constexpr (KIND != SP_YA) {
        fast_exp_tab256_init(etab, threadIdx.x);
        __syncthreads();
    }
    const int N = A.N, L = A.L;
    bool in[RPL];
    int64_t rr[RPL];
    double mr[RPL][LT], vr[RPL][LT], acc[RPL][LT];
    const double* yrow[RPL];
    const double* xbrow[RPL];
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        in[q] = row0 + 64 * q < A.rows;
        rr[q] = in[q] ? row0 + 64 * q : 0;
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            const bool use = l < L && KIND != SP_YA;
            mr[q][l] = use ? A.mu[(int64_t)l * A.ld + rr[q]] : 0.0;
            vr[q][l] = use ? A.v[(int64_t)l * A.ld + rr[q]] : 0.0;
            acc[q][l] = 0.0;
        }
        yrow[q] = A.y + rr[q] * N;
        xbrow[q] = HASXB ? A.xb + rr[q] * N : nullptr;
    }
    // the residual pass subtracts from ya at the end: fetched here, by the wave that writes (a load after the channel
    // loop is one more trip to memory in the life of a short wave)
    double yav[RPL][LT], wv[RPL][LT];
    if constexpr (KIND == SP_RES) {
#pragma unroll
        for (int q = 0; q < RPL; ++q)
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                yav[q][l] = (part == 0 && l < L) ? A.ya[(int64_t)l * A.ld + rr[q]] : 0.0;
                wv[q][l] = (part == 0 && l < L && A.sv) ? A.w[(int64_t)l * A.ld + rr[q]] : 0.0;
            }
    }
    if constexpr (KIND == SP_W) {
        // the step of the lane-per-task mean launch, applied here: mu += clip(dl - mu) (core.py:91, 96); every wave of
        // the row group advances its copy, the wave that writes w stores it (after the barrier below: the others
        // have read the old value by then)
        if (A.dmask) {
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int l = 0; l < LT; ++l)
                    if (l < L && ((A.dmask >> l) & 1u)) {
                        double st = A.dl[(int64_t)l * A.ld + rr[q]] - mr[q][l];
                        st = fmin(fmax(st, -A.dmu_bound), A.dmu_bound);
                        mr[q][l] += st;
                    }
        }
    }
    auto load_rec = [&](int i, double (&rv)[REC]) {
        const double2* rp = reinterpret_cast<const double2*>(cols + (int64_t)i * REC);
#pragma unroll
        for (int q = 0; q < REC / 2; ++q) {
            const double2 t2 = rp[q];
            rv[2 * q] = t2.x;
            rv[2 * q + 1] = t2.y;
        }
    };
    // this wave's share of the Poisson list [p_lo, np) and of the Gaussian list [g_lo, ntot)
    const int p_lo = CS == 1 ? 0 : (A.np * part) / CS;
    const int np = CS == 1 ? A.np : (A.np * (part + 1)) / CS;
    const int g_lo = CS == 1 ? A.np : A.np + ((A.ntot - A.np) * part) / CS;
    const int ntot = CS == 1 ? A.ntot : A.np + ((A.ntot - A.np) * (part + 1)) / CS;
    if constexpr (KIND == SP_YA) {
        static_assert(CS == 1 && RPL == 1, "the y pass is not split");
        auto body = [&](const double (&rv)[REC]) {
            const int n = (int)__double_as_longlong(rv[2 * LT + 2]);
            const double yc = yrow[0][n] * rv[2 * LT + 1];
#pragma unroll
            for (int l = 0; l < LT; ++l) acc[0][l] = fma(yc, rv[l], acc[0][l]);
        };
        double ra_[REC], rb_[REC];
        load_rec(0, ra_);
        int i = 0;
        for (; i + 1 < ntot; i += 2) {
            load_rec(i + 1, rb_);
            body(ra_);
            load_rec(i + 2 < ntot ? i + 2 : i + 1, ra_);
            body(rb_);
        }
        if (i < ntot) body(ra_);
    } else {
        // one Poisson channel: rate = exp(min(eta + v.a^2/2, 10)) (math.trunc_exp, vlgp/math.py:24-38)
        auto poisson = [&](const double (&rv)[REC]) {
#pragma unroll
            for (int q = 0; q < RPL; ++q) {
                double eta = HASXB ? xbrow[q][(int)__double_as_longlong(rv[2 * LT + 2])] : rv[2 * LT];
#pragma unroll
                for (int l = 0; l < LT; ++l) eta = fma(mr[q][l], rv[l], eta);
#pragma unroll
                for (int l = 0; l < LT; ++l) eta = fma(vr[q][l], rv[LT + l], eta);
                const double rate = trunc_exp_tab256(eta, etab);
#pragma unroll
                for (int l = 0; l < LT; ++l) acc[q][l] = fma(rate, rv[KIND == SP_RES ? l : LT + l], acc[q][l]);
            }
        };
        double ra_[REC], rb_[REC];
        if (np > p_lo) {
            load_rec(p_lo, ra_);
            int i = p_lo;
            for (; i + 1 < np; i += 2) {
                load_rec(i + 1, rb_);
                poisson(ra_);
                load_rec(i + 2 < np ? i + 2 : i + 1, ra_);
                poisson(rb_);
            }
            if (i < np) poisson(ra_);
        }
        if constexpr (KIND == SP_RES) {  // Gaussian channels: the residual mean is eta itself
            for (int i = g_lo; i < ntot; ++i) {
                load_rec(i, ra_);
#pragma unroll
                for (int q = 0; q < RPL; ++q) {
                    double eta = HASXB ? xbrow[q][(int)__double_as_longlong(ra_[2 * LT + 2])] : ra_[2 * LT];
#pragma unroll
                    for (int l = 0; l < LT; ++l) eta = fma(mr[q][l], ra_[l], eta);
                    const double mval = eta * ra_[2 * LT + 1];
#pragma unroll
                    for (int l = 0; l < LT; ++l) acc[q][l] = fma(mval, ra_[l], acc[q][l]);
                }
            }
        }
    }
    if constexpr (CS > 1) {
        __shared__ double part_acc[4][RPL][LT][64];
        if (part > 0) {
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int l = 0; l < LT; ++l) part_acc[wid][q][l][lane] = acc[q][l];
        }
        __syncthreads();
        if (part > 0) return;
#pragma unroll
        for (int p = 1; p < CS; ++p)
#pragma unroll
            for (int q = 0; q < RPL; ++q)
#pragma unroll
                for (int l = 0; l < LT; ++l) acc[q][l] += part_acc[wid + p][q][l][lane];
    }
#pragma unroll
    for (int q = 0; q < RPL; ++q) {
        if (in[q]) {
            const int64_t row = rr[q];
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                if (l < L) {
                    if constexpr (KIND == SP_YA) A.ya[(int64_t)l * A.ld + row] = acc[q][l];
                    else if constexpr (KIND == SP_RES) {
                        const double rav = yav[q][l] - acc[q][l];
                        if (A.ra) A.ra[(int64_t)l * A.ld + row] = rav;
                        if (A.sv) A.sv[(int64_t)l * A.ld + row] = fma(wv[q][l], mr[q][l], rav);
                    } else {
                        A.w[(int64_t)l * A.ld + row] = fma(2.0, acc[q][l], A.wconst[l]);  // (the records hold a^2 / 2)
                        if ((A.dmask >> l) & 1u) A.mu[(int64_t)l * A.ld + row] = mr[q][l];
                    }
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// per-wave view of one (unit, latent) task
struct Task {
    int m, l, T, r, rs;
    int64_t r0;
    const double* Gl;  // global: (T, r) compact prior factor of this latent
    double* Xl;    // LDS: packed X
    double* vec;   // LDS: 128 doubles
    double* u;     // LDS: 64 doubles
    double* tile;  // LDS: max(256, gcap) doubles (MFMA staging tile / staged G), then two 64-entry columns
    double* Gs;    // LDS: the staged G (T, rs): `tile`, or the workgroup's shared copy
    int gcap;      // doubles reserved for the staged G (>= 256)
};

__device__ __forceinline__ bool task_setup(const SplitArgs& A, Task& K, double* lds_wave, int lane, int bid) {
    const int wid = threadIdx.x >> 6;
    if (A.shg) {
        K.m = 4 * (bid / A.n_lat) + wid;
        if (K.m >= A.M) return false;
        const int li = bid % A.n_lat;
        K.l = A.lat[li];
        K.r0 = A.off[K.m];
        K.T = A.shg_T;
        K.r = A.shg_rk[li];
        K.rs = (K.r + 1) & ~1;
        K.Gl = A.shg_gl[li];
        K.Xl = lds_wave;
        K.tile = K.Xl + A.pkl;
        K.Gs = K.tile;
        K.gcap = A.lds_g;
        K.vec = K.tile + A.lds_g + 128;
        K.u = K.vec + 128;
        return true;
    } else {
        const int task = bid * (blockDim.x >> 6) + wid;
        if (task >= A.M * A.n_lat) return false;
        K.m = task / A.n_lat;
        K.l = A.lat[task - K.m * A.n_lat];
    }
    K.r0 = A.off[K.m];
    K.T = (int)(A.off[K.m + 1] - K.r0);
    const int pidx = A.unit_prior[K.m];
    K.r = __builtin_amdgcn_readfirstlane(A.prior_rl[pidx * A.L + K.l]);
    K.rs = (K.r + 1) & ~1;
    K.Gl = A.prior_base[pidx] + A.prior_goff[pidx * A.L + K.l];
    K.Xl = lds_wave;
    K.tile = K.Xl + A.pkl;
    K.Gs = K.tile;
    K.gcap = A.lds_g;
    K.vec = K.tile + A.lds_g + 128;  // mean kernel only
    K.u = K.vec + 128;
    return true;
}

// row t of the compact factor (r entries, zero beyond) into registers; rows are only 8-byte aligned
template <int RA>
__device__ __forceinline__ void load_g_row(double (&gt)[RA], const double* Gl, int t, int r) {
    const double* Gt = Gl + (int64_t)t * r;
#pragma unroll
    for (int i = 0; i < RA; ++i) gt[i] = i < r ? Gt[i] : 0.0;
}

// the same row from the staged copy (T, rs), rs even, columns r .. rs - 1 zero
template <int RA>
__device__ __forceinline__ void load_g_row_lds(double (&gt)[RA], const double* Gs, int t, int rs) {
    const double* Gt = Gs + t * rs;
#pragma unroll
    for (int i = 0; i < RA; i += 2) {
        double2 g2 = double2{0.0, 0.0};
        if (i < rs) g2 = *reinterpret_cast<const double2*>(Gt + i);
        gt[i] = g2.x;
        gt[i + 1] = g2.y;
    }
}

// Rank <= 16: Cholesky factor AND inverse in one elimination of the augmented matrix [H; I] (lanes 0..15 hold the
// rows of H, lanes 16..31 the unit rows): the unit rows come out as the rows of X' = L^-T, i.e. lane 16 + c holds
// X[i][c] in a[i].  Column k takes the updates of columns m <= k - 2 one step ahead, with multipliers L[k][m]
// broadcast from an LDS copy of the finished columns (off the pivot chain); the last update and the pivot are
// v_readlane broadcasts.  Steps k >= r (identity padding) are skipped.  Ld: 16 x 16 doubles of LDS.
__device__ __forceinline__ bool wave_chol_aug16(double (&a)[16], int lane, int r, double* Ld) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < r) {
            if (k > 0) {
                const double lv = tri_readlane(a[k - 1], k);
                a[k] = fma(-a[k - 1], lv, a[k]);
            }
            const double d = tri_readlane(a[k], k);
            if (!(d > 0.0) || !(d < 1e300)) ok = false;
            double y = __builtin_amdgcn_rsq(d);
            if (k + 1 < 16 && k >= 1) {  // column k + 1 <- columns 0 .. k - 1
                const double* row = Ld + (k + 1) * 16;
                double a0 = a[k + 1], a1 = 0.0;
#pragma unroll
                for (int m = 0; m + 1 < k; m += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(row + m);
                    a0 = fma(-a[m], v.x, a0);
                    a1 = fma(-a[m + 1], v.y, a1);
                }
                if (k & 1) a0 = fma(-a[k - 1], row[k - 1], a0);
                a[k + 1] = a0 + a1;
                asm volatile("" : "+v"(a[k + 1]));
            }
            double e = fma(-d * y, y, 1.0);
            y = fma(y * 0.5, e, y);
            e = fma(-d * y, y, 1.0);
            y = fma(y * 0.5, e, y);
            a[k] *= y;
            asm volatile("" : "+v"(a[k]));
            if (k + 1 < 16) {
                if (lane < 16) Ld[lane * 16 + k] = a[k];
                tri_wave_order();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    return ok;
}

// Ranks 17 .. 32: the same augmented elimination over the whole wave -- lanes 0..31 hold the rows of H, lanes 32..63
// the unit rows, RA register columns per lane; lane 32 + c ends up with column c of X = L^-1 (X[i][c] in a[i]).
// Ld: RA rows x RA doubles of LDS.  (The LDS row kernels of wave_tri.h that this replaces, factor then a separate
// triangular inverse with every pivot row broadcast from LDS, took 3.5x as long per task.)
template <int RA>
__device__ __forceinline__ bool wave_chol_aug32(double (&a)[RA], int lane, int r, double* Ld) {
#pragma unroll
    for (int k = 0; k < RA; ++k) {
        if (k < r) {
            if (k > 0) {
                const double lv = tri_readlane(a[k - 1], k);
                a[k] = fma(-a[k - 1], lv, a[k]);
            }
            const double d = tri_readlane(a[k], k);
            double y = __builtin_amdgcn_rsq(d);
            if (k + 1 < RA && k >= 1) {  // column k + 1 <- columns 0 .. k - 1
                const double* row = Ld + (k + 1) * RA;
                double a0 = a[k + 1], a1 = 0.0;
#pragma unroll
                for (int m = 0; m + 1 < k; m += 2) {
                    const double2 v = *reinterpret_cast<const double2*>(row + m);
                    a0 = fma(-a[m], v.x, a0);
                    a1 = fma(-a[m + 1], v.y, a1);
                }
                if (k & 1) a0 = fma(-a[k - 1], row[k - 1], a0);
                a[k + 1] = a0 + a1;
                asm volatile("" : "+v"(a[k + 1]));
            }
            {   // one third-order step: y (1 + e / 2 + 3 e^2 / 8), e = 1 - d y^2 (v_rsq_f64 starts at ~2^-26)
                const double e = fma(-d * y, y, 1.0);
                y = fma(y * e, fma(0.375, e, 0.5), y);
            }
            a[k] *= y;
            asm volatile("" : "+v"(a[k]));
            if (k + 1 < RA) {
                if (lane < RA) Ld[lane * RA + k] = a[k];  // (rows RA .. 31 are never pivot rows)
                tri_wave_order();
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // a pivot that is not positive and finite leaves NaN / inf on the diagonal of X (lane 32 + c, a[c])
    bool fin = true;
#pragma unroll
    for (int c = 0; c < RA; ++c)
        if (lane == 32 + c && c < r) fin = a[c] > 0.0 && a[c] < 1e300;
    return __builtin_amdgcn_ballot_w64(!fin) == 0;
}

// factor I + G'WG, invert, optionally refresh v (estep_fast.hip factor_phase, one latent)
template <int RP, int RA, bool STAGE>
__device__ __forceinline__ void factor_task(const SplitArgs& A, const Task& K, int lane) {
    const int L = A.L, l = K.l, T = K.T, r = K.r, rs = K.rs;
    const double* __restrict__ Gl = K.Gl;
    double* Xl = K.Xl;
    const double* w_s = A.w + (int64_t)l * A.ld + K.r0;
    double* v_s = A.v + (int64_t)l * A.ld + K.r0;
    const int j = lane & (RP - 1);
    bool ok;
    const int col = lane & 15, kq = lane >> 4;
    if constexpr (RP <= 16) {
        double4_t c = {0.0, 0.0, 0.0, 0.0};
        if constexpr (STAGE) {
            // one round trip to global memory: every lane fetches its row of G and its curvature, then the build
            // reads them back from LDS in the (column, time-chunk) layout of the matrix instruction
            const double wt = lane < T ? w_s[lane] : 0.0;
            double* Gs = K.Gs;  // (T, rs); per wave, the staging tile of the result reuses it afterwards
            double* wcol = K.tile + K.gcap;
            if (!A.shg) {
                double gt0[16];
                load_g_row<16>(gt0, Gl, lane < T ? lane : 0, r);
                if (lane < T) {
#pragma unroll
                    for (int i = 0; i < 16; i += 2)
                        if (i < rs) *reinterpret_cast<double2*>(Gs + lane * rs + i) = double2{gt0[i], gt0[i + 1]};
                }
            }
            if (lane < T) wcol[lane] = wt;
            tri_wave_sync();
            const bool cin = col < rs;
            for (int t0 = 0; t0 < T; t0 += 4) {
                const int t = t0 + kq;
                double g = 0.0, wg = 0.0;
                if (cin && t < T) {
                    g = Gs[t * rs + col];
                    wg = wcol[t] * g;
                }
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(wg, g, c, 0, 0, 0);
            }
        } else {
            const bool cin = col < r;
            for (int t0 = 0; t0 < T; t0 += 4) {
                const int t = t0 + kq;
                double g = 0.0, wg = 0.0;
                if (cin && t < T) {
                    g = Gl[t * r + col];
                    wg = w_s[t] * g;
                }
                c = __builtin_amdgcn_mfma_f64_16x16x4f64(wg, g, c, 0, 0, 0);
            }
        }
        tri_wave_order();
        double* ht = K.tile;
#pragma unroll
        for (int q = 0; q < 4; ++q) ht[(kq + 4 * q) * 16 + col] = c[q];
        tri_wave_sync();
        double a[16];  // lanes 0..15: rows of I + G'WG; lanes 16..31: unit rows
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
            const double2 h2 = *reinterpret_cast<const double2*>(ht + j * 16 + i);
            a[i] = (lane < 16 ? h2.x : 0.0) + (i == j ? 1.0 : 0.0);
            a[i + 1] = (lane < 16 ? h2.y : 0.0) + (i + 1 == j ? 1.0 : 0.0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(a[i]));
        tri_wave_order();
        ok = wave_chol_aug16(a, lane, r, K.tile);
        tri_wave_order();
        if (lane >= 16 && lane < 32) {  // X row-major packed in LDS for the solves: X[i][c], i >= c = lane - 16
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if (i >= lane - 16) Xl[tri_row_off(i) + lane - 16] = i < r ? a[i] : (i == lane - 16 ? 1.0 : 0.0);
        }
    } else {
        // H = G'WG on the matrix pipe: the lower block triangle of the 32 x 32 matrix, staged (with the mirror image of
        // the off-diagonal tile) as RA rows of stride RA + 2 in LDS
        constexpr int LDH = RA + 2;
        double* ht = K.tile;
        {
            // operands in two batches of eight time chunks (24 independent loads in flight per lane, then 24 matrix
            // instructions on three independent accumulators): with a load in front of every instruction the build was a
            // chain of 39 trips to L2; with all 48 operands loaded first the kernel needed 148 registers (three waves
            // per SIMD: the 4000 tasks of one latent at C3 ran as two generations)
            double4_t c0 = {0.0, 0.0, 0.0, 0.0}, c1 = c0, c2 = c0;  // tiles (0, 0), (1, 0), (1, 1)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                double g0[8], g1[8], wv[8];  // G[t][col], G[t][16 + col], w[t] at t = 4 k + kq
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    const int t = 4 * (8 * half + kk) + kq;
                    const bool in = t < T;
                    const int tc = in ? t : 0;
                    const double wt = w_s[tc];
                    const double ga = Gl[tc * r + (col < r ? col : 0)];
                    const double gb = Gl[tc * r + (16 + col < r ? 16 + col : 0)];
                    wv[kk] = in ? wt : 0.0;
                    g0[kk] = (in && col < r) ? ga : 0.0;
                    g1[kk] = (in && 16 + col < r) ? gb : 0.0;
                }
#pragma unroll
                for (int kk = 0; kk < 8; ++kk) {
                    if (4 * (8 * half + kk) < T) {
                        const double w0 = wv[kk] * g0[kk], w1 = wv[kk] * g1[kk];
                        c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(w0, g0[kk], c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1, g0[kk], c1, 0, 0, 0);
                        c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(w1, g1[kk], c2, 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int tile = 0; tile < 3; ++tile) {
                const int bi = tile == 0 ? 0 : 1, bj = tile == 2 ? 1 : 0;
                const int cb = 16 * bj + col;
                const double4_t c = tile == 0 ? c0 : (tile == 1 ? c1 : c2);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 16 * bi + kq + 4 * q;
                    if (row < RA && cb < RA) {
                        ht[row * LDH + cb] = c[q];
                        if (tile == 1) ht[cb * LDH + row] = c[q];
                    }
                }
            }
        }
        tri_wave_sync();
        double a[RA];  // lanes 0..31: rows of I + G'WG; lanes 32..63: unit rows
        {
            const int jr = lane & 31;
            const double* hrow = ht + (jr < RA ? jr : 0) * LDH;
#pragma unroll
            for (int i = 0; i < RA; i += 2) {
                const double2 h2 = *reinterpret_cast<const double2*>(hrow + i);
                a[i] = (lane < 32 ? h2.x : 0.0) + (i == jr ? 1.0 : 0.0);
                a[i + 1] = (lane < 32 ? h2.y : 0.0) + (i + 1 == jr ? 1.0 : 0.0);
            }
        }
#pragma unroll
        for (int i = 0; i < RA; ++i) asm volatile("" : "+v"(a[i]));
        tri_wave_order();
        ok = wave_chol_aug32<RA>(a, lane, r, K.tile);
        tri_wave_order();
        if (lane >= 32 && lane < 32 + RA) {  // X row-major packed in LDS: X[i][c], i >= c = lane - 32
#pragma unroll
            for (int i = 0; i < RA; ++i)
                if (i >= lane - 32) Xl[tri_row_off(i) + lane - 32] = i < r ? a[i] : (i == lane - 32 ? 1.0 : 0.0);
        }
    }
    tri_wave_sync();
    __builtin_amdgcn_sched_barrier(0);
    if (A.do_v && ok && lane < T) {
        double gt[RA];
        if (STAGE && A.shg) load_g_row_lds<RA>(gt, K.Gs, lane, rs);
        else load_g_row<RA>(gt, Gl, lane, r);  // (rank <= 16: the same loads as at the top, the compiler keeps the registers)
        double vv = 0.0;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            if (i < r) {
                const double* Xi = Xl + tri_row_off(i);
                double z0 = 0.0, z1 = 0.0;
#pragma unroll
                for (int q = 0; q + 1 <= i; q += 2) {
                    const double2 x2 = *reinterpret_cast<const double2*>(Xi + q);
                    z0 = fma(x2.x, gt[q], z0);
                    z1 = fma(x2.y, gt[q + 1], z1);
                }
                if (!(i & 1)) z0 = fma(Xi[i], gt[i], z0);
                const double z = z0 + z1;
                vv = fma(z, z, vv);
            }
        }
        v_s[lane] = vv;
    }
    // X -> global for the mean update of the next sweep
    {
        double* xd = A.xg + (int64_t)(K.m * L + l) * A.pkg;
        constexpr int PK = tri_packed_size(RA);
        for (int i = lane; i < PK; i += 64) xd[i] = Xl[i];
    }
    if (lane == 0) {
        A.failg[K.m * L + l] = ok ? 0 : 1;
        if (!ok) atomicAdd(A.fail, 1);
    }
}

// Newton step on the posterior mean of one latent (vlgp/core.py:80-95).  With K = GG', H = G'WG and the residual
// projection ra, the step delta = (I + KW)^-1 (K ra - mu) is evaluated as
//     delta = G (I + H)^-1 G' (ra + W mu) - mu
// (push-through identity; the form of estep_fast.hip, u = G G'ra - mu, delta = u - G (I + H)^-1 G'W u, is the same
// vector with two reductions over the time axis and two expansions instead of one each): one reduction over the
// rows, two triangular products with X = chol(I + H)^-1, one expansion.
template <int RP, int RA, bool STAGE>
__device__ __forceinline__ void mean_task(const SplitArgs& A, const Task& K, int lane) {
    constexpr int NCH = 64 / RP;
    const int L = A.L, l = K.l, T = K.T, r = K.r, rs = K.rs;
    const double* Gl = K.Gl;
    double* Xl = K.Xl;
    const double* w_s = A.w + (int64_t)l * A.ld + K.r0;
    const double* ra_s = A.ra + (int64_t)l * A.ld + K.r0;
    double* mu_s = A.mu + (int64_t)l * A.ld + K.r0;
    double* vec = K.vec;
    double* vec2 = vec + 64;
    double* scol = K.u;  // s = ra + w mu, one entry per row
    double gt[RA];
    double mu_t = 0.0;
    double* Gs = K.Gs;
    {
        const double* xs = A.xg + (int64_t)(K.m * L + l) * A.pkg;
        constexpr int PK = tri_packed_size(RA);
        for (int i = lane; i < PK; i += 64) Xl[i] = xs[i];
    }
    const bool shg = STAGE && A.shg;
    if (shg) load_g_row_lds<RA>(gt, Gs, lane < T ? lane : 0, rs);
    else load_g_row<RA>(gt, Gl, lane < T ? lane : 0, r);
    if (lane < T) {
        mu_t = mu_s[lane];
        scol[lane] = fma(w_s[lane], mu_t, ra_s[lane]);
        if constexpr (STAGE) {
            if (!shg) {
#pragma unroll
                for (int i = 0; i < RA; i += 2)
                    if (i < rs) *reinterpret_cast<double2*>(Gs + lane * rs + i) = double2{gt[i], gt[i + 1]};
            }
        }
    }
    tri_wave_sync();
    const int j = lane & (RP - 1), ch = lane / RP;
    // c = G' s
    double acc = 0.0;
    if (j < r) {
        if constexpr (STAGE) {
#pragma unroll 4
            for (int t = ch; t < T; t += NCH) acc = fma(Gs[t * rs + j], scol[t], acc);
        } else {
#pragma unroll 4
            for (int t = ch; t < T; t += NCH) acc = fma(Gl[t * r + j], scol[t], acc);
        }
    }
#pragma unroll
    for (int o = RP; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (lane < RA) vec2[lane] = acc;
    tri_wave_sync();
    // z = X c, sol = X' z   (lane = row, then lane = column)
    double z = 0.0;
    if (lane < RA) {
        const double* Xi = Xl + tri_row_off(lane);
        double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
#pragma unroll
        for (int q = 0; q < RA; q += 4) {
            if (q <= lane) z0 = fma(Xi[q], vec2[q], z0);
            if (q + 1 <= lane) z1 = fma(Xi[q + 1], vec2[q + 1], z1);
            if (q + 2 <= lane) z2 = fma(Xi[q + 2], vec2[q + 2], z2);
            if (q + 3 <= lane) z3 = fma(Xi[q + 3], vec2[q + 3], z3);
        }
        z = (z0 + z1) + (z2 + z3);
    }
    if (lane < RA) vec[lane] = z;
    tri_wave_sync();
    double sol = 0.0;
    if (lane < RA) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int q = 0; q < RA; q += 4) {
            if (q >= lane) s0 = fma(Xl[tri_row_off(q) + lane], vec[q], s0);
            if (q + 1 >= lane) s1 = fma(Xl[tri_row_off(q + 1) + lane], vec[q + 1], s1);
            if (q + 2 >= lane) s2 = fma(Xl[tri_row_off(q + 2) + lane], vec[q + 2], s2);
            if (q + 3 >= lane) s3 = fma(Xl[tri_row_off(q + 3) + lane], vec[q + 3], s3);
        }
        sol = (s0 + s1) + (s2 + s3);
    }
    if (lane < RA) vec2[lane] = sol;  // (c was consumed before the previous barrier)
    tri_wave_sync();
    if (lane < T) {
        double s0 = -mu_t, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < RA; i += 2)
            if (i < rs) {
                const double2 c2 = *reinterpret_cast<const double2*>(vec2 + i);
                s0 = fma(gt[i], c2.x, s0);
                s1 = fma(gt[i + 1], c2.y, s1);
            }
        double s = s0 + s1;
        s = fmin(fmax(s, -A.dmu_bound), A.dmu_bound);
        mu_s[lane] = mu_t + s;  // (the last sweep's `dmu` comes from mean_task_last)
    }
}

// The same step for rank <= 16 without a branch: lane (i = lane & 15, g = lane >> 4) keeps X[i][4g .. 4g + 3] and
// X[4g .. 4g + 3][i] in registers (straight from the packed global copy, zero outside the triangle), so that both
// triangular products are four FMAs per lane and two cross-group adds; the time loop of c = G's is fully unrolled with
// clamped addresses (s is zero beyond the unit).  The generic mean_task compiles its row <= lane predicates into ~150
// branches and as many exec-mask updates per wave -- the launch was bound by instruction issue, not by arithmetic.
// CHECK: read the failed-factor flag of the task here (else the caller has tested it).
template <bool STAGE, bool CHECK>
__device__ __forceinline__ void mean_task16(const SplitArgs& A, const Task& K, int lane) {
    const int L = A.L, l = K.l, T = K.T, r = K.r, rs = K.rs;
    const int failed = CHECK ? A.failg[K.m * L + l] : 0;
    const double* __restrict__ Gl = K.Gl;
    const double* w_s = A.w + (int64_t)l * A.ld + K.r0;
    const double* ra_s = A.ra + (int64_t)l * A.ld + K.r0;
    double* mu_s = A.mu + (int64_t)l * A.ld + K.r0;
    double* vec = K.vec;
    double* vec2 = vec + 64;
    double* scol = K.u;
    double* Gs = K.Gs;
    const int i = lane & 15, g = lane >> 4;
    const double* __restrict__ xs = A.xg + (int64_t)(K.m * L + l) * A.pkg;
    double xr[4], xc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = 4 * g + k;
        const double vr_ = xs[tri_row_off(i) + (q <= i ? q : i)];  // X[i][q]
        const double vc_ = xs[tri_row_off(q >= i ? q : i) + i];    // X[q][i]
        xr[k] = q <= i ? vr_ : 0.0;
        xc[k] = q >= i ? vc_ : 0.0;
    }
    const bool shg = STAGE && A.shg;
    const int tt = lane < T ? lane : 0;
    double mu_t = 0.0, st = 0.0;
    if (lane < T) {
        mu_t = mu_s[lane];
        st = fma(w_s[lane], mu_t, ra_s[lane]);
    }
    scol[lane] = st;
    if constexpr (STAGE) {
        if (!shg) {  // (the row of G is read again from the staged copy at the end: not held across the chain)
            double g0[16];
            load_g_row<16>(g0, Gl, tt, r);
            if (lane < T) {
#pragma unroll
                for (int q = 0; q < 16; q += 2)
                    if (q < rs) *reinterpret_cast<double2*>(Gs + lane * rs + q) = double2{g0[q], g0[q + 1]};
            }
        }
    }
    tri_wave_sync();
    // c = G' s: lane (column i, time chunk g) takes t = g, g + 4, ...
    double c;
    {
        const int ncol = STAGE ? rs : r;
        const int jj = i < ncol ? i : 0;
        double a0 = 0.0, a1 = 0.0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int t = g + 4 * k;
            const int tc = t < T ? t : T - 1;
            const double gv = STAGE ? Gs[tc * rs + jj] : Gl[tc * r + jj];
            if (k & 1) a1 = fma(gv, scol[t], a1);
            else a0 = fma(gv, scol[t], a0);
        }
        c = i < ncol ? a0 + a1 : 0.0;
    }
    c += __shfl_xor(c, 16, 64);
    c += __shfl_xor(c, 32, 64);
    if (lane < 16) vec2[lane] = c;
    tri_wave_sync();
    // z = X c, sol = X' z
    double z;
    {
        const double2 c01 = *reinterpret_cast<const double2*>(vec2 + 4 * g);
        const double2 c23 = *reinterpret_cast<const double2*>(vec2 + 4 * g + 2);
        z = fma(xr[0], c01.x, fma(xr[1], c01.y, fma(xr[2], c23.x, xr[3] * c23.y)));
    }
    z += __shfl_xor(z, 16, 64);
    z += __shfl_xor(z, 32, 64);
    if (lane < 16) vec[lane] = z;
    tri_wave_sync();
    double sol;
    {
        const double2 z01 = *reinterpret_cast<const double2*>(vec + 4 * g);
        const double2 z23 = *reinterpret_cast<const double2*>(vec + 4 * g + 2);
        sol = fma(xc[0], z01.x, fma(xc[1], z01.y, fma(xc[2], z23.x, xc[3] * z23.y)));
    }
    sol += __shfl_xor(sol, 16, 64);
    sol += __shfl_xor(sol, 32, 64);
    if (lane < 16) vec2[lane] = sol;  // (c was consumed before the previous barrier)
    double gt[16];
    if constexpr (STAGE) load_g_row_lds<16>(gt, Gs, tt, rs);
    else load_g_row<16>(gt, Gl, tt, r);
    tri_wave_sync();
    if (lane < T) {
        double s0 = -mu_t, s1 = 0.0;
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            const double2 c2 = *reinterpret_cast<const double2*>(vec2 + q);
            s0 = fma(gt[q], c2.x, s0);
            s1 = fma(gt[q + 1], c2.y, s1);
        }
        double s = s0 + s1;
        s = fmin(fmax(s, -A.dmu_bound), A.dmu_bound);
        if (!failed) mu_s[lane] = mu_t + s;  // (the last sweep's `dmu` comes from mean_task_last)
    }
    if (CHECK && failed && lane == 0) atomicAdd(A.fail, 1);
}

// Newton step on the posterior mean in the form of estep_fast.hip (mean_phase): u = G G'ra - mu, delta = u - G (I + H)^-1 G'W u.
// u vanishes at the fixed point, so delta keeps its RELATIVE accuracy when the sweeps have converged: in the last sweep
// of a call this form supplies the step handed back as `dmu` (core.py:96), while mu itself is advanced by mean_task as
// in every other sweep (so that 10 + 15 sweeps still equal 25 bit for bit).  Writes dmu only.
template <int RP, int RA, bool STAGE>
__device__ __forceinline__ void mean_task_last(const SplitArgs& A, const Task& K, int lane) {
    constexpr int NCH = 64 / RP;
    const int L = A.L, l = K.l, T = K.T, r = K.r, rs = K.rs;
    const double* Gl = K.Gl;
    double* Xl = K.Xl;
    const double* w_s = A.w + (int64_t)l * A.ld + K.r0;
    const double* ra_s = A.ra + (int64_t)l * A.ld + K.r0;
    double* mu_s = A.mu + (int64_t)l * A.ld + K.r0;
    double* vec = K.vec;
    double* vec2 = vec + 64;
    double* u = K.u;
    // STAGE (rank <= 16 in an all-rank-<=-16 launch): G, the residual and the curvature of this latent staged in LDS
    double gt[RA];
    double mu_t = 0.0;
    double* Gs = K.Gs;
    double* racol = K.tile + K.gcap;
    double* wcol = racol + 64;
    {
        const double* xs = A.xg + (int64_t)(K.m * L + l) * A.pkg;
        constexpr int PK = tri_packed_size(RA);
        for (int i = lane; i < PK; i += 64) Xl[i] = xs[i];
    }
    if constexpr (STAGE) {
        if (A.shg) load_g_row_lds<RA>(gt, Gs, lane < T ? lane : 0, rs);
        else load_g_row<RA>(gt, Gl, lane < T ? lane : 0, r);
        if (lane < T) {
            mu_t = mu_s[lane];
            racol[lane] = ra_s[lane];
            wcol[lane] = w_s[lane];
            if (!A.shg) {
#pragma unroll
                for (int i = 0; i < RA; i += 2)
                    if (i < rs) *reinterpret_cast<double2*>(Gs + lane * rs + i) = double2{gt[i], gt[i + 1]};
            }
        }
    }
    tri_wave_sync();
    const int j = lane & (RP - 1), ch = lane / RP;
    // g1 = G' (res a_l)
    double acc = 0.0;
    if (j < r) {
        if constexpr (STAGE) {
#pragma unroll 4
            for (int t = ch; t < T; t += NCH) acc = fma(Gs[t * rs + j], racol[t], acc);
        } else {
#pragma unroll 4
            for (int t = ch; t < T; t += NCH) acc = fma(Gl[t * r + j], ra_s[t], acc);
        }
    }
#pragma unroll
    for (int o = RP; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (lane < RA) vec[lane] = acc;
    tri_wave_sync();
    // u = G g1 - mu_l   (row t of G stays in registers for the last step)
    double ut = 0.0;
    {
        if constexpr (!STAGE) {
            load_g_row<RA>(gt, Gl, lane < T ? lane : 0, r);
            if (lane < T) mu_t = mu_s[lane];
        }
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < RA; i += 2) {
            if (i < rs) {
                const double2 c2 = *reinterpret_cast<const double2*>(vec + i);
                s0 = fma(gt[i], c2.x, s0);
                s1 = fma(gt[i + 1], c2.y, s1);
            }
        }
        if (lane < T) {
            ut = (s0 + s1) - mu_t;
            u[lane] = ut;
        }
    }
    tri_wave_sync();
    // rhs = (W G)' u
    acc = 0.0;
    if (j < r) {
        if constexpr (STAGE) {
#pragma unroll 4
            for (int t = ch; t < T; t += NCH) acc = fma(wcol[t] * Gs[t * rs + j], u[t], acc);
        } else {
#pragma unroll 4
            for (int t = ch; t < T; t += NCH) acc = fma(w_s[t] * Gl[t * r + j], u[t], acc);
        }
    }
#pragma unroll
    for (int o = RP; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (lane < RA) vec2[lane] = acc;
    tri_wave_sync();
    // z = X rhs, sol = X' z   (lane = row, then lane = column)
    double z = 0.0;
    if (lane < RA) {
        const double* Xi = Xl + tri_row_off(lane);
        double z0 = 0.0, z1 = 0.0, z2 = 0.0, z3 = 0.0;
#pragma unroll
        for (int q = 0; q < RA; q += 4) {
            if (q <= lane) z0 = fma(Xi[q], vec2[q], z0);
            if (q + 1 <= lane) z1 = fma(Xi[q + 1], vec2[q + 1], z1);
            if (q + 2 <= lane) z2 = fma(Xi[q + 2], vec2[q + 2], z2);
            if (q + 3 <= lane) z3 = fma(Xi[q + 3], vec2[q + 3], z3);
        }
        z = (z0 + z1) + (z2 + z3);
    }
    tri_wave_sync();
    if (lane < RA) vec[lane] = z;
    tri_wave_sync();
    double sol = 0.0;
    if (lane < RA) {
        double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
#pragma unroll
        for (int q = 0; q < RA; q += 4) {
            if (q >= lane) s0 = fma(Xl[tri_row_off(q) + lane], vec[q], s0);
            if (q + 1 >= lane) s1 = fma(Xl[tri_row_off(q + 1) + lane], vec[q + 1], s1);
            if (q + 2 >= lane) s2 = fma(Xl[tri_row_off(q + 2) + lane], vec[q + 2], s2);
            if (q + 3 >= lane) s3 = fma(Xl[tri_row_off(q + 3) + lane], vec[q + 3], s3);
        }
        sol = (s0 + s1) + (s2 + s3);
    }
    tri_wave_sync();
    if (lane < RA) vec2[lane] = sol;
    tri_wave_sync();
    if (lane < T) {
        double s0 = ut, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < RA; i += 2)
            if (i < rs) {
                const double2 c2 = *reinterpret_cast<const double2*>(vec2 + i);
                s0 = fma(-gt[i], c2.x, s0);
                s1 = fma(-gt[i + 1], c2.y, s1);
            }
        double s = s0 + s1;
        s = fmin(fmax(s, -A.dmu_bound), A.dmu_bound);
        A.dmu[(K.r0 + lane) * L + l] = s;
    }
    tri_wave_sync();
}

#include "estep_lane.h"

// MAXRA: largest register-array size compiled in (16: every latent of the launch has rank <= 16)
// LASTSW (mean only): the last sweep of the call
// (second launch bound = waves per SIMD: the rank <= 16 launches are bound by the number of resident waves --
// measured 17 + 103 / n us (mean) and 28 + 124 / n us (factor) with n workgroups per CU)
// (`bid`: the block's index among the blocks of this kind -- blockIdx.x, or its position behind the lane-per-task blocks
// of a mixed launch, esplit_mix)
template <int MAXRA, bool MEAN, bool LASTSW>
__device__ __forceinline__ void esplit_latent_body(const SplitArgs& A, double* smem, int bid) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    double* wave_base = smem;
    if constexpr (MAXRA == 16) {
        if (A.shg) {  // one latent per workgroup: its G (T, rs) staged once, zero-padded to the even stride
            const int li = bid % A.n_lat;
            const int r = A.shg_rk[li], rs = (r + 1) & ~1;
            const double* __restrict__ Gl = A.shg_gl[li];
            for (int i = threadIdx.x; i < A.shg_T * rs; i += 256) {
                const int t = i / rs, j = i - t * rs;
                smem[i] = j < r ? Gl[t * r + j] : 0.0;
            }
            __syncthreads();
            wave_base = smem + A.shg_cap;
        }
    }
    double* lds_wave = wave_base + (int64_t)wid * (A.pkl + A.lds_g + 128 + (MEAN ? 192 : 0));
    Task K;
    if (!task_setup(A, K, lds_wave, lane, bid)) return;
    if (MAXRA == 16 && A.shg) K.Gs = smem;
    if constexpr (MEAN) {
        // singular system: zero update (core.py:92-94).  The regular rank <= 16 sweeps read the flag with their other
        // loads and apply it at the store (a test up front is one more dependent trip to memory per wave).
        if ((LASTSW || K.r > 16) && A.failg[K.m * A.L + K.l]) {
            if (lane == 0) atomicAdd(A.fail, 1);
            if (A.last && lane < K.T) A.dmu[(K.r0 + lane) * A.L + K.l] = 0.0;
            return;
        }
    }
    if (K.r <= 16) {
        if constexpr (MEAN && LASTSW) mean_task_last<16, 16, MAXRA == 16>(A, K, lane);
        if constexpr (MEAN) mean_task16<MAXRA == 16, !LASTSW>(A, K, lane);
        else factor_task<16, 16, MAXRA == 16>(A, K, lane);
    } else if constexpr (MAXRA >= 20) {
        if (K.r <= 20) {  // ranks 17 .. 20: a latent whose omega drifts up usually stops here (cubic work: 0.58 of 24)
            if constexpr (MEAN && LASTSW) mean_task_last<32, 20, false>(A, K, lane);
            if constexpr (MEAN) mean_task<32, 20, false>(A, K, lane);
            else factor_task<32, 20, false>(A, K, lane);
        } else if constexpr (MAXRA < 24) {
        } else if (K.r <= 24) {
            if constexpr (MEAN && LASTSW) mean_task_last<32, 24, false>(A, K, lane);
            if constexpr (MEAN) mean_task<32, 24, false>(A, K, lane);
            else factor_task<32, 24, false>(A, K, lane);
        } else if constexpr (MAXRA >= 32) {
            if constexpr (MEAN && LASTSW) mean_task_last<32, 32, false>(A, K, lane);
            if constexpr (MEAN) mean_task<32, 32, false>(A, K, lane);
            else factor_task<32, 32, false>(A, K, lane);
        }
    }
}

#ifndef ESPLIT_LB32
#define ESPLIT_LB32 3  /* (no accumulation registers, three waves per SIMD: class-32 E-step 8.30 -> 7.9 ms, tools/variant_ab.sh; 4 spills 88 bytes and is slower; class 24 at 4 / 5 / 6: no change) */
#endif
#ifndef ESPLIT_LB24
#define ESPLIT_LB24 1
#endif
template <int MAXRA, bool MEAN, bool LASTSW = false>
__global__ void __launch_bounds__(256, (MAXRA == 16 && !LASTSW ? 8 : (!MEAN ? (MAXRA == 32 ? ESPLIT_LB32 : ESPLIT_LB24) : 1))) esplit_latent(SplitArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    esplit_latent_body<MAXRA, MEAN, LASTSW>(A, smem, blockIdx.x);
}

// MIXED launch (round 6): the lane-per-task blocks of the latents of rank <= LANE_RMAX and the wave-per-task blocks of the
// latents above it as ONE grid.  Within a sweep the latents are independent (vlgp/core.py:76-97), but as two launches on
// one stream the wave-per-task launch of a single latent that has drifted to rank 15 .. 20 was a second dependent step
// of every sweep (E-step 2.05 -> 3.1 ms at C3 for one latent at rank 18): here its blocks fill the CUs the few hundred
// lane-per-task workgroups leave idle.  Blocks [0, n_lane) are the lane-per-task ones (they live longest: first).
// Same arithmetic per task as the two launches, hence the same bits.  KIND as esplit_lane.
// RTOP: as esplit_lane; only 13 is instantiated (no scratch frame: the host sends a rank-14 latent to the wave-per-task
// blocks of a mixed launch).  The last-sweep launch (KIND 2, once per call) runs at one workgroup per CU for the same reason.
template <int KIND, int MAXRA, int RTOP>
__global__ void __launch_bounds__(256, (KIND == 2 ? 1 : 2)) esplit_mix(SplitArgs Aln, SplitArgs Alt, int n_lane) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    if ((int)blockIdx.x < n_lane) esplit_lane_body<KIND, RTOP>(Aln, smem, blockIdx.x);
    else esplit_latent_body<MAXRA, KIND != 0, KIND == 2>(Alt, smem, (int)blockIdx.x - n_lane);
}

// =========================================================================================================
// LONG units (T > 64: core.infer / api.transform on full-length trials, update_w / update_v before them) on the same
// launch sequence: the row passes above work for any unit length; the per-latent phases get one WORKGROUP per
// (unit, latent) task -- 1000 tasks of 4 waves at C3 instead of the 200 eight-wave workgroups of the persistent
// long-unit kernel (estep_long.hip: 1.6 waves per SIMD, latency-bound on L2 loads: 12.6 ms for ten sweeps).  Factor size
// = the reference's fixed rank 50 (preprocess.py:80) with identity padding above the effective rank, as there.
constexpr int LRP = 50;
constexpr int LPK = tri_packed_size(LRP);  // packed lower-triangular 50 x 50, rows padded to even
constexpr int LRED = 10 * 256;             // the ten 16 x 16 tiles of one half of the time axis

__device__ __forceinline__ void ltile_of(int tile, int& bi, int& bj) {  // lower block triangle of 4 x 4, row-major
    bi = tile < 1 ? 0 : (tile < 3 ? 1 : (tile < 6 ? 2 : 3));
    bj = tile - (bi * (bi + 1)) / 2;
}

// elong_factor, task (m, l): H = I + G'WG on the matrix pipe, Cholesky + triangular inverse by one wave (wave_tri.h,
// rows / columns in registers), X to global for the mean launch, v_t = |X g_t|^2 as Z = X G' in 16-bin column blocks.
// Rank 50 = 3 x 16 + 2: the matrix pipe sees the 48 x 48 part only (six 16 x 16 tiles of H: the time axis in two halves,
// three tiles on each of the two waves of a half; six tiles of X in the variance), the rows 48, 49 of H and of Z = X G'
// are dot products on the vector pipe -- padded to 64 they were four tiles of ten in both phases, 87 % zeros (measured:
// 320 -> 300 us per launch at C3: 40 % fewer matrix instructions buy 6 %; sharing the staged tiles of G between four
// tasks of a workgroup -- a quarter of the L2 traffic -- was slower: what bounds the launch is the latency of a wave's
// load -> multiply chain at two waves per SIMD, the register budget of the wave that factors).  (Measured alternatives at C3, 1000 tasks: the three phases as three launches with
// the matrix through global memory -- lighter waves for the build and the variance -- 180 + 34 + 155 us against 320 us
// for one kernel; the build's loads software-pipelined one step ahead of its matrix instructions: no change.)
#ifndef ELONG_LB
#define ELONG_LB 4
#endif
__global__ void __launch_bounds__(256, ELONG_LB) elong_factor(SplitArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Xp = smem;            // LPK
    double* red = smem + LPK;     // LRED: the second half's tiles (6 x 256), then its tail rows (2 x 2 x 64)
    __shared__ int s_ok;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = A.L;
    const int m = blockIdx.x / L, l = blockIdx.x - m * L;
    const int64_t r0 = A.off[m];
    const int T = (int)(A.off[m + 1] - r0);
    const int pidx = A.unit_prior[m];
    const int r = __builtin_amdgcn_readfirstlane(A.prior_rl[pidx * L + l]);
    const double* __restrict__ Gl = A.prior_base[pidx] + A.prior_goff[pidx * L + l];
    const double* __restrict__ w_s = A.w + (int64_t)l * A.ld + r0;
    double* v_s = A.v + (int64_t)l * A.ld + r0;
    const int col = lane & 15, kq = lane >> 4;
    const int rb = r < 48 ? r : 48;            // columns the matrix pipe handles
    const int nbk = (rb + 15) >> 4;            // 16-column blocks among them (wave-uniform)
    const int ne = r - rb;                     // tail rows 48 .. r - 1 (0, 1 or 2)

    for (int i = tid; i < LPK; i += 256) Xp[i] = 0.0;
    __syncthreads();
    if (tid < LRP) Xp[tri_row_off(tid) + tid] = 1.0;  // identity: rows above the effective rank are never visited
    // ---- F1 ----
    {
        const int half = wid >> 1, which = wid & 1;
        const int Th = ((T / 2) + 15) & ~15;
        const int ta = half ? Th : 0, tb = half ? T : (Th < T ? Th : T);
        // tiles of this wave: which == 0: (2,0) (2,1) (2,2); which == 1: (1,0) (1,1) (0,0)
        double4_t c[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) c[i] = double4_t{0.0, 0.0, 0.0, 0.0};
        if (which == 1 || nbk > 2) {
            for (int t0 = ta; t0 < tb; t0 += 16) {  // four k-steps of loads in flight before the matrix instructions
                double g[4][3], wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + 4 * u + kq;
                    const bool in = t < tb;
                    const int tc = in ? t : tb - 1;
                    const double* row = Gl + (int64_t)tc * r;
                    wv[u] = in ? w_s[tc] : 0.0;
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const int cb = 16 * b + col;
                        double gv = 0.0;
                        if (b < nbk) gv = row[cb < rb ? cb : 0];
                        g[u][b] = (in && cb < rb) ? gv : 0.0;
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (which == 0) {
                        const double a2 = wv[u] * g[u][2];
                        c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, g[u][0], c[0], 0, 0, 0);
                        c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, g[u][1], c[1], 0, 0, 0);
                        c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, g[u][2], c[2], 0, 0, 0);
                    } else {
                        const double a0 = wv[u] * g[u][0];
                        c[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, g[u][0], c[2], 0, 0, 0);
                        if (nbk > 1) {
                            const double a1 = wv[u] * g[u][1];
                            c[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, g[u][0], c[0], 0, 0, 0);
                            c[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, g[u][1], c[1], 0, 0, 0);
                        }
                    }
                }
            }
        }
        // tail rows 48, 49 of H on the vector pipe: lane <-> column j, the wave's quarter of the time axis
        double tl0 = 0.0, tl1 = 0.0;
        if (ne > 0) {
            const int tm = (ta + tb) >> 1;
            const int qa = which ? tm : ta, qb = which ? tb : tm;
            const int jc = lane < r ? lane : 0;
            int t = qa;
            for (; t + 4 <= qb; t += 4) {
                double gr[4], wq[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    gr[u] = Gl[(int64_t)(t + u) * r + jc];
                    wq[u] = w_s[t + u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const double wg = wq[u] * gr[u];
                    tl0 = fma(tri_readlane(gr[u], 48), wg, tl0);
                    tl1 = fma(tri_readlane(gr[u], 49), wg, tl1);  // (lane 49 holds column 0 when r == 49: unused then)
                }
            }
            for (; t < qb; ++t) {
                const double gr = Gl[(int64_t)t * r + jc];
                const double wg = w_s[t] * gr;
                tl0 = fma(tri_readlane(gr, 48), wg, tl0);
                tl1 = fma(tri_readlane(gr, 49), wg, tl1);
            }
        }
        // second half of the time axis -> LDS, first half adds it (fixed order) and plants the packed matrix
        double* tred = red + 6 * 256;  // [wave 1..3][2][64]
        if (half == 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) red[((which * 3 + i) * 4 + q) * 64 + lane] = c[i][q];
        }
        if (wid > 0) {
            tred[((wid - 1) * 2 + 0) * 64 + lane] = tl0;
            tred[((wid - 1) * 2 + 1) * 64 + lane] = tl1;
        }
        __syncthreads();
        if (half == 0) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int bi = which == 0 ? 2 : (i < 2 ? 1 : 0);
                const int bj = which == 0 ? i : (i < 2 ? i : 0);
                const int cb = 16 * bj + col;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 16 * bi + kq + 4 * q;
                    const double val = c[i][q] + red[((which * 3 + i) * 4 + q) * 64 + lane];
                    if (cb <= row && row < rb) Xp[tri_row_off(row) + cb] = val + (cb == row ? 1.0 : 0.0);
                }
            }
        }
        if (wid == 0 && ne > 0) {
            const double s0 = ((tl0 + tred[0 * 64 + lane]) + tred[2 * 64 + lane]) + tred[4 * 64 + lane];
            const double s1 = ((tl1 + tred[1 * 64 + lane]) + tred[3 * 64 + lane]) + tred[5 * 64 + lane];
            if (lane <= 48) Xp[tri_row_off(48) + lane] = s0 + (lane == 48 ? 1.0 : 0.0);
            if (ne > 1 && lane <= 49) Xp[tri_row_off(49) + lane] = s1 + (lane == 49 ? 1.0 : 0.0);
        }
    }
    __syncthreads();
    // ---- F2: one wave, rows / columns in registers ----
    if (wid == 0) {
        bool ok;
        {
            double rr[LRP];
            ok = wave_chol_rows<LRP>(rr, Xp, lane);
        }
        {
            double x[LRP];
            wave_tri_inverse_cols<LRP>(Xp, x, lane);
            tri_wave_sync();
            if (lane < LRP) {  // X overwrites L, row-major packed: X[i][c] for i >= c
#pragma unroll
                for (int i = 0; i < LRP; ++i)
                    if (i >= lane) Xp[tri_row_off(i) + lane] = x[i];
            }
        }
        if (lane == 0) {
            s_ok = ok ? 1 : 0;
            A.failg[m * L + l] = ok ? 0 : 1;
            if (!ok) atomicAdd(A.fail, 1);
        }
    }
    __syncthreads();
    {
        double* xd = A.xg + (int64_t)(m * L + l) * A.pkg;
        for (int i = tid; i < LPK; i += 256) xd[i] = Xp[i];
    }
    if (!A.do_v || !s_ok) return;  // a failed factor leaves v as it is (core.py:112-113)
    // ---- F3: v_t = |X g_t|^2.  Rows < 48 of Z = X G': X tiles (ib, kb <= ib), ib < 3, as A operands in registers, B
    // operand G'[k][n] = G[t0 + n][k], D[row = kq + 4 q][n]; rows 48, 49: dot products of the same B operands with the
    // rows of X (LDS), reduced over the four k-lanes of a time bin ----
    double xa[6][4];
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
        int ib, kb;
        ltile_of(pr, ib, kb);
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) {
            const int i = 16 * ib + col, k = 16 * kb + 4 * sq + kq;
            xa[pr][sq] = (k <= i) ? Xp[tri_row_off(i) + k] : 0.0;   // rows i >= r: the identity (planted above)
        }
    }
    const double* x48 = Xp + tri_row_off(48);
    const double* x49 = Xp + tri_row_off(49);
    const int ntb = (T + 15) / 16;
    for (int tb = wid; tb < ntb; tb += 4) {
        const int t = 16 * tb + col;
        const bool tin = t < T;
        double4_t acc[3];
#pragma unroll
        for (int ib = 0; ib < 3; ++ib) acc[ib] = double4_t{0.0, 0.0, 0.0, 0.0};
        double z0 = 0.0, z1 = 0.0;
#pragma unroll
        for (int kb = 0; kb < 3; ++kb) {
            if (16 * kb >= rb) continue;  // wave-uniform: those columns of G are zero
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {
                const int k = 16 * kb + 4 * sq + kq;
                const double gB = (tin && k < rb) ? Gl[(int64_t)t * r + k] : 0.0;
#pragma unroll
                for (int ib = kb; ib < 3; ++ib)
                    acc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[(ib * (ib + 1)) / 2 + kb][sq], gB, acc[ib], 0, 0, 0);
                if (ne > 0) {
                    z0 = fma(x48[k], gB, z0);
                    z1 = fma(x49[k], gB, z1);  // (row 49 of X is the unit row when r == 49: x49[k] = 0 for k < 49)
                }
            }
        }
        double vv = 0.0;
#pragma unroll
        for (int ib = 0; ib < 3; ++ib)
#pragma unroll
            for (int q = 0; q < 4; ++q) vv = fma(acc[ib][q], acc[ib][q], vv);
        vv += __shfl_xor(vv, 16, 64);
        vv += __shfl_xor(vv, 32, 64);
        if (ne > 0) {
            // columns 48, 49 of G: two lanes of the four carry them (kq = 0, 1)
            const int k = 48 + kq;
            const double gT = (tin && kq < 2 && k < r) ? Gl[(int64_t)t * r + k] : 0.0;
            z0 = fma(x48[kq < 1 ? 48 : 0], kq < 1 ? gT : 0.0, z0);
            z1 = fma(x49[kq < 2 ? k : 0], kq < 2 ? gT : 0.0, z1);
            z0 += __shfl_xor(z0, 16, 64);
            z0 += __shfl_xor(z0, 32, 64);
            z1 += __shfl_xor(z1, 16, 64);
            z1 += __shfl_xor(z1, 32, 64);
            vv = fma(z0, z0, vv);
            if (ne > 1) vv = fma(z1, z1, vv);
        }
        if (kq == 0 && tin) v_s[t] = vv;
    }
}

// task (m, l): the Newton step on the posterior mean in the push-through form (see mean_task),
//     delta = G (I + H)^-1 G'(ra + W mu) - mu,   (I + H)^-1 = X'X,
// one reduction over the time axis (split over the four waves, lane <-> column of G), the two triangular products by
// one wave, one expansion (thread <-> time bin).  `last`: the clipped step is handed back as dmu.
__global__ void __launch_bounds__(256, 4) elong_mean(SplitArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    double* Xp = smem;               // LPK
    double* part = smem + LPK;       // 4 x 64
    double* vec = part + 256;        // 64
    double* vec2 = vec + 64;         // 64
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int L = A.L;
    const int m = blockIdx.x / L, l = blockIdx.x - m * L;
    const int64_t r0 = A.off[m];
    const int T = (int)(A.off[m + 1] - r0);
    if (A.failg[m * L + l]) {  // singular system: zero update (core.py:92-94)
        if (tid == 0) atomicAdd(A.fail, 1);
        if (A.last)
            for (int t = tid; t < T; t += 256) A.dmu[(r0 + t) * L + l] = 0.0;
        return;
    }
    const int pidx = A.unit_prior[m];
    const int r = __builtin_amdgcn_readfirstlane(A.prior_rl[pidx * L + l]);
    const double* __restrict__ Gl = A.prior_base[pidx] + A.prior_goff[pidx * L + l];
    const double* __restrict__ w_s = A.w + (int64_t)l * A.ld + r0;
    const double* __restrict__ ra_s = A.ra + (int64_t)l * A.ld + r0;
    double* mu_s = A.mu + (int64_t)l * A.ld + r0;
    {
        const double* xs = A.xg + (int64_t)(m * L + l) * A.pkg;
        for (int i = tid; i < LPK; i += 256) Xp[i] = xs[i];
    }
    // c = G' s, s = ra + w mu.  Sixteen lanes share a row of G and walk its columns sixteen at a time (128 contiguous
    // bytes per row and load), four rows per wave and step; the time axis in four contiguous quarters over the waves.
    const int r4 = lane >> 4, c16 = lane & 15;
    {
        const int Tc = (((T + 3) / 4) + 3) & ~3;
        const int ta = wid * Tc, tb = ta + Tc < T ? ta + Tc : T;
        double acc[4] = {0.0, 0.0, 0.0, 0.0};
        for (int t0 = ta; t0 < tb; t0 += 16) {  // four steps (sixteen rows, twenty-eight loads) in flight: the loop is a chain of L2 round trips
            double gv[4][4], sv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 4 * u + r4;
                const bool in = t < tb;
                const int tc = in ? t : ta;
                const double* row = Gl + (int64_t)tc * r;
                sv[u] = in ? fma(w_s[tc], mu_s[tc], ra_s[tc]) : 0.0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int cb = 16 * bb + c16;
                    gv[u][bb] = cb < r ? row[cb] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) acc[bb] = fma(gv[u][bb], sv[u], acc[bb]);
        }
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            acc[bb] += __shfl_xor(acc[bb], 16, 64);
            acc[bb] += __shfl_xor(acc[bb], 32, 64);
        }
        if (r4 == 0) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) part[wid * 64 + 16 * bb + c16] = acc[bb];
        }
    }
    __syncthreads();
    if (wid == 0) {
        vec2[lane] = (part[lane] + part[64 + lane]) + (part[128 + lane] + part[192 + lane]);
        tri_wave_sync();
        // z = X c (lane = row), sol = X' z (lane = column); rows / columns >= r of X are the identity and c is zero there
        double z = 0.0;
        if (lane < LRP) {
            const double* Xi = Xp + tri_row_off(lane);
            double z0 = 0.0, z1 = 0.0;
            int q = 0;
            for (; q + 1 <= lane; q += 2) {
                const double2 x2 = *reinterpret_cast<const double2*>(Xi + q);
                z0 = fma(x2.x, vec2[q], z0);
                z1 = fma(x2.y, vec2[q + 1], z1);
            }
            if (q <= lane) z0 = fma(Xi[q], vec2[q], z0);
            z = z0 + z1;
        }
        vec[lane] = z;
        tri_wave_sync();
        double sol = 0.0;
        if (lane < LRP) {
            double s0 = 0.0, s1 = 0.0;
            int i = lane;
            for (; i + 1 < LRP; i += 2) {
                s0 = fma(Xp[tri_row_off(i) + lane], vec[i], s0);
                s1 = fma(Xp[tri_row_off(i + 1) + lane], vec[i + 1], s1);
            }
            if (i < LRP) s0 = fma(Xp[tri_row_off(i) + lane], vec[i], s0);
            sol = s0 + s1;
        }
        tri_wave_sync();
        vec2[lane] = lane < r ? sol : 0.0;
    }
    __syncthreads();
    // delta_t = G[t] . sol - mu_t, clipped; mu += delta: the same sixteen-lanes-per-row walk, reduced over the sixteen
    {
        double so[4];
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) so[bb] = vec2[16 * bb + c16];  // zero beyond the rank
        for (int t0 = 16 * wid; t0 < T; t0 += 64) {  // four steps (sixteen rows) per wave and turn
            double gv[4][4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int t = t0 + 4 * u + r4;
                const double* row = Gl + (int64_t)(t < T ? t : 0) * r;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) {
                    const int cb = 16 * bb + c16;
                    gv[u][bb] = cb < r ? row[cb] : 0.0;
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                double d = (gv[u][0] * so[0] + gv[u][1] * so[1]) + (gv[u][2] * so[2] + gv[u][3] * so[3]);
#pragma unroll
                for (int o = 8; o >= 1; o >>= 1) d += __shfl_xor(d, o, 64);
                const int t = t0 + 4 * u + r4;
                if (c16 == 0 && t < T) {
                    const double mt = mu_s[t];
                    double sd = d - mt;
                    sd = fmin(fmax(sd, -A.dmu_bound), A.dmu_bound);
                    mu_s[t] = mt + sd;
                    if (A.last) A.dmu[(r0 + t) * L + l] = sd;
                }
            }
        }
    }
}

// the stream the launch helpers below enqueue on: the handle's main stream, or the second E-step lane (the unit set
// split in two halves that run their sweeps side by side, launch_estep_split)
static thread_local hipStream_t t_lane = nullptr;
// (a helper reached without a lane would launch on the legacy null stream without a word)
#define NEED_LANE(ctx) \
    do { if (!t_lane) return vlgp_fail(ctx, VLGP_ERR_STATE, "split E-step launch helper called outside launch_estep_split"); } while (0)

template <int LT, int CS, int RPL>
int run_passes_cs(vlgp_ctx* ctx, const SplitArgs& A, int kind, const double* cols) {
    constexpr int RPB = (256 / CS) * RPL;
    const dim3 grid((unsigned)((A.rows + RPB - 1) / RPB)), blk(256);
    NEED_LANE(ctx);
    hipStream_t st = t_lane;
    if (A.xb) {
        if (kind == SP_RES) hipLaunchKernelGGL((esplit_pass<LT, SP_RES, true, CS, RPL>), grid, blk, 0, st, A, cols);
        else hipLaunchKernelGGL((esplit_pass<LT, SP_W, true, CS, RPL>), grid, blk, 0, st, A, cols);
    } else {
        if (kind == SP_RES) hipLaunchKernelGGL((esplit_pass<LT, SP_RES, false, CS, RPL>), grid, blk, 0, st, A, cols);
        else hipLaunchKernelGGL((esplit_pass<LT, SP_W, false, CS, RPL>), grid, blk, 0, st, A, cols);
    }
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

template <int LT>
int run_passes(vlgp_ctx* ctx, const SplitArgs& A, int kind, const double* cols) {
    NEED_LANE(ctx);
    if (kind == SP_YA) {
        const dim3 grid((unsigned)((A.rows + 255) / 256)), blk(256);
        if (A.xb) hipLaunchKernelGGL((esplit_pass<LT, SP_YA, true, 1, 1>), grid, blk, 0, t_lane, A, cols);
        else hipLaunchKernelGGL((esplit_pass<LT, SP_YA, false, 1, 1>), grid, blk, 0, t_lane, A, cols);
        HIPCHK(ctx, hipGetLastError());
        return VLGP_OK;
    }
    // channel split: four waves per row group while a wave keeps >= 16 channels (measured at 200 k rows x 100
    // channels: 39.4 / 34.4 / 32.2 us for 1 / 2 / 4; at 131 k rows 35.9 / 24.8 / 24.6 us)
    static const int forced = getenv("VLGP_PASS_SPLIT") ? atoi(getenv("VLGP_PASS_SPLIT")) : 0;
    static const int forced_rpl = getenv("VLGP_PASS_RPL") ? atoi(getenv("VLGP_PASS_RPL")) : 0;
    int cs = forced;
    if (cs != 1 && cs != 2 && cs != 4) cs = A.ntot >= 64 ? 4 : (A.ntot >= 32 ? 2 : 1);
    const int rpl = forced_rpl == 2 && LT <= 5 ? 2 : 1;
    if constexpr (LT <= 5) {
        if (rpl == 2) {
            if (cs == 4) return run_passes_cs<LT, 4, 2>(ctx, A, kind, cols);
            if (cs == 2) return run_passes_cs<LT, 2, 2>(ctx, A, kind, cols);
            return run_passes_cs<LT, 1, 2>(ctx, A, kind, cols);
        }
    }
    if (cs == 4) return run_passes_cs<LT, 4, 1>(ctx, A, kind, cols);
    if (cs == 2) return run_passes_cs<LT, 2, 1>(ctx, A, kind, cols);
    return run_passes_cs<LT, 1, 1>(ctx, A, kind, cols);
}

int run_pass(vlgp_ctx* ctx, const SplitArgs& A, int LT, int kind, const double* cols) {
    if (LT == 3) return run_passes<3>(ctx, A, kind, cols);
    if (LT == 5) return run_passes<5>(ctx, A, kind, cols);
    if (LT == 8) return run_passes<8>(ctx, A, kind, cols);
    return run_passes<10>(ctx, A, kind, cols);
}

// the y pass: coalesced form while a lane's channels fit in registers (N <= 128), else the lane-per-row pass
int run_ya(vlgp_ctx* ctx, const SplitArgs& A, int LT, const double* cols, const double* ycoef) {
    static const bool old_form = getenv("VLGP_YA_ROWLANE") != nullptr;
    if (A.N > 128 || old_form) return run_pass(ctx, A, LT, SP_YA, cols);
    const int rows_per_wave = 64;
    const dim3 grid((unsigned)((A.rows + 4 * rows_per_wave - 1) / (4 * rows_per_wave))), blk(256);
    NEED_LANE(ctx);
    hipStream_t st = t_lane;
#define ESPLIT_YA(LTV, NJV) \
    hipLaunchKernelGGL((esplit_ya<LTV, NJV>), grid, blk, 0, st, A.N, A.L, A.rows, A.ld, A.y, ycoef, A.ya, rows_per_wave)
    const bool small = A.N <= 64;
    if (LT == 3) { if (small) ESPLIT_YA(3, 4); else ESPLIT_YA(3, 8); }
    else if (LT == 5) { if (small) ESPLIT_YA(5, 4); else ESPLIT_YA(5, 8); }
    else if (LT == 8) { if (small) ESPLIT_YA(8, 4); else ESPLIT_YA(8, 8); }
    else { if (small) ESPLIT_YA(10, 4); else ESPLIT_YA(10, 8); }
#undef ESPLIT_YA
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

int run_latent_class(vlgp_ctx* ctx, const SplitArgs& A, int maxra, bool mean) {
    const int tasks = A.M * A.n_lat;
    if (tasks == 0) return VLGP_OK;
    const dim3 grid(A.shg ? (unsigned)(((A.M + 3) / 4) * A.n_lat) : (unsigned)((tasks + 3) / 4)), blk(256);
    static const int lds_pad = getenv("VLGP_LDS_PAD") ? atoi(getenv("VLGP_LDS_PAD")) : 0;  // occupancy experiments
    const size_t lds = (size_t)(4 * (A.pkl + A.lds_g + 128 + (mean ? 192 : 0)) + (A.shg ? A.shg_cap : 0)) * 8 + (size_t)lds_pad;
    NEED_LANE(ctx);
    hipStream_t st = t_lane;
#define ESPLIT_LAUNCH(RA, MEANV)                                                                                      \
    do {                                                                                                              \
        auto fn = (MEANV && A.last) ? esplit_latent<RA, MEANV, MEANV> : esplit_latent<RA, MEANV>;                                                                           \
        if (lds > 64 * 1024)                                                                                          \
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn),                                        \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                   \
        hipLaunchKernelGGL(fn, grid, blk, lds, st, A);                                                                \
    } while (0)
    if (maxra == 16) { if (mean) ESPLIT_LAUNCH(16, true); else ESPLIT_LAUNCH(16, false); }
    else if (maxra == 20) { if (mean) ESPLIT_LAUNCH(20, true); else ESPLIT_LAUNCH(20, false); }
    else if (maxra == 24) { if (mean) ESPLIT_LAUNCH(24, true); else ESPLIT_LAUNCH(24, false); }
    else { if (mean) ESPLIT_LAUNCH(32, true); else ESPLIT_LAUNCH(32, false); }
#undef ESPLIT_LAUNCH
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

// one workgroup (four waves) per (latent, group of 64 units): estep_lane.h
int run_latent_lane(vlgp_ctx* ctx, const SplitArgs& A, bool mean) {
    const int groups = (A.M + 63) / 64;
    if (groups == 0 || A.n_lat == 0) return VLGP_OK;
    const bool last = mean && A.last;
    const size_t lds = lane_lds_doubles(A.shg_T, mean, last) * 8;
    if (lds > (size_t)ctx->lds_max)  // (162,304 bytes at T = 64 on the last sweep: 1.5 KB under gfx950's 160 KB)
        return vlgp_fail(ctx, VLGP_ERR_ARG, "lane-per-task E-step launch needs %zu bytes of LDS, the device has %d", lds, ctx->lds_max);
    const dim3 grid((unsigned)(groups * A.n_lat)), blk(256);
    NEED_LANE(ctx);
    hipStream_t st = t_lane;
    int rtop = 0;
    for (int i = 0; i < A.n_lat; ++i) rtop = A.shg_rk[i] > rtop ? A.shg_rk[i] : rtop;
    auto fn = rtop <= 13 ? (!mean ? esplit_lane<0, 13> : (last ? esplit_lane<2, 13> : esplit_lane<1, 13>))
                         : (!mean ? esplit_lane<0> : (last ? esplit_lane<2> : esplit_lane<1>));
    if (lds > 64 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    hipLaunchKernelGGL(fn, grid, blk, lds, st, A);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

// the lane-per-task blocks of `Aln` and the wave-per-task blocks of `Alt` (class maxra: 20 / 24 / 32) as one grid (esplit_mix)
int run_latent_mix(vlgp_ctx* ctx, const SplitArgs& Aln, const SplitArgs& Alt, int maxra, bool mean) {
    const int n_lane = ((Aln.M + 63) / 64) * Aln.n_lat;
    const int n_lat = (Alt.M * Alt.n_lat + 3) / 4;
    if (n_lane == 0 || n_lat == 0) return vlgp_fail(ctx, VLGP_ERR_STATE, "mixed E-step launch without both kinds of blocks");
    const bool last = mean && Aln.last;
    const size_t lds_lane = lane_lds_doubles(Aln.shg_T, mean, last) * 8;
    const size_t lds_lat = (size_t)(4 * (Alt.pkl + Alt.lds_g + 128 + (mean ? 192 : 0))) * 8;
    const size_t lds = lds_lane > lds_lat ? lds_lane : lds_lat;
    if (lds > (size_t)ctx->lds_max)
        return vlgp_fail(ctx, VLGP_ERR_ARG, "mixed E-step launch needs %zu bytes of LDS, the device has %d", lds, ctx->lds_max);
    const dim3 grid((unsigned)(n_lane + n_lat)), blk(256);
    NEED_LANE(ctx);
    hipStream_t st = t_lane;
    const int kind = !mean ? 0 : (last ? 2 : 1);
    for (int i = 0; i < Aln.n_lat; ++i)
        if (Aln.shg_rk[i] > 13) return vlgp_fail(ctx, VLGP_ERR_STATE, "mixed E-step launch with a lane-per-task rank above 13");
#define ESPLIT_MIX(KINDV, RA)                                                                                       \
    do {                                                                                                            \
        auto fn = esplit_mix<KINDV, RA, 13>;                                                                          \
        if (lds > 64 * 1024)                                                                                        \
            HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn),                                      \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                 \
        hipLaunchKernelGGL(fn, grid, blk, lds, st, Aln, Alt, n_lane);                                               \
    } while (0)
#define ESPLIT_MIX_RA(RA)                                                                                           \
    do {                                                                                                            \
        if (kind == 0) ESPLIT_MIX(0, RA); else if (kind == 1) ESPLIT_MIX(1, RA); else ESPLIT_MIX(2, RA);            \
    } while (0)
    if (maxra <= 20) ESPLIT_MIX_RA(20);
    else if (maxra <= 24) ESPLIT_MIX_RA(24);
    else ESPLIT_MIX_RA(32);
#undef ESPLIT_MIX_RA
#undef ESPLIT_MIX
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

// long units: one workgroup per (unit, latent)
int run_latent_long(vlgp_ctx* ctx, const SplitArgs& A, bool mean) {
    const unsigned tasks = (unsigned)(A.M * A.L);
    if (tasks == 0) return VLGP_OK;
    NEED_LANE(ctx);
    if (mean) hipLaunchKernelGGL(elong_mean, dim3(tasks), dim3(256), (size_t)(LPK + 384) * 8, t_lane, A);
    else hipLaunchKernelGGL(elong_factor, dim3(tasks), dim3(256), (size_t)(LPK + LRED) * 8, t_lane, A);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

// One launch per rank class: the latents of rank <= 16 run the lean instantiation (staged G, small LDS footprint),
// the others the mixed one -- a single latent above 16 slows its own waves only.
struct LatentClasses {
    int n_ln = 0, ln[16];  // rank <= LANE_RMAX and one prior for the whole set: lane-per-task launches (estep_lane.h)
    int n_lo = 0, lo[16];
    int n_hi = 0, hi[16];
    int maxra_hi = 16;
    int lds_g_lo = 256;
    int single_T = 0;  // > 0: all units have this length (one prior): the rank <= 16 launch shares G per workgroup
    const Prior* single = nullptr;
    bool mix = false;  // lane-per-task and wave-per-task latents in one launch (esplit_mix)
};

int run_latent(vlgp_ctx* ctx, SplitArgs A, const LatentClasses& C, bool mean) {
    // lane-per-task latents AND others: one mixed launch (esplit_mix) instead of two or three dependent ones; the latents
    // of rank 15, 16 then ride in the class of the higher ones (VLGP_ESTEP_MIX=0: the separate launches; per call)
    const bool mix = C.mix;
    SplitArgs Ahi = A;
    const int n_hi = C.n_hi + (mix ? C.n_lo : 0);
    const int maxra_hi = (mix && C.maxra_hi < 20) ? 20 : C.maxra_hi;
    if (n_hi) {  // the long tasks first
        Ahi.n_lat = n_hi;
        for (int i = 0; i < C.n_hi; ++i) Ahi.lat[i] = C.hi[i];
        if (mix)
            for (int i = 0; i < C.n_lo; ++i) Ahi.lat[C.n_hi + i] = C.lo[i];
        Ahi.pkl = tri_packed_size(maxra_hi <= 20 ? 20 : (maxra_hi <= 24 ? 24 : 32));
        Ahi.lds_g = 256;
        Ahi.shg = 0;
        if (!mean) {  // factor: staging tile of H, then the multiplier rows (RA x (RA + 2) doubles); X overwrites it
            Ahi.lds_g = maxra_hi * (maxra_hi + 2);
            Ahi.pkl = 0;
        }
        if (!mix) CHK(run_latent_class(ctx, Ahi, maxra_hi, mean));
    }
    if (C.n_ln) {
        // ONE launch for all of them, the highest ranks first in the grid (their workgroups live longest)
        int ord[16];
        for (int i = 0; i < C.n_ln; ++i) ord[i] = C.ln[i];
        std::stable_sort(ord, ord + C.n_ln, [&](int x, int y) { return C.single->rl[x] > C.single->rl[y]; });
        A.n_lat = C.n_ln;
        for (int i = 0; i < C.n_ln; ++i) {
            A.lat[i] = ord[i];
            A.shg_rk[i] = C.single->rl[ord[i]];
            A.shg_gl[i] = C.single->d_compact + C.single->goff[ord[i]];
        }
        A.shg = 1;
        A.shg_T = C.single_T;
        static const int prio = getenv("VLGP_LANE_PRIO") ? atoi(getenv("VLGP_LANE_PRIO")) : 0;  // (measured: 3 is 5 % slower)
        static const int clk_kind = getenv("VLGP_LANE_CLOCK") ? atoi(getenv("VLGP_LANE_CLOCK")) : 0;
        A.prio = prio;
        A.clk = ctx->d_clk;
        A.clk_kind = clk_kind;
        if (mix) {
            ctx->last_estep_mix = 1;
            return run_latent_mix(ctx, A, Ahi, maxra_hi, mean);
        }
        CHK(run_latent_lane(ctx, A, mean));
    }
    if (C.n_lo) {
        A.n_lat = C.n_lo;
        for (int i = 0; i < C.n_lo; ++i) A.lat[i] = C.lo[i];
        A.pkl = tri_packed_size(16);
        A.lds_g = C.lds_g_lo;
        A.shg = 0;
        static const bool no_shg = getenv("VLGP_ESTEP_NO_SHARED_G") != nullptr;
        if (C.single_T > 0 && C.single && !no_shg) {
            for (int i = 0; i < C.n_lo; ++i) {
                A.shg_rk[i] = C.single->rl[C.lo[i]];
                A.shg_gl[i] = C.single->d_compact + C.single->goff[C.lo[i]];
            }
            // G once per workgroup (one latent x four units): 6.4 KB + 4 x 4.2 KB (factor) or 4 x 3.6 KB (mean) of LDS
            // instead of 4 x 7.7 / 4 x 9.2 KB -- six / seven workgroups per CU instead of five / four
            A.shg = 1;
            A.shg_T = C.single_T;
            A.shg_cap = C.lds_g_lo;
            A.lds_g = mean ? 0 : 256;
            // factor: the packed X overwrites the 16 x 16 staging tile (dead after the elimination); the regular mean
            // sweeps keep X in registers
            if (!mean || !A.last) A.pkl = 0;
        }
        CHK(run_latent_class(ctx, A, 16, mean));
    }
    return VLGP_OK;
}

}  // namespace

int launch_estep_split(vlgp_ctx* ctx, UnitSet& us, EstepArgs E, int* handled) {
    *handled = 0;
    const int N = ctx->N, L = ctx->L;
    const char* sw = getenv("VLGP_ESTEP_SPLIT");
    if (sw && sw[0] == '0') return VLGP_OK;
    if (getenv("VLGP_ESTEP_GENERIC")) return VLGP_OK;
    if (L > 10 || N > 1024) return VLGP_OK;
    // long units (full-length trials): the same launch sequence with one workgroup per (unit, latent) task, once there
    // are enough tasks to fill the chip (VLGP_ESTEP_LSPLIT=0/1 never / always)
    const bool lng = us.Tmax > 64;
    if (lng) {
        const char* lsw = getenv("VLGP_ESTEP_LSPLIT");
        if (lsw && lsw[0] == '0') return VLGP_OK;
        if (ctx->R > LRP) return VLGP_OK;
        if (!(lsw && lsw[0] == '1') && (int64_t)us.M * L < 128) return VLGP_OK;
    }
    // the persistent kernel (one workgroup per unit, two or three per CU) wins while its workgroups are ONE generation
    // (measured at N = 100, L = 5: 500 units 0.89 ms persistent / 1.09 ms split; 1000 units 2.09 / 1.42 ms -- the second
    // generation costs as much as the first however few units it holds)
    if (!lng && !(sw && sw[0] == '1') && (us.rows < 16LL * 1024 || us.M <= 2 * ctx->n_cu)) return VLGP_OK;
    const bool need_prior = (E.mode & (EM_FACTOR0 | EM_MEAN | EM_V)) != 0;
    int rmax = 0;
    int rlat[16] = {0};  // largest rank of each latent over the priors this set uses
    int64_t gw_lo = 0;                       // doubles of staged G per (unit, latent) among the rank <= 16 latents
    if (need_prior) {
        for (auto& kv : ctx->priors) {
            const Prior& pr = kv.second;
            if (pr.T < us.Tmin || pr.T > us.Tmax) continue;
            for (int l = 0; l < L; ++l) {
                rmax = pr.rl[l] > rmax ? pr.rl[l] : rmax;
                rlat[l] = pr.rl[l] > rlat[l] ? pr.rl[l] : rlat[l];
            }
        }
        for (auto& kv : ctx->priors) {
            const Prior& pr = kv.second;
            if (pr.T < us.Tmin || pr.T > us.Tmax) continue;
            for (int l = 0; l < L; ++l) {
                if (rlat[l] > 16) continue;
                const int64_t g = (int64_t)pr.T * ((pr.rl[l] + 1) & ~1);
                gw_lo = g > gw_lo ? g : gw_lo;
            }
        }
    }
    if (!lng && rmax > 32) return VLGP_OK;
    const int maxra = rmax <= 16 ? 16 : (rmax <= 20 ? 20 : (rmax <= 24 ? 24 : 32));
    const int LT = L <= 3 ? 3 : (L <= 5 ? 5 : (L <= 8 ? 8 : 10));
    const int REC = (2 * LT + 3 + 1) & ~1;
    const int pkg = lng ? LPK : tri_packed_size(maxra);
    // scratch of the set: ra | ya | xg | failg(int) ; records + wconst in ctx->d_ecols
    const int64_t nRL = us.rows * L;
    // lane-per-task launches (estep_lane.h): one prior for the whole set, T <= 64; VLGP_ESTEP_LANEPT=0 keeps the
    // wave-per-task kernels (per call: tests toggle it)
    const char* lpt = getenv("VLGP_ESTEP_LANEPT");
    const bool use_lane = !lng && need_prior && us.Tmin == us.Tmax && !(lpt && lpt[0] == '0') &&
                          !getenv("VLGP_ESTEP_NO_SHARED_G");
    const int64_t n_groups = ((int64_t)us.M + 63) / 64;
    const int64_t xl_len = use_lane ? n_groups * L * 64 * LANE_EMAX : 0;
    const int64_t need = 5 * nRL + (int64_t)us.M * L * pkg + ((int64_t)us.M * L + 1) / 2 + 8 + xl_len + (use_lane ? 2 * nRL : 0);
    if (us.scratch_len < need) {
        if (us.d_scratch) HIPCHK(ctx, hipFree(us.d_scratch));
        us.d_scratch = nullptr;
        us.scratch_len = 0;
        HIPCHK(ctx, hipMalloc(&us.d_scratch, (size_t)need * 8));
        us.scratch_len = need;
    }
    if (!ctx->d_ecols) HIPCHK(ctx, hipMalloc(&ctx->d_ecols, sizeof(double) * ((size_t)N * 50 + 32)));
    if (REC > 34) return VLGP_OK;
    double* cols = ctx->d_ecols;
    double* wconst = ctx->d_ecols + (int64_t)N * 34;
    double* ycoef = wconst + 32;  // (N, LT), LT <= 10
    hipLaunchKernelGGL(esplit_cols_kernel, dim3(1), dim3(256), 0, ctx->stream, N, L, LT, REC, ctx->d_a, ctx->d_b,
                       ctx->d_noise, ctx->d_gauss, cols, wconst, ycoef);
    HIPCHK(ctx, hipGetLastError());

    SplitArgs A;
    A.N = N; A.L = L; A.M = us.M; A.rows = us.rows;
    A.off = E.off; A.unit_prior = E.unit_prior; A.prior_base = E.prior_base; A.prior_rl = E.prior_rl;
    A.prior_goff = E.prior_goff;
    A.y = E.y; A.xb = E.xb; A.dmu = E.dmu;
    A.ld = us.rows;
    A.ra = us.d_scratch; A.ya = A.ra + nRL; A.mu = A.ya + nRL; A.v = A.mu + nRL; A.w = A.v + nRL;
    A.xg = A.w + nRL; A.pkg = pkg;
    {
        const unsigned nb = (unsigned)((us.rows + 255) / 256);
        hipLaunchKernelGGL(esplit_to_lm, dim3(nb), dim3(256), 0, ctx->stream, L, us.rows, E.mu, E.v, E.w, A.mu, A.v, A.w);
        HIPCHK(ctx, hipGetLastError());
    }
    A.failg = reinterpret_cast<int*>(A.xg + (int64_t)us.M * L * pkg);
    A.xl = reinterpret_cast<double*>(A.failg) + ((int64_t)us.M * L + 1) / 2 + 1;
    A.sv = use_lane ? A.xl + xl_len : nullptr;
    A.dl = use_lane ? A.sv + nRL : nullptr;
    A.dmask = 0;
    A.fail = E.fail;
    A.wconst = wconst;
    A.dmu_bound = E.dmu_bound;
    A.ntot = N; A.np = N - ctx->n_gauss;
    LatentClasses C;
    if (need_prior && us.Tmin == us.Tmax) {
        C.single_T = us.Tmax;
        for (auto& kv : ctx->priors)
            if (kv.second.T == us.Tmax) C.single = &kv.second;
    }
    for (int l = 0; l < L; ++l) {
        if (use_lane && C.single && rlat[l] <= LANE_RMAX) C.ln[C.n_ln++] = l;
        else if (rlat[l] <= 16) C.lo[C.n_lo++] = l;
        else C.hi[C.n_hi++] = l;
    }
    {
        // mixed launches carry the lane-per-task code compiled for ranks <= 13 (no scratch frame: a frame alone costs
        // every wave of the launch, and ROCr allocates it per queue at first use -- milliseconds, seen as an 8.8 ms
        // E-step in the middle of a fit); a latent at rank 14 rides with the wave-per-task ones there
        const char* mixsw = getenv("VLGP_ESTEP_MIX");
        const bool mix_on = !(mixsw && mixsw[0] == '0');
        bool others = C.n_hi > 0 || C.n_lo > 0;
        if (mix_on && C.n_ln > 0) {
            int keep = 0, moved = 0, mv[16];
            for (int i = 0; i < C.n_ln; ++i) {
                if (rlat[C.ln[i]] >= 14) mv[moved++] = C.ln[i];
                else C.ln[keep++] = C.ln[i];
            }
            if (others && moved > 0) {  // (nothing but lane-per-task latents: their launch as it is, rank 14 included)
                C.n_ln = keep;
                for (int i = 0; i < moved; ++i) C.lo[C.n_lo++] = mv[i];
            } else {  // restore
                for (int i = 0; i < moved; ++i) C.ln[keep + i] = mv[i];
            }
        }
        C.mix = mix_on && C.n_ln > 0 && (C.n_hi > 0 || C.n_lo > 0);
    }
    C.maxra_hi = maxra;
    C.lds_g_lo = (int)((gw_lo + 1) & ~1LL);  // staged G of a rank <= 16 latent (and the 16 x 16 staging tile)
    if (C.lds_g_lo < 256) C.lds_g_lo = 256;
    A.lds_g = 256; A.pkl = pkg; A.n_lat = 0;
    A.shg = 0; A.shg_cap = 0; A.shg_T = 0;
    A.do_v = 0; A.last = 0;
    *handled = lng ? 2 : 1;
    ctx->last_estep_mix = 0;
    HIPCHK(ctx, hipMemsetAsync(A.failg, 0, sizeof(int) * (size_t)us.M * L, ctx->stream));

    const int mode = E.mode;
    const bool with_mean = (mode & EM_MEAN) != 0;
    const int n_it = with_mean ? E.n_iter : ((mode & EM_W) ? 1 : 0);
    const int kind = lng ? VLGP_PROF_ESTEP_LONG
                         : (maxra <= 16 ? VLGP_PROF_ESTEP_RA16 : (maxra <= 24 ? VLGP_PROF_ESTEP_RA24 : VLGP_PROF_ESTEP_RA32));
    vlgp_prof_begin(ctx, kind);
    t_lane = ctx->stream;
    int rc = VLGP_OK;
    if (with_mean) rc = run_ya(ctx, A, LT, cols, ycoef);

    // TWO LANES.  The units are independent inside an E-step call (core.estep is a loop over units, core.py:123-126),
    // and every launch of a sweep is partly bound by the latency of one task's dependent chain (factor ~28 us, mean
    // ~17 us of their 47 / 21 us at C3) and by the ~2.5 us of a dependent launch boundary: the set is cut in two
    // parts (two by default, VLGP_ESTEP_LANES = 1 .. 4) whose sweeps are enqueued on their own streams, so that one half's latency-bound stretches sit under the other
    // half's issue-bound passes.  No dependency crosses the halves between the fork and the join; results are
    // bit-identical to the single lane (same arithmetic per unit).  VLGP_ESTEP_LANES=1 keeps one lane.
    const int lanes_env = getenv("VLGP_ESTEP_LANES") ? atoi(getenv("VLGP_ESTEP_LANES")) : 0;  // (per call: tests toggle it)
    int n_lanes = (lanes_env >= 1 && lanes_env <= VLGP_E_LANES) ? lanes_env : (us.M >= 3 * ctx->n_cu && n_it >= 2 ? 2 : 1);  // 2000 units: 2.11 -> 1.77 ms; 1000 units: no change with the wave-per-task launches, 1.60 -> 1.54 ms (and the H-step that follows 2.60 -> 2.41 ms) with the lane-per-task ones
    // (lane-per-task latents: the cuts are multiples of 64 units -- a half must not come out empty)
    while (n_lanes > 1 && us.M < (C.n_ln ? 64 : 8) * n_lanes) --n_lanes;
    for (int h = 1; h < n_lanes; ++h) {
        if (ctx->elane[h - 1]) continue;
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->elane[h - 1], hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_e_join[h - 1], hipEventDisableTiming));
    }
    if (n_lanes > 1 && !ctx->ev_e_fork) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_e_fork, hipEventDisableTiming));
    struct Half { SplitArgs pass, lat; hipStream_t st; int64_t rows; int M; };
    Half H[VLGP_E_LANES];
    {
        for (int h = 0; h < n_lanes; ++h) {
            // cuts at multiples of four units: the shared-G launches take four units per workgroup (64: a wave of the
            // lane-per-task launches)
            const int64_t cm = C.n_ln ? 63 : 3;
            int m0 = (int)(((int64_t)us.M * h / n_lanes + cm) & ~cm), m1 = (int)(((int64_t)us.M * (h + 1) / n_lanes + cm) & ~cm);
            if (m0 > us.M) m0 = us.M;
            if (m1 > us.M || h == n_lanes - 1) m1 = us.M;
            const int64_t row_lo = us.off[m0], row_hi = us.off[m1];
            Half& hf = H[h];
            hf.st = h == 0 ? ctx->stream : ctx->elane[h - 1];
            hf.M = m1 - m0;
            hf.rows = row_hi - row_lo;
            // per-unit kernels: unit tables shifted, rows stay absolute (off[] holds absolute rows)
            hf.lat = A;
            hf.lat.M = hf.M;
            hf.lat.off = A.off + m0;
            hf.lat.unit_prior = A.unit_prior ? A.unit_prior + m0 : nullptr;
            hf.lat.xg = A.xg + (int64_t)m0 * L * pkg;
            hf.lat.xl = A.xl + (int64_t)(m0 / 64) * L * 64 * LANE_EMAX;
            hf.lat.failg = A.failg + (int64_t)m0 * L;
            // row passes: every row-indexed pointer shifted to the half's first row
            hf.pass = A;
            hf.pass.rows = hf.rows;
            hf.pass.y = A.y + row_lo * N;
            hf.pass.xb = A.xb ? A.xb + row_lo * N : nullptr;
            hf.pass.mu = A.mu + row_lo; hf.pass.v = A.v + row_lo; hf.pass.w = A.w + row_lo;
            hf.pass.ra = A.ra + row_lo; hf.pass.ya = A.ya + row_lo;
            if (A.sv) { hf.pass.sv = A.sv + row_lo; hf.pass.dl = A.dl + row_lo; }
        }
    }
    // (measured: three or four lanes are SLOWER than one -- E-step 5.7 ms against 3.5 / 3.0 for one / two at C3 -- and
    // starting the second lane one or two launches late changes nothing)
    if (n_lanes > 1 && rc == VLGP_OK) {
        HIPCHK(ctx, hipEventRecord(ctx->ev_e_fork, ctx->stream));
        for (int h = 1; h < n_lanes; ++h) HIPCHK(ctx, hipStreamWaitEvent(ctx->elane[h - 1], ctx->ev_e_fork, 0));
    }
    for (int it = (mode & EM_FACTOR0) ? -1 : 0; it < n_it && rc == VLGP_OK; ++it) {
        const bool last = it == n_it - 1;
        for (int h = 0; h < n_lanes && rc == VLGP_OK; ++h) {
            Half& hf = H[h];
            t_lane = hf.st;
            bool do_factor, do_v;
            // per-kernel timing: the launches of ONE sweep per call (first lane) are bracketed (events on every launch
            // would cost more than they measure); with two lanes the other lane's launches run beside the bracketed one
            const bool sample = ctx->prof_on && h == 0 && it == (n_it > 1 ? 1 : 0);
            const double share = (double)hf.rows, tasks = (double)hf.M * L;
            if (it >= 0) {
                if (with_mean) {
                    if (sample) vlgp_prof_begin(ctx, VLGP_PROF_ESTEP_PASS, hf.st);
                    // ra itself is read by the wave-per-task mean launches and by the last sweep's dmu form only: with
                    // every latent on the lane-per-task launch the pass writes s = ra + w mu alone (4 MB less per lane
                    // and sweep at C3)
                    double* const ra_keep = hf.pass.ra;
                    if (C.n_ln && !C.n_lo && !C.n_hi && !last) hf.pass.ra = nullptr;
                    rc = run_pass(ctx, hf.pass, LT, SP_RES, cols);
                    hf.pass.ra = ra_keep;
                    if (sample) vlgp_prof_end(ctx, VLGP_PROF_ESTEP_PASS, share, hf.st);
                    hf.lat.last = last ? 1 : 0;
                    if (sample) vlgp_prof_begin(ctx, VLGP_PROF_ESTEP_MEAN, hf.st);
                    if (rc == VLGP_OK) rc = lng ? run_latent_long(ctx, hf.lat, true) : run_latent(ctx, hf.lat, C, true);
                    if (sample) vlgp_prof_end(ctx, VLGP_PROF_ESTEP_MEAN, tasks, hf.st);
                }
                if (sample) vlgp_prof_begin(ctx, VLGP_PROF_ESTEP_PASS, hf.st);
                hf.pass.dmask = 0;  // the mean launch of the lane-per-task latents left its step for this pass to apply
                if (with_mean)
                    for (int i = 0; i < C.n_ln; ++i) hf.pass.dmask |= 1u << C.ln[i];
                if (rc == VLGP_OK) rc = run_pass(ctx, hf.pass, LT, SP_W, cols);
                hf.pass.dmask = 0;
                if (sample) vlgp_prof_end(ctx, VLGP_PROF_ESTEP_PASS, share, hf.st);
                do_factor = with_mean && (E.vb || !last);
                do_v = E.vb != 0;
            } else {
                do_factor = true;
                do_v = (mode & EM_V) && !with_mean;
            }
            if (do_factor && rc == VLGP_OK) {
                hf.lat.do_v = do_v ? 1 : 0;
                if (sample) vlgp_prof_begin(ctx, VLGP_PROF_ESTEP_FACTOR, hf.st);
                rc = lng ? run_latent_long(ctx, hf.lat, false) : run_latent(ctx, hf.lat, C, false);
                if (sample) vlgp_prof_end(ctx, VLGP_PROF_ESTEP_FACTOR, tasks, hf.st);
            }
        }
    }
    t_lane = ctx->stream;
    for (int h = 1; h < n_lanes; ++h) {  // join whatever happened above: the extra lanes must never outlive the call
        (void)hipEventRecord(ctx->ev_e_join[h - 1], ctx->elane[h - 1]);
        (void)hipStreamWaitEvent(ctx->stream, ctx->ev_e_join[h - 1], 0);
    }
    if (rc == VLGP_OK) {
        const unsigned nb = (unsigned)((nRL + 255) / 256);
        hipLaunchKernelGGL(esplit_from_lm, dim3(nb), dim3(256), 0, ctx->stream, L, us.rows, A.mu, A.v, A.w, E.mu, E.v, E.w);
        if (hipGetLastError() != hipSuccess) rc = vlgp_fail(ctx, VLGP_ERR_HIP, "esplit_from_lm launch failed");
    }
    vlgp_prof_end(ctx, kind, (double)us.M * (E.n_iter > 0 ? E.n_iter : 1));
    if (ctx->last_estep_mix && !lng) *handled = 3;
    return rc;
}
