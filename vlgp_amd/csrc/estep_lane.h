// Split E-step, per-latent phases with ONE LANE PER (unit, latent) TASK (round 5).
//
// Included by estep_split.hip (inside its anonymous namespace, after SplitArgs).  Same mathematics as factor_task /
// mean_task16 / mean_task_last there (reference vlgp/core.py:76-113): H = I + G'WG, X = chol(H)^-1,
// v_t = |X g_t|^2, delta = G X'X G'(ra + W mu) - mu.
//
// Why this shape.  The wave-per-task kernels spend a 64-lane vector machine on a 9..14-wide matrix: the augmented
// elimination keeps 32 of 64 lanes busy with ~16 overhead instructions per pivot, H is built as a full 16 x 16 MFMA
// tile of which r (r + 1) / 2 entries are wanted -- measured 5.8 k SIMD-cycles per task (factor launch 47 us at C3).
// With every unit of a launch sharing one prior factor G per latent (all units of the set have the same length: the
// windows of vem), the 64 lanes of a wave can instead be 64 UNITS of one latent:
//   * the packed lower triangle of H lives in the lane's registers (r = 14: 105 doubles), every index static;
//   * G is wave-uniform: its rows arrive through scalar loads and enter the FMAs as SGPR operands, so the build is
//     r (r + 1) / 2 + r vector instructions per time bin with all 64 lanes useful, no MFMA padding;
//   * Cholesky and triangular inverse are straight-line register code (r^3 / 3 FMAs), no cross-lane traffic;
//   * the hand-over X goes to global memory entry-major ([entry][lane]: 512 contiguous bytes per entry).
// ~ (r^2 + 3 r) T + r^3 / 3 vector instructions per 64 tasks instead of ~1400 per task.
//
// A workgroup = 64 units of one latent x FOUR waves, each wave a quarter of the time bins (the loops over t are where
// the instructions are; their partial sums meet in LDS in a fixed order, wave 0 factors and hands X / the solution back).
// First form, one wave per 64 units (commit 7182169): 313 long waves at C3, 86 k cycles each (factor), of which 27 k
// waiting for its own 25 KB of staging -- a few hundred waves cannot keep enough loads in flight; four times the waves
// at a quarter of the loop length each.
// The unit state stays in the latent-major layout of the passes; a workgroup transposes its 64 units' rows through LDS
// ([unit][t], odd stride: conflict-free both ways), coalesced on the global side.
//
// Rank classes: the register arrays are sized by the template parameter R >= r (columns r .. R - 1 are identity
// padding); ranks above LANE_RMAX stay on the wave-per-task kernels (136 doubles at r = 16 do not fit the 256
// architectural registers).
#pragma once

// (-DLANE_RMAX_BUILD=16: the debug build of tools/lane_r16_fault.sh -- ranks 15, 16 compiled into these kernels, 1.5 KB of
// scratch per lane -- with which the fault recorded below esplit_lane_body was reproduced and explained; never shipped)
#ifndef LANE_RMAX_BUILD
#define LANE_RMAX_BUILD 14
#endif
constexpr int LANE_RMAX = LANE_RMAX_BUILD;
constexpr int LANE_EMAX = LANE_RMAX * (LANE_RMAX + 1) / 2;  // doubles of X per task in the hand-over buffer
constexpr int LANE_NW = 4;    // waves per workgroup
constexpr int LANE_CH = 32;   // entries of H per reduction round: 3 x 32 x 64 doubles = 48 KB of LDS
constexpr int LANE_UN = 16;   // rows per thread in the staging copies: 256 x 16 = 4096 >= 64 units x 64 bins

__device__ __forceinline__ constexpr int lt_idx(int i, int j) { return i * (i + 1) / 2 + j; }
__host__ __device__ constexpr int lane_red_doubles(bool factor) { return factor ? 3 * LANE_CH * 64 : 4 * LANE_RMAX * 64; }

// row t of the compact factor (T, r), wave-uniform address -> scalar loads; zero beyond r
// (read through the CONSTANT address space: the compiler then keeps the loads scalar whatever stores precede them --
// its no-clobber analysis gives up in a function of this size and falls back to one vector load per lane)
typedef const __attribute__((address_space(4))) double* lane_cptr;
template <int R>
__device__ __forceinline__ void lane_g_row(double (&g)[R], const double* __restrict__ Gl, int t, int r) {
    lane_cptr Gt = (lane_cptr)(Gl + t * r);
#pragma unroll
    for (int j = 0; j < R; ++j) g[j] = j < r ? Gt[j] : 0.0;
}

// LDS position of row i of a run of units of length T: [unit][t] with the odd stride TP = T | 1.
// u = i / T by a multiply-shift (inv = 2^20 / T + 1: exact for i < 4096, T <= 64).
__device__ __forceinline__ int lane_unit(int i, int inv) { return (int)(((unsigned)i * (unsigned)inv) >> 20); }
__device__ __forceinline__ int lane_pos(int i, int T, int inv) { return i + ((T & 1) ? 0 : lane_unit(i, inv)); }

// debug: cycle counters of the FIRST wave of a launch (vlgp_debug_phase_clock, VLGP_LANE_CLOCK=1 factor / 2 mean):
// factor: staging, build, reduction, Cholesky + inverse, hand-back, variance, stores
// mean:   staging, G's, reduction, solve, hand-back, expansion, update
struct LaneClock {
    unsigned long long* clk;
    unsigned long long tick;
    __device__ __forceinline__ LaneClock(const SplitArgs& A, int kind)
        : clk(blockIdx.x == 0 && threadIdx.x == 0 && A.clk_kind == kind ? A.clk : nullptr), tick(0) {
        if (clk) tick = __builtin_readcyclecounter();
    }
    __device__ __forceinline__ void lap(int slot) {
        if (clk) {
            const unsigned long long now = __builtin_readcyclecounter();
            atomicAdd(clk + slot, now - tick);
            tick = now;
        }
    }
};

struct LaneGroup {
    int l, T, TP, r, nu, nrows, m0, inv, wid, lane, t0, t1;
    int64_t r0;
    const double* __restrict__ Gl;
};

__device__ __forceinline__ LaneGroup lane_group(const SplitArgs& A, int li, int g) {
    LaneGroup K;
    K.l = A.lat[li];
    K.T = A.shg_T;
    K.TP = K.T | 1;
    K.inv = (1 << 20) / K.T + 1;
    K.r = A.shg_rk[li];
    K.m0 = 64 * g;
    K.nu = A.M - K.m0 < 64 ? A.M - K.m0 : 64;
    K.r0 = A.off[K.m0];
    K.nrows = K.nu * K.T;
    K.Gl = A.shg_gl[li];
    K.wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    K.lane = threadIdx.x & 63;
    K.t0 = (K.T * K.wid) / LANE_NW;
    K.t1 = (K.T * (K.wid + 1)) / LANE_NW;
    return K;
}

// Transposing copy of the workgroup's rows (contiguous in the latent-major global array) into LDS [unit][t]: every
// load of a thread in flight before the first use.  Indices are clamped instead of guarded (a guarded load is a branch
// with its own wait: the loads would pay their latencies one after the other -- measured 48 us per mean launch); the
// threads beyond the end rewrite the last row.
// WARM: between the requests and the first use, touch the 64-byte lines of the wave's rows of G through the scalar
// cache.  A row load that misses it is a trip to L2, ~600 cycles of which a loop body hides ~300 (measured per time
// bin: 300 cycles beyond the arithmetic in the build, 370 in the mean's loops; the loops cannot run more than one row
// ahead, a row is 2 r SGPRs); here the trips ride under the staging's own wait.
typedef const __attribute__((address_space(4))) int* lane_cptr_i;
constexpr int LANE_WARM = 24;  // lines: 13 rows x 14 doubles = 1456 bytes
template <bool WARM>
__device__ __forceinline__ void lane_stage_in(double* lds, const LaneGroup& K, const double* __restrict__ src) {
    const int tid = threadIdx.x;
    double tmp[LANE_UN];
#pragma unroll
    for (int k = 0; k < LANE_UN; ++k) {
        const int i = tid + 256 * k;
        tmp[k] = src[i < K.nrows ? i : K.nrows - 1];
    }
    __builtin_amdgcn_sched_barrier(0);
    int touch[WARM ? LANE_WARM : 1];
    if constexpr (WARM) {
        lane_cptr_i p = (lane_cptr_i)(K.Gl + K.t0 * K.r);
        const int nl = ((K.t1 - K.t0) * K.r * 8 + 63) / 64;
#pragma unroll
        for (int k = 0; k < LANE_WARM; ++k) touch[k] = p[(k < nl ? k : 0) * 16];
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int k = 0; k < LANE_UN; ++k) {
        const int i = tid + 256 * k;
        lds[lane_pos(i < K.nrows ? i : K.nrows - 1, K.T, K.inv)] = tmp[k];
    }
    if constexpr (WARM) {
        __builtin_amdgcn_sched_barrier(0);
        int acc = 0;
#pragma unroll
        for (int k = 0; k < LANE_WARM; ++k) acc += touch[k];
        asm volatile("" ::"s"(acc));
    }
}

// the rows of the units beyond the end of the set (last group): zeros
__device__ __forceinline__ void lane_zero_tail(double* lds, const LaneGroup& K) {
    if (K.lane >= K.nu)
        for (int t = K.t0; t < K.t1; ++t) lds[K.lane * K.TP + t] = 0.0;
}

// Sum over the four waves of R per-lane values, fixed order ((p0 + p1) + p2) + p3, result in every wave.
// `red`: 4 R 64 doubles.  Two barriers (the second frees `red` for the next use).
template <int R>
__device__ __forceinline__ void lane_allsum(double (&c)[R], double* red, const LaneGroup& K) {
#pragma unroll
    for (int j = 0; j < R; ++j) red[(K.wid * R + j) * 64 + K.lane] = c[j];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < R; ++j)
        c[j] = ((red[(0 * R + j) * 64 + K.lane] + red[(1 * R + j) * 64 + K.lane]) + red[(2 * R + j) * 64 + K.lane]) +
               red[(3 * R + j) * 64 + K.lane];
    __syncthreads();
}

// sol = X'(X c), the rows of X dealt to the four waves: wave q takes rows q, q + 4, ... (row slot k = row q + 4 k, read
// as 4 k + 4 entries whatever q: the code is the same for every wave, entries beyond the diagonal are zeroed), computes
// z_i = X_i. c for its rows and its share sum_i X_ij z_i of the solution; lane_allsum finishes it.  One wave reading
// all of X was 66 .. 105 loads behind one another (more than the 63 a wave can have in flight); here <= 40 per wave.
template <int R>
struct LaneXRows {
    static constexpr int NK = (R + 3) / 4;
    double x[NK][4 * NK];
    __device__ __forceinline__ void request(const double* __restrict__ xs, int q) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i = q + 4 * k;
            const int ic = i < R ? i : R - 1;
            const int base = ic * (ic + 1) / 2;
#pragma unroll
            for (int j = 0; j < 4 * k + 4; ++j)
                if (j < R) x[k][j] = xs[(base + (j <= ic ? j : ic)) * 64];
        }
    }
    __device__ __forceinline__ void solve(const double (&c)[R], double (&sol)[R], int q) {
#pragma unroll
        for (int j = 0; j < R; ++j) sol[j] = 0.0;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int i = q + 4 * k;
            double xr[4 * NK];
#pragma unroll
            for (int j = 0; j < 4 * k + 4; ++j)
                if (j < R) xr[j] = (i < R && j <= i) ? x[k][j] : 0.0;
            double z = xr[0] * c[0];
#pragma unroll
            for (int j = 1; j < 4 * k + 4; ++j)
                if (j < R) z = fma(xr[j], c[j], z);
#pragma unroll
            for (int j = 0; j < 4 * k + 4; ++j)
                if (j < R) sol[j] = fma(xr[j], z, sol[j]);
        }
    }
};

// Loop over the wave's time bins with row t + 1 of G (scalar loads) and one per-lane LDS value in flight while row t is
// consumed.  `head` makes the FIRST use of the row (the compiler's wait for the scalar loads -- they return out of
// order, so it is always "everything outstanding" -- lands there), then the next row is requested, then `tail` runs:
// with the request in front of the first use every trip would wait for the loads it has just issued (measured: ~180
// cycles per time bin on top of the arithmetic).
template <int R, class Head, class Tail>
__device__ __forceinline__ void lane_t_loop(const LaneGroup& K, const double* col, Head head, Tail tail) {
    if (K.t0 >= K.t1) return;
    double ga[R], gb[R];
    lane_g_row<R>(ga, K.Gl, K.t0, K.r);
    double xa = col[K.t0], xb = 0.0;
    for (int t = K.t0; t < K.t1; t += 2) {
        {
            auto hd = head(ga, xa, t);
            __builtin_amdgcn_sched_barrier(0);
            const int tn = t + 1 < K.t1 ? t + 1 : t;
            lane_g_row<R>(gb, K.Gl, tn, K.r);
            xb = col[tn];
            __builtin_amdgcn_sched_barrier(0);
            tail(ga, xa, t, hd);
        }
        if (t + 1 < K.t1) {
            auto hd = head(gb, xb, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            const int tn = t + 2 < K.t1 ? t + 2 : t + 1;
            lane_g_row<R>(ga, K.Gl, tn, K.r);
            xa = col[tn];
            __builtin_amdgcn_sched_barrier(0);
            tail(gb, xb, t + 1, hd);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// factor + variance of 64 units of one latent.
// LDS: 64 TP doubles (w in, v out) | 64 flags | 3 LANE_CH 64 (partial sums of H, then X on its way back)
template <int R>
__device__ __forceinline__ void lane_factor(const SplitArgs& A, int li, int g, double* lds) {
    constexpr int E = R * (R + 1) / 2;
    constexpr int NCH = (E + LANE_CH - 1) / LANE_CH;
    const LaneGroup K = lane_group(A, li, g);
    const int lane = K.lane, T = K.T, TP = K.TP;
    const double* __restrict__ w_s = A.w + (int64_t)K.l * A.ld + K.r0;
    double* v_s = A.v + (int64_t)K.l * A.ld + K.r0;
    double* flags = lds + 64 * TP;
    double* red = flags + 64;
    LaneClock ck(A, 1);
    lane_zero_tail(lds, K);
    lane_stage_in<true>(lds, K, w_s);
    __syncthreads();
    ck.lap(0);
    double h[E];
#pragma unroll
    for (int e = 0; e < E; ++e) h[e] = 0.0;
    double* wl = lds + lane * TP;
    lane_t_loop<R>(
        K, wl, [&](const double (&gr)[R], double wt, int) { return gr[0] * wt; },
        [&](const double (&gr)[R], double wt, int, double gw0) {
            h[0] = fma(gw0, gr[0], h[0]);
#pragma unroll
            for (int i = 1; i < R; ++i) {
                const double gw = gr[i] * wt;
#pragma unroll
                for (int j = 0; j <= i; ++j) h[lt_idx(i, j)] = fma(gw, gr[j], h[lt_idx(i, j)]);
            }
        });
    ck.lap(1);
    // partial sums -> wave 0, fixed order ((p0 + p1) + p2) + p3.  (Measured alternative: every wave reads all four
    // partial sums and factors the same matrix, no hand-back of X -- 16 E instead of 10 E LDS accesses of 512 bytes per
    // workgroup: reduction 10.9 k cycles against 5.0 k + 2.6 k for the hand-back at rank 11; the LDS port is the limit.)
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (K.wid > 0) {
#pragma unroll
            for (int k = 0; k < LANE_CH; ++k)
                if (c * LANE_CH + k < E) red[((K.wid - 1) * LANE_CH + k) * 64 + lane] = h[c * LANE_CH + k];
        }
        __syncthreads();
        if (K.wid == 0) {
#pragma unroll
            for (int k = 0; k < LANE_CH; ++k)
                if (c * LANE_CH + k < E) {
                    const int e = c * LANE_CH + k;
                    h[e] = ((h[e] + red[(0 * LANE_CH + k) * 64 + lane]) + red[(1 * LANE_CH + k) * 64 + lane]) +
                           red[(2 * LANE_CH + k) * 64 + lane];
                }
        }
        __syncthreads();
    }
    ck.lap(2);
    bool ok = true;
    if (K.wid == 0) {
#pragma unroll
        for (int i = 0; i < R; ++i) h[lt_idx(i, i)] += 1.0;
        // Cholesky, left-looking by column; the diagonal keeps 1 / L_jj
#pragma unroll
        for (int j = 0; j < R; ++j) {
            double d = h[lt_idx(j, j)];
#pragma unroll
            for (int k = 0; k < j; ++k) d = fma(-h[lt_idx(j, k)], h[lt_idx(j, k)], d);
            if (!(d > 0.0) || !(d < 1e300)) ok = false;
            double y = __builtin_amdgcn_rsq(d);
            double e = fma(-d * y, y, 1.0);
            y = fma(y * 0.5, e, y);
            e = fma(-d * y, y, 1.0);
            y = fma(y * 0.5, e, y);
            h[lt_idx(j, j)] = y;
#pragma unroll
            for (int i = j + 1; i < R; ++i) {
                double s = h[lt_idx(i, j)];
#pragma unroll
                for (int k = 0; k < j; ++k) s = fma(-h[lt_idx(i, k)], h[lt_idx(j, k)], s);
                h[lt_idx(i, j)] = s * y;
            }
        }
        // X = L^-1 in place, column by column: X_ij = -(1 / L_ii) sum_{k = j}^{i - 1} L_ik X_kj
#pragma unroll
        for (int j = 0; j < R; ++j) {
#pragma unroll
            for (int i = j + 1; i < R; ++i) {
                double s = h[lt_idx(i, j)] * h[lt_idx(j, j)];
#pragma unroll
                for (int k = j + 1; k < i; ++k) s = fma(h[lt_idx(i, k)], h[lt_idx(k, j)], s);
                h[lt_idx(i, j)] = -s * h[lt_idx(i, i)];
            }
        }
        flags[lane] = ok ? 1.0 : 0.0;
    }
    ck.lap(3);
    // X back to the other waves
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (K.wid == 0) {
#pragma unroll
            for (int k = 0; k < LANE_CH; ++k)
                if (c * LANE_CH + k < E) red[k * 64 + lane] = h[c * LANE_CH + k];
        }
        __syncthreads();
        if (K.wid > 0) {
#pragma unroll
            for (int k = 0; k < LANE_CH; ++k)
                if (c * LANE_CH + k < E) h[c * LANE_CH + k] = red[k * 64 + lane];
        }
        __syncthreads();
    }
    ck.lap(4);
    // (every global store comes after the last read of G)
    if (A.do_v) {
        lane_t_loop<R>(
            K, wl, [&](const double (&gr)[R], double, int) { return h[0] * gr[0]; },
            [&](const double (&gr)[R], double, int t, double z0) {
                double vv = z0 * z0;
#pragma unroll
                for (int i = 1; i < R; ++i) {
                    double z = h[lt_idx(i, 0)] * gr[0];
#pragma unroll
                    for (int j = 1; j <= i; ++j) z = fma(h[lt_idx(i, j)], gr[j], z);
                    vv = fma(z, z, vv);
                }
                wl[t] = vv;
            });
    }
    __syncthreads();
    ck.lap(5);
    // hand-over, entry-major: every wave holds X, each stores a quarter of the entries (one straight-line variant per
    // wave: a test per entry compiles into a taken branch per store)
    {
        double* xd = A.xl + ((int64_t)g * A.L + K.l) * (64 * LANE_EMAX) + lane;
        auto quarter = [&](auto Q) {
#pragma unroll
            for (int e = 0; e < E; ++e)
                if ((e & (LANE_NW - 1)) == decltype(Q)::value) xd[e * 64] = h[e];
        };
        if (K.wid == 0) quarter(std::integral_constant<int, 0>{});
        else if (K.wid == 1) quarter(std::integral_constant<int, 1>{});
        else if (K.wid == 2) quarter(std::integral_constant<int, 2>{});
        else quarter(std::integral_constant<int, 3>{});
    }
    if (K.wid == 0 && lane < K.nu) {
        A.failg[(K.m0 + lane) * A.L + K.l] = ok ? 0 : 1;
        if (!ok) atomicAdd(A.fail, 1);
    }
    __builtin_amdgcn_sched_barrier(0);  // (X is dead from here: its registers are the batch below)
    if (A.do_v) {  // (every LDS read in front of the first store: one wait instead of one per row)
        double vo[LANE_UN], fo[LANE_UN];
#pragma unroll
        for (int k = 0; k < LANE_UN; ++k) {
            const int i = threadIdx.x + 256 * k;
            const int ic = i < K.nrows ? i : K.nrows - 1;
            vo[k] = lds[lane_pos(ic, T, K.inv)];
            fo[k] = flags[lane_unit(ic, K.inv)];
        }
#pragma unroll
        for (int k = 0; k < LANE_UN; ++k) {
            const int i = threadIdx.x + 256 * k;
            if (i < K.nrows && fo[k] != 0.0) v_s[i] = vo[k];
        }
    }
    ck.lap(6);
}

// ---------------------------------------------------------------------------------------------------------
// Newton step on the posterior mean of 64 units of one latent (vlgp/core.py:80-95), push-through form:
//     delta = G (I + H)^-1 G'(ra + W mu) - mu.
// LAST (last sweep of a call): the step handed back as `dmu` comes from u = G G'ra - mu,
// delta = u - G (I + H)^-1 G'W u first (mean_task_last: u vanishes at the fixed point); mu then advances as in every
// other sweep.
// The regular sweeps read s = ra + W mu as the residual pass left it (A.sv) and write the raw target dl = G (I + H)^-1 G's;
// the curvature pass that follows applies mu += clip(dl - mu) (esplit_pass<SP_W>, A.dmask): this launch then stages one
// array instead of three and does not touch mu (measured before: staging 11.7 k of the launch's 27.8 k cycles at rank
// 11, update 4.0 k).  A failed factor (core.py:92-94: zero update) hands back dl = mu.
// LDS: s [64 TP] | flags [64] | partial sums [4 LANE_RMAX 64]; LAST: + mu, w, ra / u [64 TP each]
template <int R, bool LAST>
__device__ __forceinline__ void lane_mean(const SplitArgs& A, int li, int g, double* lds) {
    const LaneGroup K = lane_group(A, li, g);
    const int lane = K.lane, T = K.T, TP = K.TP, L = A.L, tid = threadIdx.x;
    const double* __restrict__ w_s = A.w + (int64_t)K.l * A.ld + K.r0;
    const double* __restrict__ ra_s = A.ra + (int64_t)K.l * A.ld + K.r0;
    const double* __restrict__ sv_s = A.sv + (int64_t)K.l * A.ld + K.r0;
    const double* mu_s = A.mu + (int64_t)K.l * A.ld + K.r0;
    double* dl_s = A.dl + (int64_t)K.l * A.ld + K.r0;
    double* flags = lds + 64 * TP;  // 1 = factor failed: zero update (core.py:92-94)
    double* red = flags + 64;
    LaneClock ck(A, 2);
    // every wave's rows of X are requested as soon as the staging registers are free and land while the first loop runs
    LaneXRows<R> X;
    int failed = 0;
    auto request_x = [&]() {
        X.request(A.xl + ((int64_t)g * A.L + K.l) * (64 * LANE_EMAX) + lane, K.wid);
    };
    if (K.wid == 0 && lane < K.nu) failed = A.failg[(K.m0 + lane) * L + K.l];
    lane_zero_tail(lds, K);
    double* sl = lds + lane * TP;
    double c[R], sol[R];
    if constexpr (LAST) {
        double* mbuf = red + 4 * LANE_RMAX * 64;
        double* bufw = mbuf + 64 * TP;
        double* bufu = bufw + 64 * TP;
        lane_zero_tail(bufw, K);
        lane_zero_tail(bufu, K);
        lane_zero_tail(mbuf, K);
        lane_stage_in<true>(bufu, K, ra_s);
        lane_stage_in<false>(bufw, K, w_s);
        lane_stage_in<false>(mbuf, K, mu_s);
        if (K.wid == 0) flags[lane] = failed ? 1.0 : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        request_x();
        __syncthreads();
        double* ul = bufu + lane * TP;
        double* wl = bufw + lane * TP;
        double* ml = mbuf + lane * TP;
#pragma unroll
        for (int j = 0; j < R; ++j) c[j] = 0.0;
        lane_t_loop<R>(
            K, ul, [&](const double (&gr)[R], double rt, int) { return gr[0] * rt; },
            [&](const double (&gr)[R], double rt, int, double p0) {  // g1 = G'ra
                c[0] += p0;
#pragma unroll
                for (int j = 1; j < R; ++j) c[j] = fma(gr[j], rt, c[j]);
            });
        lane_allsum<R>(c, red, K);
        double rhs[R];
#pragma unroll
        for (int j = 0; j < R; ++j) rhs[j] = 0.0;
        lane_t_loop<R>(
            K, ul, [&](const double (&gr)[R], double, int) { return gr[0] * c[0]; },
            [&](const double (&gr)[R], double rt, int t, double p0) {  // u = G g1 - mu, rhs = (W G)'u, s = ra + w mu
                double s0 = p0, s1 = 0.0;
#pragma unroll
                for (int j = 1; j < R; ++j) {
                    if (j & 1) s1 = fma(gr[j], c[j], s1);
                    else s0 = fma(gr[j], c[j], s0);
                }
                const double mt = ml[t], wt = wl[t];
                const double ut = (s0 + s1) - mt;
                ul[t] = ut;
                sl[t] = fma(wt, mt, rt);
                const double wu = wt * ut;
#pragma unroll
                for (int j = 0; j < R; ++j) rhs[j] = fma(gr[j], wu, rhs[j]);
            });
        lane_allsum<R>(rhs, red, K);
        X.solve(rhs, sol, K.wid);
        lane_allsum<R>(sol, red, K);
        const bool fl = flags[lane] != 0.0;
        lane_t_loop<R>(
            K, ul, [&](const double (&gr)[R], double ut, int) { return fma(-gr[0], sol[0], ut); },
            [&](const double (&gr)[R], double, int t, double p0) {
                double s0 = p0, s1 = 0.0;
#pragma unroll
                for (int j = 1; j < R; ++j) {
                    if (j & 1) s1 = fma(-gr[j], sol[j], s1);
                    else s0 = fma(-gr[j], sol[j], s0);
                }
                double s = s0 + s1;
                s = fmin(fmax(s, -A.dmu_bound), A.dmu_bound);
                wl[t] = fl ? 0.0 : s;
            });
        __syncthreads();
#pragma unroll 4
        for (int i = tid; i < K.nrows; i += 256) A.dmu[(K.r0 + i) * L + K.l] = bufw[lane_pos(i, T, K.inv)];
    } else {
        lane_stage_in<true>(lds, K, sv_s);
        if (K.wid == 0) flags[lane] = failed ? 1.0 : 0.0;
        __builtin_amdgcn_sched_barrier(0);
        request_x();
        __syncthreads();
    }
    ck.lap(0);
#pragma unroll
    for (int j = 0; j < R; ++j) c[j] = 0.0;
    lane_t_loop<R>(
        K, sl, [&](const double (&gr)[R], double st, int) { return gr[0] * st; },
        [&](const double (&gr)[R], double st, int, double p0) {
            c[0] += p0;
#pragma unroll
            for (int j = 1; j < R; ++j) c[j] = fma(gr[j], st, c[j]);
        });
    ck.lap(1);
    lane_allsum<R>(c, red, K);
    ck.lap(2);
    X.solve(c, sol, K.wid);
    ck.lap(3);
    lane_allsum<R>(sol, red, K);
    ck.lap(4);
    lane_t_loop<R>(
        K, sl, [&](const double (&gr)[R], double, int) { return gr[0] * sol[0]; },
        [&](const double (&gr)[R], double, int t, double p0) {
            double s0 = p0, s1 = 0.0;
#pragma unroll
            for (int j = 1; j < R; ++j) {
                if (j & 1) s1 = fma(gr[j], sol[j], s1);
                else s0 = fma(gr[j], sol[j], s0);
            }
            sl[t] = s0 + s1;
        });
    __syncthreads();
    ck.lap(5);
    {
        double dv[LANE_UN], fo[LANE_UN];
#pragma unroll
        for (int k = 0; k < LANE_UN; ++k) {
            const int i = tid + 256 * k;
            const int ic = i < K.nrows ? i : K.nrows - 1;
            dv[k] = lds[lane_pos(ic, T, K.inv)];
            fo[k] = flags[lane_unit(ic, K.inv)];
        }
        // (clamped, unconditional stores: the threads beyond the end rewrite the last row with its own value; a failed
        // unit -- rare -- is patched afterwards, same thread, same address)
        bool anyf = false;
#pragma unroll
        for (int k = 0; k < LANE_UN; ++k) {
            const int i = tid + 256 * k;
            dl_s[i < K.nrows ? i : K.nrows - 1] = dv[k];
            anyf |= fo[k] != 0.0;
        }
        if (anyf) {
            for (int k = 0; k < LANE_UN; ++k) {
                const int i = tid + 256 * k;
                const int ic = i < K.nrows ? i : K.nrows - 1;
                if (flags[lane_unit(ic, K.inv)] != 0.0) dl_s[ic] = mu_s[ic];
            }
        }
    }
    if (failed) atomicAdd(A.fail, 1);
    ck.lap(6);
}

// LDS doubles of a launch (host)
inline size_t lane_lds_doubles(int T, bool mean, bool last) {
    const size_t tp = (size_t)(T | 1);
    if (!mean) return 64 * tp + 64 + lane_red_doubles(true);
    return (last ? 4 : 1) * 64 * tp + 64 + lane_red_doubles(false);
}

// KIND 0: factor (+ variance), 1: mean, 2: mean of the last sweep.  One workgroup per (latent, group of 64 units).
// Two workgroups per CU (313 workgroups at C3 on 256 CUs); rank 14 then spills ~50 registers.  Ranks 15, 16 stay on the
// wave-per-task kernels.  (Measured alternatives: ranks 13 .. 16 as their own launch compiled for one workgroup per CU,
// the overflow in the accumulation registers -- the second launch is a second dependent step of every sweep: E-step 2.1 ->
// 3.4 ms as soon as one latent reaches rank 13, tools/estep_per_step.py; ranks 15, 16 in this kernel: 1.5 KB of scratch
// per lane, and a memory fault at launch on the second stream.)
// RTOP: the largest rank compiled in (13 or LANE_RMAX = 14).  The rank-14 case spills (its 105-double triangle plus the
// loop state exceed 256 registers) and a kernel's scratch frame is the maximum over its rank switch: launches whose
// latents all sit at rank <= 13 -- most of a fit -- take the RTOP = 13 instantiation, which has none.
template <int KIND, int RTOP = LANE_RMAX>
__device__ __forceinline__ void esplit_lane_body(const SplitArgs& A, double* smem, int bid) {
    const int li = bid % A.n_lat, g = bid / A.n_lat;
    const int r = A.shg_rk[li];
    // (s_setprio: measured 5 % slower at level 3 than at 0 with the other lane's row passes beside)
    if (A.prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (A.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (A.prio == 1) __builtin_amdgcn_s_setprio(1);
#define LANE_CASE(RV)                                            \
    do {                                                         \
        if constexpr (KIND == 0) lane_factor<RV>(A, li, g, smem); \
        else lane_mean<RV, KIND == 2>(A, li, g, smem);           \
    } while (0)
#ifdef LANE_ONLY  // (debug builds: one instantiation, to read its ISA)
    LANE_CASE(LANE_ONLY);
#else
    if (r <= 8) {
        if (r <= 4) LANE_CASE(4);
        else if (r <= 6) LANE_CASE(6);
        else LANE_CASE(8);
    } else if (r <= 12) {
        if (r == 9) LANE_CASE(9);
        else if (r == 10) LANE_CASE(10);
        else if (r == 11) LANE_CASE(11);
        else LANE_CASE(12);
    } else {
        if constexpr (RTOP >= 16) {
            if (r == 13) LANE_CASE(13);
            else if (r == 14) LANE_CASE(14);
            else if (r == 15) LANE_CASE(15);
            else LANE_CASE(16);
        } else if constexpr (RTOP >= 14) {
            if (r == 13) LANE_CASE(13);
            else LANE_CASE(14);
        } else {
            LANE_CASE(13);
        }
    }
#endif
#undef LANE_CASE
}

// (the instantiation with the rank-14 case at ONE workgroup per CU: its overflow then sits in the accumulation registers --
// 45 / 15 / 82 of them -- instead of 148 / 64 / 408 bytes of scratch per lane at two; same box, one latent at rank 14 among
// rank-11 ones: E-step 2.36 -> 2.19 ms, tools/variant_ab.sh)
#ifndef LANE_LB14
#define LANE_LB14 1
#endif
template <int KIND, int RTOP = LANE_RMAX>
__global__ void __launch_bounds__(256, (RTOP >= 14 || KIND == 2 ? LANE_LB14 : 2)) esplit_lane(SplitArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    esplit_lane_body<KIND, RTOP>(A, smem, blockIdx.x);
}
