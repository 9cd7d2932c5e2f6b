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
// The unit state stays in the latent-major layout of the passes; a wave transposes its 64 units' rows through LDS
// ([unit][t], odd stride: conflict-free both ways), coalesced on the global side.
//
// Rank classes: the register arrays are sized by the template parameter R >= r (columns r .. R - 1 are identity
// padding); ranks above LANE_RMAX = 14 stay on the wave-per-task kernels (136 doubles at r = 16 do not fit the 256
// architectural registers).
#pragma once

constexpr int LANE_RMAX = 14;
constexpr int LANE_EMAX = LANE_RMAX * (LANE_RMAX + 1) / 2;  // doubles of X per task in the hand-over buffer

__device__ __forceinline__ constexpr int lt_idx(int i, int j) { return i * (i + 1) / 2 + j; }

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
__device__ __forceinline__ int lane_pos(int i, int T, int inv) {
    const int u = (int)(((unsigned)i * (unsigned)inv) >> 20);
    return i + ((T & 1) ? 0 : u);
}
__device__ __forceinline__ int lane_unit(int i, int inv) { return (int)(((unsigned)i * (unsigned)inv) >> 20); }

// Transposing copies between the latent-major global arrays (rows of the wave's 64 units: contiguous) and LDS, UN
// rows per lane in flight: a loop of one load -> one LDS store per trip pays the global latency once PER ROW (measured:
// 74 us per mean launch of 157 waves, 50 dependent trips per wave).
template <int UN>
__device__ __forceinline__ void lane_stage_in(double* lds, int nrows, int T, int inv, int lane,
                                              const double* __restrict__ src) {
    for (int b = 0; b < nrows; b += 64 * UN) {
        // (indices clamped instead of guarded: a guarded load is a branch with its own wait, and the UN loads of a
        // batch then pay their latencies one after the other; the lanes beyond the end rewrite the last row.  The
        // scheduling barrier keeps every load of the batch in front of the first use.)
        double tmp[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const int i = b + 64 * k + lane;
            tmp[k] = src[i < nrows ? i : nrows - 1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const int i = b + 64 * k + lane;
            lds[lane_pos(i < nrows ? i : nrows - 1, T, inv)] = tmp[k];
        }
    }
}
// s = ra + w mu
template <int UN>
__device__ __forceinline__ void lane_stage_s(double* lds, int nrows, int T, int inv, int lane,
                                             const double* __restrict__ w, const double* mu,
                                             const double* __restrict__ ra) {
    for (int b = 0; b < nrows; b += 64 * UN) {
        double tw[UN], tm[UN], tr[UN];
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const int i = b + 64 * k + lane;
            const int ic = i < nrows ? i : nrows - 1;
            tw[k] = w[ic];
            tm[k] = mu[ic];
            tr[k] = ra[ic];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < UN; ++k) {
            const int i = b + 64 * k + lane;
            lds[lane_pos(i < nrows ? i : nrows - 1, T, inv)] = fma(tw[k], tm[k], tr[k]);
        }
    }
}

// Touch every 64-byte line of the wave's factor G through the SCALAR cache before the loops over its rows: a row load
// that misses costs a trip to L2 (~1100 cycles measured: 100 row loads per wave = 48 us per launch), and the loops
// cannot run more than one row ahead (a row is 2 r SGPRs).  Twelve lines in flight per trip.
typedef const __attribute__((address_space(4))) int* lane_cptr_i;
__device__ __forceinline__ void lane_warm_scalar_cache(const double* Gl, int doubles) {
    lane_cptr_i p = (lane_cptr_i)Gl;
    const int nl = (doubles * 8 + 63) / 64;
    int acc = 0;
    for (int b = 0; b < nl; b += 12) {
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int line = b + k < nl ? b + k : nl - 1;
            acc += p[line * 16];
        }
    }
    asm volatile("" ::"s"(acc));
}

// debug: cycle counters of the FIRST wave of a launch (vlgp_debug_phase_clock): factor slots 0 .. 3 (staging, build,
// Cholesky + inverse, variance + stores), mean slots 4 .. 7 (staging, G's, solve, expansion + update)
struct LaneClock {
    unsigned long long* clk;
    unsigned long long tick;
    __device__ __forceinline__ LaneClock(const SplitArgs& A) : clk(blockIdx.x == 0 && threadIdx.x == 0 ? A.clk : nullptr), tick(0) {
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
    int l, T, TP, r, nu, nrows, m0, inv;
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
    return K;
}

// ---------------------------------------------------------------------------------------------------------
// factor + variance of 64 units of one latent.  LDS: 64 TP doubles (w in, v out) + 64 flags.
template <int R>
__device__ __forceinline__ void lane_factor(const SplitArgs& A, int li, int g, double* lds) {
    constexpr int E = R * (R + 1) / 2;
    const int lane = threadIdx.x;
    const LaneGroup K = lane_group(A, li, g);
    const int T = K.T, TP = K.TP, r = K.r;
    const double* __restrict__ Gl = K.Gl;
    const double* __restrict__ w_s = A.w + (int64_t)K.l * A.ld + K.r0;
    double* v_s = A.v + (int64_t)K.l * A.ld + K.r0;
    double* flags = lds + 64 * TP;
    LaneClock ck(A);
    if (lane >= K.nu)
        for (int t = 0; t < T; ++t) lds[lane * TP + t] = 0.0;
    if (A.warm) lane_warm_scalar_cache(Gl, T * r);
    lane_stage_in<32>(lds, K.nrows, T, K.inv, lane, w_s);
    tri_wave_sync();
    ck.lap(0);
    double h[E];
#pragma unroll
    for (int e = 0; e < E; ++e) h[e] = 0.0;
    double* wl = lds + lane * TP;
    {
        double ga[R], gb[R];
        auto body = [&](const double (&gr)[R], int t) {
            const double wt = wl[t];
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const double gw = gr[i] * wt;
#pragma unroll
                for (int j = 0; j <= i; ++j) h[lt_idx(i, j)] = fma(gw, gr[j], h[lt_idx(i, j)]);
            }
        };
        lane_g_row<R>(ga, Gl, 0, r);
        int t = 0;
        for (; t + 1 < T; t += 2) {
            lane_g_row<R>(gb, Gl, t + 1, r);
            body(ga, t);
            lane_g_row<R>(ga, Gl, t + 2 < T ? t + 2 : t + 1, r);
            body(gb, t + 1);
        }
        if (t < T) body(ga, t);
    }
    ck.lap(1);
#pragma unroll
    for (int i = 0; i < R; ++i) h[lt_idx(i, i)] += 1.0;
    // Cholesky, left-looking by column; the diagonal keeps 1 / L_jj
    bool ok = true;
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
    ck.lap(2);
    // (every global store comes after the last read of G: a store in between would turn the wave-uniform scalar
    // loads of its rows into vector loads)
    flags[lane] = ok ? 1.0 : 0.0;
    if (A.do_v) {
        double ga[R], gb[R];
        auto body = [&](const double (&gr)[R], int t) {
            double vv = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double z = h[lt_idx(i, 0)] * gr[0];
#pragma unroll
                for (int j = 1; j <= i; ++j) z = fma(h[lt_idx(i, j)], gr[j], z);
                vv = fma(z, z, vv);
            }
            wl[t] = vv;
        };
        lane_g_row<R>(ga, Gl, 0, r);
        int t = 0;
        for (; t + 1 < T; t += 2) {
            lane_g_row<R>(gb, Gl, t + 1, r);
            body(ga, t);
            lane_g_row<R>(ga, Gl, t + 2 < T ? t + 2 : t + 1, r);
            body(gb, t + 1);
        }
        if (t < T) body(ga, t);
    }
    tri_wave_sync();
    // hand-over, entry-major
    {
        double* xd = A.xl + ((int64_t)g * A.L + K.l) * (64 * LANE_EMAX) + lane;
#pragma unroll
        for (int e = 0; e < E; ++e) xd[e * 64] = h[e];
    }
    if (lane < K.nu) {
        A.failg[(K.m0 + lane) * A.L + K.l] = ok ? 0 : 1;
        if (!ok) atomicAdd(A.fail, 1);
    }
    if (A.do_v) {
#pragma unroll 4
        for (int i = lane; i < K.nrows; i += 64)
            if (flags[lane_unit(i, K.inv)] != 0.0) v_s[i] = lds[lane_pos(i, T, K.inv)];
    }
    ck.lap(3);
}

// z = X c, sol = X'z with the rows of X streamed from the hand-over buffer
template <int R>
__device__ __forceinline__ void lane_solve(const double* __restrict__ xs, const double (&c)[R], double (&sol)[R]) {
#pragma unroll
    for (int j = 0; j < R; ++j) sol[j] = 0.0;
#pragma unroll
    for (int i = 0; i < R; ++i) {
        double xr[R];
#pragma unroll
        for (int j = 0; j <= i; ++j) xr[j] = xs[lt_idx(i, j) * 64];
        double z = xr[0] * c[0];
#pragma unroll
        for (int j = 1; j <= i; ++j) z = fma(xr[j], c[j], z);
#pragma unroll
        for (int j = 0; j <= i; ++j) sol[j] = fma(xr[j], z, sol[j]);
    }
}

// ---------------------------------------------------------------------------------------------------------
// Newton step on the posterior mean of 64 units of one latent (vlgp/core.py:80-95), push-through form:
//     delta = G (I + H)^-1 G'(ra + W mu) - mu.
// LAST (last sweep of a call): the step handed back as `dmu` comes from u = G G'ra - mu,
// delta = u - G (I + H)^-1 G'W u first (mean_task_last: u vanishes at the fixed point); mu then advances as in every
// other sweep.  LDS: 64 TP doubles + 64 flags; LAST: 3 x 64 TP + 64.
template <int R, bool LAST>
__device__ __forceinline__ void lane_mean(const SplitArgs& A, int li, int g, double* lds) {
    const int lane = threadIdx.x;
    const LaneGroup K = lane_group(A, li, g);
    const int T = K.T, TP = K.TP, r = K.r, L = A.L;
    const double* __restrict__ Gl = K.Gl;
    const double* __restrict__ w_s = A.w + (int64_t)K.l * A.ld + K.r0;
    const double* __restrict__ ra_s = A.ra + (int64_t)K.l * A.ld + K.r0;
    double* mu_s = A.mu + (int64_t)K.l * A.ld + K.r0;
    double* flags = lds + 64 * TP;  // 1 = factor failed: zero update (core.py:92-94)
    const double* __restrict__ xs = A.xl + ((int64_t)g * A.L + K.l) * (64 * LANE_EMAX) + lane;
    int failed = 0;
    if (lane < K.nu) failed = A.failg[(K.m0 + lane) * L + K.l];
    flags[lane] = failed ? 1.0 : 0.0;
    double* sl = lds + lane * TP;
    LaneClock ck(A);
    if (A.warm) lane_warm_scalar_cache(Gl, T * r);
    if constexpr (LAST) {
        double* bufw = flags + 64;
        double* bufm = bufw + 64 * TP;
        if (lane >= K.nu)
            for (int t = 0; t < T; ++t) {
                lds[lane * TP + t] = 0.0;
                bufw[lane * TP + t] = 0.0;
                bufm[lane * TP + t] = 0.0;
            }
        lane_stage_in<16>(lds, K.nrows, T, K.inv, lane, ra_s);
        lane_stage_in<16>(bufw, K.nrows, T, K.inv, lane, w_s);
        lane_stage_in<16>(bufm, K.nrows, T, K.inv, lane, mu_s);
        tri_wave_sync();
        double* wl = bufw + lane * TP;
        double* ml = bufm + lane * TP;
        double g1[R], rhs[R], sol[R];
#pragma unroll
        for (int j = 0; j < R; ++j) g1[j] = rhs[j] = 0.0;
        for (int t = 0; t < T; ++t) {  // g1 = G'ra
            double gr[R];
            lane_g_row<R>(gr, Gl, t, r);
            const double st = sl[t];
#pragma unroll
            for (int j = 0; j < R; ++j) g1[j] = fma(gr[j], st, g1[j]);
        }
        for (int t = 0; t < T; ++t) {  // u = G g1 - mu, rhs = (W G)'u
            double gr[R];
            lane_g_row<R>(gr, Gl, t, r);
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int j = 0; j < R; j += 2) {
                s0 = fma(gr[j], g1[j], s0);
                if (j + 1 < R) s1 = fma(gr[j + 1], g1[j + 1], s1);
            }
            const double ut = (s0 + s1) - ml[t];
            sl[t] = ut;
            const double wu = wl[t] * ut;
#pragma unroll
            for (int j = 0; j < R; ++j) rhs[j] = fma(gr[j], wu, rhs[j]);
        }
        lane_solve<R>(xs, rhs, sol);
        for (int t = 0; t < T; ++t) {
            double gr[R];
            lane_g_row<R>(gr, Gl, t, r);
            double s0 = sl[t], s1 = 0.0;
#pragma unroll
            for (int j = 0; j < R; j += 2) {
                s0 = fma(-gr[j], sol[j], s0);
                if (j + 1 < R) s1 = fma(-gr[j + 1], sol[j + 1], s1);
            }
            double s = s0 + s1;
            s = fmin(fmax(s, -A.dmu_bound), A.dmu_bound);
            sl[t] = failed ? 0.0 : s;
        }
        tri_wave_sync();
#pragma unroll 4
        for (int i = lane; i < K.nrows; i += 64) A.dmu[(K.r0 + i) * L + K.l] = lds[lane_pos(i, T, K.inv)];
        tri_wave_sync();
        // the regular update from the staged copies: s = ra + w mu
        if (lane >= K.nu)
            for (int t = 0; t < T; ++t) lds[lane * TP + t] = 0.0;
        lane_stage_in<16>(lds, K.nrows, T, K.inv, lane, ra_s);
        tri_wave_sync();
        for (int i = lane; i < K.nrows; i += 64) {
            const int o = lane_pos(i, T, K.inv);
            lds[o] = fma(bufw[o], bufm[o], lds[o]);
        }
    } else {
        if (lane >= K.nu)
            for (int t = 0; t < T; ++t) lds[lane * TP + t] = 0.0;
        lane_stage_s<26>(lds, K.nrows, T, K.inv, lane, w_s, mu_s, ra_s);
    }
    tri_wave_sync();
    ck.lap(4);
    double c[R], sol[R];
#pragma unroll
    for (int j = 0; j < R; ++j) c[j] = 0.0;
    {
        double ga[R], gb[R];
        auto body = [&](const double (&gr)[R], int t) {
            const double st = sl[t];
#pragma unroll
            for (int j = 0; j < R; ++j) c[j] = fma(gr[j], st, c[j]);
        };
        lane_g_row<R>(ga, Gl, 0, r);
        int t = 0;
        for (; t + 1 < T; t += 2) {
            lane_g_row<R>(gb, Gl, t + 1, r);
            body(ga, t);
            lane_g_row<R>(ga, Gl, t + 2 < T ? t + 2 : t + 1, r);
            body(gb, t + 1);
        }
        if (t < T) body(ga, t);
    }
    ck.lap(5);
    lane_solve<R>(xs, c, sol);
    ck.lap(6);
    {
        double ga[R], gb[R];
        auto body = [&](const double (&gr)[R], int t) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int j = 0; j < R; j += 2) {
                s0 = fma(gr[j], sol[j], s0);
                if (j + 1 < R) s1 = fma(gr[j + 1], sol[j + 1], s1);
            }
            sl[t] = s0 + s1;
        };
        lane_g_row<R>(ga, Gl, 0, r);
        int t = 0;
        for (; t + 1 < T; t += 2) {
            lane_g_row<R>(gb, Gl, t + 1, r);
            body(ga, t);
            lane_g_row<R>(ga, Gl, t + 2 < T ? t + 2 : t + 1, r);
            body(gb, t + 1);
        }
        if (t < T) body(ga, t);
    }
    tri_wave_sync();
    for (int b = 0; b < K.nrows; b += 64 * 32) {
        double mt[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int i = b + 64 * k + lane;
            mt[k] = mu_s[i < K.nrows ? i : K.nrows - 1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int i = b + 64 * k + lane;
            if (i < K.nrows) {
                double s = lds[lane_pos(i, T, K.inv)] - mt[k];
                s = fmin(fmax(s, -A.dmu_bound), A.dmu_bound);
                if (flags[lane_unit(i, K.inv)] == 0.0) mu_s[i] = mt[k] + s;
            }
        }
    }
    if (failed) atomicAdd(A.fail, 1);
    ck.lap(7);
}

// KIND 0: factor (+ variance), 1: mean, 2: mean of the last sweep.  One wave per (latent, group of 64 units).
template <int KIND>
__global__ void __launch_bounds__(64) esplit_lane(SplitArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int li = blockIdx.x % A.n_lat, g = blockIdx.x / A.n_lat;
    const int r = A.shg_rk[li];
    // a launch is a few hundred LONG waves (one per 64 tasks) that share the chip with the other lane's row passes,
    // eight short waves per SIMD: without priority a wave here gets one issue slot in nine and the launch lasts nine
    // times its own instruction stream
    if (A.prio == 3) __builtin_amdgcn_s_setprio(3);
    else if (A.prio == 2) __builtin_amdgcn_s_setprio(2);
    else if (A.prio == 1) __builtin_amdgcn_s_setprio(1);
#define LANE_CASE(RV)                                            \
    do {                                                         \
        if constexpr (KIND == 0) lane_factor<RV>(A, li, g, smem); \
        else lane_mean<RV, KIND == 2>(A, li, g, smem);           \
    } while (0)
    if (r <= 8) {
        if (r <= 4) LANE_CASE(4);
        else if (r <= 6) LANE_CASE(6);
        else LANE_CASE(8);
    } else if (r <= 11) {
        if (r == 9) LANE_CASE(9);
        else if (r == 10) LANE_CASE(10);
        else LANE_CASE(11);
    } else {
        if (r == 12) LANE_CASE(12);
        else if (r == 13) LANE_CASE(13);
        else LANE_CASE(14);
    }
#undef LANE_CASE
}
