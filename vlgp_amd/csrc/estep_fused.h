// Split E-step, FUSED sweeps (round 5): one workgroup owns a group of UG units for ALL the regular sweeps of a call.
//
// Included by estep_split.hip (inside its anonymous namespace, after estep_lane.h).  Same mathematics as the launch
// sequence there (reference vlgp/core.py:22-120): per sweep  residual pass -> mean -> curvature pass -> factor.
//
// Why.  The launch sequence pays, per sweep and lane, four dependent launches (~2.5 us each), a trip through memory for
// every per-row array between them, and the one-workgroup latency of two lane-per-task launches that leave five sixths
// of the chip idle: 25 x 98 us = 2.45 ms at C3 against ~1.2 ms of instruction issue.  The units of an E-step call are
// independent (core.estep is a loop over units, core.py:123-126), so a workgroup that keeps a few units' state ON CHIP
// needs nothing from anybody else between the first factor and the last sweep:
//   * rows <-> lanes in the passes (two rows per lane, 512 lanes: up to 1024 rows = UG units), mu in the lane's
//     registers for the whole call; v, w, s / dl per (row, latent) in LDS (3 x 32 KB at C3, UG = 16);
//   * the per-latent phases are estep_lane.h's: ONE LANE PER (unit, latent) TASK, the packed triangle of H in registers,
//     Cholesky + inverse as straight-line code -- but a task is dealt to the FOUR lanes of a quad, each a quarter of the
//     time bins (16 units x 4 quarters = one wave per latent at UG = 16), partial sums met by two DPP steps instead of
//     LDS, the rows of X dealt to the quad for the solve;
//   * the prior factors G of all latents are staged once per call in LDS (row stride 14, zero padded: a lane reads its
//     own time bin's row, so G can no longer be a scalar operand);
//   * X goes through the hand-over buffer of the lane-per-task launches (A.xl, entry-major, L2 resident), so the last
//     sweep -- which also produces dmu, core.py:96-113 -- runs through the launch sequence unchanged.
// One workgroup per CU (~130 KB of LDS); UG = ceil(M / CUs) so that a set fills the chip once.
#pragma once

constexpr int FUS_NW = 8;             // waves per workgroup
constexpr int FUS_NT = 64 * FUS_NW;   // 512 threads: two rows per lane in the passes
constexpr int FUS_GS = LANE_RMAX;     // row stride (doubles) of the staged prior factors

struct FusedArgs {
    SplitArgs A;         // lat[i] = i, shg_rk / shg_gl / shg_T filled for every latent
    const double* cols;  // channel records (esplit_cols_kernel)
    int UG;              // units per workgroup
    int RS;              // row stride of the per-latent LDS arrays (>= UG T, even)
    int n_sweeps;        // regular sweeps after the first factor
    int do_v;            // VB: the factor also refreshes v
};

// LDS doubles: exp table | G | v | w | s / dl | flags
__host__ __device__ inline size_t fused_lds_doubles(int L, int T, int UG, int RS) {
    return 256 + (size_t)L * T * FUS_GS + 3 * (size_t)L * RS + (size_t)((L * UG + 1) & ~1);
}

// sum over the four lanes of a quad; every lane ends with the same bits ((a + b) + (c + d), commutative steps)
__device__ __forceinline__ double fus_quad_sum(double v) {
    union { double d; int i[2]; } a, b;
    a.d = v;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0xB1, 0xf, 0xf, false);  // quad_perm [1, 0, 3, 2]
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0xB1, 0xf, 0xf, false);
    v += b.d;
    a.d = v;
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], 0x4E, 0xf, 0xf, false);  // quad_perm [2, 3, 0, 1]
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], 0x4E, 0xf, 0xf, false);
    v += b.d;
    return v;
}

// one (unit, latent) task seen by one lane of its quad
struct FusTask {
    int l, u, q, pair;
    int t0, t1, nb;          // this lane's time bins [t0, t1), nb = bins of the longest quarter
    bool act;                // the unit exists
    const double* Gl;        // LDS: (T, FUS_GS) factor of latent l, zero padded
    int rb;                  // l RS + u T: the unit's rows in the per-latent LDS arrays
    double* xs;              // A.xl position of the task (entry e at xs[64 e])
};

template <int R>
__device__ __forceinline__ void fus_g_row(double (&g)[R], const double* Gl, int t) {
    const double* p = Gl + t * FUS_GS;
#pragma unroll
    for (int j = 0; j < R; ++j) g[j] = p[j];
}

// factor (+ variance) of the lane's task: H = I + G'WG over the quad's four quarters, X = chol(H)^-1, v_t = |X g_t|^2
template <int R>
__device__ __forceinline__ bool fus_factor(const FusedArgs& F, const FusTask& K, const double* wL, double* vL, bool do_v) {
    constexpr int E = R * (R + 1) / 2;
    double h[E];
#pragma unroll
    for (int e = 0; e < E; ++e) h[e] = 0.0;
    const double* wrow = wL + K.rb;
#pragma unroll 1
    for (int k = 0; k < K.nb; ++k) {
        const int t = K.t0 + k;
        const bool a = t < K.t1;
        const int tc = a ? t : K.t0;
        double g[R];
        fus_g_row<R>(g, K.Gl, tc);
        const double wt = a ? wrow[tc] : 0.0;
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const double gw = g[i] * wt;
#pragma unroll
            for (int j = 0; j <= i; ++j) h[lt_idx(i, j)] = fma(gw, g[j], h[lt_idx(i, j)]);
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) h[e] = fus_quad_sum(h[e]);
    bool ok = true;
#pragma unroll
    for (int i = 0; i < R; ++i) h[lt_idx(i, i)] += 1.0;
    // Cholesky, left-looking by column; the diagonal keeps 1 / L_jj (estep_lane.h, lane_factor)
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
    // X = L^-1 in place, column by column
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
    // hand-over (one lane of the quad), entry-major as the lane-per-task launches read it
    if (K.q == 0 && K.act) {
        // (the pointer advances through an opaque register: with 66 .. 105 addresses e x 512 bytes apart -- beyond the
        // store's immediate offset -- the compiler otherwise materialises them all up front, two registers each, and
        // spills the triangle it is about to store)
        double* xp = K.xs;
#pragma unroll
        for (int e = 0; e < E; ++e) {
            *xp = h[e];
            xp += 64;
            asm volatile("" : "+v"(xp));
        }
    }
    if (do_v && ok && K.act) {
        double* vrow = vL + K.rb;
    #pragma unroll 1
    for (int k = 0; k < K.nb; ++k) {
            const int t = K.t0 + k;
            const bool a = t < K.t1;
            double g[R];
            fus_g_row<R>(g, K.Gl, a ? t : K.t0);
            double vv = 0.0;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                double z = h[lt_idx(i, 0)] * g[0];
#pragma unroll
                for (int j = 1; j <= i; ++j) z = fma(h[lt_idx(i, j)], g[j], z);
                vv = fma(z, z, vv);
            }
            if (a) vrow[t] = vv;
        }
    }
    return ok;
}

// mean of the lane's task: dl = G (I + H)^-1 G's over the unit's rows (s in, dl out, same LDS array)
template <int R>
__device__ __forceinline__ void fus_mean(const FusedArgs& F, const FusTask& K, double* sL) {
    LaneXRows<R> X;
    X.request(K.xs, K.q);  // rows q, q + 4, ... of X: the quad shares the solve
    double* srow = sL + K.rb;
    double c[R], sol[R];
#pragma unroll
    for (int j = 0; j < R; ++j) c[j] = 0.0;
#pragma unroll 1
    for (int k = 0; k < K.nb; ++k) {
        const int t = K.t0 + k;
        const bool a = t < K.t1;
        const int tc = a ? t : K.t0;
        double g[R];
        fus_g_row<R>(g, K.Gl, tc);
        const double st = a ? srow[tc] : 0.0;
#pragma unroll
        for (int j = 0; j < R; ++j) c[j] = fma(g[j], st, c[j]);
    }
#pragma unroll
    for (int j = 0; j < R; ++j) c[j] = fus_quad_sum(c[j]);
    X.solve(c, sol, K.q);
#pragma unroll
    for (int j = 0; j < R; ++j) sol[j] = fus_quad_sum(sol[j]);
#pragma unroll 1
    for (int k = 0; k < K.nb; ++k) {
        const int t = K.t0 + k;
        const bool a = t < K.t1;
        double g[R];
        fus_g_row<R>(g, K.Gl, a ? t : K.t0);
        double s0 = g[0] * sol[0], s1 = 0.0;
#pragma unroll
        for (int j = 1; j < R; ++j) {
            if (j & 1) s1 = fma(g[j], sol[j], s1);
            else s0 = fma(g[j], sol[j], s0);
        }
        if (a && K.act) srow[t] = s0 + s1;
    }
}

// Row pass over the lane's two rows.  KIND SP_RES: s = (ya - sum_n rate_n a_n) + w mu -> sL.  KIND SP_W: apply the
// mean step mu += clip(dl - mu) (dl in sL; not for a task whose factor failed, core.py:92-94), then
// w = 2 sum_n rate_n a_n^2 / 2 + wconst -> wL.  Same channel loop as esplit_pass (records by scalar loads).
template <int LT, int KIND>
__device__ __forceinline__ void fus_pass(const FusedArgs& F, double (&mr)[2][LT], const bool (&in)[2], const int (&ri)[2],
                                         const int64_t r0, const double* etab, const double* vL, double* wL, double* sL,
                                         const double* flags) {
    constexpr int REC = rec_len<LT>();
    const SplitArgs& A = F.A;
    const int L = A.L, RS = F.RS;
    double vr[2][LT], acc[2][LT], aux[2][LT];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            vr[q][l] = l < L ? vL[l * RS + ri[q]] : 0.0;
            acc[q][l] = 0.0;
            if constexpr (KIND == SP_RES) aux[q][l] = l < L ? A.ya[(int64_t)l * A.ld + r0 + ri[q]] : 0.0;
            else aux[q][l] = l < L ? sL[l * RS + ri[q]] : 0.0;
        }
    if constexpr (KIND == SP_W) {
        const int inv = (1 << 20) / A.shg_T + 1;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int u = lane_unit(ri[q], inv);
#pragma unroll
            for (int l = 0; l < LT; ++l)
                if (l < L && flags[l * F.UG + u] == 0.0) {
                    double st = aux[q][l] - mr[q][l];
                    st = fmin(fmax(st, -A.dmu_bound), A.dmu_bound);
                    mr[q][l] += st;
                }
        }
    }
    // (through the CONSTANT address space: in a kernel that also stores to global memory the compiler otherwise gives up
    // on keeping these wave-uniform loads scalar -- one vector load per lane and channel, estep_lane.h lane_g_row)
    auto load_rec = [&](int i, double (&rv)[REC]) {
        lane_cptr rp = (lane_cptr)(F.cols + (int64_t)i * REC);
#pragma unroll
        for (int q = 0; q < REC; ++q) rv[q] = rp[q];
    };
    auto poisson = [&](const double (&rv)[REC]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            double eta = rv[2 * LT];
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
    const int np = A.np, ntot = A.ntot;
    if (np > 0) {
        load_rec(0, ra_);
        int i = 0;
        for (; i + 1 < np; i += 2) {
            load_rec(i + 1, rb_);
            poisson(ra_);
            load_rec(i + 2 < np ? i + 2 : i + 1, ra_);
            poisson(rb_);
        }
        if (i < np) poisson(ra_);
    }
    if constexpr (KIND == SP_RES) {  // Gaussian channels: the residual mean is eta itself
        for (int i = np; i < ntot; ++i) {
            load_rec(i, ra_);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                double eta = ra_[2 * LT];
#pragma unroll
                for (int l = 0; l < LT; ++l) eta = fma(mr[q][l], ra_[l], eta);
                const double mval = eta * ra_[2 * LT + 1];
#pragma unroll
                for (int l = 0; l < LT; ++l) acc[q][l] = fma(mval, ra_[l], acc[q][l]);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        if (!in[q]) continue;
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            if (l >= L) continue;
            if constexpr (KIND == SP_RES) {
                const double rav = aux[q][l] - acc[q][l];
                sL[l * RS + ri[q]] = fma(wL[l * RS + ri[q]], mr[q][l], rav);
            } else {
                wL[l * RS + ri[q]] = fma(2.0, acc[q][l], A.wconst[l]);  // (the records hold a^2 / 2)
            }
        }
    }
}

// the per-latent phase of a wave, compiled for the largest rank among its lanes' latents (smaller ranks: identity padding)
template <bool MEAN>
__device__ __forceinline__ bool fus_latent_phase(const FusedArgs& F, const FusTask& K, int rw, double* vL, double* wL, double* sL, bool do_v) {
    bool ok = true;
#define FUS_CASE(RV)                                   \
    do {                                               \
        if constexpr (MEAN) fus_mean<RV>(F, K, sL);    \
        else ok = fus_factor<RV>(F, K, wL, vL, do_v);  \
    } while (0)
#ifdef FUS_ONLY
    FUS_CASE(FUS_ONLY);
#else
    if (rw <= 8) {
        if (rw <= 4) FUS_CASE(4);
        else if (rw <= 6) FUS_CASE(6);
        else FUS_CASE(8);
    } else if (rw <= 12) {
        if (rw <= 10) FUS_CASE(10);
        else if (rw == 11) FUS_CASE(11);
        else FUS_CASE(12);
    } else {
        if (rw == 13) FUS_CASE(13);
        else FUS_CASE(14);
    }
#endif
#undef FUS_CASE
    return ok;
}

template <int LT>
__global__ void __launch_bounds__(FUS_NT) efused_kernel(FusedArgs F) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SplitArgs& A = F.A;
    const int T = A.shg_T, L = A.L, UG = F.UG, RS = F.RS;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int u0 = blockIdx.x * UG;
    const int nu = A.M - u0 < UG ? A.M - u0 : UG;
    const int nrows = nu * T;
    const int64_t r0 = A.off[u0];  // (equal-length units, packed: unit u0 + u starts at r0 + u T)
    double* etab = smem;
    double* Gs = etab + 256;
    double* vL = Gs + L * T * FUS_GS;
    double* wL = vL + L * RS;
    double* sL = wL + L * RS;
    double* flags = sL + L * RS;
    fast_exp_tab256_init(etab, tid);
    // the prior factors, zero padded to FUS_GS columns
    for (int l = 0; l < L; ++l) {
        const int r = A.shg_rk[l];
        const double* gl = A.shg_gl[l];
        for (int i = tid; i < T * FUS_GS; i += FUS_NT) {
            const int t = i / FUS_GS, j = i - t * FUS_GS;
            Gs[l * T * FUS_GS + i] = j < r ? gl[t * r + j] : 0.0;
        }
    }
    // the lane's two rows: mu in registers for the whole call, v and w in LDS (zeros beyond the group's rows)
    bool in[2];
    int ri[2];
    double mr[2][LT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = tid + FUS_NT * q;
        in[q] = i < nrows;
        ri[q] = in[q] ? i : 0;
#pragma unroll
        for (int l = 0; l < LT; ++l) mr[q][l] = (l < L && in[q]) ? A.mu[(int64_t)l * A.ld + r0 + i] : 0.0;
    }
    for (int l = 0; l < L; ++l)
        for (int i = tid; i < RS; i += FUS_NT) {
            const bool ex = i < nrows;
            vL[l * RS + i] = ex ? A.v[(int64_t)l * A.ld + r0 + i] : 0.0;
            wL[l * RS + i] = ex ? A.w[(int64_t)l * A.ld + r0 + i] : 0.0;
            sL[l * RS + i] = 0.0;
        }
    for (int i = tid; i < L * UG; i += FUS_NT) flags[i] = 0.0;
    // the lane's task
    FusTask K;
    const int n_pairs = L * UG;
    K.pair = tid >> 2;
    K.q = tid & 3;
    const bool has_task = K.pair < n_pairs;
    const int pc = has_task ? K.pair : 0;
    K.l = pc / UG;
    K.u = pc - K.l * UG;
    K.act = has_task && K.u < nu;
    K.t0 = (T * K.q) / 4;
    K.t1 = (T * (K.q + 1)) / 4;
    K.nb = (T + 3) / 4;
    K.Gl = Gs + K.l * T * FUS_GS;
    K.rb = K.l * RS + K.u * T;
    {
        const int um = u0 + (K.act ? K.u : 0);
        K.xs = A.xl + ((int64_t)(um >> 6) * L + K.l) * (64 * LANE_EMAX) + (um & 63);
    }
    // the waves that hold tasks, and the rank class of each (wave-uniform)
    const int n_lw = (4 * n_pairs + 63) / 64;
    int rw = 0;
    if (wid < n_lw) {
        const int l_lo = (16 * wid) / UG;
        int l_hi = (16 * wid + 15) / UG;
        if (l_hi > L - 1) l_hi = L - 1;
        for (int l = l_lo; l <= l_hi; ++l) rw = A.shg_rk[l] > rw ? A.shg_rk[l] : rw;
    }
    int nfail = 0;
    // debug: cycles per phase of the first workgroup (VLGP_LANE_CLOCK=3, vlgp_debug_phase_clock): staging | (unused) |
    // residual pass | mean | curvature pass | factor | write-back
    LaneClock ck(A, 3);
    __syncthreads();
    ck.lap(0);
    // sweep -1 is the factor from the incoming w alone (v is not refreshed there: the launch sequence's sweep -1).
    // (One call site per phase: a phase reached from two places is compiled out of line, its LDS pointers become generic
    // ones and the triangle of H an array in scratch.)
    for (int sw = -1; sw < F.n_sweeps; ++sw) {
        if (sw >= 0) {
#ifndef FUS_NOPASS
            fus_pass<LT, SP_RES>(F, mr, in, ri, r0, etab, vL, wL, sL, flags);
#endif
            __syncthreads();
            ck.lap(2);
            if (wid < n_lw) {
#ifndef FUS_NOMEAN
                fus_latent_phase<true>(F, K, rw, vL, wL, sL, false);
#endif
                if (K.q == 0 && K.act && flags[K.pair] != 0.0) ++nfail;  // (the mean launch counts a failed task once more)
            }
            __syncthreads();
            ck.lap(3);
#ifndef FUS_NOPASS
            fus_pass<LT, SP_W>(F, mr, in, ri, r0, etab, vL, wL, sL, flags);
#endif
            __syncthreads();
            ck.lap(4);
        }
        if (wid < n_lw) {
#ifndef FUS_NOFACTOR
            const bool ok = fus_latent_phase<false>(F, K, rw, vL, wL, sL, sw >= 0 && F.do_v != 0);
#else
            const bool ok = true;
#endif
            if (K.q == 0 && has_task) flags[K.pair] = ok ? 0.0 : 1.0;
            if (K.q == 0 && K.act && !ok) ++nfail;
        }
        __syncthreads();
        ck.lap(5);
    }
    // back to the latent-major arrays of the launch sequence
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int l = 0; l < LT; ++l)
            if (l < L && in[q]) A.mu[(int64_t)l * A.ld + r0 + ri[q]] = mr[q][l];
    for (int l = 0; l < L; ++l)
        for (int i = tid; i < nrows; i += FUS_NT) {
            A.v[(int64_t)l * A.ld + r0 + i] = vL[l * RS + i];
            A.w[(int64_t)l * A.ld + r0 + i] = wL[l * RS + i];
        }
    if (K.q == 0 && K.act) A.failg[(u0 + K.u) * L + K.l] = flags[K.pair] != 0.0 ? 1 : 0;
    if (nfail) atomicAdd(A.fail, nfail);
    ck.lap(6);
}
