// Split E-step, factor + variance of the latents at ranks 17 .. 32 with a GROUP OF LANES per (unit, latent) task (round 6).
//
// Included by estep_split.hip (inside its anonymous namespace, after estep_lane.h).  Same mathematics as factor_task there
// (reference vlgp/core.py:102-113): H = G'WG, P = (I + H)^-1, v_t = g_t'P g_t; what the mean launches need from it is
// P c for one vector c per sweep, so P itself is handed over (A.xsym) instead of the Cholesky factor's inverse.
//
// Why this shape.  The wave-per-task kernel (esplit_latent<.., false>) issues ~6 k vector instructions per task at rank 29
// -- a 64-lane machine on a 29-wide matrix, the elimination one pivot at a time over 32 + 32 rows -- and its class-32
// launch is 144 us per half of C3.  The lane-per-task form (estep_lane.h) needs r (r + 1) / 2 doubles per lane: 105 at
// rank 14 is the end of it.  In between: NL = 4 / 8 lanes share a task, the rows of the lower triangle dealt to them
// round-robin (lane q of a group owns rows q, q + NL, ...), every index static, G wave-uniform (all units of a launch share
// the prior factor of their latent) and staged once per workgroup.  The inverse is a symmetric Gauss-Jordan sweep per
// pivot; the pivot's row / column reaches the lanes of its group through R doubles of LDS.
//
//   R (compiled rank, ranks above the actual one are identity padding) and NL:  20 / 4,  24 / 4,  32 / 8
//   registers of the matrix per lane: NL K (K + 1) / 2, K = R / NL:              60      84      80 doubles
//
// A row slot k of a lane (row i = NL k + q) keeps the columns 0 .. NL (k + 1) - 1: the lower triangle plus the rest of its
// diagonal block, so that every lane of a group runs the same code on the same register indices; the entries above the
// diagonal are scratch and are zeroed before P is used.
#pragma once
#ifndef QUAD_LB
#define QUAD_LB 1
#endif

template <int CTRL, int BANK>
__device__ __forceinline__ double quad_dpp(double old, double v) {
    union { double d; int i[2]; } a, o, r;
    a.d = v;
    o.d = old;
    r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], a.i[0], CTRL, 0xf, BANK, false);
    r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], a.i[1], CTRL, 0xf, BANK, false);
    return r.d;
}

// sum over the lanes of a group, the same bits in every lane (a butterfly of commutative additions)
template <int NL>
__device__ __forceinline__ double quad_sum(double v) {
    v += quad_dpp<0xB1, 0xf>(v, v);                       // quad_perm [1, 0, 3, 2]
    if constexpr (NL >= 4) v += quad_dpp<0x4E, 0xf>(v, v);  // quad_perm [2, 3, 0, 1]
    if constexpr (NL >= 8) {
        double p = quad_dpp<0x114, 0xA>(v, v);  // upper quads <- lower
        p = quad_dpp<0x104, 0x5>(p, v);         // lower quads <- upper
        v += p;
    }
    return v;
}

template <int R, int NL>
struct QuadGeom {
    static_assert(R % NL == 0 && 64 % NL == 0, "rows are dealt to the lanes of a group round-robin");
    static constexpr int K = R / NL;                    // row slots per lane
    static constexpr int E = NL * K * (K + 1) / 2;      // doubles of H per lane
    static constexpr int UPW = 64 / NL;                 // units per wave
    __host__ __device__ static constexpr int off(int k) { return NL * k * (k + 1) / 2; }
    __host__ __device__ static constexpr int width(int k) { return NL * (k + 1); }
};

// Sweep KP of the symmetric Gauss-Jordan inversion (see quad_factor), then the next: the pivot index is a template
// parameter -- as a `#pragma unroll` loop the body with its pins is past the unroller's budget, stays rolled, and the matrix,
// indexed at run time, becomes a scratch array.
template <int R, int NL, int KP, class HArr>
__device__ __forceinline__ void quad_sweeps(HArr& h, double* xs, double* xq, int q, bool& ok) {
    using Q = QuadGeom<R, NL>;
    constexpr int K = Q::K, k = KP, kk = KP / NL, qk = KP % NL;
    {   // (branch-free: a lane that has nothing to contribute writes to the second, unused half of its group's buffer)
        double* dst = q == qk ? xs : xs + R;
#pragma unroll
        for (int j = 0; j <= k; j += 2) {
            if (j + 1 <= k) *reinterpret_cast<double2*>(dst + j) = double2{h[Q::off(kk) + j], h[Q::off(kk) + j + 1]};
            else dst[j] = h[Q::off(kk) + j];
        }
    }
    // (every address = one of two lane-dependent bases + a constant: addresses formed per entry are all materialised
    // up front by the compiler and spill the matrix)
    (q > qk ? xq : xq + R)[NL * kk] = h[Q::off(kk) + k];
#pragma unroll
    for (int k2 = kk + 1; k2 < K; ++k2) xq[NL * k2] = h[Q::off(k2) + k];
    tri_wave_order();
    const double d = xs[k];
    if (!(d > 0.0) || !(d < 1e300)) ok = false;
    double dinv = __builtin_amdgcn_rcp(d);
    {
        double e = fma(-d, dinv, 1.0);
        dinv = fma(dinv, e, dinv);
        e = fma(-d, dinv, 1.0);
        dinv = fma(dinv, e, dinv);
    }
    double f[K];
#pragma unroll
    for (int k2 = 0; k2 < K; ++k2) f[k2] = (k2 == kk && q == qk) ? 1.0 - dinv : xq[NL * k2] * dinv;
#pragma unroll
    for (int jb = 0; jb < R; jb += 2) {
        const double2 p2 = *reinterpret_cast<const double2*>(xs + jb);
#pragma unroll
        for (int k2 = jb / NL; k2 < K; ++k2) {
            h[Q::off(k2) + jb] = fma(-f[k2], p2.x, h[Q::off(k2) + jb]);
            h[Q::off(k2) + jb + 1] = fma(-f[k2], p2.y, h[Q::off(k2) + jb + 1]);
        }
    }
#pragma unroll
    for (int k2 = kk; k2 < K; ++k2) h[Q::off(k2) + k] = f[k2];
    if (q == qk) h[Q::off(kk) + k] = -dinv;
    // evaluate the sweep HERE: left alone, the compiler defers the updates of the entries no later sweep reads soon and
    // carries their operands -- every p and f of up to thirty sweeps -- in accumulation registers and 3.8 KB of scratch
#pragma unroll
    for (int k2 = 0; k2 < K; ++k2)
#pragma unroll
        for (int j = 0; j < Q::width(k2); ++j) asm volatile("" : "+v"(h[Q::off(k2) + j]));
    tri_wave_order();
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KP + 1 < R) quad_sweeps<R, NL, KP + 1>(h, xs, xq, q, ok);
}

// LDS doubles of a factor launch of class (R, NL): G staged once per workgroup (T x R, zero beyond the rank) | per wave
// its units' w / v ([unit][t], odd stride) | per group the R doubles of the sweeps' exchange (+ R of trash)
__host__ __device__ inline size_t quad_lds_doubles(int T, int R, int NL) {
    return (size_t)T * R + 4 * (size_t)(64 / NL) * (T | 1) + 4 * (size_t)(64 / NL) * 2 * R;
}

// factor + variance of 4 x 64 / NL units of one latent: one workgroup, the four waves independent of each other
template <int R, int NL>
__device__ __forceinline__ void quad_factor(const SplitArgs& A, int l, int r, const double* __restrict__ Gl, int m0, double* lds) {
    using Q = QuadGeom<R, NL>;
    constexpr int K = Q::K, E = Q::E, UPW = Q::UPW;
    const int T = A.shg_T, TP = T | 1, L = A.L;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = lane % NL, gu = lane / NL;           // lane of the group, unit of the wave
    double* Gs = lds;                                   // T x R
    double* wl = lds + (size_t)T * R + (size_t)wid * UPW * TP + gu * TP;  // this unit's w, later v
    // ---- staging: G (zero-padded to R columns), w of the wave's units ----
    for (int x = threadIdx.x; x < T * R; x += 256) {
        const int t = x / R, j = x - t * R;
        Gs[x] = j < r ? Gl[t * r + j] : 0.0;
    }
    const int mw = m0 + wid * UPW;                      // first unit of this wave
    const int nu = A.M - mw < UPW ? (A.M - mw > 0 ? A.M - mw : 0) : UPW;
    {
        double* wb = lds + (size_t)T * R + (size_t)wid * UPW * TP;
        const double* __restrict__ w_s = A.w + (int64_t)l * A.ld + (nu > 0 ? A.off[mw] : 0);
        for (int x = lane; x < UPW * T; x += 64) {
            const int u = x / T, t = x - u * T;
            wb[u * TP + t] = u < nu ? w_s[x] : 0.0;   // (the units of a set are contiguous rows of T)
        }
    }
    __syncthreads();
    // ---- build H = G'WG (lower triangle + the rest of every diagonal block) ----
    double h[E];
#pragma unroll
    for (int e = 0; e < E; ++e) h[e] = 0.0;
    for (int t = 0; t < T; ++t) {
        const double wt = wl[t];
        const double* gr = Gs + t * R;
        double gw[K];
#pragma unroll
        for (int k = 0; k < K; ++k) gw[k] = gr[NL * k + q] * wt;
#pragma unroll
        for (int jb = 0; jb < R; jb += 2) {
            const double2 g2 = *reinterpret_cast<const double2*>(gr + jb);
#pragma unroll
            for (int k = jb / NL; k < K; ++k) {  // slots whose width covers column jb (NL is even: jb + 1 as well)
                h[Q::off(k) + jb] = fma(gw[k], g2.x, h[Q::off(k) + jb]);
                h[Q::off(k) + jb + 1] = fma(gw[k], g2.y, h[Q::off(k) + jb + 1]);
            }
        }
    }
    // the unit diagonal: lane q owns (row NL k + q, column NL k + q)
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int s = 0; s < NL; ++s) h[Q::off(k) + NL * k + s] += (s == q) ? 1.0 : 0.0;
    // ---- P = (I + H)^-1 by symmetric Gauss-Jordan sweeps on the lower triangle.  Sweep k needs row / column k of the
    // matrix in every lane of the group (p_m = a_{max(m, k), min(m, k)}): the owner of row k puts its entries m <= k, every
    // lane the column-k entries of its rows m > k, into R doubles of LDS per group -- one round trip per sweep instead of
    // a DPP broadcast per entry (a Cholesky + triangular inverse through DPP was the first version: ~12 k of the kernel's
    // 16 k instructions at R = 32 went into 2 x 528 broadcasts).  With f_i = a_ik / a_kk (f_k = 1 - 1 / a_kk for the pivot
    // row itself) ONE update a_ij -= f_i p_j covers every entry; afterwards column k <- f, a_kk <- -1 / a_kk.  All R sweeps
    // done, h = -P.  Rows beyond the rank are the identity and sweep as such (pivot 1, no update).
    double* xs = lds + (size_t)T * R + 4 * (size_t)UPW * TP + ((size_t)wid * UPW + gu) * 2 * R;
    double* xq = xs + q;
    bool ok = true;
    quad_sweeps<R, NL, 0>(h, xs, xq, q, ok);
    // entries above the diagonal -> 0; the diagonal of the own rows, once
    double hd[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        hd[k] = h[Q::off(k) + NL * k];
#pragma unroll
        for (int s = 1; s < NL; ++s) {
            if (q == s) hd[k] = h[Q::off(k) + NL * k + s];
            if (q < s) h[Q::off(k) + NL * k + s] = 0.0;
        }
    }
    // ---- variance v_t = g_t'P g_t = sum_i g_i (2 sum_{j <= i} P_ij g_j - P_ii g_i), with h = -P ----
    if (A.do_v) {
        for (int t = 0; t < T; ++t) {
            const double* gr = Gs + t * R;
            double z[K], gi[K];
#pragma unroll
            for (int k = 0; k < K; ++k) {
                z[k] = 0.0;
                gi[k] = gr[NL * k + q];
            }
#pragma unroll
            for (int jb = 0; jb < R; jb += 2) {
                const double2 g2 = *reinterpret_cast<const double2*>(gr + jb);
#pragma unroll
                for (int k = jb / NL; k < K; ++k) {
                    z[k] = fma(h[Q::off(k) + jb], g2.x, z[k]);
                    z[k] = fma(h[Q::off(k) + jb + 1], g2.y, z[k]);
                }
            }
            double vv = 0.0;
#pragma unroll
            for (int k = 0; k < K; ++k) vv = fma(gi[k], fma(hd[k], gi[k], -2.0 * z[k]), vv);
            vv = quad_sum<NL>(vv);
            if (q == 0) wl[t] = vv;
        }
    }
    // ---- hand-over: P = -h as packed lower-triangular rows (the layout of the wave-per-task launches' X; the mean
    // launches are told by A.xsym that it is the symmetric inverse itself) ----
    const int m = mw + gu;
    if (m < A.M) {
        double* xd = A.xg + (int64_t)(m * L + l) * A.pkg;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = NL * k + q;
            double* xr = xd + tri_row_off(i);
#pragma unroll
            for (int j = 0; j < Q::width(k); ++j)
                if (j <= i) xr[j] = -h[Q::off(k) + j];
        }
        if (q == 0) {
            A.failg[m * L + l] = ok ? 0 : 1;
            if (!ok) atomicAdd(A.fail, 1);
        }
    }
    if (A.do_v) {
        // v: [unit][t] in LDS -> the latent-major global array, coalesced over the wave's contiguous rows
        __builtin_amdgcn_wave_barrier();
        double* wb = lds + (size_t)T * R + (size_t)wid * UPW * TP;
        double* v_s = A.v + (int64_t)l * A.ld + (nu > 0 ? A.off[mw] : 0);
        // a failed factor leaves v as it is (core.py:109-113): its lanes hold ok = false
        const unsigned long long okmask = __ballot(ok);
        for (int x = lane; x < nu * T; x += 64) {
            const int u = x / T, t = x - u * T;
            if ((okmask >> (u * NL)) & 1ull) v_s[x] = wb[u * TP + t];
        }
    }
}

// One workgroup = one latent x (4 x 64 / NL) units, every latent of a launch in the class (R, NL) of the highest rank among
// them: 15, 16 -> (16, 2); 17 .. 20 -> (20, 4); 21 .. 24 -> (24, 4); 25 .. 32 -> (32, 8).  A.qblk0[i]: blocks per latent.
template <int R, int NL>
__device__ __forceinline__ void esplit_quad_body(const SplitArgs& A, double* smem, int bid) {
    const int nb = A.qblk0[0];
    const int li = bid / nb, b = bid - li * nb;
    quad_factor<R, NL>(A, A.lat[li], A.shg_rk[li], A.shg_gl[li], b * (4 * (64 / NL)), smem);
}

template <int R, int NL>
__global__ void __launch_bounds__(256, QUAD_LB) esplit_quad(SplitArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    esplit_quad_body<R, NL>(A, smem, blockIdx.x);
}
