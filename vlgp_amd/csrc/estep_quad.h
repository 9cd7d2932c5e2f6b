// Split E-step, factor + variance of the latents at ranks 15 .. 32 with a GROUP OF LANES per (unit, latent) task (round 6).
//
// Included by estep_split.hip (inside its anonymous namespace, after estep_lane.h).  Same mathematics as factor_task there
// (reference vlgp/core.py:102-113): H = I + G'WG, X = chol(H)^-1, v_t = |X g_t|^2.
//
// Why this shape.  The wave-per-task kernel (esplit_latent<.., false>) issues ~6 k vector instructions per task at rank 29
// -- a 64-lane machine on a 29-wide matrix, the elimination one pivot at a time over 32 + 32 rows -- and its class-32
// launch is 144 us per half of C3.  The lane-per-task form (estep_lane.h) needs r (r + 1) / 2 doubles per lane: 105 at
// rank 14 is the end of it.  In between: NL = 2 / 4 / 8 lanes share a task, the rows of the lower triangle dealt to them
// round-robin (lane q of a group owns rows q, q + NL, ...), every index static, G wave-uniform (all units of a launch share
// the prior factor of their latent).  What one lane needs from its neighbours -- the pivot column in the Cholesky sweep,
// the finished entries of a column of X in the inversion -- travels by DPP (quad_perm inside a quad, one row_shr / row_shl
// by four across the two quads of a group of eight): no LDS, no barrier.
//
//   R (compiled rank, ranks above the actual one are identity padding) and NL:  16 / 2,  20 / 4,  24 / 4,  32 / 8
//   registers of H per lane: NL K (K + 1) / 2, K = R / NL:                       72      60      84      80 doubles
//
// A row slot k of a lane (row i = NL k + q) keeps the columns 0 .. NL (k + 1) - 1: the lower triangle plus the rest of its
// diagonal block, so that every lane of a group runs the same code on the same register indices; the entries above the
// diagonal are scratch and are zeroed before X is used.
#pragma once

template <int CTRL, int BANK>
__device__ __forceinline__ double quad_dpp(double old, double v) {
    union { double d; int i[2]; } a, o, r;
    a.d = v;
    o.d = old;
    r.i[0] = __builtin_amdgcn_update_dpp(o.i[0], a.i[0], CTRL, 0xf, BANK, false);
    r.i[1] = __builtin_amdgcn_update_dpp(o.i[1], a.i[1], CTRL, 0xf, BANK, false);
    return r.d;
}

// the value lane S of every group of NL lanes holds, in all lanes of the group
template <int NL, int S>
__device__ __forceinline__ double quad_bcast_s(double v) {
    if constexpr (NL == 2) {
        return quad_dpp<(S | (S << 2) | ((2 + S) << 4) | ((2 + S) << 6)), 0xf>(v, v);
    } else if constexpr (NL == 4) {
        return quad_dpp<S * 0x55, 0xf>(v, v);
    } else {
        const double t = quad_dpp<(S & 3) * 0x55, 0xf>(v, v);  // every quad: its lane S & 3
        if constexpr (S < 4) return quad_dpp<0x114, 0xA>(t, t);  // row_shr:4 into the upper quads
        else return quad_dpp<0x104, 0x5>(t, t);                  // row_shl:4 into the lower quads
    }
}
template <int NL>
__device__ __forceinline__ double quad_bcast(double v, int s) {  // s: compile-time after unrolling
    switch (s) {
        case 0: return quad_bcast_s<NL, 0>(v);
        case 1: return quad_bcast_s<NL, 1>(v);
        case 2: if constexpr (NL > 2) return quad_bcast_s<NL, 2>(v); else return v;
        case 3: if constexpr (NL > 2) return quad_bcast_s<NL, 3>(v); else return v;
        case 4: if constexpr (NL > 4) return quad_bcast_s<NL, 4>(v); else return v;
        case 5: if constexpr (NL > 4) return quad_bcast_s<NL, 5>(v); else return v;
        case 6: if constexpr (NL > 4) return quad_bcast_s<NL, 6>(v); else return v;
        default: if constexpr (NL > 4) return quad_bcast_s<NL, 7>(v); else return v;
    }
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

// LDS doubles of a factor launch of class (R, NL): G staged once per workgroup (T x R, zero beyond the rank) | per wave
// its units' w / v ([unit][t], odd stride)
__host__ __device__ inline size_t quad_lds_doubles(int T, int R, int NL) { return (size_t)T * R + 4 * (size_t)(64 / NL) * (T | 1); }

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
    // ---- Cholesky, right-looking, column by column; the diagonal keeps 1 / L_jj ----
    bool ok = true;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (j < r) {  // (columns beyond the rank: identity)
            constexpr int dummy = 0;
            (void)dummy;
            const int kj = j / NL, qj = j % NL;
            const double d = quad_bcast<NL>(h[Q::off(kj) + j], qj);
            if (!(d > 0.0) || !(d < 1e300)) ok = false;
            double y = __builtin_amdgcn_rsq(d);
            {
                double e = fma(-d * y, y, 1.0);
                y = fma(y * 0.5, e, y);
                e = fma(-d * y, y, 1.0);
                y = fma(y * 0.5, e, y);
            }
#pragma unroll
            for (int k = kj; k < K; ++k) h[Q::off(k) + j] *= y;       // column j of L for the rows below (and scratch above)
            if (q == qj) h[Q::off(kj) + j] = y;                       // the owner of row j: 1 / L_jj
#pragma unroll
            for (int c = j + 1; c < R; ++c) {
                if (c < r) {
                    const int kc = c / NL, qc = c % NL;
                    const double b = quad_bcast<NL>(h[Q::off(kc) + j], qc);  // L_cj
#pragma unroll
                    for (int k = kc; k < K; ++k) h[Q::off(k) + c] = fma(-h[Q::off(k) + j], b, h[Q::off(k) + c]);
                }
            }
        }
    }
    // the owner's slot of column j holds 1 / L_jj, and a trailing update read it as if it were L_ij for i = j: those
    // products landed in entries above the diagonal only (c > j = i), which are scratch.
    // ---- X = L^-1 in place, column by column: X_ij = -(1 / L_ii) sum_{k = j}^{i - 1} L_ik X_kj ----
#pragma unroll
    for (int j = 0; j < R; ++j) {
        if (j < r) {
            const int kj = j / NL, qj = j % NL;
            const double xjj = quad_bcast<NL>(h[Q::off(kj) + j], qj);  // X_jj = 1 / L_jj
            double s[K];
#pragma unroll
            for (int k = 0; k < K; ++k) s[k] = k >= kj ? h[Q::off(k) + j] * xjj : 0.0;  // L_ij X_jj for the rows below j
#pragma unroll
            for (int c = j + 1; c < R; ++c) {
                if (c < r) {
                    const int kc = c / NL, qc = c % NL;
                    // row c is complete: X_cj = -(1 / L_cc) s_c, in its owner
                    const double xcj = -s[kc] * h[Q::off(kc) + c];
                    if (q == qc) h[Q::off(kc) + j] = xcj;
                    const double xb = quad_bcast<NL>(xcj, qc);
#pragma unroll
                    for (int k = kc; k < K; ++k) s[k] = fma(h[Q::off(k) + c], xb, s[k]);  // rows below c: + L_ic X_cj
                    // (rows of slot kc at or above c added a product with a diagonal / scratch entry to an s they no longer
                    // need: each s[k] is consumed when its own row completes, and slot kc's rows complete in order q)
                }
            }
        }
    }
    // entries above the diagonal -> 0; rows / columns beyond the rank stay the identity (their diagonal is 1, the rest 0)
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int s = 1; s < NL; ++s)
            if (q < s) h[Q::off(k) + NL * k + s] = 0.0;
    // ---- variance v_t = |X g_t|^2 ----
    if (A.do_v) {
        for (int t = 0; t < T; ++t) {
            const double* gr = Gs + t * R;
            double z[K];
#pragma unroll
            for (int k = 0; k < K; ++k) z[k] = 0.0;
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
            for (int k = 0; k < K; ++k) vv = fma(z[k], z[k], vv);
            vv = quad_sum<NL>(vv);
            if (q == 0) wl[t] = vv;
        }
    }
    // ---- hand-over: packed lower-triangular rows, as the wave-per-task mean launches read them ----
    const int m = mw + gu;
    if (m < A.M) {
        double* xd = A.xg + (int64_t)(m * L + l) * A.pkg;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const int i = NL * k + q;
            double* xr = xd + tri_row_off(i);
#pragma unroll
            for (int j = 0; j < Q::width(k); ++j)
                if (j <= i) xr[j] = h[Q::off(k) + j];
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
__global__ void __launch_bounds__(256, 1) esplit_quad(SplitArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    esplit_quad_body<R, NL>(A, smem, blockIdx.x);
}
