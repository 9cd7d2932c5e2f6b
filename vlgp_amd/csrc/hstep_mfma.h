// H-step objective, per-segment part, on the matrix pipe (gfx950, v_mfma_f64_16x16x4).
//
// One wavefront = one segment.  For A = I + S K S (S = diag sqrt(w), eigenvalues >= 1) the round needs
//     tr(A^-1)   and   sum_jk s_j s_k dK_jk (A^-1)_jk            (see hstep.hip, "Fast path")
// i.e. a weighted sum over ALL entries of A^-1: O(T^3) whatever the algebra.  The register-resident row
// kernels (wave_tri.h) spend one LDS broadcast per two FMAs and leave half of the lanes idle in every
// triangular loop (measured: 10 % of the fp64 peak, latency- and LDS-bound).  Here the cubic work runs as
// 16 x 16 x 4 matrix instructions on 16-row blocks and only the panel factorisations stay on the vector pipe:
//
//   * right-looking blocked Cholesky of the matrix AUGMENTED with identity rows, [A; I] -> [L; L^-T]:
//     the same panel elimination that turns the rows of A into rows of L turns the unit rows into rows
//     of X' = L^-T, so the triangular inverse costs no separate sweep and no diagonal-block inverse.
//   * the Schur complements are never stored: N = sum_k L_:k L_:k' (for A) and M = sum_k X'_:k L_:k' (for
//     X') accumulate in MFMA result registers and are subtracted from the freshly generated entries of A
//     (Toeplitz first column from a small LDS table) when their panel comes up.
//   * A^-1 = X'X accumulates panel by panel as well (P += X'_:k X'_:k'), so nothing but the current
//     panel ever sits in LDS: 66 rows of 18 doubles per segment.
//   * panel factorisation: every lane holds one row of the panel in registers (two register sets: rows of
//     A, rows of X'); pivot and multipliers are wave-uniform v_readlane broadcasts from the lanes that
//     hold the diagonal block.
//   * T = 16 NB + E (50 = 3 x 16 + 2): the matrix instructions see NB x NB blocks only.  The last E rows and
//     columns ("tail") never enter a 16 x 16 block: the last panel is 16 + E columns wide, and the tail's
//     Schur terms (two columns of N and M) and its two columns of A^-1 are dot products in the one-row-per-
//     lane layout -- a handful of vector FMAs instead of a fourth, almost empty, block row (130 -> 78 MFMA).
//
// fp64 MFMA and fp64 vector FMA share the SIMD's double-precision units on gfx950 (tools/mfma_f64_bench.hip:
// a wave of back-to-back MFMAs leaves a co-resident wave one v_fma_f64 per 37 cycles), so the two kinds of
// work add up rather than overlap; what more waves per SIMD hide is the latency of the 50 dependent pivots.
//
// Layouts (v_mfma_f64_16x16x4, cdna_hip_programming.md 3): operand lane (r = lane & 15, g = lane >> 4)
// holds M[r][4 kk + g] for chunk kk -- the SAME registers serve as the A operand for M and as the B
// operand for M'; result register p of lane (c = lane & 15, g) holds D[g + 4 p][c].
#pragma once
#include <hip/hip_runtime.h>
#include "wave_tri.h"

typedef double hm_d4 __attribute__((ext_vector_type(4)));

template <int TP>
struct HmGeom {
    static constexpr int NB = TP / 16;        // 16-row blocks seen by the matrix pipe
    static constexpr int E = TP - 16 * NB;    // tail rows / columns
    static constexpr int WL = 16 + E;         // width of the last panel
    static_assert(NB >= 2 && NB <= 4 && E <= 8, "compiled window must be 16 NB + E, NB = 2..4, E <= 8");
    static constexpr int LDB = WL <= 18 ? 18 : (WL <= 22 ? 22 : 26);  // row stride: 2 mod 4 -> operand reads conflict-free
    static constexpr int ROWS = 16 * NB + 16 + E;  // rows of A at / below panel k (16 (NB - k) + E) + block rows of X' (16 (k + 1))
    static constexpr int O_SV = ROWS * LDB;    // sqrt(w), 64
    static constexpr int O_KVM = O_SV + 64;    // kvm[63 + d] = first column of K at |d|, d = -63..63
    static constexpr int O_DKV = O_KVM + 128;  // first column of dK / dln omega, 64
    static constexpr int O_NT = O_DKV + 64;    // ntail[E][64]: columns 16 NB + t of N
    static constexpr int O_Z = O_NT + (E > 0 ? E : 1) * 64;  // 32 zeros
    static constexpr int TASK = O_Z + 32;
    __host__ __device__ static constexpr int width(int k) { return k == NB - 1 ? WL : 16; }
    __host__ __device__ static constexpr int chunks(int k) { return (width(k) + 3) / 4; }
};

__device__ __forceinline__ double hm_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    double e = fma(-d * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    e = fma(-d * y, y, 1.0);
    y = fma(y * 0.5, e, y);
    return y;
}

// buf: TASK doubles of LDS owned by this wave; on entry the tables at O_SV, O_KVM (jitter eps added at distance 0),
// O_DKV and the zeros at O_Z are filled (sqrt(w) zero beyond the rows present).  A pivot of A that is not positive and finite makes tr and cs NaN.  tr = tr(A^-1) over all
// TP rows (the caller subtracts the identity padding's share), cs = sum_jk s_j s_k dK_jk (A^-1)_jk; both
// per-lane partials.
//
// KMODE (the K block of the round): the same elimination applied to K itself -- sv holds 1 for the rows present and
// 0 beyond, rows >= tr_k carry a unit diagonal (identity padding) -- and instead of the two sums the routine leaves
// K^-1 as a full symmetric TP x ldk matrix in `kl` (LDS) and returns log det chol(K) in tr.
//
// EXPL (with KMODE, tr_k = TP): the matrix is not generated from the tables but read from `aex` (LDS, full symmetric
// TP x lda, its own diagonal included): the inverse of an explicit SPD matrix, used for the Schur complement of the
// long-window segment kernel (hstep.hip, hstep_seg_big).  In place (kl == aex) is fine: every panel's entries are read
// before the inverse is written at the end.
template <int TP, bool KMODE = false, bool EXPL = false>
__device__ __forceinline__ bool hstep_task_mfma(double* buf, double eps, int lane, double& tr, double& cs,
                                                int tr_k = 0, double* kl = nullptr, int ldk = 0,
                                                const double* aex = nullptr, int lda = 0) {
    static_assert(!EXPL || KMODE, "the explicit-matrix form returns the inverse (KMODE)");
    using G = HmGeom<TP>;
    constexpr int NB = G::NB, E = G::E, LDB = G::LDB, WL = G::WL, TB = 16 * NB;
    const double* sv = buf + G::O_SV;
    const double* kvm = buf + G::O_KVM;
    const double* dkv = buf + G::O_DKV;
    double* ntail = buf + G::O_NT;
    const double* zrow = buf + G::O_Z;
    const int c = lane & 15, g = lane >> 4;
    hm_d4 N[NB][NB], M[NB][NB], P[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            N[i][j] = hm_d4{0.0, 0.0, 0.0, 0.0};
            M[i][j] = hm_d4{0.0, 0.0, 0.0, 0.0};
            P[i][j] = hm_d4{0.0, 0.0, 0.0, 0.0};
        }
    double mt[E > 0 ? E : 1];  // columns 16 NB + t of M, lane <-> row of X'
#pragma unroll
    for (int t = 0; t < (E > 0 ? E : 1); ++t) mt[t] = 0.0;
    double ra[WL], rx[WL];
    double logdet = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int col0 = 16 * k;
        const int W = G::width(k);
        const bool last = k == NB - 1;
        const int rowsA = TP - col0;             // rows of A at and below the diagonal block (lane <-> row col0 + lane)
        const int nX = last ? TP : 16 * (k + 1);  // rows of X' with entries in this panel (lane <-> row)
        double* bufA = buf;
        double* bufX = buf + (TP - col0) * LDB;
        // ---- 1. Schur terms of this panel from the result registers -> LDS ----
        // N and M accumulate the NEGATIVE Schur terms (neg:[0,1,0] on the matrix instruction is free), and the unit
        // diagonals are planted here, four entries per lane: the diagonal block of A gets diag(one) - N, the rows
        // col0 .. col0 + 15 of X' get I.  The row assembly below is then one FMA (A) and a plain load (X') per entry
        // instead of compare + select + subtract on top (6 of 11 vector operations per column pair).
        if (k > 0) {
#pragma unroll
            for (int bi = k; bi < NB; ++bi)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    double val = N[bi][k][p];
                    if (bi == k) {
                        const double one = KMODE ? (col0 + c < tr_k ? 0.0 : 1.0) : 1.0;
                        val += (g + 4 * p == c) ? one : 0.0;
                    }
                    bufA[(16 * (bi - k) + g + 4 * p) * LDB + c] = val;
                }
#pragma unroll
            for (int i = 0; i < k; ++i)
#pragma unroll
                for (int p = 0; p < 4; ++p) bufX[(16 * i + g + 4 * p) * LDB + c] = M[i][k][p];
#pragma unroll
            for (int p = 0; p < 4; ++p) bufX[(col0 + g + 4 * p) * LDB + c] = (g + 4 * p == c) ? 1.0 : 0.0;
            tri_wave_sync();
        }
        // ---- 2. one row per lane: ra = row (col0 + lane) of A - N, rx = row `lane` of I - M ----
        {
            const int i = col0 + lane;
            const double si = sv[i < 64 ? i : 63];
            const double one = KMODE ? (i < tr_k ? 0.0 : 1.0) : 1.0;
            // row of N: block rows from bufA, tail rows from ntail (N is symmetric)
            const double* pn = (E > 0 && i >= TB) ? ntail + (i - TB < E ? i - TB : 0) * 64 + col0
                                                  : bufA + (lane < TB - col0 ? lane : 0) * LDB;
            // an opaque copy of the lane index per panel: the unit-diagonal selects below are otherwise common
            // subexpressions of all panels, computed once at the top and kept in ~40 registers to the end
            int ln = lane;
            asm volatile("" : "+v"(ln));
            const double* pm = lane < col0 + 16 ? bufX + lane * LDB : zrow;  // (k > 0) -M, I on the diagonal block rows, nothing below
            const double* pk = kvm + 63 + lane;  // kvm[63 + lane - q] = K[|i - (col0 + q)|], jitter included at distance 0
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const double2 sj = *reinterpret_cast<const double2*>(sv + col0 + q);
                double2 nv = double2{0.0, 0.0}, mv = double2{0.0, 0.0};
                if (k > 0) {
                    nv = *reinterpret_cast<const double2*>(pn + q);
                    mv = *reinterpret_cast<const double2*>(pm + q);
                }
                // A = diag(one) + S K S - N: the addend comes ready from LDS (first panel: the unit diagonal by select)
                if constexpr (EXPL) {
                    const double2 av = *reinterpret_cast<const double2*>(aex + (i < TP ? i : TP - 1) * lda + col0 + q);
                    ra[q] = av.x + nv.x;
                    ra[q + 1] = av.y + nv.y;
                    if (k > 0) {
                        rx[q] = mv.x;
                        rx[q + 1] = mv.y;
                    } else {
                        rx[q] = (ln == q ? 1.0 : 0.0);
                        rx[q + 1] = (ln == q + 1 ? 1.0 : 0.0);
                    }
                } else if (k > 0) {
                    ra[q] = fma(si * sj.x, pk[-q], nv.x);
                    ra[q + 1] = fma(si * sj.y, pk[-q - 1], nv.y);
                    rx[q] = mv.x;
                    rx[q + 1] = mv.y;
                } else {
                    ra[q] = fma(si * sj.x, pk[-q], (ln == q ? one : 0.0));
                    ra[q + 1] = fma(si * sj.y, pk[-q - 1], (ln == q + 1 ? one : 0.0));
                    rx[q] = (ln == q ? 1.0 : 0.0);
                    rx[q + 1] = (ln == q + 1 ? 1.0 : 0.0);
                }
            }
            if (last) {
#pragma unroll
                for (int t = 0; t < E; ++t) {  // tail columns: N from ntail[t][row], M from mt[t]
                    const double sj = sv[TB + t];
                    const double nv = ntail[t * 64 + (i < 64 ? i : 63)];
                    ra[16 + t] = fma(si * sj, pk[-16 - t], (ln == 16 + t ? one : 0.0) + nv);  // ntail, mt hold -N, -M
                    rx[16 + t] = (ln == TB + t ? 1.0 : 0.0) + mt[t];
                }
            }
        }
        // pin the panel entries here: left alone, the compiler sinks their assembly into the elimination steps
        // and keeps every loaded table value alive across them (measured: 530 spilled VGPRs)
#pragma unroll
        for (int q = 0; q < WL; ++q)
            if (q < W) asm volatile("" : "+v"(ra[q]), "+v"(rx[q]));
        __builtin_amdgcn_sched_barrier(0);
        // ---- 3. elimination of the panel.  Column j takes the updates of columns m <= j - 2 "lazily", one step
        // ahead and with multipliers L[j][m] broadcast from an LDS copy of the diagonal block (they are final by
        // then, so the LDS round trip is off the pivot chain); only the last update (m = j - 1) and the pivot
        // itself are v_readlane broadcasts.
        {
            double* Ld = buf;  // rows of the diagonal block (lanes < W), stride LDD; the panel buffer is free meanwhile
            constexpr int LDD = (WL + 1) & ~1;
            double yprod = 1.0;  // KMODE: product of the reciprocal pivots of this panel
#pragma unroll
            for (int j = 0; j < WL; ++j) {
                if (j < W) {
                    if (j > 0) {
                        const double lv = tri_readlane(ra[j - 1], j);
                        ra[j] = fma(-ra[j - 1], lv, ra[j]);
                        rx[j] = fma(-rx[j - 1], lv, rx[j]);
                    }
                    const double d = tri_readlane(ra[j], j);
                    double y = __builtin_amdgcn_rsq(d);
                    if (j + 1 < W && j >= 1) {  // column j + 1 <- columns 0 .. j - 1
                        const double* row = Ld + (j + 1) * LDD;
                        // one chain per register set: with three waves per SIMD the FMA latency is covered
                        double a0 = ra[j + 1], x0 = rx[j + 1];
#pragma unroll
                        for (int m = 0; m + 1 < j; m += 2) {
                            const double2 v = *reinterpret_cast<const double2*>(row + m);
                            a0 = fma(-ra[m], v.x, a0);
                            x0 = fma(-rx[m], v.x, x0);
                            a0 = fma(-ra[m + 1], v.y, a0);
                            x0 = fma(-rx[m + 1], v.y, x0);
                        }
                        if (j & 1) {
                            const double v = row[j - 1];
                            a0 = fma(-ra[j - 1], v, a0);
                            x0 = fma(-rx[j - 1], v, x0);
                        }
                        ra[j + 1] = a0;
                        rx[j + 1] = x0;
                        // evaluate the X' half HERE: unpinned, the compiler sinks the whole rx chain below the panel
                        // and carries every multiplier to it through scratch
                        asm volatile("" : "+v"(ra[j + 1]), "+v"(rx[j + 1]));
                    }
                    {   // one third-order step: y (1 + e / 2 + 3 e^2 / 8), e = 1 - d y^2 (v_rsq_f64 starts at ~2^-26)
                        const double e = fma(-d * y, y, 1.0);
                        y = fma(y * e, fma(0.375, e, 0.5), y);
                    }
                    ra[j] *= y;
                    rx[j] *= y;
                    if (KMODE) {
                        yprod *= y;
                        asm volatile("" : "+v"(yprod));  // (else the product is formed at the end from fifty kept values)
                    }
                    asm volatile("" : "+v"(ra[j]), "+v"(rx[j]));
                    if (j + 1 < W) {
                        if (lane < W) Ld[lane * LDD + j] = ra[j];
                        tri_wave_order();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (KMODE) logdet -= log(yprod);  // log det chol(K) = -sum log(1 / L_jj)
        }
        // ---- 4. finished rows back to LDS (block rows for the operands, tail rows for the dot products) ----
        if (!last && lane >= 16 && lane < rowsA) {
            double* pa = bufA + lane * LDB;
#pragma unroll
            for (int q = 0; q < 16; q += 2) *reinterpret_cast<double2*>(pa + q) = double2{ra[q], ra[q + 1]};
        }
        if (lane < (nX < TB ? nX : TB)) {
            double* px = bufX + lane * LDB;
#pragma unroll
            for (int q = 0; q + 1 < WL; q += 2)
                if (q < W) *reinterpret_cast<double2*>(px + q) = double2{rx[q], rx[q + 1]};
        }
        tri_wave_sync();
        // ---- 5. tail columns of the Schur terms: dot products with the tail rows of L (broadcast from LDS) ----
        if (!last && E > 0) {
#pragma unroll
            for (int t = 0; t < E; ++t) {
                const double* lt = bufA + (TB + t - col0) * LDB;
                double n0 = 0.0, n1 = 0.0, m0 = 0.0, m1 = 0.0;
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    const double2 l2 = *reinterpret_cast<const double2*>(lt + q);
                    n0 = fma(ra[q], l2.x, n0);
                    n1 = fma(ra[q + 1], l2.y, n1);
                    m0 = fma(rx[q], l2.x, m0);
                    m1 = fma(rx[q + 1], l2.y, m1);
                }
                mt[t] -= (lane < nX) ? m0 + m1 : 0.0;
                asm volatile("" : "+v"(mt[t]));  // evaluate now (else the dot product waits, operands and all, for the last panel)
                if (lane >= 16 && lane < rowsA) {
                    double* pn = ntail + t * 64 + col0 + lane;
                    *pn = (k > 0 ? *pn : 0.0) - (n0 + n1);
                }
            }
        }
        constexpr int MAXCH = (WL + 3) / 4;
        double opL[NB][4], opX[NB][MAXCH];
        const int nch = G::chunks(k);
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) opL[bi][kk] = (bi > k) ? bufA[(16 * (bi - k) + c) * LDB + 4 * kk + g] : 0.0;
#pragma unroll
            for (int kk = 0; kk < MAXCH; ++kk) {
                opX[bi][kk] = 0.0;
                if (bi <= k && kk < nch) {
                    if (4 * kk + 4 <= W) {
                        opX[bi][kk] = bufX[(16 * bi + c) * LDB + 4 * kk + g];
                    } else {  // last chunk of the wide panel: columns beyond W are zero
                        const bool in = 4 * kk + g < W;
                        const double v = bufX[(16 * bi + c) * LDB + (in ? 4 * kk + g : 0)];
                        opX[bi][kk] = in ? v : 0.0;
                    }
                }
            }
        }
        tri_wave_order();
        __builtin_amdgcn_sched_barrier(0);
        // ---- 6. rank-W updates on the matrix pipe; what the next panel needs goes first ----
        if (!last) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int bi = k + 1; bi < NB; ++bi)
                    N[bi][k + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(opL[bi][kk], opL[k + 1][kk], N[bi][k + 1], 0, 0, 2);
#pragma unroll
                for (int i = 0; i <= k; ++i)
                    M[i][k + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(opX[i][kk], opL[k + 1][kk], M[i][k + 1], 0, 0, 2);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int bj = k + 2; bj < NB; ++bj) {
#pragma unroll
                    for (int bi = bj; bi < NB; ++bi)
                        N[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(opL[bi][kk], opL[bj][kk], N[bi][bj], 0, 0, 2);
#pragma unroll
                    for (int i = 0; i <= k; ++i)
                        M[i][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(opX[i][kk], opL[bj][kk], M[i][bj], 0, 0, 2);
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < MAXCH; ++kk) {
            if (kk < nch) {
#pragma unroll
                for (int i = 0; i <= k; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j)
                        P[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(opX[i][kk], opX[j][kk], P[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (KMODE) {
        // ---- K^-1 = P as a full symmetric matrix in LDS ----
#pragma unroll
        for (int bj = 0; bj < NB; ++bj)
#pragma unroll
            for (int bi = bj; bi < NB; ++bi)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int i = 16 * bi + g + 4 * p, j = 16 * bj + c;
                    kl[i * ldk + j] = P[bi][bj][p];
                    if (bi != bj) kl[j * ldk + i] = P[bi][bj][p];
                }
        if (E > 0) {
#pragma unroll
            for (int t = 0; t < E; ++t) {
                double p = 0.0;
#pragma unroll
                for (int u = t; u < E; ++u) p = fma(rx[16 + u], tri_readlane(rx[16 + u], TB + t), p);
                if (lane < TP) {
                    kl[lane * ldk + TB + t] = p;
                    kl[(TB + t) * ldk + lane] = p;
                }
            }
        }
        tr = logdet;
        cs = 0.0;
        return true;
    }
    // ---- weighted sums over A^-1: block part from P (lower blocks; off-diagonal blocks count twice) ----
    tr = 0.0;
    cs = 0.0;
    double cd = 0.0;
#pragma unroll
    for (int bj = 0; bj < NB; ++bj) {
        const double sj = sv[16 * bj + c];
#pragma unroll
        for (int bi = bj; bi < NB; ++bi) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int il = g + 4 * p;
                const int i = 16 * bi + il;
                const int dd = i - (16 * bj + c);
                const double wgt = (sv[i] * sj) * dkv[dd < 0 ? -dd : dd];
                if (bi == bj) {
                    cd = fma(P[bi][bj][p], wgt, cd);
                    tr += (il == c) ? P[bi][bj][p] : 0.0;
                } else {
                    cs = fma(P[bi][bj][p], wgt, cs);
                }
            }
        }
    }
    // ---- tail columns of A^-1 in the row-per-lane layout (rx = rows of X' after the last panel) ----
    if (E > 0) {
        const double si = sv[lane];
#pragma unroll
        for (int t = 0; t < E; ++t) {
            double p = 0.0;  // (A^-1)[lane][TB + t] = sum_{u >= t} X'[lane][TB + u] X'[TB + t][TB + u]
#pragma unroll
            for (int u = t; u < E; ++u) p = fma(rx[16 + u], tri_readlane(rx[16 + u], TB + t), p);
            const int dd = TB + t - lane;
            const double wgt = (si * sv[TB + t]) * dkv[dd > 0 ? dd : 0];
            if (lane < TB + t) cs = fma(p, wgt, cs);
            if (lane == TB + t) tr += p;
        }
    }
    cs = fma(2.0, cs, cd);
    return true;  // a pivot that is not positive and finite turns tr and cs into NaN (the caller checks)
}


// ---------------------------------------------------------------------------------------------------------
// TP = 50, ONE register set: buffer row = lane.
//
// In panel k (columns col0 = 16 k ...) the elimination touches the rows of A at and below the diagonal block
// (nA = 50 - col0 of them) and the rows of X' = L^-T that have entries in these columns (col0 + W).  That is 66 rows
// (68 in the last, 18-wide, panel) and the routine above gives the two kinds a register set each, every vector
// instruction of the elimination issued twice.  But the last rows of the diagonal block of X' are trivial -- row
// col0 + 15 is (0 .. 0, y15), row col0 + 14 is (0 .. 0, y14, -y14 y15 L[15][14]) -- so 64 lanes are enough:
//     lane < nA:   row col0 + lane of A                       (rows of the panel buffer 0 .. nA - 1)
//     lane >= nA:  row lane - nA of X' (the 16 k rows of the earlier blocks, then rows 0 .. 13 of this block:
//                  unit rows on entry)                        (rows nA .. 63 of the panel buffer)
//     "special" rows col0 + 14 .. col0 + W - 1 of X': a few wave-uniform operations from the reciprocal pivots and
//                  the multipliers L[j][m], j > m >= 14, after the elimination (rows 64, 65 of the panel buffer).
// The row a lane holds moves 16 lanes down per panel for BOTH kinds, so the panel buffer is simply indexed by the lane,
// and the tail-column accumulators (columns 48, 49 of N for the rows of A, of M for the rows of X') ride along in two
// registers per lane, moved 16 lanes down between panels (no LDS tables).  Half the elimination FMAs, half the panel
// assembly, 36 registers fewer than the two-set routine; same matrix instructions.
//
// LDS: 10 KB per segment so that FOUR workgroups of four waves share a CU (the round is bound by the latency of its
// dependent chains: PMC, profiles/r2): per wave the panel buffer (66 x 18) and sqrt(w) (52); per workgroup -- all its
// segments belong to one evaluation -- the first column of K at distances -17 .. 49 (kvs[17 + d], jitter at 0) and of
// dK (52).  The 2 x 16 transposition scratch of the tail accumulators aliases rows 64, 65 of the panel buffer (written
// after the elimination, the scratch is read before it).
struct HmGeom50 {
    static constexpr int LDB = 18, ROWS = 66;
    static constexpr int O_SV = ROWS * LDB;  // 1188
    static constexpr int SVN = 52;
    static constexpr int TASK = O_SV + SVN;  // 1240 doubles per wave
    static constexpr int KVN = 68, DKN = 52; // shared per workgroup
    static constexpr int SHARED = KVN + DKN;
};

template <bool KMODE = false>
__device__ __forceinline__ bool hstep_task_mfma50(double* buf, const double* kvs, const double* dkv, double eps, int lane,
                                                  double& tr, double& cs, int tr_k = 0, double* kl = nullptr, int ldk = 0) {
    constexpr int TP = 50;
    using G = HmGeom<TP>;
    constexpr int NB = G::NB, E = G::E, LDB = G::LDB, WL = G::WL, TB = 16 * NB;
    static_assert(NB == 3 && E == 2 && LDB == 18 && WL == 18 && HmGeom50::LDB == LDB, "written for 50 = 3 x 16 + 2");
    const double* sv = buf + HmGeom50::O_SV;
    double* zs = buf + 64 * LDB;  // 2 x 16: tail-column accumulators of the rows of the diagonal block, as rows
    const int c = lane & 15, g = lane >> 4;
    hm_d4 N[NB][NB], M[NB][NB], P[NB][NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            N[i][j] = hm_d4{0.0, 0.0, 0.0, 0.0};
            M[i][j] = hm_d4{0.0, 0.0, 0.0, 0.0};
            P[i][j] = hm_d4{0.0, 0.0, 0.0, 0.0};
        }
    double tl[E] = {0.0, 0.0};  // minus the tail columns (48, 49) of N (lanes of A) / of M (lanes of X') for this lane's row
    double r[WL];
    double xs[4][4];  // special rows of the last panel: xs[a][b] = X'[46 + a][46 + b], b >= a
    double logdet = 0.0;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int col0 = 16 * k;
        const int W = G::width(k);
        const bool last = k == NB - 1;
        const int nA = TP - col0;  // lanes of A; X' row of lane >= nA: lane - nA
        double* bufA = buf;
        double* bufX = buf + nA * LDB;
        // ---- 1. Schur terms of this panel from the result registers -> LDS (negative, unit diagonals planted) ----
        if (k > 0) {
#pragma unroll
            for (int bi = k; bi < NB; ++bi)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    double val = N[bi][k][p];
                    if (bi == k) {
                        const double one = KMODE ? (col0 + c < tr_k ? 0.0 : 1.0) : 1.0;
                        val += (g + 4 * p == c) ? one : 0.0;
                    }
                    bufA[(16 * (bi - k) + g + 4 * p) * LDB + c] = val;
                }
#pragma unroll
            for (int i = 0; i < k; ++i)
#pragma unroll
                for (int p = 0; p < 4; ++p) bufX[(16 * i + g + 4 * p) * LDB + c] = M[i][k][p];
#pragma unroll
            for (int p = 0; p < 4; ++p) bufX[(col0 + g + 4 * p) * LDB + c] = (g + 4 * p == c) ? 1.0 : 0.0;
            // the tail rows of A read their row of N from the tail-column accumulators of the rows col0 .. col0 + 15
            if (lane < 16) {
                zs[lane] = tl[0];
                zs[16 + lane] = tl[1];
            }
            tri_wave_sync();
        }
        // ---- 2. one row per lane ----
        {
            const bool isA = lane < nA;
            const int i = col0 + lane;  // row of A (lanes of A)
            const double si = isA ? sv[i] : 0.0;  // lanes of X': r = addend exactly (table values are finite)
            int ln = lane;  // opaque per panel (see the two-set routine)
            asm volatile("" : "+v"(ln));
            const double* pk = kvs + 17 + (lane < TP ? lane : TP - 1);  // pk[-q] = K[|i - (col0 + q)|], jitter at distance 0
            const double* pr = (lane >= TB - col0 && lane < nA) ? zs + (lane - (TB - col0)) * 16 : buf + lane * LDB;
            // first panel: unit entry by select (rows of A: diag(one); lanes >= 50: unit rows 0 .. 13 of X')
            const int upos = ln < nA ? ln : ln - nA;
            const double uval = KMODE ? ((isA && i < tr_k) ? 0.0 : 1.0) : 1.0;
#pragma unroll
            for (int q = 0; q < 16; q += 2) {
                const double2 sj = *reinterpret_cast<const double2*>(sv + col0 + q);
                if (k > 0) {
                    const double2 av = *reinterpret_cast<const double2*>(pr + q);
                    r[q] = fma(si * sj.x, pk[-q], av.x);
                    r[q + 1] = fma(si * sj.y, pk[-q - 1], av.y);
                } else {
                    r[q] = fma(si * sj.x, pk[-q], (upos == q ? uval : 0.0));
                    r[q + 1] = fma(si * sj.y, pk[-q - 1], (upos == q + 1 ? uval : 0.0));
                }
            }
            if (last) {
#pragma unroll
                for (int t = 0; t < E; ++t) {  // tail columns: -N (rows of A, + the unit diagonal) / -M (rows of X') from tl
                    const double sj = sv[TB + t];
                    r[16 + t] = fma(si * sj, pk[-16 - t], (ln == 16 + t ? uval : 0.0) + tl[t]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < WL; ++q)
            if (q < W) asm volatile("" : "+v"(r[q]));
        __builtin_amdgcn_sched_barrier(0);
        // ---- 3. elimination of the panel (lazy updates one step ahead, multipliers from the LDS copy of the finished
        // columns of the diagonal block = rows 0 .. W - 1 of the panel buffer) ----
        double ysp[4] = {0.0, 0.0, 0.0, 0.0};  // reciprocal pivots of columns 14 ..
        {
            double* Ld = buf;
            constexpr int LDD = LDB;
            double yprod = 1.0;
#pragma unroll
            for (int j = 0; j < WL; ++j) {
                if (j < W) {
                    if (j > 0) {
                        const double lv = tri_readlane(r[j - 1], j);
                        r[j] = fma(-r[j - 1], lv, r[j]);
                    }
                    const double d = tri_readlane(r[j], j);
                    double y = __builtin_amdgcn_rsq(d);
                    if (j + 1 < W && j >= 1) {  // column j + 1 <- columns 0 .. j - 1
                        const double* row = Ld + (j + 1) * LDD;
                        double a0 = r[j + 1], a1 = 0.0;
#pragma unroll
                        for (int m = 0; m + 1 < j; m += 2) {
                            const double2 v = *reinterpret_cast<const double2*>(row + m);
                            a0 = fma(-r[m], v.x, a0);
                            a1 = fma(-r[m + 1], v.y, a1);
                        }
                        if (j & 1) a0 = fma(-r[j - 1], row[j - 1], a0);
                        r[j + 1] = a0 + a1;
                        asm volatile("" : "+v"(r[j + 1]));
                    }
                    {
                        const double e = fma(-d * y, y, 1.0);
                        y = fma(y * e, fma(0.375, e, 0.5), y);
                    }
                    r[j] *= y;
                    if (j >= 14) ysp[j - 14] = y;
                    if (KMODE) {
                        yprod *= y;
                        asm volatile("" : "+v"(yprod));
                    }
                    asm volatile("" : "+v"(r[j]));
                    if (j + 1 < W) {
                        if (lane < W) Ld[lane * LDD + j] = r[j];
                        tri_wave_order();
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            if (KMODE) logdet -= log(yprod);
        }
        // ---- 3b. the special rows of X' (wave-uniform): row 14 + a of this block, columns 14 + b, b >= a ----
        //     x[a][a] = y_a,   x[a][b] = -y_b sum_{a <= m < b} x[a][m] L[14 + b][14 + m]
        double sp[4][4];
        {
            const int NS = W - 14;  // 2, or 4 in the last panel
            double lm[4][4];        // lm[b][m] = L[14 + b][14 + m], m < b (columns <= W - 2 of the LDS copy)
#pragma unroll
            for (int b = 1; b < 4; ++b)
#pragma unroll
                for (int m = 0; m < 3; ++m) lm[b][m] = (b < NS && m < b) ? buf[(14 + b) * LDB + 14 + m] : 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    sp[a][b] = 0.0;
                    if (a < NS && b < NS && b >= a) {
                        if (b == a) {
                            sp[a][b] = ysp[a];
                        } else {
                            double s = 0.0;
#pragma unroll
                            for (int m = a; m < b; ++m) s = fma(sp[a][m], lm[b][m], s);
                            sp[a][b] = -ysp[b] * s;
                        }
                    }
                }
            if (last) {
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) xs[a][b] = sp[a][b];
            }
        }
        tri_wave_order();
        // ---- 4. finished rows back to LDS: row of the panel buffer = lane (rows of A below the diagonal block, rows of
        // X'); the special rows go to rows 64, 65, lane <-> column ----
        if (lane >= (last ? nA : 16)) {
            double* pw = buf + lane * LDB;
#pragma unroll
            for (int q = 0; q + 1 < WL; q += 2)
                if (q < W) *reinterpret_cast<double2*>(pw + q) = double2{r[q], r[q + 1]};
        }
        if (lane < W) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                double v = 0.0;
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    if (b >= a && 14 + b < W) v = (lane == 14 + b) ? sp[a][b] : v;
                buf[(64 + a) * LDB + lane] = v;
            }
        }
        tri_wave_sync();
        // ---- 5. tail columns of the Schur terms: dot products with the tail rows of L (broadcast from LDS) ----
        if (!last) {
#pragma unroll
            for (int t = 0; t < E; ++t) {
                const double* lt = bufA + (TB + t - col0) * LDB;
                double d0 = 0.0, d1 = 0.0;
                double2 l2 = double2{0.0, 0.0};
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    l2 = *reinterpret_cast<const double2*>(lt + q);
                    d0 = fma(r[q], l2.x, d0);
                    d1 = fma(r[q + 1], l2.y, d1);
                }
                tl[t] -= d0 + d1;
                // rows col0 + 14, col0 + 15 of X' (l2 = L[tail][14], L[tail][15] now)
                const double m14 = fma(sp[0][0], l2.x, sp[0][1] * l2.y);
                const double m15 = sp[1][1] * l2.y;
                // the row of a lane moves 16 lanes down for the next panel; the special rows arrive at lanes 48, 49, new
                // unit rows (no history) above them
                double nx = __shfl_down(tl[t], 16, 64);
                nx = lane >= 50 ? 0.0 : nx;
                nx = lane == 48 ? -m14 : nx;
                nx = lane == 49 ? -m15 : nx;
                tl[t] = nx;
                asm volatile("" : "+v"(tl[t]));
            }
        }
        constexpr int MAXCH = (WL + 3) / 4;
        double opL[NB][4], opX[NB][MAXCH];
        const int nch = G::chunks(k);
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) opL[bi][kk] = (bi > k) ? bufA[(16 * (bi - k) + c) * LDB + 4 * kk + g] : 0.0;
#pragma unroll
            for (int kk = 0; kk < MAXCH; ++kk) {
                opX[bi][kk] = 0.0;
                if (bi <= k && kk < nch) {
                    if (4 * kk + 4 <= W) {
                        opX[bi][kk] = bufX[(16 * bi + c) * LDB + 4 * kk + g];
                    } else {
                        const bool in = 4 * kk + g < W;
                        const double v = bufX[(16 * bi + c) * LDB + (in ? 4 * kk + g : 0)];
                        opX[bi][kk] = in ? v : 0.0;
                    }
                }
            }
        }
        tri_wave_order();
        __builtin_amdgcn_sched_barrier(0);
        // ---- 6. rank-W updates on the matrix pipe; what the next panel needs goes first ----
        if (!last) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int bi = k + 1; bi < NB; ++bi)
                    N[bi][k + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(opL[bi][kk], opL[k + 1][kk], N[bi][k + 1], 0, 0, 2);
#pragma unroll
                for (int i = 0; i <= k; ++i)
                    M[i][k + 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(opX[i][kk], opL[k + 1][kk], M[i][k + 1], 0, 0, 2);
            }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
                for (int bj = k + 2; bj < NB; ++bj) {
#pragma unroll
                    for (int bi = bj; bi < NB; ++bi)
                        N[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(opL[bi][kk], opL[bj][kk], N[bi][bj], 0, 0, 2);
#pragma unroll
                    for (int i = 0; i <= k; ++i)
                        M[i][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(opX[i][kk], opL[bj][kk], M[i][bj], 0, 0, 2);
                }
            }
        }
#pragma unroll
        for (int kk = 0; kk < MAXCH; ++kk) {
            if (kk < nch) {
#pragma unroll
                for (int i = 0; i <= k; ++i)
#pragma unroll
                    for (int j = 0; j <= i; ++j)
                        P[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(opX[i][kk], opX[j][kk], P[i][j], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- tail columns of A^-1: (A^-1)[rho][48 + t] = sum_{u >= t} X'[rho][48 + u] X'[48 + t][48 + u].  Lanes >= 18 hold
    // the rows rho = lane - 18 of X' (r[16], r[17] = columns 48, 49); lanes 0 .. 3 stand in for the special rows 46 .. 49 ----
    const bool xrow = lane >= 18 || lane < 4;
    const int rho = lane >= 18 ? lane - 18 : 46 + lane;
    double x48 = r[16], x49 = r[17];
    if (lane < 18) {
        x48 = lane == 0 ? xs[0][2] : (lane == 1 ? xs[1][2] : (lane == 2 ? xs[2][2] : 0.0));
        x49 = lane == 0 ? xs[0][3] : (lane == 1 ? xs[1][3] : (lane == 2 ? xs[2][3] : xs[3][3]));
    }
    const double pt[E] = {fma(x48, xs[2][2], x49 * xs[2][3]), x49 * xs[3][3]};
    if constexpr (KMODE) {
#pragma unroll
        for (int bj = 0; bj < NB; ++bj)
#pragma unroll
            for (int bi = bj; bi < NB; ++bi)
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int i = 16 * bi + g + 4 * p, j = 16 * bj + c;
                    kl[i * ldk + j] = P[bi][bj][p];
                    if (bi != bj) kl[j * ldk + i] = P[bi][bj][p];
                }
#pragma unroll
        for (int t = 0; t < E; ++t) {
            if (xrow) {
                kl[rho * ldk + TB + t] = pt[t];
                kl[(TB + t) * ldk + rho] = pt[t];
            }
        }
        tr = logdet;
        cs = 0.0;
        return true;
    }
    tr = 0.0;
    cs = 0.0;
    double cd = 0.0;
#pragma unroll
    for (int bj = 0; bj < NB; ++bj) {
        const double sj = sv[16 * bj + c];
#pragma unroll
        for (int bi = bj; bi < NB; ++bi) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int il = g + 4 * p;
                const int i = 16 * bi + il;
                const int dd = i - (16 * bj + c);
                const double wgt = (sv[i] * sj) * dkv[dd < 0 ? -dd : dd];
                if (bi == bj) {
                    cd = fma(P[bi][bj][p], wgt, cd);
                    tr += (il == c) ? P[bi][bj][p] : 0.0;
                } else {
                    cs = fma(P[bi][bj][p], wgt, cs);
                }
            }
        }
    }
    {
        const double si = sv[rho];
#pragma unroll
        for (int t = 0; t < E; ++t) {
            const int dd = TB + t - rho;
            const double wgt = (si * sv[TB + t]) * dkv[dd > 0 ? dd : 0];
            if (xrow && rho < TB + t) cs = fma(pt[t], wgt, cs);
            if (xrow && rho == TB + t) tr += pt[t];
        }
    }
    cs = fma(2.0, cs, cd);
    return true;
}


// The traces of the K block against the second moments C = sum_i mu_i mu_i' of one latent, on the matrix pipe:
//     quad = tr(K^-1 C),   gq = tr(K^-1 dK K^-1 C) = <C K^-1, K^-1 dK>_F
// with K^-1 in LDS (`kl`, full symmetric, stride ldk = 2 mod 4), C in global memory (TP x TP, zero beyond the rows
// present) and dK Toeplitz from its first column `dkv` (zero beyond tr).  Wave `wid` of `nwv` takes the 16-row
// block rows wid, wid + nwv, ...; returns this wave's partial sums (per lane; the caller reduces over the wave).
// `pre` (optional): the C operands of block row `wid`, fetched by the caller BEFORE it waited for K^-1 (hstep_kblock_fetch:
// the waves that do not factor are idle meanwhile, and thirteen dependent trips to L2 leave the critical path of a
// round -- the K block is what a round costs when the segments do not fill the chip: C1, C2, a rank's shard at 8 GPUs)
template <int TP>
__device__ __forceinline__ void hstep_kblock_fetch(const double* __restrict__ C, int lane, int wid,
                                                   double (&pre)[(TP + 3) / 4]) {
    constexpr int NK = (TP + 3) / 4;
    const int c = lane & 15, g = lane >> 4;
    const int ra_ = 16 * wid + c;
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
        const int kcol = 4 * kk + g;
        pre[kk] = (ra_ < TP && kcol < TP) ? C[ra_ * TP + kcol] : 0.0;
    }
}

template <int TP>
__device__ __forceinline__ void hstep_kblock_products(const double* kl, int ldk, const double* __restrict__ C,
                                                      const double* dkv, int tr_k, int lane, int wid, int nwv,
                                                      double& quad, double& gq,
                                                      const double (*pre)[(TP + 3) / 4] = nullptr) {
    constexpr int NBK = (TP + 15) / 16, NK = (TP + 3) / 4;
    const int c = lane & 15, g = lane >> 4;
    quad = 0.0;
    gq = 0.0;
    if (pre != nullptr && nwv >= NBK) {  // one block row per wave, operands of C already in registers
        const int a = wid;
        if (a >= NBK) return;
        hm_d4 Eb[NBK], Gb[NBK];
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            Eb[b] = hm_d4{0.0, 0.0, 0.0, 0.0};
            Gb[b] = hm_d4{0.0, 0.0, 0.0, 0.0};
        }
        const int ra_ = 16 * a + c;
        const bool rin = ra_ < TP;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int kcol = 4 * kk + g;
            const bool kin = kcol < TP;
            const double opC = (*pre)[kk];
            const double opKa = (rin && kin) ? kl[ra_ * ldk + kcol] : 0.0;
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                const int rb_ = 16 * b + c;
                const bool bin = rb_ < TP && kin;
                const double opKb = bin ? kl[rb_ * ldk + kcol] : 0.0;
                const int dd = kcol - rb_;
                const double opD = (bin && kcol < tr_k && rb_ < tr_k) ? dkv[dd < 0 ? -dd : dd] : 0.0;
                Eb[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(opC, opKb, Eb[b], 0, 0, 0);
                Gb[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(opKa, opD, Gb[b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int b = 0; b < NBK; ++b)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                gq = fma(Eb[b][p], Gb[b][p], gq);
                if (b == a && g + 4 * p == c) quad += Eb[b][p];
            }
        return;
    }
    for (int a = wid; a < NBK; a += nwv) {
        hm_d4 Eb[NBK], Gb[NBK];
#pragma unroll
        for (int b = 0; b < NBK; ++b) {
            Eb[b] = hm_d4{0.0, 0.0, 0.0, 0.0};
            Gb[b] = hm_d4{0.0, 0.0, 0.0, 0.0};
        }
        const int ra_ = 16 * a + c;  // operand row of this lane
        const bool rin = ra_ < TP;
#pragma unroll 2
        for (int kk = 0; kk < NK; ++kk) {
            const int kcol = 4 * kk + g;
            const bool kin = kcol < TP;
            const double opC = (rin && kin) ? C[ra_ * TP + kcol] : 0.0;
            const double opKa = (rin && kin) ? kl[ra_ * ldk + kcol] : 0.0;
#pragma unroll
            for (int b = 0; b < NBK; ++b) {
                const int rb_ = 16 * b + c;
                const bool bin = rb_ < TP && kin;
                const double opKb = bin ? kl[rb_ * ldk + kcol] : 0.0;
                const int dd = kcol - rb_;
                const double opD = (bin && kcol < tr_k && rb_ < tr_k) ? dkv[dd < 0 ? -dd : dd] : 0.0;
                Eb[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(opC, opKb, Eb[b], 0, 0, 0);
                Gb[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(opKa, opD, Gb[b], 0, 0, 0);
            }
        }
#pragma unroll
        for (int b = 0; b < NBK; ++b)
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                gq = fma(Eb[b][p], Gb[b][p], gq);
                if (b == a && g + 4 * p == c) quad += Eb[b][p];
            }
    }
}
