// H-step objective, per-segment part, in an EXACT low-rank (Woodbury) form -- round 4.
//
// Reference math: vlgp/gp.py:12-43 (elbo), 126-147 (construct_posterior_cov).  Per segment i and evaluation
// (sigma^2, omega, eps) the round needs, with A = I + S K S, S = diag sqrt(w_i), K = K_s + eps I,
// K_s = sigma^2 exp(-omega D^2):
//     tr  = tr(A^-1)                         ( = tr(K^-1 S_i) of the reference )
//     cs  = sum_jk s_j s_k dK_jk (A^-1)_jk   ( = d/dln(omega) log det A; enters tr(K^-1 S_i K^-1 dK) )
// The dense kernels (hstep_mfma.h) factor the 50 x 50 matrix A per segment: T^3 / 2 multiply-adds whatever the
// smoothness of K_s.  But K_s is numerically low-rank -- the squared-exponential kernel on a 50-bin window has
// 12 ... 28 eigenvalues above 1e-13 at the omegas a fit visits after its first iterations -- and only w changes
// from segment to segment.  With K_s = U U' (U: T x r, pivoted Cholesky run until the residual diagonal drops
// below `tol`, the tail is below the rounding of the dense kernels), d = 1 / (1 + eps w), wt = w d:
//     A^-1 = D - D S U M^-1 U' S D,    D = diag d,    M = I_r + U' diag(wt) U        (eigenvalues of M >= 1)
//     tr   = sum_t d_t - < M^-1, U' diag(wt d) U >
//     cs   = < M^-1, Ud' diag(wt) U + U' diag(wt) Ud >,   Ud = dU / dln(omega)
// (cs is the derivative of log det M along omega; Ud comes out of the same pivoted Cholesky recursion by forward
// differentiation, so that Ud U' + U Ud' = dK_s up to the same tail).  Checked against 40-digit arithmetic in
// tools/lr_proto.py: 1e-14 ... 1e-12 relative for tol 1e-14 ... 1e-12.
//
// Two more structural facts are used:
//   * K_s is centro-symmetric: in the basis of even / odd time courses (e_t + e_t') / sqrt 2, (e_t - e_t') / sqrt 2,
//     t' = T - 1 - t, it is block diagonal, each block has about half the rank, and diag(wt) becomes
//     [[wp, wm], [wm, wp]] with wp/wm = (wt_t +- wt_t') / 2 on HALF the time axis: the sums over t are 25 long.
//   * the three r x r matrices of a segment are sums over t of (weight_t) x (a product of two table entries that
//     does not depend on the segment): B[seg][pair (i, j)] = sum_t wgt[seg][t] P[t][pair], a GEMM across SEGMENTS
//     with the segments on the 16 rows of v_mfma_f64_16x16x4, 16 (i <= j) pairs on its columns and t on its
//     depth -- symmetric halves never computed, no per-segment operand staging.
// A workgroup takes 16 segments of one evaluation:
//   phase 1  M = I + B0 for the 16 segments (tiles of 16 pairs split over the waves) -> packed lower triangles in LDS
//   phase 2  M^-1 in place by symmetric Gauss-Jordan sweeps, one lane per row, 2 or 4 segments side by side in a
//            wave, the pivot column (= row) handed round through a small LDS buffer
//   phase 3  the tiles of U'diag(wt d)U and of the symmetrised Ud'diag(wt)U stream through the matrix pipe once
//            more and are contracted with M^-1 as they leave it (never stored)
// Cost per segment at r = 24: 57 matrix instructions (19 tiles x 3 x 7 / 16 ... per segment 25) and ~ 600 vector
// instructions against 78 + 1960 of the dense kernel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "hstep_mfma.h"

#define LR_RCAP 32    // largest total rank (even + odd) the low-rank round takes; above it the dense kernels run
#define LR_RH 20      // per-parity capacity of the factor kernel
#define LR_TROWS 32   // folded time rows of a table (windows up to 64 bins)
#define LR_NPAIR 576  // pair codes per evaluation: 16 ceil(same / 16) + 16 ceil(cross / 16) <= 528 + 30

struct LrMeta {
    int re, ro;     // ranks of the even and the odd block
    int r;          // re + ro (clamped to LR_RCAP)
    int ns_tiles;   // tiles of same-parity pairs (weights wp)
    int n_tiles;    // all tiles; [ns_tiles, n_tiles) hold the cross pairs (weights wm)
    int overflow;   // 1: the rank did not fit the capacity the host predicted -> results of this evaluation invalid
    int pad0, pad1;
};

struct HLrTabArgs {
    int T;
    double dt, tol;
    int rcap[16];           // columns the round kernel has room for (host prediction)
    double* tab;            // (n_eval, 2, LR_TROWS, LR_RCAP): U | dU/dln(omega), sigma folded in, zero padded
    LrMeta* meta;           // (n_eval)
    unsigned short* pairs;  // (n_eval, LR_NPAIR): (i << 8) | j, i >= j; 0xffff = padding
};

__device__ __forceinline__ double lr_rcp(double x) {  // 1 / x for x >= 1 (no denormals, no scaling needed)
    double y = __builtin_amdgcn_rcp(x);
    double e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    e = fma(-x, y, 1.0);
    y = fma(y, e, y);
    return y;
}

// Wave-wide maximum, the same value in every lane.  Four row_shr steps on the DPP path leave the maximum of each row
// of sixteen lanes in its last lane (a lane without a source keeps its own value); the four row results meet through
// v_readlane.  About 100 cycles against ~400 for six ds_bpermute rounds: the pivot search is on the critical chain
// of every step of the factorisation below.
__device__ __forceinline__ double lr_wave_max(double v) {
    union { double d; int i[2]; } a, b;
#define LR_DPP_STEP(ctrl)                                                          \
    a.d = v;                                                                       \
    b.i[0] = __builtin_amdgcn_update_dpp(a.i[0], a.i[0], ctrl, 0xf, 0xf, false);   \
    b.i[1] = __builtin_amdgcn_update_dpp(a.i[1], a.i[1], ctrl, 0xf, 0xf, false);   \
    v = fmax(v, b.d);
    LR_DPP_STEP(0x111)  // row_shr:1
    LR_DPP_STEP(0x112)  // row_shr:2
    LR_DPP_STEP(0x114)  // row_shr:4
    LR_DPP_STEP(0x118)  // row_shr:8
#undef LR_DPP_STEP
    const double r0 = tri_readlane(v, 15), r1 = tri_readlane(v, 31), r2 = tri_readlane(v, 47), r3 = tri_readlane(v, 63);
    return fmax(fmax(r0, r1), fmax(r2, r3));
}

// Sum over the sixteen lanes of a row, on the DPP path (row_shr 1, 2, 4, 8; a lane without a source adds 0): the total
// ends up in lane 15 of each row.
__device__ __forceinline__ double lr_row_sum_to15(double v) {
    union { double d; int i[2]; } a, b;
#define LR_DPP_ADD(ctrl)                                                           \
    a.d = v;                                                                       \
    b.i[0] = __builtin_amdgcn_update_dpp(0, a.i[0], ctrl, 0xf, 0xf, true);         \
    b.i[1] = __builtin_amdgcn_update_dpp(0, a.i[1], ctrl, 0xf, 0xf, true);         \
    v += b.d;
    LR_DPP_ADD(0x111)
    LR_DPP_ADD(0x112)
    LR_DPP_ADD(0x114)
    LR_DPP_ADD(0x118)
#undef LR_DPP_ADD
    return v;
}

// The pivoted Cholesky of one folded block with its omega-tangent, by one wave (par 0: even block, 1: odd block, other
// waves fall through with r = 0): factor columns g, gd in registers (lane <-> folded time row tau), rank r, `capped` = the
// capacity LR_RH ran out before the tolerance was met.  kv, dkv: exp(-omega d^2) and its derivative along ln(omega) by
// distance (64 doubles of LDS each, filled by the caller).
__device__ __forceinline__ void lr_factor_wave(int T, double tol, int par, int lane, const double* kv, const double* dkv,
                                               double (&g)[LR_RH], double (&gd)[LR_RH], int& r_out, int& capped_out,
                                               bool& rowin_out, int& tau_out) {
    const int h = T >> 1, nt = (T + 1) >> 1;
    const bool oddT = (T & 1) != 0;
    struct { double tol; } A{tol};
    const int n = par == 0 ? nt : (par == 1 ? h : 0);
    const bool rowin = lane < n;
    const int tau = rowin ? lane : 0;
    constexpr double RS2 = 0.70710678118654752440;
    const double stau = (par == 0 && oddT && tau == h) ? RS2 : 1.0;
    auto kent = [&](int p, double& kk, double& dk) {
        const int d1 = tau > p ? tau - p : p - tau, d2 = T - 1 - tau - p;
        const double a = kv[d1], b = kv[d2], da = dkv[d1], db = dkv[d2];
        if (par == 0) {
            const double s = stau * ((oddT && p == h) ? RS2 : 1.0);
            kk = s * (a + b);
            dk = s * (da + db);
        } else {
            kk = a - b;
            dk = da - db;
        }
    };
#pragma unroll
    for (int k = 0; k < LR_RH; ++k) {
        g[k] = 0.0;
        gd[k] = 0.0;
    }
    double d, dd;
    kent(tau, d, dd);
    if (!rowin) {
        d = -1.0;
        dd = 0.0;
    }
    int r = 0;
    auto wave_argmax = [&](double& bv, int& bi) {  // largest residual diagonal, lowest lane on ties; branch-free
        bv = lr_wave_max(d);
        const unsigned long long m = __ballot(d == bv);
        bi = m ? __builtin_ctzll(m) : 0;
    };
#pragma unroll
    for (int k = 0; k < LR_RH; ++k) {
        double bv;
        int bi;
        wave_argmax(bv, bi);
        if (!(bv > A.tol)) break;  // (a NaN stops too)
        const int p = __builtin_amdgcn_readfirstlane(bi);
        const double ddp = tri_readlane(dd, p);
        double ginv, gpp;
        tri_rsqrt(bv, &ginv, &gpp);
        const double gdot = 0.5 * ddp * ginv;
        double col, cold;
        kent(p, col, cold);
        double cold2 = 0.0, col2 = 0.0;  // (independent partial sums: the chains are what a step costs)
#pragma unroll
        for (int j = 0; j < k; ++j) {
            const double gpj = tri_readlane(g[j], p), gdpj = tri_readlane(gd[j], p);
            if (j & 1) col2 = fma(-g[j], gpj, col2);
            else col = fma(-g[j], gpj, col);
            cold = fma(-gd[j], gpj, cold);
            cold2 = fma(-g[j], gdpj, cold2);
        }
        col += col2;
        cold += cold2;
        const double gk = rowin ? col * ginv : 0.0;
        const double gdk = rowin ? (cold - gk * gdot) * ginv : 0.0;
        g[k] = gk;
        gd[k] = gdk;
        d = fma(-gk, gk, d);
        dd = fma(-2.0 * gk, gdk, dd);
        if (lane == p || !rowin) {
            d = -1.0;
            dd = 0.0;
        }
        r = k + 1;
    }
    int capped = 0;
    if (r == LR_RH) {
        double bv;
        int bi;
        wave_argmax(bv, bi);
        capped = bv > A.tol;
    }
    r_out = r;
    capped_out = capped;
    rowin_out = rowin;
    tau_out = tau;
}

// ---------------------------------------------------------------------------------------------------------
// Tables of one evaluation: pivoted Cholesky of the even and of the odd block of K_0 = exp(-omega D^2) (one wave
// each, lane <-> folded time row, factor columns in registers, pivot row entries by v_readlane), differentiated
// along ln(omega) step by step.  Even block: Ke[t][p] = s_t s_p (k(t - p) + k(t + p - (T - 1))), s = 1 / sqrt 2 at
// the middle row of an odd window and 1 elsewhere; odd block: Ko[t][p] = k(t - p) - k(t + p - (T - 1)), t, p < T / 2.
// ---------------------------------------------------------------------------------------------------------
// Called by every thread of a block of NT >= 128 threads; the first two waves factor, the others help with the fills.
// kv, dkv: 2 x 64 doubles of LDS; s_i: 4 ints of LDS.
template <int NT>
__device__ __forceinline__ void lr_tables_block(const HLrTabArgs& A, int e, double sigmasq, double omega, double* kv,
                                                double* dkv, int* s_i) {
    int* s_r = s_i;
    int* s_cap = s_i + 2;
    const int lane = threadIdx.x & 63, par = threadIdx.x >> 6;
    const bool fac = par < 2;  // this wave factors a block (0: even, 1: odd)
    const int T = A.T, h = T >> 1, nt = (T + 1) >> 1;
    const bool oddT = (T & 1) != 0;
    double* U = A.tab + (int64_t)e * 2 * LR_TROWS * LR_RCAP;
    double* Ud = U + LR_TROWS * LR_RCAP;
    unsigned short* pairs = A.pairs + (int64_t)e * LR_NPAIR;
    for (int i = threadIdx.x; i < 2 * LR_TROWS * LR_RCAP; i += NT) U[i] = 0.0;
    for (int q = threadIdx.x; q < LR_NPAIR; q += NT) pairs[q] = 0xffffu;
    if (threadIdx.x < 64) {
        const double d = lane * A.dt, d2 = d * d;
        const double k = lane < T ? exp(-omega * d2) : 0.0;
        kv[lane] = k;
        dkv[lane] = -omega * d2 * k;
    }
    __syncthreads();
    double g[LR_RH], gd[LR_RH];
    int r, capped, tau;
    bool rowin;
    lr_factor_wave(T, A.tol, par, lane, kv, dkv, g, gd, r, capped, rowin, tau);
    if (lane == 0 && fac) {
        s_r[par] = r;
        s_cap[par] = capped;
    }
    __syncthreads();
    int re = s_r[0], ro = s_r[1];
    int overflow = s_cap[0] | s_cap[1];
    if (re + ro > A.rcap[e] || re + ro > LR_RCAP) overflow = 1;
    if (overflow) re = ro = 0;  // the round kernel's LDS is carved for rcap columns: leave it nothing to do
    const int coff = par == 0 ? 0 : re;
    const int rw = par == 0 ? re : ro;
    const double sig = sqrt(sigmasq);
    if (rowin && fac) {
#pragma unroll
        for (int k = 0; k < LR_RH; ++k)
            if (k < rw) {
                U[tau * LR_RCAP + coff + k] = sig * g[k];
                Ud[tau * LR_RCAP + coff + k] = sig * gd[k];
            }
    }
    // pair list: same-parity pairs (even-even, then odd-odd), padded to a tile boundary, then the cross pairs
    const int rt = re + ro;
    const int n_ee = re * (re + 1) / 2, n_oo = ro * (ro + 1) / 2;
    const int ns16 = (n_ee + n_oo + 15) & ~15, nc16 = (re * ro + 15) & ~15;
    for (int i = par; i < rt; i += NT / 64) {  // row i by wave, column j by lane (no integer divisions)
        const int j = lane;
        if (j <= i) {
            const bool io = i >= re, jo = j >= re;
            int pos;
            if (io == jo)
                pos = io ? n_ee + (i - re) * (i - re + 1) / 2 + (j - re) : i * (i + 1) / 2 + j;
            else
                pos = ns16 + (i - re) * re + j;  // i odd block, j even block
            pairs[pos] = (unsigned short)((i << 8) | j);
        }
    }
    if (threadIdx.x == 0) {
        LrMeta m;
        m.re = re;
        m.ro = ro;
        m.r = rt;
        m.ns_tiles = ns16 >> 4;
        m.n_tiles = (ns16 + nc16) >> 4;
        m.overflow = overflow;
        m.pad0 = m.pad1 = 0;
        A.meta[e] = m;
    }
}

// LDS carve of a segment group (doubles), for rank r, ntp folded time rows, nw waves
struct LrGeom {
    int LDU, NPS;
    int o_u, o_ud, o_mp, o_xb, o_red, o_codes, total;
};
// doubles per wave of phase 2's exchange buffer for register class rc: one slot of GB BS entries per segment a wave
// holds (block-triangle layout of lr_group: GB = 4 blocks a side up to class 16, 5 above)
// class 32 keeps the lane-per-row sweeps: a 7 x 7 block per lane (98 registers) on top of the rest of the kernel
// spills inside the sweep loop -- measured 284 us against 151 us for a round of five evaluations at rank 29
__host__ __device__ constexpr bool lr_block_inverse(int rc) {
#ifdef LR_ROWWISE
    return false;
#else
    return rc <= 28;
#endif
}
__host__ __device__ constexpr int lr_xs_per_wave(int rc) {
    if (!lr_block_inverse(rc)) return 64;
    const int gb = rc <= 16 ? 4 : 5, bs = (rc + gb - 1) / gb, lps = gb * (gb + 1) / 2, spw = 64 / lps;
    return (spw * gb * bs + 1) & ~1;
}
__host__ __device__ inline LrGeom lr_geom(int r, int ntp, int nw, bool tab_global, int rc) {
    LrGeom G;
    G.LDU = tab_global ? LR_RCAP : ((r + 1) | 1);  // column r: zeros (padding pairs point at it)
    G.NPS = (r * (r + 1) / 2 + 2) | 1;     // packed lower triangle + a zero slot (padding pairs) + a trash slot
    G.o_u = 0;
    G.o_ud = tab_global ? 0 : ntp * G.LDU;
    G.o_mp = tab_global ? 0 : ((2 * ntp * G.LDU + 1) & ~1);
    G.o_xb = (G.o_mp + 16 * G.NPS + 1) & ~1;
    // exchange buffers of phase 2; the partial sums of the final reduction (nw x 32 doubles) take the same place later.
    // (Every double counts here: at ranks 23 ... 27 the packed matrices leave ~5 KB for everything else if three
    // workgroups are to share a CU.)
    G.o_red = G.o_xb;
    const int xw = nw * lr_xs_per_wave(rc), rw = nw * 32 + 2;
    G.o_codes = G.o_xb + (xw > rw ? xw : rw);  // pair codes of the evaluation (LR_NPAIR unsigned shorts)
    int ncode = r * (r + 1) / 2 + 32;  // same-parity + cross pairs = r (r + 1) / 2, each list padded to a tile of 16
    if (ncode > LR_NPAIR) ncode = LR_NPAIR;
    G.total = G.o_codes + (ncode + 3) / 4;
    return G;
}

// One group of 16 segments (seg0 ... seg0 + 15, those >= M masked) of one evaluation, by a workgroup of NW waves.
// RC: register class (rank <= RC); NK: depth steps (4 NK >= folded rows).  Returns the group's sums of tr and cs
// in lane 0 of wave 0 (other threads: garbage).
// TABF: the tables are not read -- the workgroup factors the two folded blocks itself (lr_factor_wave by its first two
// waves) straight into its LDS carve, from (sigma^2, omega) of evaluation e; fa (rcap, tol, dt, meta) are the table
// kernel's arguments, lds_doubles the size of the dynamic LDS (scratch of the factorisation at its end), publish: this
// workgroup writes the evaluation's meta record for the round's final block (overflow flag).
template <int RC, int NK, int NW, bool TABG = false, bool TABF = false>
__device__ __forceinline__ void lr_group(const double* __restrict__ tab_e, const LrMeta mt,
                                         const unsigned short* __restrict__ pairs_e, const double* __restrict__ w,
                                         const int64_t* __restrict__ off, int L, int l, int M, int T, double eps,
                                         int seg0, double* lds, int lane, int wid, double& out_tr, double& out_cs,
                                         unsigned long long* clk = nullptr, const HLrTabArgs* fa = nullptr, int e = 0,
                                         double sigmasq = 0.0, double omega = 0.0, int lds_doubles = 0, bool publish = false) {
    static_assert(!(TABG && TABF), "fused tables live in LDS");
    // debug (vlgp_debug_phase_clock): cycles per phase of one workgroup, thread 0
    long long tck = clk ? clock64() : 0;
    auto stamp = [&](int slot) {
        if (clk && threadIdx.x == 0) {
            const long long now = clock64();
            atomicAdd(clk + slot, (unsigned long long)(now - tck));
            tck = now;
        }
    };
    const int c = lane & 15, g = lane >> 4;
    // ---- loads that do not depend on the evaluation's meta record go out first: the weights' rows of w and (tables in
    // LDS) the table entries this thread will copy.  The meta record, the tables and w are three global round trips
    // (~1.5 us each with the chip full) that used to run one after the other in a workgroup that lives ~30 us. ----
    const int h = T >> 1, nt = (T + 1) >> 1;
    const int gs = seg0 + c;
    const bool sval = gs < M;
    const int64_t r0 = sval ? (int64_t)gs * T : 0;  // (every unit has T rows: off[m] = m T, no load)
    (void)off;
    double wpre1[NK], wpre2[NK];
    auto load_w = [&]() {
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int tau = 4 * kk + g;
            wpre1[kk] = (sval && tau < nt) ? w[(r0 + tau) * L + l] : 0.0;
            wpre2[kk] = (sval && tau < h) ? w[(r0 + T - 1 - tau) * L + l] : 0.0;
        }
    };
    if constexpr (!TABF) load_w();  // (fused tables: behind the factorisation, whose forty column registers need the room)
    constexpr int TPRE = (TABG || TABF) ? 1 : (4 * NK * LR_RCAP + 64 * NW - 1) / (64 * NW);
    double tpre[TPRE][2];
    LrMeta mloc = mt;
    constexpr int FR = TABF ? LR_RH : 1;
    double fg[FR], fgd[FR];
    int f_r = 0, f_cap = 0, f_tau = 0, f_re = 0;
    bool f_rowin = false;
    if constexpr (TABF) {
        double* kv = lds + lds_doubles - 136;
        double* dkv = kv + 64;
        int* s_i = reinterpret_cast<int*>(dkv + 64);
        if (threadIdx.x < 64) {  // (as lr_tables_block)
            const double d = lane * fa->dt, d2 = d * d;
            const double k = lane < T ? exp(-omega * d2) : 0.0;
            kv[lane] = k;
            dkv[lane] = -omega * d2 * k;
        }
        __syncthreads();
        lr_factor_wave(T, fa->tol, wid, lane, kv, dkv, fg, fgd, f_r, f_cap, f_rowin, f_tau);
        if (lane == 0 && wid < 2) {
            s_i[wid] = f_r;
            s_i[2 + wid] = f_cap;
        }
        __syncthreads();
        int re = s_i[0], ro = s_i[1];
        int overflow = s_i[2] | s_i[3];
        if (re + ro > fa->rcap[e] || re + ro > LR_RCAP) overflow = 1;
        if (overflow) re = ro = 0;
        const int n_ee = re * (re + 1) / 2, n_oo = ro * (ro + 1) / 2;
        const int ns16 = (n_ee + n_oo + 15) & ~15, nc16 = (re * ro + 15) & ~15;
        mloc.re = re; mloc.ro = ro; mloc.r = re + ro;
        mloc.ns_tiles = ns16 >> 4;
        mloc.n_tiles = (ns16 + nc16) >> 4;
        mloc.overflow = overflow;
        mloc.pad0 = mloc.pad1 = 0;
        f_re = re;
        if (publish && threadIdx.x == 0) fa->meta[e] = mloc;
        __syncthreads();  // (the scratch at the end of the LDS is dead from here: the pair codes take its place)
        load_w();
    }
    if constexpr (!TABG && !TABF) {
#pragma unroll
        for (int q = 0; q < TPRE; ++q) {
            const int x = threadIdx.x + q * 64 * NW;
            const bool ok = x < 4 * NK * LR_RCAP;
            tpre[q][0] = ok ? tab_e[x] : 0.0;
            tpre[q][1] = ok ? tab_e[LR_TROWS * LR_RCAP + x] : 0.0;
        }
    }
    // wave-uniform by construction; said explicitly so that the rank guards below are scalar branches
    const int r = __builtin_amdgcn_readfirstlane(mloc.r);
    const int ns_tiles = __builtin_amdgcn_readfirstlane(mloc.ns_tiles), n_tiles = __builtin_amdgcn_readfirstlane(mloc.n_tiles);
    // TABG: the tables stay in global memory (L1 / L2: 13 KB per evaluation, shared by its 250 workgroups) and the LDS
    // they would take goes to a third workgroup per CU (ranks 25 ... 31); needs a zero column, i.e. r < LR_RCAP
    const LrGeom G = lr_geom(r, 4 * NK, NW, TABG, RC);
    const double* Ul = TABG ? tab_e : lds + G.o_u;
    const double* Udl = TABG ? tab_e + LR_TROWS * LR_RCAP : lds + G.o_ud;
    double* Mp = lds + G.o_mp;
    double* xb = lds + G.o_xb + wid * lr_xs_per_wave(RC);
    double* red = lds + G.o_red;
    unsigned short* codes = reinterpret_cast<unsigned short*>(lds + G.o_codes);
    if constexpr (!TABF)
        for (int x = threadIdx.x; x < 16 * n_tiles; x += 64 * NW) codes[x] = pairs_e[x];  // (the tile loops read them from LDS)
    const int LDU = G.LDU, NPS = G.NPS;
    const int NPZ = r * (r + 1) / 2;  // the zero slot of a packed matrix; NPZ + 1: trash
    if constexpr (TABF) {
        double* Uw = lds + G.o_u;
        double* Udw = lds + G.o_ud;
        for (int x = threadIdx.x; x < 4 * NK * LDU; x += 64 * NW) {
            Uw[x] = 0.0;
            Udw[x] = 0.0;
        }
        for (int x = threadIdx.x; x < 16 * n_tiles; x += 64 * NW) codes[x] = 0xffffu;
        __syncthreads();
        const int re = f_re, ro = r - f_re;
        if (f_rowin && wid < 2) {
            const int coff = wid == 0 ? 0 : re, rw = wid == 0 ? re : ro;
            const double sig = sqrt(sigmasq);
#pragma unroll
            for (int k = 0; k < LR_RH; ++k)
                if (k < rw) {
                    Uw[f_tau * LDU + coff + k] = sig * fg[k];
                    Udw[f_tau * LDU + coff + k] = sig * fgd[k];
                }
        }
        const int n_ee = re * (re + 1) / 2;
        const int ns16 = 16 * ns_tiles;
        for (int i = wid; i < r; i += NW) {  // (as lr_tables_block: row i by wave, column j by lane)
            const int j = lane;
            if (j <= i) {
                const bool io = i >= re, jo = j >= re;
                int pos;
                if (io == jo)
                    pos = io ? n_ee + (i - re) * (i - re + 1) / 2 + (j - re) : i * (i + 1) / 2 + j;
                else
                    pos = ns16 + (i - re) * re + j;
                codes[pos] = (unsigned short)((i << 8) | j);
            }
        }
    }
    if constexpr (!TABG && !TABF) {
        double* Uw = lds + G.o_u;
        double* Udw = lds + G.o_ud;
#pragma unroll
        for (int q = 0; q < TPRE; ++q) {  // (LR_RCAP = 32 columns per table row: shifts)
            const int x = threadIdx.x + q * 64 * NW;
            const int tau = x >> 5, cc = x & 31;
            if (x < 4 * NK * LR_RCAP && cc < LDU) {  // columns >= r of the table are zero already
                Uw[tau * LDU + cc] = tpre[q][0];
                Udw[tau * LDU + cc] = tpre[q][1];
            }
        }
        if (LDU > LR_RCAP && threadIdx.x < 4 * NK) {  // full rank: the zero column lies beyond the table's
            Uw[threadIdx.x * LDU + LR_RCAP] = 0.0;
            Udw[threadIdx.x * LDU + LR_RCAP] = 0.0;
        }
    }
    // ---- phase 0: folded weights of segment c at the depth positions of this lane ----
    // (evaluated before phase 1 and again before phase 3: 4 NK values that would otherwise sit in registers through the
    // sweeps of phase 2, which need the room for the matrix rows)
    double ap[NK], am[NK], bp[NK], bm[NK];
    double dsum = 0.0;
    auto weights = [&](const double* p1, const double* p2) {
        dsum = 0.0;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int tau = 4 * kk + g;
            const bool in1 = sval && tau < nt, in2 = sval && tau < h;
            const double w1 = p1 ? p1[kk] : (in1 ? w[(r0 + tau) * L + l] : 0.0);
            const double w2 = p2 ? p2[kk] : (in2 ? w[(r0 + T - 1 - tau) * L + l] : 0.0);
            const double d1 = lr_rcp(fma(eps, w1, 1.0)), d2 = lr_rcp(fma(eps, w2, 1.0));
            const double t1 = w1 * d1, t2 = w2 * d2;
            const double q1 = t1 * d1, q2 = t2 * d2;
            const double fa = in2 ? 0.5 : 1.0;  // the middle row of an odd window is its own mirror image
            ap[kk] = fa * (t1 + t2);
            bp[kk] = fa * (q1 + q2);
            am[kk] = in2 ? 0.5 * (t1 - t2) : 0.0;
            bm[kk] = in2 ? 0.5 * (q1 - q2) : 0.0;
            dsum += (in1 ? d1 : 0.0) + (in2 ? d2 : 0.0);
        }
    };
    weights(wpre1, wpre2);
    __syncthreads();
    stamp(0);
    // ---- phase 1: M = I + U' diag(wt) U, 16 pairs x 16 segments per tile; two tiles of one kind at a time (two
    // independent accumulate chains on the matrix pipe) ----
    auto tile_codes = [&](int q, int& i, int& j, int& idx, double& dg) {
        const unsigned code = codes[q * 16 + c];
        const bool valid = code != 0xffffu;
        i = valid ? (int)(code >> 8) : r;
        j = valid ? (int)(code & 255u) : r;
        idx = valid ? i * (i + 1) / 2 + j : NPZ;
        dg = (valid && i == j) ? 1.0 : 0.0;
    };
    auto phase1 = [&](int qb, int qe, const double(&aw)[NK]) {
        for (int q = qb + wid; q < qe; q += 2 * NW) {
            const bool two = q + NW < qe;
            int i0, j0, x0, i1, j1, x1;
            double g0, g1;
            tile_codes(q, i0, j0, x0, g0);
            tile_codes(two ? q + NW : q, i1, j1, x1, g1);
            const double* u0i = Ul + g * LDU + i0;
            const double* u0j = Ul + g * LDU + j0;
            const double* u1i = Ul + g * LDU + i1;
            const double* u1j = Ul + g * LDU + j1;
            hm_d4 acc0 = hm_d4{0.0, 0.0, 0.0, 0.0}, acc1 = hm_d4{0.0, 0.0, 0.0, 0.0};
            if (two) {
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) {
                    const int o = 4 * kk * LDU;
                    const double b0 = u0i[o] * u0j[o], b1 = u1i[o] * u1j[o];
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[kk], b0, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[kk], b1, acc1, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int kk = 0; kk < NK; ++kk) {
                    const int o = 4 * kk * LDU;
                    const double b0 = u0i[o] * u0j[o];
                    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[kk], b0, acc0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) Mp[(g + 4 * p) * NPS + x0] = acc0[p] + g0;
            if (two) {
#pragma unroll
                for (int p = 0; p < 4; ++p) Mp[(g + 4 * p) * NPS + x1] = acc1[p] + g1;
            }
        }
    };
    phase1(0, ns_tiles, ap);
    phase1(ns_tiles, n_tiles, am);
    __syncthreads();
    stamp(1);
    if constexpr (lr_block_inverse(RC)) {
    // ---- phase 2: M^-1 by symmetric Gauss-Jordan sweeps on the LOWER BLOCK TRIANGLE, six segments side by side.  The
    // matrix is cut into 4 x 4 blocks of BS x BS entries (BS = RC / 4); lane <-> one of the ten blocks (I >= J) of one
    // segment, its BS^2 entries in registers (diagonal blocks hold both triangles).  Sweep k (block K = k / BS, local
    // index kl): the lanes of block column K / block row K hand column k (= row k) to the others through `xs`, then every
    // lane updates its block with the BS multipliers of its rows and the BS pivot-row entries of its columns:
    // BS^2 FMAs and 2 BS LDS reads per lane and sweep for SIX segments, against RC FMAs and RC reads for two in the
    // lane-per-row layout this replaces (LR_ROWWISE), and one pass per wave instead of two. ----
    {
        // GB x GB blocks of BS x BS entries: four blocks a side up to rank 24 (ten lanes per segment, six segments per
        // wave), five above (fifteen lanes, four segments: a 7 x 7 or 8 x 8 block per lane does not fit the registers)
        constexpr int GB = RC <= 16 ? 4 : 5, BS = (RC + GB - 1) / GB, LPS = GB * (GB + 1) / 2, SPW = 64 / LPS;
        const int sg = lane / LPS, li = lane - sg * LPS;
        int I = 0;
#pragma unroll
        for (int q = 1; q < GB; ++q) I += li >= q * (q + 1) / 2 ? 1 : 0;
        const int J = li - I * (I + 1) / 2;
        for (int P = wid; P * SPW < 16; P += NW) {
            int seg = P * SPW + sg;
            const bool act = sg < SPW && seg < 16;
            if (!act) seg = 0;
            int iv = BS * I, jv = BS * J;  // opaque per pass: the packed indices are not worth registers across the sweeps
            asm volatile("" : "+v"(iv), "+v"(jv));
            double* Ms = Mp + seg * NPS;
            double* xs = xb + (act ? sg : 0) * (GB * BS);  // (idle lanes write nothing and read a neighbour's slot)
            double blk[BS][BS];
#pragma unroll
            for (int a_ = 0; a_ < BS; ++a_)
#pragma unroll
                for (int b_ = 0; b_ < BS; ++b_) {
                    const int i = iv + a_, j = jv + b_;
                    const int hi = i > j ? i : j, lo = i > j ? j : i;
                    const double v = Ms[(hi < r) ? hi * (hi + 1) / 2 + lo : NPZ];
                    blk[a_][b_] = hi >= r ? (i == j ? 1.0 : 0.0) : v;  // unit diagonal beyond the rank
                }
            auto sweeps = [&]() {
                // (the block index K is a RUN-TIME loop: only the BS sweeps of one block are unrolled.  Unrolled over all
                // GB BS sweeps the kernel is ~90 KB of straight-line code that every wave walks through once, against a
                // 64 KB instruction cache shared by two CUs.)
#pragma nounroll
                for (int K = 0; K < GB; ++K) {
                const bool wc = J == K;            // this block holds column k for its rows (local column kl)
                const bool wr = I == K && J < K;   // this block holds row k for its columns (local row kl)
#pragma unroll
                for (int kl = 0; kl < BS; ++kl) {
                    const int k = K * BS + kl;
                    if (k >= r) return;
                    constexpr int dummy = 0;
                    if ((wc || wr) && act) {
                        double* dst = xs + (wc ? BS * I : BS * J);
#pragma unroll
                        for (int a_ = 0; a_ < BS; ++a_) dst[a_] = wc ? blk[a_][kl] : blk[kl][a_];
                    }
                    tri_wave_order();
                    const double dinv = lr_rcp(xs[k]);
                    double f[BS], pc[BS];
#pragma unroll
                    for (int a_ = 0; a_ < BS; ++a_) {
                        f[a_] = xs[BS * I + a_] * dinv;
                        pc[a_] = xs[BS * J + a_];
                    }
                    if (I == K) f[kl] = 1.0 - dinv;    // the pivot row: a_kj - (1 - d) a_kj = d a_kj
#pragma unroll
                    for (int a_ = 0; a_ < BS; ++a_)
#pragma unroll
                        for (int b_ = 0; b_ < BS; ++b_) blk[a_][b_] = fma(-f[a_], pc[b_], blk[a_][b_]);
                    if (wc) {
#pragma unroll
                        for (int a_ = 0; a_ < BS; ++a_) blk[a_][kl] = f[a_];
                        if (I == K) blk[kl][kl] = -dinv;
                    }
                    tri_wave_order();
                    (void)dummy;
                }
                }
            };
            sweeps();
            // blk = -(M^-1) block; stored with the off-diagonal entries doubled: the contraction runs over i >= j
            if (act) {
#pragma unroll
                for (int a_ = 0; a_ < BS; ++a_)
#pragma unroll
                    for (int b_ = 0; b_ < BS; ++b_) {
                        const int i = iv + a_, j = jv + b_;
                        const bool keep = i >= j && i < r;
                        Ms[keep ? i * (i + 1) / 2 + j : NPZ + 1] = i == j ? -blk[a_][b_] : -2.0 * blk[a_][b_];
                    }
            }
        }
    }
    } else {
    // ---- phase 2: M^-1 by symmetric Gauss-Jordan sweeps, lane <-> row, SPP segments side by side.  Sweep k: every
    // lane hands its entry of column k (= row k, the matrix stays symmetric) to the others through `xs`; the lane that
    // holds row k hands over 1 / pivot instead.  Rows beyond the rank idle.  A sweep is one dependent chain (column
    // out through LDS, back in, row update, the next pivot's reciprocal): ~700 cycles for ~50 instructions
    // (tools/lr_phase_clock.py); measured alternatives, all slower: the passes of a wave interleaved (two chains in
    // flight: the register file does not hold two rows per lane plus the column entries in flight, 3x slower with the
    // spills), blocks of four pivots (a quarter of the LDS round trips, but the 4 x 4 pivot block inverted by every
    // lane and four dependent FMAs per entry: 1.6x slower), eight waves per workgroup (one pass each: faster alone,
    // slower when the chip is full). ----
    {
        constexpr int LPS = RC <= 16 ? 16 : 32, SPP = 64 / LPS, NPASS = 16 / SPP;
        const int sl = lane / LPS, row = lane % LPS;
        const bool rowok = row < r;
        for (int P = wid; P < NPASS; P += NW) {
            int rowv = row;  // opaque per pass: the packed indices of a row are not worth 2 RC registers across passes
            asm volatile("" : "+v"(rowv));
            double* Ms = Mp + (P * SPP + sl) * NPS;
            double* xs = xb + sl * LPS;
            double a[RC];
#pragma unroll
            for (int j = 0; j < RC; ++j) {  // (no branches: rows / columns beyond the rank read the zero slot)
                const int hi = rowv > j ? rowv : j, lo = rowv > j ? j : rowv;
                a[j] = Ms[(rowok && j < r) ? hi * (hi + 1) / 2 + lo : NPZ];
            }
            double nd = lr_rcp(a[0]);
            // (no rank guards inside a sweep: entries beyond the rank are zeros on both sides, and every guarded chunk of
            // the column would cost its own LDS round trip; the sweeps themselves stop at the rank)
            auto sweeps = [&]() {
#pragma unroll
                for (int k = 0; k < RC; ++k) {
                    if (k >= r) return;
                    xs[row] = row == k ? nd : a[k];
                    tri_wave_order();
                    const double dinv = xs[k];
                    const double f = row == k ? 1.0 - dinv : a[k] * dinv;
                    if (k + 1 < RC) {  // the next pivot first: its reciprocal is on the chain to the next sweep
                        const double pn = xs[k + 1];
                        a[k + 1] = fma(-f, pn, a[k + 1]);
                    }
#pragma unroll
                    for (int j = 0; j < RC; j += 2) {
                        const double2 pv = *reinterpret_cast<const double2*>(xs + j);
                        if (j != k + 1) a[j] = fma(-f, pv.x, a[j]);
                        if (j + 1 < RC && j + 1 != k + 1) a[j + 1] = fma(-f, pv.y, a[j + 1]);
                    }
                    a[k] = row == k ? -dinv : f;
                    if (k + 1 < RC) nd = lr_rcp(a[k + 1]);
                    // every LDS read of the sweep up front (ONE round trip), then the arithmetic: left alone the scheduler
                    // recycles a handful of registers and takes six dependent round trips per sweep
                    // (rank class 32: sixteen 16-byte reads in flight on top of the 32-entry row leave the allocator no
                    // room -- measured 1.5x slower with the hint)
                    if constexpr (RC <= 24) {
                        __builtin_amdgcn_sched_group_barrier(0x100, RC / 2 + 2, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4 * RC, 0);
                    }
                    tri_wave_order();
                }
            };
            sweeps();
            // a = -(M^-1)[row][:]; stored with the off-diagonal entries doubled: the contraction runs over i >= j
            const int base = rowv * (rowv + 1) / 2;
#pragma unroll
            for (int j = 0; j < RC; ++j)  // (no branches: the upper half and the idle rows go to the trash slot)
                if (j < r) Ms[(rowok && j <= rowv) ? base + j : NPZ + 1] = j == rowv ? -a[j] : -2.0 * a[j];
        }
    }
    }
    __syncthreads();
    stamp(2);
    // ---- phase 3: < M^-1, U' diag(wt d) U > and < M^-1, Ud' diag(wt) U + U' diag(wt) Ud > ----
    double ts[4] = {0.0, 0.0, 0.0, 0.0}, cs[4] = {0.0, 0.0, 0.0, 0.0};
    // One kind of tiles at a time -- same-parity pairs with (w+, w+ d), then mixed pairs with (w-, w- d) -- so that only
    // ONE pair of weight arrays (2 NK doubles) is live beside the accumulators: the four arrays at once were the
    // register peak of the kernel.  Same tiles in the same order per wave as one loop over all of them.
    double wa[NK], wb[NK];
    auto weights_kind = [&](int kind) {
        double ds = 0.0;
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int tau = 4 * kk + g;
            const bool in1 = sval && tau < nt, in2 = sval && tau < h;
            const double w1 = in1 ? w[(r0 + tau) * L + l] : 0.0;
            const double w2 = in2 ? w[(r0 + T - 1 - tau) * L + l] : 0.0;
            const double d1 = lr_rcp(fma(eps, w1, 1.0)), d2 = lr_rcp(fma(eps, w2, 1.0));
            const double t1 = w1 * d1, t2 = w2 * d2;
            const double q1 = t1 * d1, q2 = t2 * d2;
            const double fa = in2 ? 0.5 : 1.0;  // the middle row of an odd window is its own mirror image
            if (kind == 0) {
                wa[kk] = fa * (t1 + t2);
                wb[kk] = fa * (q1 + q2);
                ds += (in1 ? d1 : 0.0) + (in2 ? d2 : 0.0);
            } else {
                wa[kk] = in2 ? 0.5 * (t1 - t2) : 0.0;
                wb[kk] = in2 ? 0.5 * (q1 - q2) : 0.0;
            }
        }
        if (kind == 0) dsum = ds;
    };
    auto tile3 = [&](int q) {
        const unsigned code = codes[q * 16 + c];
        const bool valid = code != 0xffffu;
        const int i = valid ? (int)(code >> 8) : r, j = valid ? (int)(code & 255u) : r;
        const double* ui = Ul + g * LDU + i;
        const double* uj = Ul + g * LDU + j;
        const double* di = Udl + g * LDU + i;
        const double* dj = Udl + g * LDU + j;
        hm_d4 accB = hm_d4{0.0, 0.0, 0.0, 0.0}, accD = hm_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < NK; ++kk) {
            const int o = 4 * kk * LDU;
            const double vi = ui[o], vj = uj[o];
            const double pb = vi * vj;
            const double pd = fma(di[o], vj, vi * dj[o]);
            accB = __builtin_amdgcn_mfma_f64_16x16x4f64(wb[kk], pb, accB, 0, 0, 0);
            accD = __builtin_amdgcn_mfma_f64_16x16x4f64(wa[kk], pd, accD, 0, 0, 0);
        }
        const int idx = valid ? i * (i + 1) / 2 + j : NPZ;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const double m = Mp[(g + 4 * p) * NPS + idx];
            ts[p] = fma(m, accB[p], ts[p]);
            cs[p] = fma(m, accD[p], cs[p]);
        }
    };
    weights_kind(0);
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    for (int q = wid; q < ns_tiles; q += NW) tile3(q);
    weights_kind(1);
    for (int q = ns_tiles + ((wid - ns_tiles % NW) + NW) % NW; q < n_tiles; q += NW) tile3(q);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double tsum = lr_row_sum_to15(ts[p]), csum = lr_row_sum_to15(cs[p]);
        if (c == 15) {
            red[(wid * 16 + g + 4 * p) * 2 + 0] = tsum;
            red[(wid * 16 + g + 4 * p) * 2 + 1] = csum;
        }
    }
    __syncthreads();
    if (wid == 0) {
        double t = 0.0, cc = 0.0;
#pragma unroll
        for (int wv = 0; wv < NW; ++wv) {
            t += red[(wv * 16 + c) * 2 + 0];
            cc += red[(wv * 16 + c) * 2 + 1];
        }
        double tA = (sval && g == 0) ? dsum - t : 0.0;
        double cv = (sval && g == 0) ? cc : 0.0;
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) {  // lanes 0..15 hold the 16 segments
            tA += __shfl_xor(tA, o, 64);
            cv += __shfl_xor(cv, o, 64);
        }
        out_tr = tA;
        out_cs = cv;
    }
    stamp(3);
}
