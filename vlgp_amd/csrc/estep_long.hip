// E-step for LONG units (full-length trials: core.infer, update_w, update_v of the reference,
// vlgp/core.py:22-120, 260-266, 419-471) on gfx950.
//
// Same algorithm as estep.hip (one persistent workgroup per unit, I + G'WG factored once per
// sweep, v_t = |X g_t|^2, y read once per launch), but organised for T in the hundreds or
// thousands with the prior factor G (T x r, r <= 50) streamed from L2 instead of LDS:
//   * H = I + G'WG : (latent, 16x16 tile) tasks spread over all eight waves, accumulated on the
//     matrix pipe (v_mfma_f64_16x16x4, four time bins per instruction), written straight into the
//     packed LDS layout the factorisation reads;
//   * factor + inverse: one wave per latent, rows/columns in registers (wave_tri.h, size 50 with
//     identity padding above the effective rank);
//   * v_t = |X g_t|^2 : Z = X G' as (latent, 16 time bins) MFMA tasks, X tiles held in registers,
//     column norms reduced in the MFMA result layout;
//   * the T-long reductions of the mean update (G'r, G'(w u)) split the time axis over the waves
//     and sum the partials in a fixed order; the T-long expansions (G vec) map threads to bins.
// The generic kernel did all per-latent work with ONE wave per latent (latency-bound: 12 ms per
// sweep for 200 x 1000-bin trials); this one takes ~0.3 ms.
#include <stdlib.h>

#include <type_traits>

#include "ctx.h"

#include "estep_args.h"
#include "fast_exp.h"
#include "wave_tri.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

namespace {

constexpr int RPL = 50;                       // factor size = the reference's fixed rank (preprocess.py:80)
constexpr int PKL = tri_packed_size(RPL);     // packed lower-triangular 50 x 50 (rows padded to even)
constexpr int NWL = 8;                        // waves per workgroup
constexpr int NTL = NWL * 64;

enum { LPASS_YA = 0, LPASS_RES = 1, LPASS_W = 2 };

__device__ __forceinline__ void tile_of(int tile, int& bi, int& bj) {
    // lower block triangle of a 4 x 4 block matrix, row-major
    bi = tile < 1 ? 0 : (tile < 3 ? 1 : (tile < 6 ? 2 : 3));
    bj = tile - (bi * (bi + 1)) / 2;
}

template <int LT>
__global__ void __launch_bounds__(NTL, 2) estep_long_kernel(EstepArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int N = A.N, L = A.L;
    const int m = blockIdx.x;
    const int64_t r0 = A.off[m];
    const int T = (int)(A.off[m + 1] - r0);
    const int pidx = A.unit_prior ? A.unit_prior[m] : -1;
    const double* Gbase = pidx >= 0 ? A.prior_base[pidx] : nullptr;

    // ---- LDS ---------------------------------------------------------------
    double* p = smem;
    double* a_s = p;      p += LT * N;   // rows l >= L are zero: the (T x N) passes need no l < L tests
    double* asq_s = p;    p += LT * N;
    double* bvec = p;     p += N;
    double* cn = p;       p += N;
    double* wconst = p;   p += (L + 1) & ~1;
    double* vec_s = p;    p += L * 64;
    double* vec2_s = p;   p += L * 64;
    double* Xp = p;       p += (int64_t)L * PKL;
    int* ip = reinterpret_cast<int*>(p);
    int* gauss_s = ip;  ip += N;
    int* rl_s = ip;     ip += L;
    int* goff_s = ip;   ip += L;
    int* fail_s = ip;   ip += L;

    // ---- unit state in HBM / L2 ----------------------------------------------
    double* mu_g = A.mu + r0 * L;
    double* v_g = A.v + r0 * L;
    double* w_g = A.w + r0 * L;
    double* ra_g = A.scratch + 3 * r0 * L;
    double* ya_g = ra_g + (int64_t)T * L;
    double* u_g = ya_g + (int64_t)T * L;                               // [l][T]
    double* part_g = A.lc_global + (int64_t)m * A.lc_stride;           // [wave][l][64]

    for (int i = tid; i < LT * N; i += NTL) {
        const double av = i < L * N ? A.a[i] : 0.0;
        a_s[i] = av;
        asq_s[i] = av * av;
    }
    for (int n = tid; n < N; n += NTL) {
        const int g = A.gauss[n];
        gauss_s[n] = g;
        bvec[n] = A.b[n];
        cn[n] = g ? 1.0 / A.noise[n] : 1.0;
    }
    if (tid < L) {
        rl_s[tid] = pidx >= 0 ? A.prior_rl[pidx * L + tid] : 0;
        goff_s[tid] = pidx >= 0 ? (int)A.prior_goff[pidx * L + tid] : 0;
        fail_s[tid] = 0;
    }
    __syncthreads();
    if (tid < L) {  // Gaussian channels contribute a constant to w (core.py:103-104)
        double s = 0.0;
        for (int n = 0; n < N; ++n)
            if (gauss_s[n]) s = fma(asq_s[tid * N + n], cn[n], s);
        wconst[tid] = s;
    }
    __syncthreads();

    const int RG = A.rg;
    const int sub = tid & (RG - 1), rgid = tid / RG, nrg = NTL / RG;
    const bool has_xb = A.xb != nullptr;

    // ---- (T x N) passes: lanes of a row group stride over channels; RB rows in flight per
    // group so that the y / mu / v loads of several rows overlap (the unit state is in HBM/L2) ----
    auto tn_pass = [&](auto kind_c) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr int RB = LT <= 5 ? 4 : 2;
        for (int row0 = rgid; row0 < T; row0 += nrg * RB) {
            double mr[RB][LT], vr[RB][LT], acc[RB][LT];
            int rows[RB];
#pragma unroll
            for (int q = 0; q < RB; ++q) {
                const int row = row0 + q * nrg;
                rows[q] = row < T ? row : T - 1;  // clamped: loads stay in bounds, the store is masked
#pragma unroll
                for (int l = 0; l < LT; ++l) {
                    // unconditional (clamped) loads: a select keeps them in one block, a branch per
                    // load serialises their latencies
                    const int lc = l < L ? l : 0;
                    const double mv = KIND != LPASS_YA ? mu_g[rows[q] * L + lc] : 0.0;
                    const double vv = KIND != LPASS_YA ? v_g[rows[q] * L + lc] : 0.0;
                    mr[q][l] = l < L ? mv : 0.0;
                    vr[q][l] = l < L ? vv : 0.0;
                    acc[q][l] = 0.0;
                }
            }
            for (int n = sub; n < N; n += RG) {
                double yv[RB];
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    if constexpr (KIND == LPASS_YA) yv[q] = A.y[(r0 + rows[q]) * N + n];
                    else yv[q] = has_xb ? A.xb[(r0 + rows[q]) * N + n] : 0.0;
                }
                const int g = gauss_s[n];
                const double cnn = cn[n];
                const double bn = has_xb ? 0.0 : bvec[n];
                double al[LT], aq[LT];
#pragma unroll
                for (int l = 0; l < LT; ++l) {
                    al[l] = a_s[l * N + n];
                    aq[l] = asq_s[l * N + n];
                }
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    if constexpr (KIND == LPASS_YA) {
                        const double yc = yv[q] * cnn;
#pragma unroll
                        for (int l = 0; l < LT; ++l) acc[q][l] = fma(yc, al[l], acc[q][l]);
                    } else {
                        double eta = yv[q] + bn;
                        double lin = 0.0;
#pragma unroll
                        for (int l = 0; l < LT; ++l) {
                            eta = fma(mr[q][l], al[l], eta);
                            lin = fma(vr[q][l], aq[l], lin);
                        }
                        if constexpr (KIND == LPASS_RES) {
                            const double pois = fast_exp(clamp10(fma(0.5, lin, eta)));
                            const double mval = g ? eta * cnn : pois;
#pragma unroll
                            for (int l = 0; l < LT; ++l) acc[q][l] = fma(mval, al[l], acc[q][l]);
                        } else {
                            const double rate = g ? 0.0 : fast_exp(clamp10(fma(0.5, lin, eta)));
#pragma unroll
                            for (int l = 0; l < LT; ++l) acc[q][l] = fma(rate, aq[l], acc[q][l]);
                        }
                    }
                }
            }
            for (int o = RG >> 1; o > 0; o >>= 1) {
#pragma unroll
                for (int q = 0; q < RB; ++q)
#pragma unroll
                    for (int l = 0; l < LT; ++l) acc[q][l] += __shfl_xor(acc[q][l], o, 64);
            }
            if (sub == 0) {
#pragma unroll
                for (int q = 0; q < RB; ++q) {
                    const int row = row0 + q * nrg;
                    if (row < T) {
#pragma unroll
                        for (int l = 0; l < LT; ++l)
                            if (l < L) {
                                if constexpr (KIND == LPASS_YA) ya_g[row * L + l] = acc[q][l];
                                else if constexpr (KIND == LPASS_RES) ra_g[row * L + l] = ya_g[row * L + l] - acc[q][l];
                                else w_g[row * L + l] = acc[q][l] + wconst[l];
                            }
                    }
                }
            }
        }
    };

    // optional per-phase cycle counters (thread 0 of every block; vlgp_debug_phase_clock)
    unsigned long long tick = A.clk ? __builtin_readcyclecounter() : 0, tick2 = tick;
    auto lap = [&](int slot) {
        if (A.clk && tid == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            atomicAdd(A.clk + slot, now - tick);
            tick = now;
        }
    };
    auto lap2 = [&](int slot) {  // sub-phases of the factor phase: slots 6 (build) and 7 (factor + inverse)
        if (A.clk && tid == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            if (slot >= 0) atomicAdd(A.clk + slot, now - tick2);
            tick2 = now;
        }
    };

    // ---- factor I + G'WG, invert the factor, optionally refresh v ---------------
    auto factor_phase = [&](bool do_v) {
        lap2(-1);
        // identity everywhere first: tiles above the effective rank are never visited
        for (int i = tid; i < L * PKL; i += NTL) Xp[i] = 0.0;
        __syncthreads();
        for (int i = tid; i < L * RPL; i += NTL) {
            const int l = i / RPL, d = i - l * RPL;
            Xp[(int64_t)l * PKL + tri_row_off(d) + d] = 1.0;
        }
        __syncthreads();
        const int col = lane & 15, kq = lane >> 4;
        // F1: H tiles on the matrix pipe, one (latent, block row bi) task per wave at a time: the A
        // operand w_t G[t][16 bi + col] is loaded once and serves the tiles bj = 0..bi of the row.
        // Lane (col, kq) feeds A[col][kq] and B[kq][col] = G[t][16 bj + col], t = t0 + kq;
        // D[row = kq + 4 q][col] comes back in c[bj][q].  Heaviest rows first (fixed schedule).
        for (int task = wid; task < L * 4; task += NWL) {
            const int bi = 3 - task / L, l = task - (task / L) * L;
            const int r = rl_s[l];
            if (16 * bi >= r) continue;  // wave-uniform
            const double* Gl = Gbase + goff_s[l];
            const int ca = 16 * bi + col;
            const bool ina = ca < r;
            const int cas = ina ? ca : 0;
            double4_t c[4];
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) c[bj] = double4_t{0.0, 0.0, 0.0, 0.0};
            for (int t0 = 0; t0 < T; t0 += 16) {  // four k-steps of loads in flight before the MFMAs
                double ga[4], gb[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int t = t0 + 4 * u + kq;
                    const int tc = t < T ? t : T - 1;
                    const double* row = Gl + tc * r;
                    const double wv = t < T ? w_g[tc * L + l] : 0.0;
                    ga[u] = ina ? wv * row[cas] : 0.0;
#pragma unroll
                    for (int bj = 0; bj < 4; ++bj) {
                        const int cb = 16 * bj + col;
                        gb[u][bj] = (bj < bi) ? row[cb] : (bj == bi ? (ina ? row[cas] : 0.0) : 0.0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int bj = 0; bj < 4; ++bj)
                        if (bj <= bi) c[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[u], gb[u][bj], c[bj], 0, 0, 0);
            }
            double* Xl = Xp + (int64_t)l * PKL;
#pragma unroll
            for (int bj = 0; bj < 4; ++bj) {
                if (bj > bi) continue;
                const int cb = 16 * bj + col;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int row = 16 * bi + kq + 4 * q;
                    if (cb <= row && row < r) Xl[tri_row_off(row) + cb] = c[bj][q] + (cb == row ? 1.0 : 0.0);
                }
            }
        }
        __syncthreads();
        lap2(6);
        // F2: one wave per latent, rows / columns in registers
        for (int l = wid; l < L; l += NWL) {
            double* Xl = Xp + (int64_t)l * PKL;
            bool ok;
            {
                double rr[RPL];
                ok = wave_chol_rows<RPL>(rr, Xl, lane);
            }
            {
                double x[RPL];
                wave_tri_inverse_cols<RPL>(Xl, x, lane);
                tri_wave_sync();
                if (lane < RPL) {  // X overwrites L, row-major packed: X[i][c] for i >= c
#pragma unroll
                    for (int i = 0; i < RPL; ++i)
                        if (i >= lane) Xl[tri_row_off(i) + lane] = x[i];
                }
            }
            if (lane == 0) {
                fail_s[l] = ok ? 0 : 1;
                if (!ok) atomicAdd(A.fail, 1);
            }
        }
        __syncthreads();
        lap2(7);
        if (!do_v) return;
        // F3: v_t = |X g_t|^2.  Z = X G' in 16-bin column blocks; X tiles (ib, kb <= ib) live in
        // registers as A operands (A[m][k]: lane (m = lane & 15, k = 4 s + (lane >> 4))), the B
        // operand is G'[k][n] = G[t0 + n][k].  D[row = kq + 4 q][n]: sum of squares over rows.
        const int ntb = (T + 15) / 16;
        const int ntask = L * ntb;
        const int lo = (int)((int64_t)ntask * wid / NWL), hi = (int)((int64_t)ntask * (wid + 1) / NWL);
        int cur = -1;
        double xa[10][4];
        for (int task = lo; task < hi; ++task) {
            const int l = task / ntb, tb = task - l * ntb;
            if (fail_s[l]) continue;  // v keeps its value (core.py:112-113)
            const int r = rl_s[l];
            if (l != cur) {
                cur = l;
                const double* Xl = Xp + (int64_t)l * PKL;
#pragma unroll
                for (int pr = 0; pr < 10; ++pr) {
                    int ib, kb;
                    tile_of(pr, ib, kb);
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int i = 16 * ib + col, k = 16 * kb + 4 * s + kq;
                        double val = (i == k) ? 1.0 : 0.0;
                        if (i < RPL) val = (k <= i) ? Xl[tri_row_off(i) + k] : 0.0;
                        xa[pr][s] = val;
                    }
                }
            }
            const double* Gl = Gbase + goff_s[l];
            const int t = 16 * tb + col;
            const bool tin = t < T;
            double4_t acc[4];
#pragma unroll
            for (int ib = 0; ib < 4; ++ib) acc[ib] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) {
                if (16 * kb >= r) continue;  // wave-uniform: those columns of G are zero
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k = 16 * kb + 4 * s + kq;
                    const double gB = (tin && k < r) ? Gl[(int64_t)t * r + k] : 0.0;
#pragma unroll
                    for (int ib = kb; ib < 4; ++ib)
                        acc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[(ib * (ib + 1)) / 2 + kb][s], gB, acc[ib], 0, 0, 0);
                }
            }
            double vv = 0.0;
#pragma unroll
            for (int ib = 0; ib < 4; ++ib)
#pragma unroll
                for (int q = 0; q < 4; ++q) vv = fma(acc[ib][q], acc[ib][q], vv);
            vv += __shfl_xor(vv, 16, 64);
            vv += __shfl_xor(vv, 32, 64);
            if (kq == 0 && tin) v_g[t * L + l] = vv;
        }
        __syncthreads();
    };

    // ---- Newton step on the posterior mean ----------------------------------------
    auto mean_phase = [&](bool last) {
        const int Tc = (T + NWL - 1) / NWL;
        const int ta = wid * Tc, tb = (ta + Tc < T) ? ta + Tc : T;
        // S1: g1_l = G_l' ra_l, time axis split over the waves (lane = column)
        for (int l = 0; l < L; ++l) {
            const int r = rl_s[l];
            const double* Gl = Gbase + goff_s[l];
            double acc = 0.0, acc1 = 0.0;
            if (lane < r) {
                int t = ta;
                for (; t + 8 <= tb; t += 8) {
                    double gv[8], rv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        gv[q] = Gl[(int64_t)(t + q) * r + lane];
                        rv[q] = ra_g[(t + q) * L + l];
                    }
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        acc = fma(gv[q], rv[q], acc);
                        acc1 = fma(gv[q + 1], rv[q + 1], acc1);
                    }
                }
                for (; t < tb; ++t) acc = fma(Gl[(int64_t)t * r + lane], ra_g[t * L + l], acc);
            }
            part_g[(wid * L + l) * 64 + lane] = acc + acc1;
        }
        __syncthreads();
        for (int i = tid; i < L * 64; i += NTL) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NWL; ++w) s += part_g[w * L * 64 + i];
            vec_s[i] = s;
        }
        __syncthreads();
        // S2: u_l = G_l g1_l - mu_l (thread = time bin; a wave-reduction variant with coalesced rows
        // measured 1.7x slower: the shuffles cost more than the strided row reads)
        for (int t = tid; t < T; t += NTL)
            for (int l = 0; l < L; ++l) {
                const int r = rl_s[l];
                const double* Gt = Gbase + goff_s[l] + (int64_t)t * r;
                const double* vl = vec_s + l * 64;
                double s0 = 0.0, s1 = 0.0;
                int i = 0;
                for (; i + 8 <= r; i += 8) {
                    double gv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) gv[q] = Gt[i + q];
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        s0 = fma(gv[q], vl[i + q], s0);
                        s1 = fma(gv[q + 1], vl[i + q + 1], s1);
                    }
                }
                for (; i < r; ++i) s0 = fma(Gt[i], vl[i], s0);
                u_g[(int64_t)l * T + t] = (s0 + s1) - mu_g[t * L + l];
            }
        __syncthreads();
        // S3: rhs_l = G_l' (w_l u_l)
        for (int l = 0; l < L; ++l) {
            const int r = rl_s[l];
            const double* Gl = Gbase + goff_s[l];
            double acc = 0.0, acc1 = 0.0;
            if (lane < r) {
                int t = ta;
                for (; t + 8 <= tb; t += 8) {
                    double gv[8], rv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        gv[q] = Gl[(int64_t)(t + q) * r + lane];
                        rv[q] = w_g[(t + q) * L + l] * u_g[(int64_t)l * T + t + q];
                    }
#pragma unroll
                    for (int q = 0; q < 8; q += 2) {
                        acc = fma(gv[q], rv[q], acc);
                        acc1 = fma(gv[q + 1], rv[q + 1], acc1);
                    }
                }
                for (; t < tb; ++t)
                    acc = fma(w_g[t * L + l] * Gl[(int64_t)t * r + lane], u_g[(int64_t)l * T + t], acc);
            }
            part_g[(wid * L + l) * 64 + lane] = acc + acc1;
        }
        __syncthreads();
        for (int i = tid; i < L * 64; i += NTL) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NWL; ++w) s += part_g[w * L * 64 + i];
            vec_s[i] = s;
        }
        __syncthreads();
        // S4: sol = X'(X rhs), one wave per latent
        for (int l = wid; l < L; l += NWL) {
            const double* Xl = Xp + (int64_t)l * PKL;
            double* vl = vec_s + l * 64;
            double* v2 = vec2_s + l * 64;
            double z = 0.0;
            if (lane < RPL) {
                const double* Xi = Xl + tri_row_off(lane);
                for (int c = 0; c <= lane; ++c) z = fma(Xi[c], vl[c], z);
            }
            v2[lane] = z;
            tri_wave_sync();
            double sol = 0.0;
            if (lane < RPL)
                for (int i = lane; i < RPL; ++i) sol = fma(Xl[tri_row_off(i) + lane], v2[i], sol);
            tri_wave_sync();
            vl[lane] = sol;
        }
        __syncthreads();
        // S5: delta = u - G sol, clipped; mu += delta
        for (int t = tid; t < T; t += NTL)
            for (int l = 0; l < L; ++l) {
                double s = 0.0;
                if (!fail_s[l]) {  // singular system: zero update (core.py:92-94)
                    const int r = rl_s[l];
                    const double* Gt = Gbase + goff_s[l] + (int64_t)t * r;
                    const double* vl = vec_s + l * 64;
                    double s0 = u_g[(int64_t)l * T + t], s1 = 0.0;
                    int i = 0;
                    for (; i + 8 <= r; i += 8) {
                        double gv[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) gv[q] = Gt[i + q];
#pragma unroll
                        for (int q = 0; q < 8; q += 2) {
                            s0 = fma(-gv[q], vl[i + q], s0);
                            s1 = fma(-gv[q + 1], vl[i + q + 1], s1);
                        }
                    }
                    for (; i < r; ++i) s0 = fma(-Gt[i], vl[i], s0);
                    s = fmin(fmax(s0 + s1, -A.dmu_bound), A.dmu_bound);
                    mu_g[t * L + l] += s;
                }
                if (last) A.dmu[(r0 + t) * L + l] = s;
            }
        if (tid < L && fail_s[tid]) atomicAdd(A.fail, 1);
        __syncthreads();
    };

    // ---- schedule (as estep.hip) -----------------------------------------------------
    const int mode = A.mode;
    lap(0);
    if (mode & EM_MEAN) tn_pass(std::integral_constant<int, LPASS_YA>{});
    if (mode & EM_FACTOR0) factor_phase((mode & EM_V) && !(mode & EM_MEAN));
    __syncthreads();
    lap(1);
    if (mode & EM_MEAN) {
        for (int it = 0; it < A.n_iter; ++it) {
            const bool last = it == A.n_iter - 1;
            tn_pass(std::integral_constant<int, LPASS_RES>{});
            __syncthreads();
            lap(2);
            mean_phase(last);
            lap(3);
            tn_pass(std::integral_constant<int, LPASS_W>{});
            __syncthreads();
            lap(4);
            if (A.vb || !last) factor_phase(A.vb != 0);
            __syncthreads();
            lap(5);
        }
    } else if (mode & EM_W) {
        tn_pass(std::integral_constant<int, LPASS_W>{});
        __syncthreads();
    }
}

template <int LT>
int launch_long_t(vlgp_ctx* ctx, const EstepArgs& A, int M, size_t lds) {
    auto fn = estep_long_kernel<LT>;
    if (lds > 64 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    hipLaunchKernelGGL(fn, dim3(M), dim3(NTL), lds, ctx->stream, A);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

}  // namespace

// Sets *handled = 1 and launches when the long-unit kernel applies (rank <= 50, L <= 10, LDS fits).
int launch_estep_long(vlgp_ctx* ctx, UnitSet& us, EstepArgs A, int* handled) {
    *handled = 0;
    const int N = ctx->N, L = ctx->L;
    if (ctx->R > RPL || L > 10 || getenv("VLGP_ESTEP_GENERIC")) return VLGP_OK;
    const int LT = L <= 3 ? 3 : (L <= 5 ? 5 : (L <= 8 ? 8 : 10));
    const int64_t doubles = 2LL * LT * N + 2LL * N + ((L + 1) & ~1) + 2LL * L * 64 + (int64_t)L * PKL;
    const int64_t ints = ((int64_t)N + 3 * L + 1) / 2 + 1;
    const size_t lds = (size_t)(doubles + ints) * 8;
    if (lds > 160 * 1024) return VLGP_OK;
    const int64_t need = 3 * us.rows * L;
    const int64_t part = (int64_t)NWL * L * 64;
    if (us.scratch_len < need + part * us.M) {
        if (us.d_scratch) HIPCHK(ctx, hipFree(us.d_scratch));
        us.d_scratch = nullptr;
        HIPCHK(ctx, hipMalloc(&us.d_scratch, (size_t)(need + part * us.M) * 8));
        us.scratch_len = need + part * us.M;
    }
    A.scratch = us.d_scratch;
    A.lc_global = us.d_scratch + need;  // here: per-unit partial-sum area
    A.lc_stride = part;
    A.rg = N >= 16 ? 4 : 1;  // lanes per row in the (T x N) passes (measured: 4 beats 64 by 6x at N = 100)
    vlgp_prof_begin(ctx, VLGP_PROF_ESTEP_LONG);
    int rc;
    if (LT == 3) rc = launch_long_t<3>(ctx, A, us.M, lds);
    else if (LT == 5) rc = launch_long_t<5>(ctx, A, us.M, lds);
    else if (LT == 8) rc = launch_long_t<8>(ctx, A, us.M, lds);
    else rc = launch_long_t<10>(ctx, A, us.M, lds);
    vlgp_prof_end(ctx, VLGP_PROF_ESTEP_LONG, (double)us.M * (A.n_iter > 0 ? A.n_iter : 1));
    if (rc != VLGP_OK) return rc;
    *handled = 1;
    return VLGP_OK;
}
