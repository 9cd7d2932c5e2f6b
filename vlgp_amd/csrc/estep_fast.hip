// Fast E-step kernel for window-sized units (T <= 64 bins, effective rank <= 32).
//
// Same mathematics and phase order as estep.hip (see its header; reference
// vlgp/core.py:22-120); what changes is the mapping onto the CU:
//
//   (T x N) passes   lane <-> time bin, each wave owns a contiguous block of
//                    channels.  The per-channel record (a_l, a_l^2, b, 1/noise)
//                    is wave-uniform -> one broadcast ds_read_b128 per two
//                    values, no cross-lane reduction at all; the per-(t, l)
//                    sums of the waves meet once in LDS.  Gaussian channels are
//                    a wave-uniform branch (no exp, no divergence).
//   factor phases    one wave per latent, the matrix I + G'WG (padded with the
//                    identity to RP = 16 or 32) is built with lanes =
//                    (row, time-chunk), factored and inverted with each lane's
//                    row / column in registers (wave_tri.h): the only LDS
//                    traffic is the broadcast pivot row.
//   solves           (I+H)^-1 rhs = X'(X rhs) and v_t = |X g_t|^2 as dense
//                    products with X rows broadcast from LDS and the G row of
//                    lane t in registers.
#include <stdlib.h>

#include <type_traits>

#include "estep_args.h"
#include "wave_tri.h"
#include "fast_exp.h"

enum { FP_YA = 0, FP_RES = 1, FP_W = 2 };
typedef double double4_t __attribute__((ext_vector_type(4)));

// RP: lane mapping of the per-latent phases (power of two); RA <= RP: size of the register arrays and of
// the unrolled factor / solve loops (RA = 24 serves ranks 17..24 with 3/4 of the registers of RA = 32,
// which is what lets two workgroups share a CU).
template <int LT, int RP, int RA = RP>
__global__ void __launch_bounds__((LT > 8 ? 640 : 512), (LT > 8 ? 3 : (RP <= 16 ? 4 : (RA <= 24 ? 3 : 1))))
estep_fast_kernel(EstepArgs A, const double* __restrict__ cols_g) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int REC = 2 * LT + 2;          // a[LT], a^2[LT], b, c  (even -> 16-byte records)
    constexpr int PK = tri_packed_size(RA);  // packed lower-triangular RA x RA
    constexpr int NCH = 64 / RP;             // time chunks per wave in the (row, chunk) mapping
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wid = tid >> 6, nw = nthr >> 6;
    const int N = A.N, L = A.L;
    const int m = blockIdx.x;
    const int64_t r0 = A.off[m];
    const int T = (int)(A.off[m + 1] - r0);
    const int Tc = A.lds_T;
    const int pidx = A.unit_prior ? A.unit_prior[m] : -1;
    const double* Gbase = pidx >= 0 ? A.prior_base[pidx] : nullptr;

    double* p = smem;
    double* mu_s = p;    p += Tc * L;
    double* v_s = p;     p += Tc * L;
    double* w_s = p;     p += Tc * L;
    double* ra_s = p;    p += Tc * L;
    double* ya_s = p;    p += Tc * L;
    p += ((p - smem) & 1);  // keep 16-byte alignment for the vector-read regions below
    // One scratch region, three tenants that never overlap in time (workgroup barriers in
    // between): `part` in the (T x N) passes, `vec`/`u` in the mean update, `tile` in the
    // factor build.
    double* scr = p;     p += A.lds_scr;
    double* part = scr;
    double* vec_s = scr;
    double* u_s = scr + nw * 128;
    double* tile_s = scr;
    double* Xp = p;      p += (int64_t)L * PK;
    double* G_s = p;     p += A.lds_gsz;
    int* ip = reinterpret_cast<int*>(p);
    int* gflag = ip;     ip += N;
    int* clist = ip;     ip += N;   // Poisson channels first, then Gaussian ones
    int* ncnt = ip;      ip += 2;   // [0] = number of Poisson channels
    int* rl_s = ip;      ip += L;
    int* goff_s = ip;    ip += L;
    int* fail_s = ip;    ip += L;

    // ---- stage ---------------------------------------------------------------
    for (int n = tid; n < N; n += nthr) gflag[n] = A.gauss[n];
    if (tid == 64 || (nthr <= 64 && tid == 0)) {
        int np = 0;
        for (int n = 0; n < N; ++n)
            if (!A.gauss[n]) clist[np++] = n;
        ncnt[0] = np;
        for (int n = 0; n < N; ++n)
            if (A.gauss[n]) clist[np++] = n;
    }
    if (tid == 0) {
        int go = 0;
        for (int l = 0; l < L; ++l) {
            const int r = pidx >= 0 ? A.prior_rl[pidx * L + l] : 0;
            rl_s[l] = r;
            goff_s[l] = go;
            go += T * ((r + 1) & ~1);
            fail_s[l] = 0;
        }
    }
    for (int i = tid; i < T * L; i += nthr) {
        mu_s[i] = A.mu[r0 * L + i];
        v_s[i] = A.v[r0 * L + i];
        w_s[i] = A.w[r0 * L + i];
    }
    __syncthreads();
    for (int l = 0; l < L; ++l) {
        const int r = rl_s[l], rs = (r + 1) & ~1;
        if (r == 0) continue;
        const double* src = Gbase + A.prior_goff[pidx * L + l];
        double* dst = G_s + goff_s[l];
        for (int i = tid; i < T * rs; i += nthr) {
            const int t = i / rs, c = i - t * rs;
            dst[i] = c < r ? src[t * r + c] : 0.0;
        }
    }
    __syncthreads();

    // ---- (T x N) passes ----------------------------------------------------------
    auto tn_pass_x = [&](auto kind_c, auto hasxb_c) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr bool HASXB = decltype(hasxb_c)::value;
        const int t = lane;
        const bool in = t < T;
        double mr[LT], vr[LT], accA[LT], accB[LT];
#pragma unroll
        for (int l = 0; l < LT; ++l) {
            const bool use = in && l < L && KIND != FP_YA;
            mr[l] = use ? mu_s[t * L + l] : 0.0;
            vr[l] = use ? v_s[t * L + l] : 0.0;
            accA[l] = 0.0;
            accB[l] = 0.0;
        }
        const double* yrow = A.y + (r0 + (in ? t : 0)) * N;
        const double* xbrow = HASXB ? A.xb + (r0 + (in ? t : 0)) * N : nullptr;
        // the record is wave-uniform: with a scalar channel index the loads below are scalar
        // (s_load_dwordx4) and the values feed the FMAs straight from SGPRs
        auto load_rec = [&](int n_any, double (&rv)[REC]) {
            const int n = __builtin_amdgcn_readfirstlane(n_any);
            const double2* rp = reinterpret_cast<const double2*>(cols_g + (int64_t)n * REC);
#pragma unroll
            for (int q = 0; q < REC / 2; ++q) {
                const double2 t2 = rp[q];
                rv[2 * q] = t2.x;
                rv[2 * q + 1] = t2.y;
            }
        };
        // one Poisson channel: rate = exp(min(eta + v.a^2/2, 10)); branch-free so that two
        // channels issued back to back interleave their dependent fp64 chains
        auto poisson_col = [&](int n, double (&acc)[LT]) {
            double rv[REC];
            load_rec(n, rv);
            double eta = HASXB ? xbrow[n] : rv[2 * LT];
            double lin = 0.0;
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                eta = fma(mr[l], rv[l], eta);
                lin = fma(vr[l], rv[LT + l], lin);
            }
            const double rate = fast_exp(clamp10(fma(0.5, lin, eta)));
#pragma unroll
            for (int l = 0; l < LT; ++l) acc[l] = fma(rate, rv[KIND == FP_RES ? l : LT + l], acc[l]);
        };
        const int np = ncnt[0];
        if constexpr (KIND == FP_YA) {
            const int n0 = (N * wid) / nw, n1 = (N * (wid + 1)) / nw;
            for (int n = n0; n < n1; ++n) {
                double rv[REC];
                load_rec(n, rv);
                const double yc = yrow[n] * rv[2 * LT + 1];
#pragma unroll
                for (int l = 0; l < LT; ++l) accA[l] = fma(yc, rv[l], accA[l]);
            }
        } else {
            const int p0 = (np * wid) / nw, p1 = (np * (wid + 1)) / nw;
            int i = p0;
            for (; i + 1 < p1; i += 2) {
                const int na = clist[i], nb = clist[i + 1];
                poisson_col(na, accA);
                poisson_col(nb, accB);
            }
            if (i < p1) poisson_col(clist[i], accA);
            if constexpr (KIND == FP_RES) {  // Gaussian channels: residual mean is eta itself, no rate
                const int ng = N - np;
                const int g0 = np + (ng * wid) / nw, g1 = np + (ng * (wid + 1)) / nw;
                for (int q = g0; q < g1; ++q) {
                    const int n = clist[q];
                    double rv[REC];
                    load_rec(n, rv);
                    double eta = HASXB ? xbrow[n] : rv[2 * LT];
#pragma unroll
                    for (int l = 0; l < LT; ++l) eta = fma(mr[l], rv[l], eta);
                    const double mval = eta * rv[2 * LT + 1];
#pragma unroll
                    for (int l = 0; l < LT; ++l) accB[l] = fma(mval, rv[l], accB[l]);
                }
            }
        }
        if (in) {
#pragma unroll
            for (int l = 0; l < LT; ++l)
                if (l < L) part[((int64_t)wid * Tc + t) * L + l] = accA[l] + accB[l];
        }
        __syncthreads();
        for (int idx = tid; idx < T * L; idx += nthr) {
            double s = 0.0;
            for (int w = 0; w < nw; ++w) s += part[(int64_t)w * Tc * L + idx];
            if constexpr (KIND == FP_YA) ya_s[idx] = s;
            else if constexpr (KIND == FP_RES) ra_s[idx] = ya_s[idx] - s;
            else w_s[idx] = s + A.wconst_g[idx % L];
        }
        __syncthreads();
    };
    auto tn_pass = [&](auto kind_c) {
        if (A.xb) tn_pass_x(kind_c, std::true_type{});
        else tn_pass_x(kind_c, std::false_type{});
    };

    // ---- factor I + G'WG, invert, optionally refresh v -------------------------------
    unsigned long long tick2 = 0;
    auto lap2 = [&](int slot) {
        if (A.clk && tid == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            if (slot >= 0) atomicAdd(A.clk + slot, now - tick2);
            tick2 = now;
        }
    };
    auto factor_phase = [&](bool do_v) {
        for (int l = wid; l < L; l += nw) {
            lap2(-1);
            const int r = __builtin_amdgcn_readfirstlane(rl_s[l]);
            const int rs = (r + 1) & ~1;
            const double* Gl = G_s + __builtin_amdgcn_readfirstlane(goff_s[l]);
            double* Xl = Xp + (int64_t)l * PK;
            const int j = lane & (RP - 1), ch = lane / RP;
            bool ok;
            // H = G' diag(w) G on the matrix pipe: v_mfma_f64_16x16x4, four time bins per
            // instruction.  Lane (col = lane & 15, kq = lane >> 4) feeds A[col][kq] = w_t G[t][col]
            // and B[kq][col] = G[t][col], t = t0 + kq; D[row = kq + 4p][col] comes back in c[p].
            const int col = lane & 15, kq = lane >> 4;
            if constexpr (RP <= 16) {
                double4_t c = {0.0, 0.0, 0.0, 0.0};
                const bool cin = col < rs;
                for (int t0 = 0; t0 < T; t0 += 4) {
                    const int t = t0 + kq;
                    double g = 0.0, wg = 0.0;
                    if (cin && t < T) {
                        g = Gl[t * rs + col];
                        wg = w_s[t * L + l] * g;
                    }
                    c = __builtin_amdgcn_mfma_f64_16x16x4f64(wg, g, c, 0, 0, 0);
                }
                double* ht = tile_s + wid * 256;  // 16 x 16 staging tile of this wave
#pragma unroll
                for (int q = 0; q < 4; ++q) ht[(kq + 4 * q) * 16 + col] = c[q];
                tri_wave_sync();
                double a[RP];
#pragma unroll
                for (int i = 0; i < RP; i += 2) {
                    const double2 h2 = *reinterpret_cast<const double2*>(ht + j * 16 + i);
                    a[i] = h2.x + (i == j ? 1.0 : 0.0);
                    a[i + 1] = h2.y + (i + 1 == j ? 1.0 : 0.0);
                }
                lap2(6);
                // all in registers: every lane holds row (lane mod RP) of I + G'WG
                double x[RP];
                ok = wave_chol_inv_regs<RP>(a, x, j, r);
                if (lane < RP) {  // X row-major packed in LDS for the solves: X[i][c], i >= c
#pragma unroll
                    for (int i = 0; i < RP; ++i)
                        if (i >= lane) Xl[tri_row_off(i) + lane] = x[i];
                }
            } else {
#pragma unroll
                for (int tile = 0; tile < 3; ++tile) {  // lower block triangle of the 32 x 32 matrix
                    const int bi = tile == 0 ? 0 : 1, bj = tile == 2 ? 1 : 0;
                    const int ca = 16 * bi + col, cb = 16 * bj + col;
                    double4_t c = {0.0, 0.0, 0.0, 0.0};
                    for (int t0 = 0; t0 < T; t0 += 4) {
                        const int t = t0 + kq;
                        double ga = 0.0, gb = 0.0;
                        if (t < T) {
                            if (ca < rs) ga = w_s[t * L + l] * Gl[t * rs + ca];
                            if (cb < rs) gb = Gl[t * rs + cb];
                        }
                        c = __builtin_amdgcn_mfma_f64_16x16x4f64(ga, gb, c, 0, 0, 0);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int row = 16 * bi + kq + 4 * q;
                        if (cb <= row && row < RA) Xl[tri_row_off(row) + cb] = c[q] + (cb == row ? 1.0 : 0.0);
                    }
                }
                lap2(6);
            }
            if constexpr (RP > 16) {
                tri_wave_sync();
                __builtin_amdgcn_sched_barrier(0);
                {
                    double rr[RA];
                    ok = wave_chol_rows<RA>(rr, Xl, lane);
                }
                {
                    double x[RA];
                    wave_tri_inverse_cols<RA>(Xl, x, lane);
                    tri_wave_sync();
                    if (lane < RA) {  // X overwrites L, row-major packed: X[i][c] for i >= c
#pragma unroll
                        for (int i = 0; i < RA; ++i)
                            if (i >= lane) Xl[tri_row_off(i) + lane] = x[i];
                    }
                }
            }
            tri_wave_sync();
            __builtin_amdgcn_sched_barrier(0);
            lap2(7);
            if (do_v && ok && lane < T) {
                const double* Gt = Gl + lane * rs;
                double gt[RA];
#pragma unroll
                for (int i = 0; i < RA; i += 2) {
                    double2 g2 = {0.0, 0.0};
                    if (i < rs) g2 = *reinterpret_cast<const double2*>(Gt + i);
                    gt[i] = g2.x;
                    gt[i + 1] = g2.y;
                }
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
                v_s[lane * L + l] = vv;
            }
            if (lane == 0) {
                fail_s[l] = ok ? 0 : 1;
                if (!ok) atomicAdd(A.fail, 1);
            }
        }
    };

    // ---- Newton step on the posterior mean -------------------------------------------
    auto mean_phase = [&](bool last) {
        double* vec = vec_s + wid * 128;
        double* vec2 = vec + 64;
        for (int l = wid; l < L; l += nw) {
            const int r = __builtin_amdgcn_readfirstlane(rl_s[l]);
            const int rs = (r + 1) & ~1;
            const double* Gl = G_s + __builtin_amdgcn_readfirstlane(goff_s[l]);
            const double* Xl = Xp + (int64_t)l * PK;
            double* u = u_s + (int64_t)l * Tc;
            if (__builtin_amdgcn_readfirstlane(fail_s[l])) {  // singular system: zero update (core.py:92-94)
                if (lane == 0) atomicAdd(A.fail, 1);
                if (last && lane < T) A.dmu[(r0 + lane) * L + l] = 0.0;
                continue;
            }
            const int j = lane & (RP - 1), ch = lane / RP;
            // g1 = G' (res a_l)
            double acc = 0.0;
            if (j < rs) {
#pragma unroll 4
                for (int t = ch; t < T; t += NCH) acc = fma(Gl[t * rs + j], ra_s[t * L + l], acc);
            }
#pragma unroll
            for (int o = RP; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
            if (lane < RA) vec[lane] = acc;
            tri_wave_sync();
            // u = G g1 - mu_l   (row t of G stays in registers for the last step)
            double gt[RA];
            double ut = 0.0;
            {
                const double* Gt = Gl + (lane < T ? lane : 0) * rs;
                double s0 = 0.0, s1 = 0.0;
#pragma unroll
                for (int i = 0; i < RA; i += 2) {
                    double2 g2 = {0.0, 0.0};
                    if (i < rs) {
                        g2 = *reinterpret_cast<const double2*>(Gt + i);
                        const double2 c2 = *reinterpret_cast<const double2*>(vec + i);
                        s0 = fma(g2.x, c2.x, s0);
                        s1 = fma(g2.y, c2.y, s1);
                    }
                    gt[i] = g2.x;
                    gt[i + 1] = g2.y;
                }
                if (lane < T) {
                    ut = (s0 + s1) - mu_s[lane * L + l];
                    u[lane] = ut;
                }
            }
            tri_wave_sync();
            // rhs = (W G)' u
            acc = 0.0;
            if (j < rs) {
#pragma unroll 4
                for (int t = ch; t < T; t += NCH) acc = fma(w_s[t * L + l] * Gl[t * rs + j], u[t], acc);
            }
#pragma unroll
            for (int o = RP; o < 64; o <<= 1) acc += __shfl_xor(acc, o, 64);
            if (lane < RA) vec2[lane] = acc;
            tri_wave_sync();
            // z = X rhs, sol = X' z   (lane = row, then lane = column)
            double z = 0.0;
            if (lane < RA) {  // four partial sums: the dependent FMA chain is RA / 4 long instead of RA
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
                if (last) A.dmu[(r0 + lane) * L + l] = s;
                mu_s[lane * L + l] += s;
            }
            tri_wave_sync();
        }
    };

    // ---- schedule (identical to estep.hip) ---------------------------------------------
    const int mode = A.mode;
    unsigned long long tick = A.clk ? __builtin_readcyclecounter() : 0;
    auto lap = [&](int slot) {
        if (A.clk && tid == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            atomicAdd(A.clk + slot, now - tick);
            tick = now;
        }
    };
    lap(0);
    // One call site per phase (the unrolled factor code is large): iteration -1
    // is the initial factorisation from the incoming w; update_w is a single
    // curvature pass; update_v is iteration -1 alone.
    const bool with_mean = (mode & EM_MEAN) != 0;
    const int n_it = with_mean ? A.n_iter : ((mode & EM_W) ? 1 : 0);
    if (with_mean) tn_pass(std::integral_constant<int, FP_YA>{});
    for (int it = (mode & EM_FACTOR0) ? -1 : 0; it < n_it; ++it) {
        const bool last = it == n_it - 1;
        bool do_factor, do_v;
        if (it >= 0) {
            if (with_mean) {
                tn_pass(std::integral_constant<int, FP_RES>{});
                lap(2);
                mean_phase(last);
                __syncthreads();
                lap(3);
            }
            tn_pass(std::integral_constant<int, FP_W>{});
            lap(4);
            do_factor = with_mean && (A.vb || !last);
            do_v = A.vb != 0;
        } else {
            do_factor = true;
            do_v = (mode & EM_V) && !with_mean;
        }
        if (do_factor) factor_phase(do_v);
        __syncthreads();
        lap(it >= 0 ? 5 : 1);
    }

    const bool wr_mu = mode & EM_MEAN;
    const bool wr_w = mode & (EM_MEAN | EM_W);
    const bool wr_v = (mode & EM_V) != 0;
    for (int i = tid; i < T * L; i += nthr) {
        if (wr_mu) A.mu[r0 * L + i] = mu_s[i];
        if (wr_w) A.w[r0 * L + i] = w_s[i];
        if (wr_v) A.v[r0 * L + i] = v_s[i];
    }
}

// per-channel records for the passes: (a_l, a_l^2)[LT padded], b, 1/noise (or 1), and the
// constant Gaussian-channel part of w; one tiny launch per E-step call
__global__ void __launch_bounds__(256)
estep_cols_kernel(int N, int L, int LT, const double* a, const double* b, const double* noise, const int* gauss,
                  double* cols, double* wconst) {
    const int REC = 2 * LT + 2;
    for (int n = threadIdx.x; n < N; n += 256) {
        double* rec = cols + (int64_t)n * REC;
        for (int l = 0; l < LT; ++l) {
            const double av = l < L ? a[l * N + n] : 0.0;
            rec[l] = av;
            rec[LT + l] = av * av;
        }
        rec[2 * LT] = b[n];
        rec[2 * LT + 1] = gauss[n] ? 1.0 / noise[n] : 1.0;
    }
    if ((int)threadIdx.x < L) {  // w = U (a')^2 with U = 1/noise on Gaussian channels (core.py:103-104)
        double s = 0.0;
        for (int n = 0; n < N; ++n)
            if (gauss[n]) s = fma(a[threadIdx.x * N + n] * a[threadIdx.x * N + n], 1.0 / noise[n], s);
        wconst[threadIdx.x] = s;
    }
}

// ---------------------------------------------------------------------------
template <int LT, int RP, int RA>
static int launch_fast_t(vlgp_ctx* ctx, const EstepArgs& A, int M, int nthr, size_t lds) {
    auto fn = estep_fast_kernel<LT, RP, RA>;
    if (lds > 64 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    if (getenv("VLGP_DEBUG_OCC")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, nthr, lds);
        fprintf(stderr, "estep_fast<%d,%d,%d>: %d threads, %zu B LDS -> %d blocks per CU\n", LT, RP, RA, nthr, lds, nb);
    }
    constexpr int kind = RA <= 16 ? VLGP_PROF_ESTEP_RA16 : (RA <= 24 ? VLGP_PROF_ESTEP_RA24 : VLGP_PROF_ESTEP_RA32);
    vlgp_prof_begin(ctx, kind);
    hipLaunchKernelGGL(fn, dim3(M), dim3(nthr), lds, ctx->stream, A, A.cols_g);
    vlgp_prof_end(ctx, kind, (double)M * (A.n_iter > 0 ? A.n_iter : 1));
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

template <int RP, int RA>
static int launch_fast_l(vlgp_ctx* ctx, const EstepArgs& A, int M, int nthr, size_t lds) {
    if (A.L <= 3) return launch_fast_t<3, RP, RA>(ctx, A, M, nthr, lds);
    if (A.L <= 5) return launch_fast_t<5, RP, RA>(ctx, A, M, nthr, lds);
    if (A.L <= 8) return launch_fast_t<8, RP, RA>(ctx, A, M, nthr, lds);
    if constexpr (RA == 16) return launch_fast_t<10, RP, RA>(ctx, A, M, nthr, lds);  // ten waves, one workgroup per CU
    return vlgp_fail(ctx, VLGP_ERR_STATE, "no fast E-step instantiation for %d latents at this rank", A.L);
}

int launch_estep_fast(vlgp_ctx* ctx, UnitSet& us, EstepArgs A, int* handled) {
    *handled = 0;
    const int N = ctx->N, L = ctx->L;
    if (getenv("VLGP_ESTEP_GENERIC")) return VLGP_OK;
    if (us.Tmax > 64 || L > 10) return VLGP_OK;
    const bool need_prior = (A.mode & (EM_FACTOR0 | EM_MEAN | EM_V)) != 0;
    int rmax = 0;
    int64_t gsz = 0;
    if (need_prior) {
        for (auto& kv : ctx->priors) {
            const Prior& pr = kv.second;
            if (pr.T < us.Tmin || pr.T > us.Tmax) continue;
            int64_t g = 0;
            for (int l = 0; l < L; ++l) {
                rmax = pr.rl[l] > rmax ? pr.rl[l] : rmax;
                g += (int64_t)pr.T * ((pr.rl[l] + 1) & ~1);
            }
            gsz = g > gsz ? g : gsz;
        }
    }
    if (rmax > 32 || (L > 8 && rmax > 16)) return VLGP_OK;  // nine or ten latents: only the rank <= 16 instantiation
    const int RP = rmax <= 16 ? 16 : 32;
    const int LT = L <= 3 ? 3 : (L <= 5 ? 5 : (L <= 8 ? 8 : 10));
    const int nw = L < 4 ? 4 : L;  // L <= 8
    const int Tc = us.Tmax;
    const int RA = rmax <= 16 ? 16 : (rmax <= 24 && !getenv("VLGP_ESTEP_NO_RA24") ? 24 : 32);
    const int64_t PK = tri_packed_size(RA);
    int64_t scr = (int64_t)nw * Tc * L;                       // partial sums of the passes
    if (scr < (int64_t)nw * 256) scr = (int64_t)nw * 256;     // 16 x 16 MFMA staging tiles
    if (scr < (int64_t)nw * 128 + (int64_t)Tc * L) scr = (int64_t)nw * 128 + (int64_t)Tc * L;  // vec + u
    scr = (scr + 1) & ~1LL;
    int64_t d = 5LL * Tc * L + 1 + scr + L * PK + gsz + (2 * N + 3 * L + 3) / 2 + 2;
    if (d * 8 > 160 * 1024) return VLGP_OK;
    A.lds_gsz = (int)gsz;
    A.lds_T = Tc;
    A.lds_scr = (int)scr;
    if (!ctx->d_ecols) HIPCHK(ctx, hipMalloc(&ctx->d_ecols, sizeof(double) * ((size_t)N * 50 + 32)));
    A.cols_g = ctx->d_ecols;
    A.wconst_g = ctx->d_ecols + (int64_t)N * 34;
    hipLaunchKernelGGL(estep_cols_kernel, dim3(1), dim3(256), 0, ctx->stream, N, L, LT, ctx->d_a, ctx->d_b, ctx->d_noise,
                       ctx->d_gauss, ctx->d_ecols, ctx->d_ecols + (int64_t)N * 34);
    HIPCHK(ctx, hipGetLastError());
    *handled = 1;
    const int nthr = nw * 64;
    if (RA == 16) return launch_fast_l<16, 16>(ctx, A, us.M, nthr, (size_t)d * 8);
    if (RA == 24) return launch_fast_l<32, 24>(ctx, A, us.M, nthr, (size_t)d * 8);
    return launch_fast_l<32, 32>(ctx, A, us.M, nthr, (size_t)d * 8);
}
