// E-step of vLGP on gfx950: one persistent workgroup per unit (trial or segment).
//
// Replaces core.infer_single_trial / update_w / update_v of the reference
// (vlgp/core.py:22-120, 419-471).  Per inner sweep the reference does, for each
// latent l with prior factor G (T x r) and curvature w:
//     H = G' diag(w) G;  u = G G' (res a_l) - mu_l
//     mu_l += clip(u - G (I+H)^-1 G' diag(w) u)                  (core.py:85-97)
//     w = U (a')^2                                               (core.py:100-104)
//     v_l = diag(G (I+H)^-1 G')                                  (core.py:105-113)
// This kernel computes the same quantities with the minimal algorithm:
//   * I+H is factored ONCE per sweep (Cholesky, then the triangular inverse X)
//     and shared by the variance update of sweep i and the mean update of
//     sweep i+1 (both use the same w);
//   * v_t = |X g_t|^2 and (I+H)^-1 rhs = X'(X rhs);
//   * zero columns of G (ichol_gauss stops early, math.py:105) are dropped:
//     every latent works at its effective rank;
//   * sum_n (y - r) a_ln is split into the sweep-invariant sum_n y a_ln and the
//     rate term, so y is read from HBM exactly once per launch.
// Layout: lanes map to channels n inside a row group (coalesced y/xb reads,
// conflict-free LDS reads of a), one wavefront owns one latent in the
// factor/solve phases (wave-synchronous, no workgroup barriers inside).
#include "ctx.h"

#include "estep_args.h"
#include "fast_exp.h"

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ double shfl_xor_f64(double x, int m) { return __shfl_xor(x, m, 64); }

enum { PASS_YA = 0, PASS_RES = 1, PASS_W = 2 };

template <bool SMALL, int LT>
__global__ void __launch_bounds__(SMALL ? 512 : 1024) estep_kernel(EstepArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int lane = tid & 63, wid = tid >> 6, nw = nthr >> 6;
    const int N = A.N, L = A.L;
    const int m = blockIdx.x;
    const int64_t r0 = A.off[m];
    const int T = (int)(A.off[m + 1] - r0);
    // update_w needs no prior: unit_prior may be null, every rank is then 0
    const int pidx = A.unit_prior ? A.unit_prior[m] : -1;
    const double* Gbase = pidx >= 0 ? A.prior_base[pidx] : nullptr;

    // ---- carve LDS ------------------------------------------------------
    double* p = smem;
    double* a_s = p;      p += LT * N;   // rows l >= L are zero: the (T x N) passes need no l < L tests
    double* asq_s = p;    p += LT * N;
    double* bvec = p;     p += N;
    double* cn = p;       p += N;
    double* wconst = p;   p += (L + 1) & ~1;
    double* vec_s = p;    p += nw * 64;
    double* vec2_s = p;   p += nw * 64;
    double *mu_s, *v_s, *w_s, *ra_s, *ya_s, *u_s, *Lc_s;
    const double* G_s;
    double* G_w = nullptr;
    if constexpr (SMALL) {
        const int Tc = A.lds_T;
        mu_s = p; p += Tc * L;
        v_s = p;  p += Tc * L;
        w_s = p;  p += Tc * L;
        ra_s = p; p += Tc * L;
        ya_s = p; p += Tc * L;
        u_s = p;  p += Tc * L;
        G_w = p;  p += A.lds_gsz;
        G_s = G_w;
        Lc_s = p; p += A.lds_lcsz;
    } else {
        mu_s = A.mu + r0 * L;
        v_s = A.v + r0 * L;
        w_s = A.w + r0 * L;
        ra_s = A.scratch + 3 * r0 * L;
        ya_s = ra_s + (int64_t)T * L;
        u_s = ya_s + (int64_t)T * L;
        G_s = Gbase;
        if (A.lc_global) {
            Lc_s = A.lc_global + (int64_t)m * A.lc_stride;
        } else {
            Lc_s = p; p += A.lds_lcsz;
        }
    }
    int* ip = reinterpret_cast<int*>(p);
    int* gauss_s = ip;  ip += N;
    int* rl_s = ip;     ip += L;
    int* goff_s = ip;   ip += L;
    int* lcoff_s = ip;  ip += L;
    int* fail_s = ip;   ip += L;

    // ---- stage parameters and the unit state -----------------------------
    for (int i = tid; i < LT * N; i += nthr) {
        const double av = i < L * N ? A.a[i] : 0.0;
        a_s[i] = av;
        asq_s[i] = av * av;
    }
    for (int n = tid; n < N; n += nthr) {
        const int g = A.gauss[n];
        gauss_s[n] = g;
        bvec[n] = A.b[n];  // row 0 of b: used only when x == 1
        cn[n] = g ? 1.0 / A.noise[n] : 1.0;
    }
    if (tid == 0) {
        int go = 0, lo = 0;
        for (int l = 0; l < L; ++l) {
            const int r = pidx >= 0 ? A.prior_rl[pidx * L + l] : 0;
            rl_s[l] = r;
            lcoff_s[l] = lo;
            lo += r * (r | 1);
            if constexpr (SMALL) {
                goff_s[l] = go;
                go += T * (r | 1);
            } else {
                goff_s[l] = pidx >= 0 ? (int)A.prior_goff[pidx * L + l] : 0;
            }
            fail_s[l] = 0;
        }
    }
    if constexpr (SMALL) {
        for (int i = tid; i < T * L; i += nthr) {
            mu_s[i] = A.mu[r0 * L + i];
            v_s[i] = A.v[r0 * L + i];
            w_s[i] = A.w[r0 * L + i];
        }
    }
    __syncthreads();
    if (tid < L) {  // Gaussian channels contribute a constant to w (core.py:103-104)
        double s = 0.0;
        for (int n = 0; n < N; ++n)
            if (gauss_s[n]) s = fma(asq_s[tid * N + n], cn[n], s);
        wconst[tid] = s;
    }
    if constexpr (SMALL) {
        for (int l = 0; l < L; ++l) {
            const int r = rl_s[l], gs = r | 1;
            const double* src = Gbase + (pidx >= 0 ? A.prior_goff[pidx * L + l] : 0);
            double* dst = G_w + goff_s[l];
            for (int i = tid; i < T * r; i += nthr) {
                const int t = i / r, c = i - t * r;
                dst[t * gs + c] = src[i];
            }
        }
    }
    __syncthreads();

    const int RG = A.rg;
    const int sub = tid & (RG - 1), rgid = tid / RG, nrg = nthr / RG;

    // ---- (T x N) passes: lanes of a row group stride over channels --------
    // Branch-free body: the loadings are zero-padded to LT rows, Poisson / Gaussian is a select, the
    // regressor term comes from one of two uniform sources (a per-latent `l < L` branch or a
    // global-vs-LDS pointer select costs a wait per load; see estep_long.hip).
    const bool has_xb = A.xb != nullptr;
    auto tn_pass = [&](auto kind_c) {
        constexpr int KIND = decltype(kind_c)::value;
        for (int row = rgid; row < T; row += nrg) {
            double mr[LT], vr[LT], acc[LT];
#pragma unroll
            for (int l = 0; l < LT; ++l) {
                const int lc = l < L ? l : 0;
                const double mv = KIND != PASS_YA ? mu_s[row * L + lc] : 0.0;
                const double vv = KIND != PASS_YA ? v_s[row * L + lc] : 0.0;
                mr[l] = l < L ? mv : 0.0;
                vr[l] = l < L ? vv : 0.0;
                acc[l] = 0.0;
            }
            const double* yrow = A.y + (r0 + row) * N;
            const double* xbrow = A.xb + (has_xb ? (r0 + row) * N : 0);
            for (int n = sub; n < N; n += RG) {
                double al[LT], aq[LT];
#pragma unroll
                for (int l = 0; l < LT; ++l) {
                    al[l] = a_s[l * N + n];
                    aq[l] = asq_s[l * N + n];
                }
                const double cnn = cn[n];
                const int g = gauss_s[n];
                if constexpr (KIND == PASS_YA) {
                    const double yc = yrow[n] * cnn;
#pragma unroll
                    for (int l = 0; l < LT; ++l) acc[l] = fma(yc, al[l], acc[l]);
                } else {
                    double eta = has_xb ? xbrow[n] : bvec[n];
                    double lin = 0.0;
#pragma unroll
                    for (int l = 0; l < LT; ++l) {
                        eta = fma(mr[l], al[l], eta);
                        lin = fma(vr[l], aq[l], lin);
                    }
                    const double pois = exp(clamp10(fma(0.5, lin, eta)));
                    if constexpr (KIND == PASS_RES) {
                        const double mval = g ? eta * cnn : pois;
#pragma unroll
                        for (int l = 0; l < LT; ++l) acc[l] = fma(mval, al[l], acc[l]);
                    } else {
                        const double rate = g ? 0.0 : pois;
#pragma unroll
                        for (int l = 0; l < LT; ++l) acc[l] = fma(rate, aq[l], acc[l]);
                    }
                }
            }
            for (int o = RG >> 1; o > 0; o >>= 1) {
#pragma unroll
                for (int l = 0; l < LT; ++l) acc[l] += shfl_xor_f64(acc[l], o);
            }
            if (sub == 0) {
#pragma unroll
                for (int l = 0; l < LT; ++l)
                    if (l < L) {
                        if constexpr (KIND == PASS_YA) ya_s[row * L + l] = acc[l];
                        else if constexpr (KIND == PASS_RES) ra_s[row * L + l] = ya_s[row * L + l] - acc[l];
                        else w_s[row * L + l] = acc[l] + wconst[l];
                    }
            }
        }
    };

    // ---- factor I + G'WG, invert the factor, optionally refresh v ----------
    auto factor_phase = [&](bool do_v) {
        for (int l = wid; l < L; l += nw) {
            const int r = rl_s[l];
            const int gs = SMALL ? (r | 1) : r, ls = r | 1;
            const double* Gl = G_s + goff_s[l];
            double* Lc = Lc_s + lcoff_s[l];
            const int ne = r * (r + 1) / 2;
            for (int e = lane; e < ne; e += 64) {
                int i = (int)((sqrt(8.0 * e + 1.0) - 1.0) * 0.5);
                while (i * (i + 1) / 2 > e) --i;
                while ((i + 1) * (i + 2) / 2 <= e) ++i;
                const int j = e - i * (i + 1) / 2;
                double s = (i == j) ? 1.0 : 0.0;
                for (int t = 0; t < T; ++t)
                    s = fma(w_s[t * L + l] * Gl[t * gs + i], Gl[t * gs + j], s);
                Lc[i * ls + j] = s;
            }
            wave_sync();
            bool ok = true;
            for (int k = 0; k < r; ++k) {  // left-looking Cholesky, lane = row
                double s = 0.0;
                const bool act = lane >= k && lane < r;
                if (act) {
                    s = Lc[lane * ls + k];
                    for (int i = 0; i < k; ++i) s = fma(-Lc[lane * ls + i], Lc[k * ls + i], s);
                }
                const double d = __shfl(s, k, 64);
                if (!(d > 0.0) || !(d < 1e300)) { ok = false; break; }
                const double sd = sqrt(d);
                if (act) Lc[lane * ls + k] = (lane == k) ? sd : s / sd;
                wave_sync();
            }
            if (ok) {
                for (int i = 0; i < r; ++i) {  // X = Lc^-1 in place, lane = column
                    const double lii = Lc[i * ls + i];
                    double acc = 0.0;
                    if (lane < i)
                        for (int j = lane; j < i; ++j) acc = fma(Lc[i * ls + j], Lc[j * ls + lane], acc);
                    wave_sync();
                    if (lane < i) Lc[i * ls + lane] = -acc / lii;
                    else if (lane == i) Lc[i * ls + i] = 1.0 / lii;
                    wave_sync();
                }
                if (do_v) {
                    for (int t = lane; t < T; t += 64) {
                        double acc = 0.0;
                        for (int i = 0; i < r; ++i) {
                            double z = 0.0;
                            for (int j = 0; j <= i; ++j) z = fma(Lc[i * ls + j], Gl[t * gs + j], z);
                            acc = fma(z, z, acc);
                        }
                        v_s[t * L + l] = acc;
                    }
                }
            }
            if (lane == 0) {
                fail_s[l] = ok ? 0 : 1;
                if (!ok) atomicAdd(A.fail, 1);
            }
        }
    };

    // ---- Newton step on the posterior mean --------------------------------
    auto mean_phase = [&](bool last) {
        double* vec = vec_s + wid * 64;
        double* vec2 = vec2_s + wid * 64;
        for (int l = wid; l < L; l += nw) {
            const int r = rl_s[l];
            const int gs = SMALL ? (r | 1) : r, ls = r | 1;
            const double* Gl = G_s + goff_s[l];
            const double* X = Lc_s + lcoff_s[l];
            double* u = u_s + (int64_t)l * T;
            if (fail_s[l]) {  // singular system: zero update (core.py:92-94)
                if (lane == 0) atomicAdd(A.fail, 1);
                if (last)
                    for (int t = lane; t < T; t += 64) A.dmu[(r0 + t) * L + l] = 0.0;
                continue;
            }
            double g1 = 0.0;
            if (lane < r)
                for (int t = 0; t < T; ++t) g1 = fma(Gl[t * gs + lane], ra_s[t * L + l], g1);
            vec[lane] = g1;
            wave_sync();
            for (int t = lane; t < T; t += 64) {
                double s = 0.0;
                for (int i = 0; i < r; ++i) s = fma(Gl[t * gs + i], vec[i], s);
                u[t] = s - mu_s[t * L + l];
            }
            wave_sync();
            double rhs = 0.0;
            if (lane < r)
                for (int t = 0; t < T; ++t) rhs = fma(w_s[t * L + l] * Gl[t * gs + lane], u[t], rhs);
            vec[lane] = rhs;
            wave_sync();
            double z = 0.0;
            if (lane < r)
                for (int j = 0; j <= lane; ++j) z = fma(X[lane * ls + j], vec[j], z);
            vec2[lane] = z;
            wave_sync();
            double sol = 0.0;
            if (lane < r)
                for (int j = lane; j < r; ++j) sol = fma(X[j * ls + lane], vec2[j], sol);
            vec[lane] = sol;
            wave_sync();
            for (int t = lane; t < T; t += 64) {
                double s = u[t];
                for (int i = 0; i < r; ++i) s = fma(-Gl[t * gs + i], vec[i], s);
                s = fmin(fmax(s, -A.dmu_bound), A.dmu_bound);
                if (last) A.dmu[(r0 + t) * L + l] = s;
                mu_s[t * L + l] += s;
            }
            wave_sync();
        }
    };

    // ---- schedule ----------------------------------------------------------
    const int mode = A.mode;
    unsigned long long tick = A.clk ? __builtin_readcyclecounter() : 0;
    auto lap = [&](int slot) {
        if (A.clk && tid == 0) {
            const unsigned long long now = __builtin_readcyclecounter();
            atomicAdd(A.clk + slot, now - tick);
            tick = now;
        }
    };
    lap(0);  // staging
    if (mode & EM_MEAN) tn_pass(std::integral_constant<int, PASS_YA>{});
    if (mode & EM_FACTOR0) factor_phase((mode & EM_V) && !(mode & EM_MEAN));
    __syncthreads();
    lap(1);  // y.a pass + first factor
    if (mode & EM_MEAN) {
        for (int it = 0; it < A.n_iter; ++it) {
            const bool last = it == A.n_iter - 1;
            tn_pass(std::integral_constant<int, PASS_RES>{});
            __syncthreads();
            lap(2);
            mean_phase(last);
            __syncthreads();
            lap(3);
            tn_pass(std::integral_constant<int, PASS_W>{});
            __syncthreads();
            lap(4);
            if (A.vb || !last) factor_phase(A.vb != 0);
            __syncthreads();
            lap(5);
        }
    } else if (mode & EM_W) {
        tn_pass(std::integral_constant<int, PASS_W>{});
        __syncthreads();
    }

    if constexpr (SMALL) {
        const bool wr_mu = mode & EM_MEAN;
        const bool wr_w = mode & (EM_MEAN | EM_W);
        const bool wr_v = (mode & EM_V) != 0;
        for (int i = tid; i < T * L; i += nthr) {
            if (wr_mu) A.mu[r0 * L + i] = mu_s[i];
            if (wr_w) A.w[r0 * L + i] = w_s[i];
            if (wr_v) A.v[r0 * L + i] = v_s[i];
        }
    }
}

// ---------------------------------------------------------------------------
template <bool SMALL, int LT>
static int launch_t(vlgp_ctx* ctx, const EstepArgs& A, int M, int nthr, size_t lds) {
    auto fn = estep_kernel<SMALL, LT>;
    if (lds > 64 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fn, dim3(M), dim3(nthr), lds, ctx->stream, A);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

template <bool SMALL>
static int launch_l(vlgp_ctx* ctx, const EstepArgs& A, int M, int nthr, size_t lds) {
    const int L = A.L;
    if (L <= 2) return launch_t<SMALL, 2>(ctx, A, M, nthr, lds);
    if (L <= 3) return launch_t<SMALL, 3>(ctx, A, M, nthr, lds);
    if (L <= 5) return launch_t<SMALL, 5>(ctx, A, M, nthr, lds);
    if (L <= 8) return launch_t<SMALL, 8>(ctx, A, M, nthr, lds);
    if (L <= 10) return launch_t<SMALL, 10>(ctx, A, M, nthr, lds);
    if (L <= 16) return launch_t<SMALL, 16>(ctx, A, M, nthr, lds);
    // beyond sixteen latents the per-latent register arrays of the (T x N) passes spill: slow, same arithmetic
    if (L <= 32) return launch_t<SMALL, 32>(ctx, A, M, nthr, lds);
    if (L <= 64) return launch_t<SMALL, 64>(ctx, A, M, nthr, lds);
    return vlgp_fail(ctx, VLGP_ERR_ARG, "E-step kernel supports at most 64 latents, got %d", L);
}

static int pick_rg(int T, int N, int nthr) {
    int best = 64;
    double best_u = -1.0;
    for (int rg = 8; rg <= 64; rg <<= 1) {
        const int rows = nthr / rg;
        const double ut = (double)T / (double)(((T + rows - 1) / rows) * rows);
        const double un = (double)N / (double)(((N + rg - 1) / rg) * rg);
        if (ut * un > best_u + 1e-9) {
            best_u = ut * un;
            best = rg;
        }
    }
    return best;
}

int launch_estep(vlgp_ctx* ctx, UnitSet& us, int mode, int n_iter, double dmu_bound, int vb) {
    const int N = ctx->N, L = ctx->L;
    const bool need_prior = (mode & (EM_FACTOR0 | EM_MEAN | EM_V)) != 0;
    if (need_prior) CHK(vlgp_bind_priors(ctx, us));
    if (!us.x_ones) CHK(vlgp_refresh_xb(ctx, us));

    // LDS demand over the priors this set uses
    int64_t gsz = 0, lcsz = 0;
    for (auto& kv : ctx->priors) {
        const Prior& pr = kv.second;
        if (!need_prior || pr.T < us.Tmin || pr.T > us.Tmax) continue;
        int64_t g = 0, lc = 0;
        for (int l = 0; l < L; ++l) {
            g += (int64_t)pr.T * (pr.rl[l] | 1);
            lc += (int64_t)pr.rl[l] * (pr.rl[l] | 1);
        }
        gsz = g > gsz ? g : gsz;
        lcsz = lc > lcsz ? lc : lcsz;
    }
    gsz = (gsz + 1) & ~1LL;
    lcsz = (lcsz + 1) & ~1LL;

    const int64_t LDS_MAX = 160 * 1024;
    const int LTl = L <= 2 ? 2 : (L <= 3 ? 3 : (L <= 5 ? 5 : (L <= 8 ? 8 : (L <= 10 ? 10 : (L <= 16 ? 16 : (L <= 32 ? 32 : 64))))));  // as launch_l dispatches
    const int64_t common = 2LL * LTl * N + 2LL * N + ((L + 1) & ~1);
    const int64_t ints = ((int64_t)N + 4 * L + 1) / 2 + 1;

    EstepArgs A;
    A.N = N; A.L = L;
    A.mode = mode; A.n_iter = n_iter; A.vb = vb; A.dmu_bound = dmu_bound;
    A.off = us.d_off; A.unit_prior = need_prior ? us.d_unit_prior : nullptr;
    A.prior_base = ctx->d_prior_base; A.prior_rl = ctx->d_prior_rl; A.prior_goff = ctx->d_prior_goff;
    A.y = us.y; A.xb = us.x_ones ? nullptr : us.d_xb;
    A.mu = us.mu; A.v = us.v; A.w = us.w; A.dmu = us.dmu;
    A.a = ctx->d_a; A.b = ctx->d_b; A.noise = ctx->d_noise; A.gauss = ctx->d_gauss;
    A.scratch = nullptr; A.lc_global = nullptr; A.lc_stride = 0;
    A.fail = ctx->d_fail;
    A.clk = ctx->d_clk;
    A.cols_g = nullptr; A.wconst_g = nullptr;
    A.lds_T = us.Tmax; A.lds_gsz = (int)gsz; A.lds_lcsz = (int)lcsz;

    // SPLIT: many short units -> chip-wide launches per phase (estep_split.hip); declines for small sets
    {
        int handled = 0;
        CHK(launch_estep_split(ctx, us, A, &handled));
        if (handled) { ctx->last_estep_path = handled == 2 ? VLGP_PATH_ESTEP_LSPLIT : (handled == 3 ? VLGP_PATH_ESTEP_SPLIT_MIXED : VLGP_PATH_ESTEP_SPLIT); return VLGP_OK; }
    }

    // FAST: register-resident factorisations (estep_fast.hip); declines when it does not apply
    {
        int handled = 0;
        CHK(launch_estep_fast(ctx, us, A, &handled));
        if (handled) { ctx->last_estep_path = VLGP_PATH_ESTEP_FAST; return VLGP_OK; }
    }

    // LONG units: all waves on the per-latent phases, MFMA builds (estep_long.hip)
    if (us.Tmax > 64) {
        int handled = 0;
        CHK(launch_estep_long(ctx, us, A, &handled));
        if (handled) { ctx->last_estep_path = VLGP_PATH_ESTEP_LONG; return VLGP_OK; }
    }

    // SMALL: whole unit state lives in LDS
    int nw_s = L < 4 ? 4 : (L > 8 ? 8 : L);
    const int64_t small_d = common + 2LL * nw_s * 64 + 6LL * us.Tmax * L + gsz + lcsz + ints;
    if (us.Tmax <= 256 && small_d * 8 <= LDS_MAX) {
        const int nthr = nw_s * 64;
        A.rg = pick_rg(us.Tmax, N, nthr);
        ctx->last_estep_path = VLGP_PATH_ESTEP_GENERIC;
        vlgp_prof_begin(ctx, VLGP_PROF_ESTEP_GENERIC);
        int rc = launch_l<true>(ctx, A, us.M, nthr, (size_t)small_d * 8);
        vlgp_prof_end(ctx, VLGP_PROF_ESTEP_GENERIC, (double)us.M * (A.n_iter > 0 ? A.n_iter : 1));
        return rc;
    }

    // too big for LDS residency (many latents at high rank): the long-unit kernel streams G from L2
    {
        int handled = 0;
        CHK(launch_estep_long(ctx, us, A, &handled));
        if (handled) { ctx->last_estep_path = VLGP_PATH_ESTEP_LONG; return VLGP_OK; }
    }

    // LONG: unit state in HBM/L2, factors in LDS when they fit
    const int nthr = 1024, nw = nthr / 64;
    int64_t long_d = common + 2LL * nw * 64 + ints;
    const int64_t need = 3 * us.rows * L;
    int64_t lc_total = 0;
    if ((long_d + lcsz) * 8 <= LDS_MAX) {
        long_d += lcsz;
    } else {
        lc_total = lcsz * us.M;
        A.lc_stride = lcsz;
    }
    if (us.scratch_len < need + lc_total) {
        if (us.d_scratch) HIPCHK(ctx, hipFree(us.d_scratch));
        us.d_scratch = nullptr;
        HIPCHK(ctx, hipMalloc(&us.d_scratch, (size_t)(need + lc_total) * 8));
        us.scratch_len = need + lc_total;
    }
    A.scratch = us.d_scratch;
    if (lc_total) A.lc_global = us.d_scratch + need;
    A.rg = 64;
    ctx->last_estep_path = VLGP_PATH_ESTEP_GENERIC;
    vlgp_prof_begin(ctx, VLGP_PROF_ESTEP_GENERIC);
    int rc = launch_l<false>(ctx, A, us.M, nthr, (size_t)long_d * 8);
    vlgp_prof_end(ctx, VLGP_PROF_ESTEP_GENERIC, (double)us.M * (A.n_iter > 0 ? A.n_iter : 1));
    return rc;
}
