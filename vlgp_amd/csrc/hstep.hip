// H-step objective of vLGP on gfx950: the function scipy's L-BFGS-B minimises
// in gp.optimze1d (vlgp/gp.py:100-123) = construct_posterior_cov
// (gp.py:126-147) + elbo (gp.py:12-43) with gradient mask [0, 1, 0].
//
// For one latent and log-parameters (sigma^2, omega, eps), with K the T x T
// squared-exponential kernel over one window, per segment i
//     S_i = (K^-1 + diag w_i)^-1
//     ll_i  = -1/2 mu_i' K^-1 mu_i - 1/2 tr(K^-1 S_i) - sum log diag chol(K)
//     dll_i = 1/2 tr[(alpha alpha' - K^-1 + K^-1 S_i K^-1) dK/dln(omega)]
// Shared per evaluation (prep kernel, one workgroup): K^-1, Q = K^-1 dK K^-1,
// tr(K^-1 dK), log det.  Per segment (one wavefront each, factor in LDS):
// Cholesky of K^-1 + W, its triangular inverse X, and the two Frobenius
// products <X'X, K^-1>, <X'X, Q> accumulated without materialising S.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "ctx.h"
#include "wave_tri.h"
#include "hstep_mfma.h"
#include "hstep_lr.h"

#define HS_MAXT 1024  // generic kernels: rows are lane-strided; one T x T matrix per wave, in LDS while it fits (T <= 128),
                      // else in global memory (round 4: any window; slow, but the reference's own cost is T^3 per segment too)

struct HPrepArgs {
    int T;
    double dt;
    const double* logp;  // (n_eval, 3)
    double* kinv;        // (n_eval, T, T)
    double* q;           // (n_eval, T, T)
    double* dk;          // (n_eval, T, T)
    double* scal;        // (n_eval, 4): logdet, tr(Kinv dK), omega_used, ok
    double* tm;          // (n_eval, T, T) scratch for K^-1 dK when the four matrices do not fit LDS, else null
    double* km;          // (n_eval, T, T | 1) the factor's matrix in global memory when even it does not fit LDS, else null
};

__device__ __forceinline__ void hs_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
}

// wave-level in-place Cholesky (lower) + triangular inverse of an n x n matrix in
// LDS with row stride ls; returns false on a non-positive pivot.  On success the
// lower triangle holds X = chol^-1 and *logdet = sum log diag(chol).
__device__ bool wave_chol_inv(double* Amat, int n, int ls, int lane, double* logdet) {
    double ld = 0.0;
    for (int k = 0; k < n; ++k) {  // left-looking; lane owns rows lane, lane + 64, ...
        for (int r = k + ((lane - k) & 63); r < n; r += 64) {
            double s = Amat[r * ls + k];
            for (int i = 0; i < k; ++i) s = fma(-Amat[r * ls + i], Amat[k * ls + i], s);
            Amat[r * ls + k] = s;
        }
        hs_wave_sync();
        const double d = Amat[k * ls + k];
        if (!(d > 0.0) || !(d < 1e300)) return false;
        const double sd = sqrt(d);
        ld += log(sd);
        hs_wave_sync();
        for (int r = k + ((lane - k) & 63); r < n; r += 64) Amat[r * ls + k] = (r == k) ? sd : Amat[r * ls + k] / sd;
        hs_wave_sync();
    }
    for (int i = 0; i < n; ++i) {  // X = L^-1 in place, row by row; lane owns columns lane, lane + 64, ...
        const double lii = Amat[i * ls + i];
        double acc[HS_MAXT / 64];
#pragma unroll
        for (int m = 0; m < HS_MAXT / 64; ++m) {
            const int c = lane + 64 * m;
            double a = 0.0;
            if (c < i)
                for (int j = c; j < i; ++j) a = fma(Amat[i * ls + j], Amat[j * ls + c], a);
            acc[m] = a;
        }
        hs_wave_sync();
#pragma unroll
        for (int m = 0; m < HS_MAXT / 64; ++m) {
            const int c = lane + 64 * m;
            if (c < i) Amat[i * ls + c] = -acc[m] / lii;
            else if (c == i) Amat[i * ls + i] = 1.0 / lii;
        }
        hs_wave_sync();
    }
    *logdet = ld;
    return true;
}

__global__ void __launch_bounds__(256) hstep_prep_kernel(HPrepArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double red[256];
    __shared__ int s_ok;
    __shared__ double s_logdet;
    const int T = A.T, ls = T | 1;
    const int e = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int64_t base = (int64_t)e * T * T;
    double* Km = A.km ? A.km + (int64_t)e * T * ls : smem;  // T x ls : K, then chol, then X
    // small windows keep all four matrices in LDS; large ones work on the output buffers in L2
    double* Ki = A.tm ? A.kinv + base : Km + T * ls;     // T x T  : K^-1
    double* Dk = A.tm ? A.dk + base : Ki + T * T;        // T x T  : dK/dln omega
    double* Tm = A.tm ? A.tm + base : Dk + T * T;        // T x T  : K^-1 dK
    const double sigmasq = exp(A.logp[3 * e + 0]);
    double omega = exp(A.logp[3 * e + 1]);
    const double eps = exp(A.logp[3 * e + 2]);

    for (int attempt = 0; attempt < 64; ++attempt) {
        for (int i = tid; i < T * T; i += 256) {
            const int r = i / T, c = i - r * T;
            const double d = (r - c) * A.dt;
            const double d2 = d * d;
            const double kv = sigmasq * exp(-omega * d2);
            Km[r * ls + c] = kv + (r == c ? eps : 0.0);
            Dk[i] = -kv * d2 * omega;
        }
        __syncthreads();
        if (wid == 0) {
            double ld;
            const bool ok = wave_chol_inv(Km, T, ls, lane, &ld);
            if (lane == 0) { s_ok = ok; s_logdet = ld; }
        }
        __syncthreads();
        if (s_ok) break;
        omega += 2.302585092994046;  // gp.py:135 adds log(10) to omega itself
        __syncthreads();
    }
    // K^-1 = X'X
    for (int i = tid; i < T * T; i += 256) {
        const int r = i / T, c = i - r * T;
        const int k0 = r > c ? r : c;
        double s = 0.0;
        for (int k = k0; k < T; ++k) s = fma(Km[k * ls + r], Km[k * ls + c], s);
        Ki[i] = s;
    }
    __syncthreads();
    double tr = 0.0;
    for (int i = tid; i < T * T; i += 256) {
        const int r = i / T, c = i - r * T;
        double s = 0.0;
        for (int k = 0; k < T; ++k) s = fma(Ki[r * T + k], Dk[k * T + c], s);
        Tm[i] = s;
        tr += Ki[i] * Dk[i];
    }
    red[tid] = tr;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    __syncthreads();
    for (int i = tid; i < T * T; i += 256) {
        const int r = i / T, c = i - r * T;
        double s = 0.0;
        for (int k = 0; k < T; ++k) s = fma(Tm[r * T + k], Ki[k * T + c], s);
        A.q[base + i] = s;
        if (!A.tm) {
            A.kinv[base + i] = Ki[i];
            A.dk[base + i] = Dk[i];
        }
    }
    if (tid == 0) {
        A.scal[4 * e + 0] = s_logdet;
        A.scal[4 * e + 1] = red[0];
        A.scal[4 * e + 2] = omega;
        A.scal[4 * e + 3] = s_ok ? 1.0 : 0.0;
    }
}

struct HSegArgs {
    int T, L, M;
    const int64_t* off;
    const double* mu;
    const double* w;
    const int* latent;   // (n_eval)
    const double* kinv;
    const double* q;
    const double* dk;
    const double* scal;
    double* out;         // (n_eval, M, 2)
    double* bmat;        // (n_eval, M, T, T | 1) one matrix per (evaluation, segment) in global memory, or null: LDS
};

// one wavefront per (segment, evaluation)
__global__ void __launch_bounds__(256) hstep_seg_kernel(HSegArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int T = A.T, ls = T | 1;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int seg = blockIdx.x * nw + wid;
    const int e = blockIdx.y;
    if (seg >= A.M) return;
    double* B = A.bmat ? A.bmat + ((int64_t)e * A.M + seg) * T * ls : smem + (int64_t)wid * (T * ls + 2 * T);
    double* muv = A.bmat ? smem + (int64_t)wid * 2 * T : B + T * ls;
    double* alv = muv + T;
    const int l = A.latent[e];
    const int64_t r0 = A.off[seg];
    const double* Ki = A.kinv + (int64_t)e * T * T;
    const double* Q = A.q + (int64_t)e * T * T;
    const double* Dk = A.dk + (int64_t)e * T * T;

    for (int t = lane; t < T; t += 64) muv[t] = A.mu[(r0 + t) * A.L + l];
    for (int i = lane; i < T * T; i += 64) {
        const int r = i / T, c = i - r * T;
        B[r * ls + c] = Ki[i];
    }
    hs_wave_sync();
    for (int t = lane; t < T; t += 64) B[t * ls + t] += A.w[(r0 + t) * A.L + l];
    // alpha = K^-1 mu (K^-1 symmetric: read columns for coalescing)
    for (int t = lane; t < T; t += 64) {
        double al = 0.0;
        for (int j = 0; j < T; ++j) al = fma(Ki[j * T + t], muv[j], al);
        alv[t] = al;
    }
    hs_wave_sync();
    double quad = 0.0, gq = 0.0;
    for (int t = lane; t < T; t += 64) {
        double s = 0.0;
        for (int j = 0; j < T; ++j) s = fma(Dk[j * T + t], alv[j], s);
        quad = fma(muv[t], alv[t], quad);
        gq = fma(s, alv[t], gq);
    }
    double ld;
    const bool ok = wave_chol_inv(B, T, ls, lane, &ld);
    double sk = 0.0, sq = 0.0;
    if (ok) {
        const int ne = T * (T + 1) / 2;
        for (int idx = lane; idx < ne; idx += 64) {
            int i = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
            while (i * (i + 1) / 2 > idx) --i;
            while ((i + 1) * (i + 2) / 2 <= idx) ++i;
            const int j = idx - i * (i + 1) / 2;
            double s = 0.0;
            for (int k = i; k < T; ++k) s = fma(B[k * ls + i], B[k * ls + j], s);
            const double f = (i == j) ? 1.0 : 2.0;
            sk = fma(f * s, Ki[i * T + j], sk);
            sq = fma(f * s, Q[i * T + j], sq);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        quad += __shfl_xor(quad, o, 64);
        gq += __shfl_xor(gq, o, 64);
        sk += __shfl_xor(sk, o, 64);
        sq += __shfl_xor(sq, o, 64);
    }
    if (lane == 0) {
        const double logdet = A.scal[4 * e + 0], trkd = A.scal[4 * e + 1];
        double ll = -0.5 * quad - 0.5 * sk - logdet;
        double dll = 0.5 * (gq - trkd + sq);
        if (!ok) { ll = nan(""); dll = nan(""); }
        A.out[((int64_t)e * A.M + seg) * 2 + 0] = ll;
        A.out[((int64_t)e * A.M + seg) * 2 + 1] = dll;
    }
}

// ---------------------------------------------------------------------------------------------------------
// Windows 65 .. 128 (gp.py:77-80 with a window above the compiled 64): one WORKGROUP (four waves) per (segment,
// evaluation), everything cubic on the matrix pipe.  A = I + S K S (S = diag sqrt(w)) is split at row 64,
//     A = [A11 A21'; A21 A22],   P11 = A11^-1,  B = A21 P11,  Q = (A22 - B A21')^-1,  C = Q B,
//     A^-1 = [P11 + B'C, -C'; -C, Q],
// with the two 64 x 64 inverses from the blocked augmented elimination of hstep_mfma.h (wave 0; the second one on the
// explicit Schur complement, identity-padded to 64) and the four 64 x 64 x 64 products as v_mfma_f64_16x16x4 block rows
// over the four waves, operands of the Toeplitz blocks generated from the first columns of K and dK in LDS.  What the
// round needs are tr(A^-1) and cs = sum_jk s_j s_k dK_jk (A^-1)_jk (see "Fast path" below):
//     tr = tr P11 + <B, C> + tr Q - (padding),   cs = <W11, P11> + <B W11, C> - 2 <W21, C> + <W22, Q>,  W = ss' o dK.
// The quadratic terms alpha = K^-1 mu, alpha' dK alpha use K^-1 of hstep_prep_kernel and run on the idle waves while
// wave 0 factors.  Replaces hstep_seg_kernel (one wave per segment, the whole T x T matrix walked entry by entry in LDS:
// 12.9 ms per launch at window 100 on the C3 data).
struct HBigArgs {
    HSegArgs S;
    const double* logp;  // (n_eval, 3) on the device
    double dt;
};

// D (64 x 64) = A (64 x 64) B (64 x 64): wave `wid` owns block row wid; fa(i, k), fb(k, j) produce the operands,
// fo(i, j, value) consumes the result
template <class FA, class FB, class FO>
__device__ __forceinline__ void big_gemm64(int lane, int wid, FA fa, FB fb, FO fo) {
    const int c = lane & 15, g = lane >> 4;
    hm_d4 acc[4];
#pragma unroll
    for (int bj = 0; bj < 4; ++bj) acc[bj] = hm_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
        const int k = 4 * kk + g;
        const double opa = fa(16 * wid + c, k);
        double opb[4];
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) opb[bj] = fb(k, 16 * bj + c);
#pragma unroll
        for (int bj = 0; bj < 4; ++bj) acc[bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(opa, opb[bj], acc[bj], 0, 0, 0);
    }
#pragma unroll
    for (int bj = 0; bj < 4; ++bj)
#pragma unroll
        for (int p = 0; p < 4; ++p) fo(16 * wid + g + 4 * p, 16 * bj + c, acc[bj][p]);
}

// The K part of a round for windows 65 .. 128: K^-1 (full, to global memory, for alpha = K^-1 mu) and log det chol(K)
// of one evaluation per workgroup through the same split -- K11^-1, B = K21 K11^-1, Q = (K22 - B K21')^-1, C = Q B,
// K^-1 = [K11^-1 + B'C, -C'; -C, Q] -- with the reference's retry (omega += log 10 while K does not factor, gp.py:135).
// Replaces hstep_prep_kernel there (entry-by-entry loops over T x T matrices: 1.5 ms at window 100, 3.5 ms at 128 per
// launch); the segment kernel below needs neither K^-1 dK K^-1 nor tr(K^-1 dK).
__global__ void __launch_bounds__(256, 1) hstep_prep_big(HPrepArgs A) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    using G = HmGeom<64>;
    constexpr int LD = 66;
    double* P11 = smem;
    double* Bm = P11 + 64 * LD;
    double* Sm = Bm + 64 * LD;
    double* Cm = Sm + 64 * LD;
    double* buf = Cm + 64 * LD;
    double* sv = buf + G::TASK;
    double* kv = sv + 128;
    __shared__ int s_bad;
    __shared__ double s_ld[2];
    const int T = A.T, T2 = T - 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = blockIdx.x;
    const double sigmasq = exp(A.logp[3 * e + 0]), eps = exp(A.logp[3 * e + 2]);
    double omega = exp(A.logp[3 * e + 1]);
    bool ok = false;
    for (int attempt = 0; attempt < 64; ++attempt) {
        if (tid == 0) s_bad = 0;
        if (tid < 128) {
            const double d = tid * A.dt;
            sv[tid] = tid < T ? 1.0 : 0.0;
            kv[tid] = sigmasq * exp(-omega * d * d) + (tid == 0 ? eps : 0.0);
        }
        __syncthreads();
        if (wid == 0) {
            buf[G::O_SV + lane] = 1.0;
            buf[G::O_KVM + 63 + lane] = kv[lane];
            buf[G::O_KVM + 63 - lane] = kv[lane];
            buf[G::O_DKV + lane] = 0.0;
            if (lane < 32) buf[G::O_Z + lane] = 0.0;
            tri_wave_sync();
            double ld, unused;
            hstep_task_mfma<64, true>(buf, eps, lane, ld, unused, 64, P11, LD);  // tr_k = 64: K11 itself, no unit diagonal
            if (lane == 0) {
                s_ld[0] = ld;
                if (!(ld == ld && fabs(ld) < 1e300)) s_bad = 1;
            }
        }
        __syncthreads();
        big_gemm64(lane, wid,
                   [&](int i, int k) { return sv[64 + i] * kv[64 + i - k]; },
                   [&](int k, int j) { return P11[k * LD + j]; },
                   [&](int i, int j, double v) { Bm[i * LD + j] = v; });
        __syncthreads();
        big_gemm64(lane, wid,
                   [&](int i, int k) { return Bm[i * LD + k]; },
                   [&](int k, int j) { return sv[64 + j] * kv[64 + j - k]; },
                   [&](int i, int j, double v) {
                       const int dd = i > j ? i - j : j - i;
                       const double pad = (i == j && i >= T2) ? 1.0 : 0.0;  // identity below the rows present
                       Sm[i * LD + j] = fma(sv[64 + i] * sv[64 + j], kv[dd], pad) - v;
                   });
        __syncthreads();
        if (wid == 0) {
            double ld, unused;
            hstep_task_mfma<64, true, true>(buf, eps, lane, ld, unused, 64, Sm, LD, Sm, LD);
            if (lane == 0) {
                s_ld[1] = ld;
                if (!(ld == ld && fabs(ld) < 1e300)) s_bad = 1;
            }
        }
        __syncthreads();
        ok = s_bad == 0;
        if (ok) break;
        omega += 2.302585092994046;  // gp.py:135 adds log(10) to omega itself
        __syncthreads();
    }
    big_gemm64(lane, wid,
               [&](int i, int k) { return Sm[i * LD + k]; },
               [&](int k, int j) { return Bm[k * LD + j]; },
               [&](int i, int j, double v) { Cm[i * LD + j] = v; });
    __syncthreads();
    double* Ki = A.kinv + (int64_t)e * T * T;
    big_gemm64(lane, wid,
               [&](int j, int i) { return Bm[i * LD + j]; },
               [&](int i, int k) { return Cm[i * LD + k]; },
               [&](int j, int k, double v) { Ki[j * T + k] = P11[j * LD + k] + v; });
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int i = idx >> 6, k = idx & 63;
        if (i < T2) {
            const double cv = -Cm[i * LD + k];
            Ki[(64 + i) * T + k] = cv;
            Ki[k * T + 64 + i] = cv;
            if (k < T2) Ki[(64 + i) * T + 64 + k] = Sm[i * LD + k];
        }
    }
    if (tid == 0) {
        A.scal[4 * e + 0] = s_ld[0] + s_ld[1];
        A.scal[4 * e + 1] = 0.0;
        A.scal[4 * e + 2] = omega;
        A.scal[4 * e + 3] = ok ? 1.0 : 0.0;
    }
}

__global__ void __launch_bounds__(256, 1) hstep_seg_big(HBigArgs H) {
    extern __shared__ __attribute__((aligned(16))) double smem[];
    using G = HmGeom<64>;
    constexpr int LD = 66;
    const HSegArgs& A = H.S;
    double* P11 = smem;              // 64 x LD: A11^-1, later C = Q B
    double* Bm = P11 + 64 * LD;      // 64 x LD: B = A21 P11 (rows >= T - 64 are zero)
    double* Sm = Bm + 64 * LD;       // 64 x LD: Schur complement, then Q
    double* buf = Sm + 64 * LD;      // task buffer of the factor routine
    double* sv = buf + G::TASK;      // sqrt(w), zero beyond T
    double* kv = sv + 128;           // first column of K (jitter at distance 0)
    double* dkv = kv + 128;          // first column of dK / dln omega
    double* muv = dkv + 128;
    double* alv = muv + 128;
    __shared__ double red[4][8];
    __shared__ int s_bad;
    const int T = A.T, T2 = T - 64;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int seg = blockIdx.x, e = blockIdx.y;
    const int l = A.latent[e];
    const int64_t r0 = A.off[seg];
    const double sigmasq = exp(H.logp[3 * e + 0]), omega = A.scal[4 * e + 2], eps = exp(H.logp[3 * e + 2]);
    const double* Ki = A.kinv + (int64_t)e * T * T;
    if (tid == 0) s_bad = 0;
    if (tid < 128) {
        const double w = tid < T ? A.w[(r0 + tid) * A.L + l] : 0.0;
        const double d = tid * H.dt, d2 = d * d;
        const double kk = sigmasq * exp(-omega * d2);
        sv[tid] = sqrt(w);
        kv[tid] = kk + (tid == 0 ? eps : 0.0);
        dkv[tid] = -kk * d2 * omega;
        muv[tid] = tid < T ? A.mu[(r0 + tid) * A.L + l] : 0.0;
    }
    __syncthreads();
    double acc_s[8];  // this thread's shares of: quad, gq, tr, cs (and spares)
#pragma unroll
    for (int i = 0; i < 8; ++i) acc_s[i] = 0.0;
    // ---- S1: wave 0 inverts A11 = I + S1 K11 S1; the other waves form alpha = K^-1 mu ----
    if (wid == 0) {
        buf[G::O_SV + lane] = sv[lane];
        buf[G::O_KVM + 63 + lane] = kv[lane];
        buf[G::O_KVM + 63 - lane] = kv[lane];
        buf[G::O_DKV + lane] = dkv[lane];
        if (lane < 32) buf[G::O_Z + lane] = 0.0;
        tri_wave_sync();
        double ld, unused;
        hstep_task_mfma<64, true>(buf, eps, lane, ld, unused, 0, P11, LD);
        if (lane == 0 && !(ld == ld && fabs(ld) < 1e300)) s_bad = 1;
    } else {
        for (int t = tid - 64; t < T; t += 192) {  // K^-1 symmetric: columns for coalescing
            double a0 = 0.0, a1 = 0.0;
            int j = 0;
            for (; j + 8 <= T; j += 8) {
                double kx[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) kx[q] = Ki[(j + q) * T + t];
#pragma unroll
                for (int q = 0; q < 8; q += 2) {
                    a0 = fma(kx[q], muv[j + q], a0);
                    a1 = fma(kx[q + 1], muv[j + q + 1], a1);
                }
            }
            for (; j < T; ++j) a0 = fma(Ki[j * T + t], muv[j], a0);
            alv[t] = a0 + a1;
        }
    }
    __syncthreads();
    // ---- S2: B = A21 P11 ----
    big_gemm64(lane, wid,
               [&](int i, int k) { return (sv[64 + i] * sv[k]) * kv[64 + i - k]; },
               [&](int k, int j) { return P11[k * LD + j]; },
               [&](int i, int j, double v) { Bm[i * LD + j] = v; });
    __syncthreads();
    // ---- S3: Schur complement A22 - B A21' (rows >= T2: s = 0, B = 0 -> the identity) ----
    big_gemm64(lane, wid,
               [&](int i, int k) { return Bm[i * LD + k]; },
               [&](int k, int j) { return (sv[64 + j] * sv[k]) * kv[64 + j - k]; },
               [&](int i, int j, double v) {
                   const int dd = i > j ? i - j : j - i;
                   Sm[i * LD + j] = fma(sv[64 + i] * sv[64 + j], kv[dd], (i == j ? 1.0 : 0.0)) - v;
               });
    __syncthreads();
    // ---- S4: wave 0 inverts the Schur complement in place; the others take the sums over P11 and the quadratic terms ----
    if (wid == 0) {
        double ld, unused;
        hstep_task_mfma<64, true, true>(buf, eps, lane, ld, unused, 64, Sm, LD, Sm, LD);
        if (lane == 0 && !(ld == ld && fabs(ld) < 1e300)) s_bad = 1;
    } else {
        for (int idx = tid - 64; idx < 64 * 64; idx += 192) {
            const int j = idx >> 6, k = idx & 63;
            const int dd = j > k ? j - k : k - j;
            const double pv = P11[j * LD + k];
            acc_s[3] = fma((sv[j] * sv[k]) * dkv[dd], pv, acc_s[3]);
            if (j == k) acc_s[2] += pv;
        }
        for (int t = tid - 64; t < T; t += 192) {
            double s0 = 0.0;
            for (int j = 0; j < T; ++j) s0 = fma(dkv[j > t ? j - t : t - j], alv[j], s0);
            acc_s[0] = fma(muv[t], alv[t], acc_s[0]);
            acc_s[1] = fma(s0, alv[t], acc_s[1]);
        }
    }
    __syncthreads();
    // ---- S5: C = Q B (over P11) ----
    big_gemm64(lane, wid,
               [&](int i, int k) { return Sm[i * LD + k]; },
               [&](int k, int j) { return Bm[k * LD + j]; },
               [&](int i, int j, double v) { P11[i * LD + j] = v; });
    __syncthreads();
    // ---- S6: <B W11, C>, <B, C>, <W21, C>, <W22, Q>, tr Q ----
    big_gemm64(lane, wid,
               [&](int i, int k) { return Bm[i * LD + k]; },
               [&](int k, int j) { return (sv[k] * sv[j]) * dkv[k > j ? k - j : j - k]; },
               [&](int i, int j, double v) { acc_s[3] = fma(v, P11[i * LD + j], acc_s[3]); });
    for (int idx = tid; idx < 64 * 64; idx += 256) {
        const int i = idx >> 6, k = idx & 63;
        const double cv = P11[i * LD + k];
        acc_s[2] = fma(Bm[i * LD + k], cv, acc_s[2]);
        acc_s[3] = fma(-2.0 * (sv[64 + i] * sv[k]) * dkv[64 + i - k], cv, acc_s[3]);
        const double qv = Sm[i * LD + k];
        const int dd = i > k ? i - k : k - i;
        acc_s[3] = fma((sv[64 + i] * sv[64 + k]) * dkv[dd], qv, acc_s[3]);
        if (i == k) acc_s[2] += qv;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        double v = acc_s[q];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wid][q] = v;
    }
    __syncthreads();
    if (tid == 0) {
        double t4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) t4[q] = (red[0][q] + red[1][q]) + (red[2][q] + red[3][q]);
        const double quad = t4[0], gq = t4[1], tr = t4[2] - (double)(64 - T2), cs = t4[3];
        double ll = -0.5 * quad - 0.5 * tr - A.scal[4 * e + 0];
        double dll = 0.5 * (gq - cs);
        if (s_bad) { ll = nan(""); dll = nan(""); }
        A.out[((int64_t)e * A.M + seg) * 2 + 0] = ll;
        A.out[((int64_t)e * A.M + seg) * 2 + 1] = dll;
    }
}

// out[e][c] = sum_i in[e][i][c]  (fixed order: strided partial sums, then a tree)
__global__ void __launch_bounds__(256) hstep_reduce_kernel(int M, const double* in, double* out,
                                                           const double* scal = nullptr, double* ok_out = nullptr) {
    __shared__ double r0[256], r1[256];
    const int e = blockIdx.x;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < M; i += 256) {
        a += in[((int64_t)e * M + i) * 2 + 0];
        b += in[((int64_t)e * M + i) * 2 + 1];
    }
    r0[threadIdx.x] = a;
    r1[threadIdx.x] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) {
            r0[threadIdx.x] += r0[threadIdx.x + o];
            r1[threadIdx.x] += r1[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[2 * e + 0] = r0[0];
        out[2 * e + 1] = r1[0];
        if (scal) ok_out[e] = scal[4 * e + 3];
    }
}


// ===========================================================================
// Fast path (window known at compile time): register-resident rows, see
// wave_tri.h.  Uses the algebraically equivalent, better conditioned form
//     A_i = I + W_i^1/2 K W_i^1/2          (eigenvalues >= 1)
//     tr(K^-1 S_i)            = tr(A_i^-1)
//     tr(K^-1 S_i K^-1 dK)    = tr(K^-1 dK) - sum_jk sqrt(w_j w_k) dK_jk (A_i^-1)_jk
// (W^1/2 S W^1/2 = I - A^-1, diag(dK) = 0), so that K^-1 is needed only for
// alpha = K^-1 mu, and K, dK enter through their first columns (Toeplitz).
// ===========================================================================
struct HFastArgs {
    int L, M;
    int Tr;              // rows actually present per segment (<= the compiled window T): rows Tr..T-1 are
                         // identity padding (K block-diagonal with I, w = mu = 0), which leaves every term but
                         // tr(A^-1) unchanged; that one is corrected by T - Tr
    double dt;
    const int64_t* off;
    const double* mu;
    const double* w;
    const double* wlm;   // w latent-major, (L, wld), or null: inside a vlgp_hstep_begin bracket the round kernels read a
    int64_t wld;         // segment's 50 weights as 400 contiguous bytes instead of 8 of every 8 L (2.8x over-fetch, PMC)
    int latent[16];      // by value: no host->device copies on the round's critical path
    double logp[48];
    double* kinv;   // (n_eval, T, T)
    double* kcol;   // (n_eval, 2, 64): first column of K, first column of dK
    double* scal;   // (n_eval, 4): logdet, -, omega_used, ok
    double* out;    // (n_eval, M, 2)
};

// ---------------------------------------------------------------------------
// Second moments of the latent means, C_l = sum_i mu_i[:, l] mu_i[:, l]' (T x T per
// latent, summed over the set's units).  The quadratic terms of the objective depend
// on the units only through C_l:
//     sum_i mu_i' K^-1 mu_i              = tr(K^-1 C_l)
//     sum_i alpha_i' dK alpha_i          = tr(K^-1 dK K^-1 C_l),   alpha_i = K^-1 mu_i
// and mu does not change during one gp.optimize call, so C_l is built once per
// H-step (vlgp_hstep_begin) and every evaluation's K block reduces it in ~10 us.
// ---------------------------------------------------------------------------
template <int T>
__global__ void __launch_bounds__(256) hstep_moment_kernel(int L, int M, int Tr, const int64_t* off, const double* mu,
                                                           int nchunk, double* part) {
    // thread <-> (row j of C_l, a run of CW columns): one LDS read of mu_j and CW / 2 wide reads of the run per segment for
    // CW multiply-adds (an entry per thread in index order cost two reads per multiply-add)
    constexpr int CW = (T * T + 255) / 256;    // 10 at T = 50, 16 at T = 64
    constexpr int NG = (T + CW - 1) / CW;      // column runs: 5, 4
    static_assert(T * NG <= 256, "one thread per (row, run)");
    constexpr int SG = 16;  // segments per stage: their loads are in flight together, the next stage's under this one's sums
    __shared__ __attribute__((aligned(16))) double s[SG][64];
    const int l = blockIdx.y, c = blockIdx.x, tid = threadIdx.x;
    const int per = (M + nchunk - 1) / nchunk;
    const int m0 = c * per, m1 = (m0 + per < M) ? m0 + per : M;
    const bool act = tid < T * NG;
    const int j = act ? tid % T : 0, k0 = act ? (tid / T) * CW : 0;
    double acc[CW];
#pragma unroll
    for (int i = 0; i < CW; ++i) acc[i] = 0.0;
    // (segments enter every sum in ascending order: the same bits whatever the stage size or the thread mapping)
    const int w4 = tid >> 6, t = tid & 63;
    double nx[SG / 4];
    auto fetch = [&](int m) {
#pragma unroll
        for (int q = 0; q < SG / 4; ++q) {
            const int sg = w4 + 4 * q;
            nx[q] = (m + sg < m1 && t < Tr) ? mu[(off[m + sg] + t) * L + l] : 0.0;
        }
    };
    if (m0 < m1) fetch(m0);
    for (int m = m0; m < m1; m += SG) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < SG / 4; ++q) s[w4 + 4 * q][t] = nx[q];
        __syncthreads();
        if (m + SG < m1) fetch(m + SG);
#pragma unroll
        for (int sg = 0; sg < SG; ++sg) {
            const double a = s[sg][j];
#pragma unroll
            for (int i = 0; i < CW; ++i) acc[i] = fma(a, (k0 + i < 64) ? s[sg][k0 + i] : 0.0, acc[i]);
        }
    }
    if (act) {
#pragma unroll
        for (int i = 0; i < CW; ++i)
            if (k0 + i < T) part[((int64_t)l * nchunk + c) * T * T + j * T + k0 + i] = acc[i];
    }
}

__global__ void __launch_bounds__(256) hstep_moment_reduce(int nchunk, int TT, const double* part, double* mom) {
    const int idx = blockIdx.x * 256 + threadIdx.x, l = blockIdx.y;
    if (idx >= TT) return;
    double a = 0.0;
    for (int c = 0; c < nchunk; ++c) a += part[((int64_t)l * nchunk + c) * TT + idx];  // fixed order
    mom[(int64_t)l * TT + idx] = a;
}

// ---------------------------------------------------------------------------
// One launch per L-BFGS-B round.  Blocks [0, n_eval) factor K and reduce the
// moment matrices (hstep_round_kblock); the others take segments of one evaluation
// and compute only the A_i terms, which need nothing from the K blocks, so all
// blocks run concurrently.  Every block leaves one partial sum and takes a
// ticket (release); the block that draws the last ticket adds the partials in a
// fixed order and publishes (ll, dll, ok) per evaluation -- to mapped host memory
// when the host polls (single rank), else to `red` for the all-reduce.
// ---------------------------------------------------------------------------
struct HRoundArgs {
    HFastArgs F;
    int n_eval, nb;        // nb = segment blocks per evaluation
    int lds_doubles;       // dynamic LDS of the low-rank launch (fused tables: scratch at its end)
    unsigned seq;          // launch sequence number published with the results
    unsigned* sync;        // [16]: finished-block counter
    const double* mom;     // (L, T, T) second moments of mu
    double* qsum;          // (n_eval, 2): tr(K^-1 C), tr(K^-1 dK K^-1 C)
    double* red;           // device: (ll, dll) x n_eval, then ok x n_eval
    double* host;          // mapped pinned copy of `red` + sequence word at [48], or null
    // low-rank round (hstep_round_lr): tables written by hstep_lr_tables, launched in front
    HLrTabArgs lr;         // lr.tab == null in the dense rounds
    // a round may be several launches (one per rank class of the low-rank kernel): the ticket counts the blocks of all
    unsigned total_blocks;  // blocks of the whole round
    int k_blocks;           // this launch starts with the n_eval K blocks (the first launch of a round), else 0
    int lr_nev;             // evaluations this launch takes the segments of ...
    int lr_ev[16];          // ... and which
    // a MIXED round is two launches, the low-rank kernel for the evaluations within rank LR_RCAP and the dense kernel for
    // the rest: their segment blocks differ in size (16 / 4 segments), so the partial sums of evaluation e sit at
    // out + 2 (e slot_stride + block) and there are nbe[e] of them; lr_mask: the evaluations that went through the tables
    int slot_stride;
    int nbe[16];
    unsigned lr_mask;
    unsigned long long* clk;  // debug: per-phase cycle counters of the first segment block (vlgp_debug_phase_clock), or null
    int prio;                 // 1: the round's waves run at high instruction priority (hstep_wave_prio)
};

// K block of a round (shared by the dense and the low-rank round kernels): wave 0 takes K -> K^-1 and log det through
// the blocked elimination of hstep_mfma.h (KMODE); then every wave takes block rows of the two products against the
// second moments of this latent.  `lds`: TASKW + T * LDK doubles, `kvs`: the SHR doubles of shared tables (one-set layout).
template <int T, int NW, bool ONESET>
struct HRoundK {
    using G = HmGeom<T>;
    static constexpr int LDK = T | 2;                  // row stride of K^-1 in LDS: 2 mod 4 -> conflict-free operand reads
    static constexpr int TASKW = ONESET ? HmGeom50::TASK : G::TASK;           // per-wave task buffer
    static constexpr int SHR = ONESET ? HmGeom50::SHARED : 0;                // per-workgroup tables (one-set layout)
    static constexpr int KBLK = TASKW + T * LDK + SHR;  // wave 0's task buffer | K^-1 (| tables)
};

template <int T, int NW, bool ONESET>
__device__ __forceinline__ void hstep_round_kblock(const HRoundArgs& R, int e, double* lds, double* kvs, double (*part)[2],
                                                   int lane, int wid) {
    using G = HmGeom<T>;
    using KG = HRoundK<T, NW, ONESET>;
    constexpr int LDK = KG::LDK, TASKW = KG::TASKW;
    const HFastArgs& A = R.F;
    double* buf = lds;
    double* Kl = lds + TASKW;
    double* dks = kvs + HmGeom50::KVN;
    // the operands of C for this wave's block row of the products: in flight while wave 0 factors
    double preC[(T + 3) / 4];
    constexpr bool PRE = NW >= (T + 15) / 16;
    // (wave 0 fetches after its factorisation: thirteen values held across it cost 40 spilled registers)
    if constexpr (PRE) {
        if (wid != 0) hstep_kblock_fetch<T>(R.mom + (int64_t)A.latent[e] * T * T, lane, wid, preC);
    }
    if (wid == 0) {
        const double sigmasq = exp(A.logp[3 * e + 0]), omega = exp(A.logp[3 * e + 1]), eps = exp(A.logp[3 * e + 2]);
        if constexpr (ONESET) {
            const double d = lane * A.dt, d2 = d * d;
            const double kk = sigmasq * exp(-omega * d2);
            if (lane < HmGeom50::SVN) buf[HmGeom50::O_SV + lane] = lane < A.Tr ? 1.0 : 0.0;  // rows >= Tr: identity padding
            if (lane < T) kvs[17 + lane] = kk + (lane == 0 ? eps : 0.0);
            if (lane >= 1 && lane <= 17) kvs[17 - lane] = kk;
            if (lane < HmGeom50::DKN) dks[lane] = -kk * d2 * omega;
        } else {
            const double d = lane * A.dt, d2 = d * d;
            const double kk = sigmasq * exp(-omega * d2);
            buf[G::O_SV + lane] = lane < A.Tr ? 1.0 : 0.0;  // rows >= Tr: identity padding
            buf[G::O_KVM + 63 + lane] = kk + (lane == 0 ? eps : 0.0);
            buf[G::O_KVM + 63 - lane] = kk + (lane == 0 ? eps : 0.0);
            buf[G::O_DKV + lane] = -kk * d2 * omega;
            if (lane < 32) buf[G::O_Z + lane] = 0.0;
        }
        tri_wave_sync();
        double logdet, unused;
        if constexpr (ONESET) hstep_task_mfma50<true>(buf, kvs, dks, eps, lane, logdet, unused, A.Tr, Kl, LDK);
        else hstep_task_mfma<T, true>(buf, eps, lane, logdet, unused, A.Tr, Kl, LDK);
        if (lane == 0) {
            A.scal[4 * e + 0] = logdet;
            A.scal[4 * e + 1] = 0.0;
            A.scal[4 * e + 2] = omega;
            A.scal[4 * e + 3] = (logdet == logdet && fabs(logdet) < 1e300) ? 1.0 : 0.0;  // a bad pivot -> NaN / inf
        }
        if constexpr (PRE) hstep_kblock_fetch<T>(R.mom + (int64_t)A.latent[e] * T * T, lane, 0, preC);
    }
    __syncthreads();
    double quad, gq;
    hstep_kblock_products<T>(Kl, LDK, R.mom + (int64_t)A.latent[e] * T * T, ONESET ? dks : buf + G::O_DKV, A.Tr, lane,
                             wid, NW, quad, gq, PRE ? &preC : nullptr);
    for (int o = 32; o > 0; o >>= 1) {
        quad += __shfl_xor(quad, o, 64);
        gq += __shfl_xor(gq, o, 64);
    }
    if (lane == 0) {
        part[wid][0] = quad;
        part[wid][1] = gq;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double q0 = part[0][0], q1 = part[0][1];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            q0 += part[w][0];
            q1 += part[w][1];
        }
        R.qsum[2 * e + 0] = q0;
        R.qsum[2 * e + 1] = q1;
    }
}

// Completion of a round: one partial per segment block (part[w] summed over the NW waves), then the block that draws
// the last ticket adds the partials of every evaluation in a fixed order and publishes (ll, dll, flag).  `rs`: 2 x 64 NW
// doubles of LDS.  flag: 1 = fine, 0 = K did not factor, 2 = the low-rank tables overflowed (R.lr.meta)
template <int NW>
__device__ __forceinline__ void hstep_round_finish(const HRoundArgs& R, double* rs, double (*part)[2], int* s_last,
                                                   int64_t out_slot) {
    const HFastArgs& A = R.F;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (out_slot >= 0) {
            double* o = A.out + 2 * out_slot;
            double t0 = part[0][0], t1 = part[0][1];
#pragma unroll
            for (int w = 1; w < NW; ++w) {
                t0 += part[w][0];
                t1 += part[w][1];
            }
            o[0] = t0;
            o[1] = t1;
        }
        const unsigned ticket = __hip_atomic_fetch_add(&R.sync[16], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        *s_last = ticket == R.total_blocks - 1;
    }
    __syncthreads();
    if (!*s_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    constexpr int NT = 64 * NW;
    for (int e = 0; e < R.n_eval; ++e) {
        const double2* in = reinterpret_cast<const double2*>(A.out) + (int64_t)e * R.slot_stride;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll 8
        for (int m = threadIdx.x; m < R.nbe[e]; m += NT) {
            const double2 v = in[m];
            s0 += v.x;
            s1 += v.y;
        }
        __syncthreads();
        rs[threadIdx.x] = s0;
        rs[NT + threadIdx.x] = s1;
        __syncthreads();
        for (int o = NT / 2; o > 0; o >>= 1) {
            if ((int)threadIdx.x < o) {
                rs[threadIdx.x] += rs[threadIdx.x + o];
                rs[NT + threadIdx.x] += rs[NT + threadIdx.x + o];
            }
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            double okf = A.scal[4 * e + 3];
            if (R.lr.tab != nullptr && ((R.lr_mask >> e) & 1u) && R.lr.meta[e].overflow) okf = okf != 0.0 ? 2.0 : 0.0;
            const double ll = -0.5 * R.qsum[2 * e + 0] - 0.5 * rs[0] - (double)A.M * A.scal[4 * e + 0];
            const double dll = 0.5 * (R.qsum[2 * e + 1] - rs[NT]);
            R.red[2 * e + 0] = ll;
            R.red[2 * e + 1] = dll;
            R.red[2 * R.n_eval + e] = okf;
            if (R.host) {
                R.host[2 * e + 0] = ll;
                R.host[2 * e + 1] = dll;
                R.host[2 * R.n_eval + e] = okf;
            }
        }
    }
    if (threadIdx.x == 0) {
        R.sync[16] = 0;  // next launch is stream-ordered after this one
        if (R.host) {
            __threadfence_system();
            __hip_atomic_store(reinterpret_cast<unsigned long long*>(R.host + 48), (unsigned long long)R.seq,
                               __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The round kernels run beside the M-step lane's launches (ctx.h: mstream), on the same SIMDs.  When the rounds are the
// critical path of that window (C3: ≈ 40 dependent rounds against 25 independent Newton launches with ≈ 0.6 ms of slack)
// their waves take the instruction arbiter's highest user priority.  Same box, four runs each: 139.3 EM it/s at
// priority 0, 140.8 / 140.4 / 141.8 at 1 / 2 / 3 (H-step 4.16 -> 3.97 ms, the M-step lane 3.34 -> 3.63 ms beside it).
// Where the M-step lane is the longer one (C5: 28.6 against 24.5 ms) the rounds stay at the default: the host decides
// per bracket from the durations of the previous EM iteration (vlgp_hstep_begin, ctx->h_prio).
__device__ __forceinline__ void hstep_wave_prio(int prio) {
    if (prio) __builtin_amdgcn_s_setprio(3);
}

// (rows, L) -> (L, rows) copy of w for the rounds of one vlgp_hstep_begin bracket
__global__ void __launch_bounds__(256) hstep_w_latent_major(int L, int64_t rows, const double* __restrict__ w, double* __restrict__ wlm) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;  // latent-major index
    if (i >= rows * L) return;
    const int64_t l = i / rows, row = i - l * rows;
    wlm[i] = w[row * L + l];
}

// The round on the matrix pipe (hstep_mfma.h): one wave per segment, NW waves per block.  Blocks [0, n_eval) are
// the K blocks (all NW waves share the trace phase).
// ONESET (T = 50 only): the one-register-set task routine hstep_task_mfma50 (buffer row = lane); else the two-set routine.
template <int T, int NW, bool ONESET = false>
__global__ void __launch_bounds__(64 * NW, NW >= 4 ? (T <= 50 ? (ONESET ? 4 : 3) : 2) : 1) hstep_round_mfma(HRoundArgs R) {
    static_assert(!ONESET || T == 50, "the one-register-set routine is written for 50 = 3 x 16 + 2");
    hstep_wave_prio(R.prio);
    using G = HmGeom<T>;
    using KG = HRoundK<T, NW, ONESET>;
    constexpr int TASKW = KG::TASKW, SHR = KG::SHR, KBLK = KG::KBLK;
    constexpr int LDSN = NW * TASKW + SHR > KBLK ? NW * TASKW + SHR : KBLK;
    static_assert(!ONESET || LDSN * 8 + 128 <= 40960, "four workgroups per CU");
    __shared__ __attribute__((aligned(16))) double lds[LDSN];
    __shared__ double part[NW][2];
    __shared__ int s_last;
    const HFastArgs& A = R.F;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int64_t out_slot = -1;
    if ((int)blockIdx.x < R.k_blocks) {
        hstep_round_kblock<T, NW, ONESET>(R, blockIdx.x, lds, lds + LDSN - SHR, part, lane, wid);
    } else {
        const int b = blockIdx.x - R.k_blocks;
        const int ei = b / R.nb, bx = b - ei * R.nb;
        const int e = R.lr_ev[ei];  // (all of them in order, or the evaluations a mixed round leaves to this kernel)
        out_slot = (int64_t)e * R.slot_stride + bx;
        const int seg = bx * NW + wid;
        double tr = 0.0, cs = 0.0;
        if (seg < A.M) {
            double* buf = lds + wid * TASKW;
            double* kvs = lds + LDSN - SHR;  // one-set layout: every wave of the block writes the same table values
            double* dks = kvs + HmGeom50::KVN;
            const int l = A.latent[e];
            const int64_t r0row = A.off[seg];
            const double sigmasq = exp(A.logp[3 * e + 0]), omega = exp(A.logp[3 * e + 1]), eps = exp(A.logp[3 * e + 2]);
            if constexpr (ONESET) {
                const double w = lane < A.Tr ? (A.wlm ? A.wlm[(int64_t)l * A.wld + r0row + lane] : A.w[(r0row + lane) * A.L + l])
                                             : 0.0;  // rows >= Tr: identity padding
                const double d = lane * A.dt, d2 = d * d;
                const double kk = sigmasq * exp(-omega * d2);
                if (lane < HmGeom50::SVN) buf[HmGeom50::O_SV + lane] = sqrt(w);
                if (lane < T) kvs[17 + lane] = kk + (lane == 0 ? eps : 0.0);
                if (lane >= 1 && lane <= 17) kvs[17 - lane] = kk;
                if (lane < HmGeom50::DKN) dks[lane] = -kk * d2 * omega;
            } else {
                const double w = lane < A.Tr ? (A.wlm ? A.wlm[(int64_t)l * A.wld + r0row + lane] : A.w[(r0row + lane) * A.L + l])
                                             : 0.0;  // rows >= Tr: identity padding
                const double d = lane * A.dt, d2 = d * d;
                const double kk = sigmasq * exp(-omega * d2);
                buf[G::O_SV + lane] = sqrt(w);
                buf[G::O_KVM + 63 + lane] = kk + (lane == 0 ? eps : 0.0);
                buf[G::O_KVM + 63 - lane] = kk + (lane == 0 ? eps : 0.0);
                buf[G::O_DKV + lane] = -kk * d2 * omega;
                if (lane < 32) buf[G::O_Z + lane] = 0.0;
            }
            tri_wave_sync();
            bool ok;
            if constexpr (ONESET) ok = hstep_task_mfma50<false>(buf, kvs, dks, eps, lane, tr, cs);
            else ok = hstep_task_mfma<T>(buf, eps, lane, tr, cs);
            if (!ok) { tr = nan(""); cs = nan(""); }  // failed factorisation: propagates into ll, dll
            for (int o = 32; o > 0; o >>= 1) {
                tr += __shfl_xor(tr, o, 64);
                cs += __shfl_xor(cs, o, 64);
            }
            tr -= (double)(T - A.Tr);  // the identity padding's share of tr(A^-1)
        }
        if (lane == 0) {
            part[wid][0] = tr;
            part[wid][1] = cs;
        }
    }
    hstep_round_finish<NW>(R, lds, part, &s_last, out_slot);
}

// Tables of the evaluations of a low-rank round: one block per evaluation (hstep_lr.h, lr_tables_block).
__global__ void __launch_bounds__(128) hstep_lr_tables(HRoundArgs R) {
    __shared__ double kv[64], dkv[64];
    __shared__ int s_i[4];
    hstep_wave_prio(R.prio);
    const int e = R.lr_ev[blockIdx.x];  // (the evaluations of the round that take the low-rank kernel)
    lr_tables_block<128>(R.lr, e, exp(R.F.logp[3 * e + 0]), exp(R.F.logp[3 * e + 1]), kv, dkv, s_i);
}

// The round in the low-rank form (hstep_lr.h): blocks [0, n_eval) are the K blocks as above, every other block takes
// sixteen segments of one evaluation.  The tables of the evaluations come from hstep_lr_tables, launched in front.
// T: compiled window of the K block (50: windows <= 50, 64: <= 64); RC: register class of the ranks in this round.
// (four waves per SIMD up to class 24: since phase 3 takes one kind of tiles at a time the class fits 128 registers
// without scratch; ranks up to 20 then have four workgroups per CU -- their LDS allows it -- and two instead of one fit
// beside a workgroup of the M-step lane.  Same box: 139.0 against 137.7 EM it/s.)
#ifndef HLR64_LB
#define HLR64_LB 4
#endif
template <int T, int NW, int RC, bool TABG = false, bool TABF = false>
__global__ void __launch_bounds__(64 * NW, RC <= 24 ? (T == 64 ? HLR64_LB : 4) : 3) hstep_round_lr(HRoundArgs R) {
    hstep_wave_prio(R.prio);
    constexpr bool ONESET = T == 50;
    constexpr int NK = T == 50 ? 7 : 8;
    using KG = HRoundK<T, NW, ONESET>;
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    __shared__ double part[NW][2];
    __shared__ int s_last;
    const HFastArgs& A = R.F;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    int64_t out_slot = -1;
    if ((int)blockIdx.x < R.k_blocks) {
        hstep_round_kblock<T, NW, ONESET>(R, blockIdx.x, lds_dyn, lds_dyn + KG::KBLK - KG::SHR, part, lane, wid);
    } else {
        const int b = blockIdx.x - R.k_blocks;
        const int ei = b / R.nb, bx = b - ei * R.nb;
        const int e = R.lr_ev[ei];
        out_slot = (int64_t)e * R.slot_stride + bx;
        double tr = 0.0, cs = 0.0;
        const double eps = exp(A.logp[3 * e + 2]);
        // (w indexed as w[row * L + l]: the latent-major copy enters as its latent's column with L = 1, l = 0)
        if constexpr (TABF)
            lr_group<RC, NK, NW, false, true>(nullptr, LrMeta{}, nullptr,
                             A.wlm ? A.wlm + (int64_t)A.latent[e] * A.wld : A.w, A.off, A.wlm ? 1 : A.L,
                             A.wlm ? 0 : A.latent[e], A.M, A.Tr, eps, 16 * bx, lds_dyn, lane, wid, tr, cs,
                             b == 0 ? R.clk : nullptr, &R.lr, e, exp(A.logp[3 * e + 0]), exp(A.logp[3 * e + 1]),
                             R.lds_doubles, bx == 0);
        else
        lr_group<RC, NK, NW, TABG>(R.lr.tab + (int64_t)e * 2 * LR_TROWS * LR_RCAP, R.lr.meta[e], R.lr.pairs + (int64_t)e * LR_NPAIR,
                             A.wlm ? A.wlm + (int64_t)A.latent[e] * A.wld : A.w, A.off, A.wlm ? 1 : A.L,
                             A.wlm ? 0 : A.latent[e], A.M, A.Tr, eps, 16 * bx, lds_dyn, lane, wid, tr, cs,
                             b == 0 ? R.clk : nullptr);
        if (lane == 0) {
            part[wid][0] = wid == 0 ? tr : 0.0;
            part[wid][1] = wid == 0 ? cs : 0.0;
        }
    }
    hstep_round_finish<NW>(R, lds_dyn, part, &s_last, out_slot);
}

// ---- low-rank round: host side ------------------------------------------------------------------------------
// Rank of the folded even / odd blocks of exp(-omega D^2) under the pivoted Cholesky of hstep_lr_tables (same pivot
// rule, no tangent), used to predict the LDS a round needs; > LR_RCAP when a block exceeds the kernel's capacity.
static int lr_host_rank(int T, double dt, double omega, double tol) {
    const int h = T / 2, nt = (T + 1) / 2;
    const bool odd = (T & 1) != 0;
    std::vector<double> kv(T);
    for (int d = 0; d < T; ++d) kv[d] = exp(-omega * (d * dt) * (d * dt));
    const double RS2 = 0.70710678118654752440;
    int total = 0;
    for (int par = 0; par < 2; ++par) {
        const int n = par == 0 ? nt : h;
        auto kent = [&](int t, int p) {
            const int d1 = t > p ? t - p : p - t, d2 = T - 1 - t - p;
            if (par == 0) {
                const double s = ((odd && t == h) ? RS2 : 1.0) * ((odd && p == h) ? RS2 : 1.0);
                return s * (kv[d1] + kv[d2]);
            }
            return kv[d1] - kv[d2];
        };
        std::vector<double> G((size_t)n * (n > 0 ? n : 1), 0.0), d(n);
        for (int t = 0; t < n; ++t) d[t] = kent(t, t);
        int r = 0;
        for (; r < n; ++r) {
            int p = -1;
            double bv = -1.0;
            for (int t = 0; t < n; ++t)
                if (d[t] > bv) { bv = d[t]; p = t; }
            if (!(bv > tol)) break;
            const double ginv = 1.0 / sqrt(bv);
            for (int t = 0; t < n; ++t) {
                double col = kent(t, p);
                for (int j = 0; j < r; ++j) col -= G[(size_t)t * n + j] * G[(size_t)p * n + j];
                const double gk = col * ginv;
                G[(size_t)t * n + r] = gk;
                if (d[t] >= 0.0) d[t] -= gk * gk;
            }
            d[p] = -1.0;
        }
        if (r > LR_RH) return LR_RCAP + 1;
        total += r;
    }
    return total;
}

// largest omega whose predicted rank is <= r, r = 0 .. LR_RCAP (the rank grows with omega)
static const std::vector<double>& lr_thresholds(vlgp_ctx* ctx, int T, double dt, double tol) {
    for (auto& t : ctx->lr_thr)
        if (t.T == T && t.dt == dt && t.tol == tol) return t.om;
    vlgp_ctx::LrThr t;
    t.T = T; t.dt = dt; t.tol = tol;
    t.om.assign(LR_RCAP + 1, 0.0);
    const double lo0 = log(1e-14), hi0 = log(1e8);
    for (int r = 0; r <= LR_RCAP; ++r) {
        double lo = r > 0 && t.om[r - 1] > 0.0 ? log(t.om[r - 1]) : lo0, hi = hi0;
        if (lr_host_rank(T, dt, exp(lo), tol) > r) { t.om[r] = 0.0; continue; }
        for (int it = 0; it < 48; ++it) {
            const double mid = 0.5 * (lo + hi);
            if (lr_host_rank(T, dt, exp(mid), tol) <= r) lo = mid;
            else hi = mid;
        }
        t.om[r] = exp(lo);
    }
    ctx->lr_thr.push_back(t);
    return ctx->lr_thr.back().om;
}

constexpr int LR_NW = 4;  // waves per workgroup of the low-rank round (sixteen segments)
static inline int n_lr_or_all(bool lr, int n_lr, int n_eval) { return lr ? n_lr : n_eval; }

template <int T, int RC, bool TABG = false, bool TABF = false>
static int launch_round_lr(vlgp_ctx* ctx, const HRoundArgs& R, int grid, size_t lds_bytes) {
    constexpr int NW = LR_NW;
    // the dynamic-LDS ceiling is a per-DEVICE attribute of the function: remembered per handle (one handle = one device),
    // not per process (a second engine on another device never got it)
    const void* fn = reinterpret_cast<const void*>(hstep_round_lr<T, NW, RC, TABG, TABF>);
    bool have = false;
    for (const void* f : ctx->lds_attr_done) have = have || f == fn;
    if (!have) {
        HIPCHK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024 - 512)));
        ctx->lds_attr_done.push_back(fn);
    }
    hipLaunchKernelGGL((hstep_round_lr<T, NW, RC, TABG, TABF>), dim3(grid), dim3(64 * NW), lds_bytes, ctx->stream, R);
    return VLGP_OK;
}

static int launch_hstep_impl(vlgp_ctx* ctx, UnitSet& us, int window, double dt, int n_eval, const int* latent,
                             const double* logp, double* ll, double* dll, bool force_dense);

int launch_hstep(vlgp_ctx* ctx, UnitSet& us, int window, double dt, int n_eval, const int* latent,
                 const double* logp, double* ll, double* dll) {
    return launch_hstep_impl(ctx, us, window, dt, n_eval, latent, logp, ll, dll, false);
}

// What one gp.optimize run needs of the units before its first round and keeps until its last: the second moments
// C_l of mu (hstep_moment_kernel + reduce) and, for the round kernels' coalesced reads, w latent-major.  Both depend on
// the units only, so vlgp_hstep_prepare can enqueue them the moment the E-step is done -- under the host's way to the
// first objective call -- instead of in front of the first round.  Buffers of their own (not the round workspace).
static bool hstep_round_kernels_apply(vlgp_ctx* ctx, const UnitSet& us, int T) {
    const HstepSwitches& sw = ctx->hsw;
    return us.Tmin == T && us.Tmax == T && T <= 64 && T >= (sw.dense ? 24 : 4) && !sw.generic;
}
static int hstep_prepare_units(vlgp_ctx* ctx, UnitSet& us, int T, bool want_wlm, bool want_mom, hipStream_t st) {
    const int L = ctx->L, M = us.M;
    if (want_wlm && (!ctx->hwlm_valid || ctx->hmom_us != &us)) {
        const int64_t n = us.rows * L;
        if (ctx->hwlm_len < n) {
            if (ctx->d_hwlm) HIPCHK(ctx, hipFree(ctx->d_hwlm));
            ctx->d_hwlm = nullptr; ctx->hwlm_len = 0;
            HIPCHK(ctx, hipMalloc(&ctx->d_hwlm, sizeof(double) * (size_t)n));
            ctx->hwlm_len = n;
        }
        hipLaunchKernelGGL(hstep_w_latent_major, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, L,
                           us.rows, us.w, ctx->d_hwlm);
        HIPCHK(ctx, hipGetLastError());
        ctx->hwlm_valid = true;
    }
    if (!want_mom) return VLGP_OK;
    const int HT = T <= 50 ? 50 : 64;
    constexpr int NCH = 64;
    if (!ctx->d_hmom || ctx->hmom_len < (int64_t)L * HT * HT) {
        if (ctx->d_hmom) HIPCHK(ctx, hipFree(ctx->d_hmom));
        if (ctx->d_hmpart) HIPCHK(ctx, hipFree(ctx->d_hmpart));
        ctx->d_hmom = nullptr; ctx->d_hmpart = nullptr;
        HIPCHK(ctx, hipMalloc(&ctx->d_hmom, sizeof(double) * L * HT * HT));
        HIPCHK(ctx, hipMalloc(&ctx->d_hmpart, sizeof(double) * (size_t)L * NCH * HT * HT));
        ctx->hmom_len = (int64_t)L * HT * HT;
        ctx->hmom_us = nullptr;
    }
    if (ctx->hmom_us != &us || ctx->hmom_T != T) {
        if (HT == 50)
            hipLaunchKernelGGL((hstep_moment_kernel<50>), dim3(NCH, L), dim3(256), 0, st, L, M, T,
                               us.d_off, us.mu, NCH, ctx->d_hmpart);
        else
            hipLaunchKernelGGL((hstep_moment_kernel<64>), dim3(NCH, L), dim3(256), 0, st, L, M, T,
                               us.d_off, us.mu, NCH, ctx->d_hmpart);
        hipLaunchKernelGGL(hstep_moment_reduce, dim3((HT * HT + 255) / 256, L), dim3(256), 0, st, NCH,
                           HT * HT, ctx->d_hmpart, ctx->d_hmom);
        HIPCHK(ctx, hipGetLastError());
        ctx->hmom_us = &us;
        ctx->hmom_T = T;
    }
    return VLGP_OK;
}
int hstep_prepare(vlgp_ctx* ctx, UnitSet& us, int T, hipStream_t st) {
    static const bool no_wlm = getenv("VLGP_HSTEP_NO_WLM") != nullptr;
    if (!hstep_round_kernels_apply(ctx, us, T)) return VLGP_OK;  // (the objective call reports what is wrong, if anything)
    ctx->hmom_us = nullptr;  // units as they are when everything queued so far is done
    ctx->hwlm_valid = false;
    return hstep_prepare_units(ctx, us, T, !no_wlm, true, st);
}

static int launch_hstep_impl(vlgp_ctx* ctx, UnitSet& us, int window, double dt, int n_eval, const int* latent,
                             const double* logp, double* ll, double* dll, bool force_dense) {
    const int T = window, L = ctx->L, M = us.M;
    if (T > HS_MAXT) return vlgp_fail(ctx, VLGP_ERR_ARG, "H-step kernel supports window <= %d, got %d", HS_MAXT, T);
    const bool huge = (size_t)(T * (T | 1)) * 8 > 150 * 1024;  // (T > ~138): not even one matrix per workgroup fits LDS
    if (us.Tmin != T || us.Tmax != T)
        return vlgp_fail(ctx, VLGP_ERR_STATE, "H-step needs every unit to have exactly window=%d rows", T);
    // the T = 50 kernels, identity-padded; below ~half the compiled size the padding costs more than the
    // generic kernels (measured: window 25 14 vs 24 ms per H-step, window 20 21 vs 13 ms)
    // windows 24..50 run the kernels compiled for 50, 51..64 those compiled for 64 (matrix-pipe round only)
    // round 4: windows of 4 ... 23 bins take the round kernels too when the low-rank round runs (its cost follows the
    // rank, not the compiled window; the K block pads to 50 as it always did): the generic kernels below 24 remain
    // for the dense switches only
    // (round 4: the round-1 / round-2 kernels behind VLGP_HSTEP_UNFUSED / _PADDED / _LEAN / _TWOSET are gone; what is left
    // is the low-rank round, the dense matrix-pipe round (VLGP_HSTEP_DENSE=1 forces it), and the generic kernels
    // (VLGP_HSTEP_GENERIC=1), which also implement the reference's omega retry)
    const HstepSwitches& sw = ctx->hsw;  // (the environment is read at vlgp_create / vlgp_debug_reload_switches)
    const bool lr_allowed = !force_dense && !sw.dense;
    const bool fast = T <= 64 && T >= (lr_allowed ? 4 : 24) && !sw.generic;
    const int TC = T <= 50 ? 50 : 64;  // compiled window
    const int64_t TT = fast ? (int64_t)TC * TC : (int64_t)T * T;
    // workspace: kinv | q | dk | scal | seg_out | red | logp | latent(int)
    const int64_t o_kinv = 0, o_q = o_kinv + n_eval * TT, o_dk = o_q + n_eval * TT, o_scal = o_dk + n_eval * TT;
    const int64_t o_out = o_scal + 4 * n_eval, o_red = o_out + 2LL * n_eval * M, o_logp = o_red + 3 * n_eval;
    const int64_t o_lat = o_logp + 3 * n_eval, o_qsum = o_lat + n_eval + 8, o_mpart = o_qsum + 2 * n_eval + 2;
    const int64_t o_tm = o_mpart;  // (the moments' partial sums have a buffer of their own: hstep_prepare_units)
    // low-rank round: tables | meta | pair codes (doubles; each region 16-byte aligned)
    const int64_t o_km = o_tm + (T > 64 ? n_eval * TT : 0);
    const int64_t o_bmat = o_km + (huge ? (int64_t)n_eval * T * (T | 1) : 0);
    const int64_t o_lrtab = (o_bmat + (huge ? (int64_t)n_eval * M * T * (T | 1) : 0) + 1) & ~1LL;
    const int64_t o_lrmeta = o_lrtab + (fast ? (int64_t)n_eval * 2 * LR_TROWS * LR_RCAP : 0);
    const int64_t o_lrpairs = o_lrmeta + (fast ? (int64_t)n_eval * 4 : 0);
    const int64_t total = o_lrpairs + (fast ? ((int64_t)n_eval * LR_NPAIR * 2 + 7) / 8 : 0);
    CHK(vlgp_ensure_work(ctx, total));
    CHK(vlgp_ensure_pinned(ctx, 12 * n_eval + 32));
    double* W = ctx->d_work;
    double* hp = ctx->h_pinned;
    if (n_eval > 16) return vlgp_fail(ctx, VLGP_ERR_ARG, "at most 16 evaluations per call, got %d", n_eval);
    for (int i = 0; i < n_eval; ++i)
        if (latent[i] < 0 || latent[i] >= L) return vlgp_fail(ctx, VLGP_ERR_ARG, "latent index out of range");

    if (fast) {
        HFastArgs F;
        F.L = L; F.M = M; F.Tr = T; F.dt = dt; F.off = us.d_off; F.mu = us.mu; F.w = us.w;
        for (int i = 0; i < n_eval; ++i) F.latent[i] = latent[i];
        for (int i = 0; i < 3 * n_eval; ++i) F.logp[i] = logp[i];
        F.kinv = W + o_kinv; F.kcol = W + o_q; F.scal = W + o_scal; F.out = W + o_out;
        F.wlm = nullptr; F.wld = 0;
        static const bool no_wlm = getenv("VLGP_HSTEP_NO_WLM") != nullptr;
        if (ctx->hmom_bracket && !no_wlm) {  // mu, w are fixed inside the bracket: one transposed copy of w serves every round
            CHK(hstep_prepare_units(ctx, us, T, true, false, ctx->stream));
            F.wlm = ctx->d_hwlm; F.wld = us.rows;
        }
        double* hres = hp + 4 * n_eval + 8;
        {
            if (!ctx->d_hsync) {
                HIPCHK(ctx, hipMalloc(&ctx->d_hsync, 32 * sizeof(unsigned)));
                HIPCHK(ctx, hipMemsetAsync(ctx->d_hsync, 0, 32 * sizeof(unsigned), ctx->stream));
                HIPCHK(ctx, hipHostMalloc(&ctx->h_hres, 64 * sizeof(double), hipHostMallocMapped));
                memset(ctx->h_hres, 0, 64 * sizeof(double));
                HIPCHK(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_hres), ctx->h_hres, 0));
            }
            // second moments of mu: once per vlgp_hstep_begin bracket (or already there: vlgp_hstep_prepare), else per call
            if (!ctx->hmom_bracket) ctx->hmom_us = nullptr;
            CHK(hstep_prepare_units(ctx, us, T, false, true, ctx->stream));
            HRoundArgs R;
            R.F = F;
            constexpr int MFMA_NW = 4;  // waves (= segments) per block of the matrix-pipe round kernel
            const bool mfma = true;
            // the exact low-rank round (hstep_lr.h) when every evaluation's kernel matrix has numerical rank <= LR_RCAP
            // (omega below about 2e-2 on a 50-bin window); VLGP_HSTEP_DENSE=1 keeps the dense matrix-pipe round
            const bool lr_off = sw.dense;
            const double lr_tol = sw.lr_tol;
            bool lr = !lr_off && !force_dense;
            int rcap[16], rmax = 0;
            // evaluations whose kernel matrix has a numerical rank above LR_RCAP (omega above ~2e-2 at window 50: a fit's first
            // iterations) take the dense kernel; the others of the same round keep the low-rank one (a MIXED round: two
            // launches on the stream, one ticket counter, one mailbox -- round 5; until then one rough latent sent its whole
            // round to the dense kernel)
            unsigned rough = 0;
            int n_lr = 0;
            if (lr) {
                const std::vector<double>& om = lr_thresholds(ctx, T, dt, 0.5 * lr_tol);
                for (int e = 0; e < n_eval; ++e) {
                    const double omega = exp(logp[3 * e + 1]);
                    int r = 0;
                    while (r <= LR_RCAP && !(omega <= om[r])) ++r;
                    if (r > LR_RCAP) { rough |= 1u << e; rcap[e] = 0; continue; }
                    rcap[e] = r < 4 ? 4 : r;
                    if (rcap[e] > rmax) rmax = rcap[e];
                    ++n_lr;
                }
                if (n_lr == 0) lr = false;
            }
            const int n_rough = lr ? n_eval - n_lr : 0;
            if (lr && !sw.lowrank) {
                // Which round is faster depends on how much there is to do (measured on MI355X, tools/lr_round_bench.py,
                // us per round over n = n_eval x M segment-evaluations, 4000 of them = one "generation"):
                //   dense      41 + 28 (n / 4000 - 1): one wave per segment, 4096 resident waves, ~25 us wave lifetime
                //   low-rank   13 (tables launch) + base + marg (n / 4000 - 1), by rank class 16 / 24 / 28 / 32:
                //              base 26 / 35 / 42 / 51 (one workgroup's latency), marg 8.7 / 12 / 15.5 / 22.6
                //              (re-measured with the block-triangle inverse and the early loads; class 32 keeps the
                //              lane-per-row inverse and only beats the dense round from about ten evaluations)
                // A few hundred segments (C1, C2, the shard a rank holds at 8 GPUs) never fill the chip: the dense round's
                // single launch wins there.  VLGP_HSTEP_LOWRANK=1 takes the low-rank round regardless (tests).
                static const double base[4] = {26.0, 35.0, 42.0, 51.0}, marg[4] = {8.7, 12.0, 15.5, 22.6};
                const int ci = rmax <= 16 ? 0 : (rmax <= 24 ? 1 : (rmax <= 28 ? 2 : 3));
                // (several ranks: the rule must pick the SAME round on every rank -- a rank alone on the low-rank round
                // would also be alone in its overflow fallback and its extra collectives, ADVICE round 4 -- so it looks
                // at the mean shard, a rank-invariant number once the row totals have been exchanged)
                const double m_rule = (ctx->world > 1 && us.rows_all_ranks > 0.0) ? us.rows_all_ranks / ((double)T * ctx->world) : (double)M;
                const double gens = (double)n_lr * m_rule / 4000.0;  // (the low-rank part of a mixed round)
                const double extra = gens > 1.0 ? gens - 1.0 : 0.0;
                const double t_dense = 41.0 + 28.0 * extra, t_lr = 13.0 + base[ci] + marg[ci] * extra;
                if (T >= 24 && t_lr > 0.95 * t_dense) lr = false;
            }
            const bool mixed = lr && n_rough > 0;
            const int nb_lr = (M + 15) / 16, nb_dense = (M + MFMA_NW - 1) / MFMA_NW;
            R.n_eval = n_eval; R.nb = lr ? nb_lr : nb_dense;
            R.slot_stride = mixed ? nb_dense : R.nb;
            R.lr_mask = 0;
            for (int e = 0; e < n_eval; ++e) {
                const bool e_lr = lr && !((rough >> e) & 1u);
                R.nbe[e] = e_lr ? nb_lr : nb_dense;
                if (e_lr) R.lr_mask |= 1u << e;
                R.lr_ev[e] = e;
            }
            R.seq = ++ctx->h_seq; R.sync = ctx->d_hsync;
            R.mom = ctx->d_hmom; R.qsum = W + o_qsum;
            R.red = W + o_red;
            R.lr.tab = nullptr; R.lr.meta = nullptr; R.lr.pairs = nullptr;
            R.total_blocks = (unsigned)(n_eval + n_lr_or_all(lr, n_lr, n_eval) * (lr ? nb_lr : nb_dense) + (mixed ? n_rough * nb_dense : 0));
            R.k_blocks = n_eval; R.lr_nev = n_eval; R.clk = ctx->d_clk;
            R.prio = ctx->h_prio;
            if (lr) {
                R.lr.T = T; R.lr.dt = dt; R.lr.tol = lr_tol;
                for (int e = 0; e < n_eval; ++e) R.lr.rcap[e] = rcap[e];
                R.lr.tab = W + o_lrtab;
                R.lr.meta = reinterpret_cast<LrMeta*>(W + o_lrmeta);
                R.lr.pairs = reinterpret_cast<unsigned short*>(W + o_lrpairs);
                // (Measured: the tables built by blocks of the round kernel itself, the segment blocks waiting on a flag,
                // is SLOWER than this extra launch -- 57 against 26 + 12 us for one evaluation: the waiting blocks fill
                // the chip before the table blocks finish.)
                const bool fuse_tables = sw.fuse_tables && rmax <= 24;  // (the classes without tables in global memory)
                if (!fuse_tables) {
                    HRoundArgs Rt = R;
                    int k = 0;
                    for (int e = 0; e < n_eval; ++e)
                        if ((R.lr_mask >> e) & 1u) Rt.lr_ev[k++] = e;
                    vlgp_prof_begin(ctx, VLGP_PROF_HSTEP_TAB);
                    hipLaunchKernelGGL(hstep_lr_tables, dim3(n_lr), dim3(128), 0, ctx->stream, Rt);
                    vlgp_prof_end(ctx, VLGP_PROF_HSTEP_TAB, (double)n_lr);
                }
                HIPCHK(ctx, hipGetLastError());
                ctx->hstat[0] += n_lr;
                for (int e = 0; e < n_eval; ++e) ctx->hstat[1] += rcap[e];
                ctx->hstat[2] += n_rough;
            } else if (mfma) {
                ctx->hstat[2] += n_eval;
            }
            ctx->last_hstep_path = lr ? (mixed ? VLGP_PATH_HSTEP_MIXED : VLGP_PATH_HSTEP_LOWRANK) : VLGP_PATH_HSTEP_DENSE;
            // single rank: the kernel publishes to the host mailbox.  Several ranks: same, then the ranks add
            // their sums on the host (vlgp_hx_allreduce); without the exchange segment the sums go through the
            // device all-reduce and a copy instead
            const bool mailbox = ctx->world == 1 || ctx->hx != nullptr;
            R.host = mailbox ? ctx->d_hres : nullptr;
            const int prof_kind = lr ? VLGP_PROF_HSTEP_LR : VLGP_PROF_HSTEP;
            vlgp_prof_begin(ctx, prof_kind);
            if (lr) {
                // one launch per rank class present in the round (registers and LDS are sized by the class: a smooth
                // latent's segments must not run at the occupancy of a rough one's); the K blocks ride in the first
                constexpr int NW = LR_NW;
                const int NK = TC == 50 ? 7 : 8;
                const int kblk = TC == 50 ? HRoundK<50, NW, true>::KBLK : HRoundK<64, NW, false>::KBLK;
                // ONE launch, compiled for the largest rank of the round.  (Measured: one launch per rank class --
                // registers and LDS sized by the class -- serialises three under-filled launches, 137 us against
                // 97 us for a steady-state round of C3.)
                const int cls_of_max = rmax <= 16 ? 0 : (rmax <= 24 ? 1 : (rmax <= 28 ? 2 : 3));
                bool first = true;
                for (int ci = cls_of_max; ci <= cls_of_max; ++ci) {
                    HRoundArgs Rc = R;
                    Rc.lr_nev = 0;
                    int rm = rmax;
                    for (int e = 0; e < n_eval; ++e)
                        if ((R.lr_mask >> e) & 1u) Rc.lr_ev[Rc.lr_nev++] = e;
                    // longest first: the workgroups of the highest ranks are dispatched before the cheap ones (shorter tail)
                    std::stable_sort(Rc.lr_ev, Rc.lr_ev + Rc.lr_nev, [&](int x, int y) { return rcap[x] > rcap[y]; });
                    if (Rc.lr_nev == 0) continue;
                    Rc.k_blocks = first ? n_eval : 0;
                    // tables in LDS unless leaving them in global memory lets one more workgroup share a CU
                    auto wgs = [&](int doubles) {
                        int n = doubles;
                        if (first && n < kblk) n = kblk;
                        if (n < 2 * 64 * NW) n = 2 * 64 * NW;
                        return (160 * 1024) / ((n * 8 + 80 + 511) & ~511);  // (+ the kernel's static LDS; 512-byte granules)
                    };
                    const int rc_cls = ci == 0 ? 16 : (ci == 1 ? 24 : (ci == 2 ? 28 : 32));
                    const int need_l = lr_geom(rm, 4 * NK, NW, false, rc_cls).total, need_g = lr_geom(rm, 4 * NK, NW, true, rc_cls).total;
                    const bool tabg = ci >= 2 && rm < LR_RCAP && wgs(need_g) > wgs(need_l) && wgs(need_l) < 3;
                    int need = tabg ? need_g : need_l;
                    if (first && need < kblk) need = kblk;
                    if (need < 2 * 64 * NW) need = 2 * 64 * NW;
                    const size_t lds_bytes = (size_t)need * 8;
                    const int grid = Rc.k_blocks + Rc.lr_nev * R.nb;
                    Rc.lds_doubles = need;
                    const bool fuse = sw.fuse_tables && rmax <= 24;
                    if (fuse && TC == 50 && ci == 0) CHK((launch_round_lr<50, 16, false, true>(ctx, Rc, grid, lds_bytes)));
                    else if (fuse && TC == 50 && ci == 1) CHK((launch_round_lr<50, 24, false, true>(ctx, Rc, grid, lds_bytes)));
                    else if (fuse && ci == 0) CHK((launch_round_lr<64, 16, false, true>(ctx, Rc, grid, lds_bytes)));
                    else if (fuse && ci == 1) CHK((launch_round_lr<64, 24, false, true>(ctx, Rc, grid, lds_bytes)));
                    else if (TC == 50) {
                        if (ci == 0) CHK((launch_round_lr<50, 16>(ctx, Rc, grid, lds_bytes)));
                        else if (ci == 1) CHK((launch_round_lr<50, 24>(ctx, Rc, grid, lds_bytes)));
                        else if (ci == 2 && tabg) CHK((launch_round_lr<50, 28, true>(ctx, Rc, grid, lds_bytes)));
                        else if (ci == 2) CHK((launch_round_lr<50, 28>(ctx, Rc, grid, lds_bytes)));
                        else if (tabg) CHK((launch_round_lr<50, 32, true>(ctx, Rc, grid, lds_bytes)));
                        else CHK((launch_round_lr<50, 32>(ctx, Rc, grid, lds_bytes)));
                    } else {
                        if (ci == 0) CHK((launch_round_lr<64, 16>(ctx, Rc, grid, lds_bytes)));
                        else if (ci == 1) CHK((launch_round_lr<64, 24>(ctx, Rc, grid, lds_bytes)));
                        else if (ci == 2 && tabg) CHK((launch_round_lr<64, 28, true>(ctx, Rc, grid, lds_bytes)));
                        else if (ci == 2) CHK((launch_round_lr<64, 28>(ctx, Rc, grid, lds_bytes)));
                        else if (tabg) CHK((launch_round_lr<64, 32, true>(ctx, Rc, grid, lds_bytes)));
                        else CHK((launch_round_lr<64, 32>(ctx, Rc, grid, lds_bytes)));
                    }
                    first = false;
                }
                if (mixed) {  // the rough evaluations: the dense kernel behind it on the stream (no K blocks: the first launch had them)
                    HRoundArgs Rd = R;
                    Rd.nb = nb_dense;
                    Rd.k_blocks = 0;
                    Rd.lr_nev = 0;
                    for (int e = 0; e < n_eval; ++e)
                        if ((rough >> e) & 1u) Rd.lr_ev[Rd.lr_nev++] = e;
                    if (TC == 50)
                        hipLaunchKernelGGL((hstep_round_mfma<50, MFMA_NW, true>), dim3(Rd.lr_nev * nb_dense), dim3(64 * MFMA_NW), 0,
                                           ctx->stream, Rd);
                    else
                        hipLaunchKernelGGL((hstep_round_mfma<64, MFMA_NW>), dim3(Rd.lr_nev * nb_dense), dim3(64 * MFMA_NW), 0,
                                           ctx->stream, Rd);
                }
            } else if (TC == 50)
                hipLaunchKernelGGL((hstep_round_mfma<50, MFMA_NW, true>), dim3(n_eval + n_eval * R.nb), dim3(64 * MFMA_NW), 0,
                                   ctx->stream, R);
            else
                hipLaunchKernelGGL((hstep_round_mfma<64, MFMA_NW>), dim3(n_eval + n_eval * R.nb), dim3(64 * MFMA_NW), 0,
                                   ctx->stream, R);
            vlgp_prof_end(ctx, prof_kind, (double)n_eval * M);
            HIPCHK(ctx, hipGetLastError());
            if (mailbox) {
                volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(ctx->h_hres + 48);
                unsigned spins = 0;
                while (*flag != (unsigned long long)R.seq) {
                    if ((++spins & 0xfff) == 0) {  // a faulted kernel must not hang the host
                        const hipError_t qe = hipStreamQuery(ctx->stream);
                        if (qe == hipSuccess) break;
                        if (qe != hipErrorNotReady)
                            return vlgp_fail(ctx, VLGP_ERR_HIP, "H-step round kernel failed: %s", hipGetErrorString(qe));
                    }
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                for (int i = 0; i < 3 * n_eval; ++i) hres[i] = ctx->h_hres[i];
                if (ctx->world > 1) {
                    // the status flags travel with the sums (as counts: failed factorisations + 1024 x overflows), so that
                    // every rank takes the same fallback
                    for (int e = 0; e < n_eval; ++e) {
                        const double f = hres[2 * n_eval + e];
                        hres[2 * n_eval + e] = f == 1.0 ? 0.0 : (f == 2.0 ? 1024.0 : 1.0);
                    }
                    CHK(vlgp_hx_allreduce(ctx, hres, 3 * n_eval));
                    for (int e = 0; e < n_eval; ++e) {
                        const double c = hres[2 * n_eval + e];
                        hres[2 * n_eval + e] = c == 0.0 ? 1.0 : (fmod(c, 1024.0) != 0.0 ? 0.0 : 2.0);
                    }
                }
            } else {
                CHK(vlgp_allreduce(ctx, W + o_red, 2LL * n_eval));
                HIPCHK(ctx, hipMemcpyAsync(hres, W + o_red, sizeof(double) * 3 * n_eval, hipMemcpyDeviceToHost, ctx->stream));
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            }
        }
        bool all_ok = true, lr_over = false;
        for (int e = 0; e < n_eval; ++e) {
            all_ok = all_ok && hres[2 * n_eval + e] != 0.0;
            lr_over = lr_over || hres[2 * n_eval + e] == 2.0;
        }
        // a rank beyond the host's prediction (never seen; the prediction runs at half the tolerance): dense round
        if (all_ok && lr_over) ctx->hstat[3] += 1.0;
        if (all_ok && lr_over) return launch_hstep_impl(ctx, us, window, dt, n_eval, latent, logp, ll, dll, true);
        if (all_ok) {
            for (int e = 0; e < n_eval; ++e) {
                ll[e] = hres[2 * e + 0];
                dll[3 * e + 0] = 0.0;
                dll[3 * e + 1] = hres[2 * e + 1];
                dll[3 * e + 2] = 0.0;
            }
            return VLGP_OK;
        }
        // K did not factor for some evaluation: fall through to the generic path,
        // which implements the reference's omega bump
        if (sw.debug_occ) fprintf(stderr, "hstep: K failed to factor in a round of %d evaluations -> generic kernels\n", n_eval);
    }
    // generic path: its sums go through the main communicator; with several ranks the M-step lane's
    // collectives must not be in flight at the same time (two communicators, no common order)
    if (ctx->world > 1) CHK(vlgp_join_m(ctx));
    for (int i = 0; i < 3 * n_eval; ++i) hp[i] = logp[i];
    int* hlat = reinterpret_cast<int*>(hp + 3 * n_eval);
    for (int i = 0; i < n_eval; ++i) hlat[i] = latent[i];
    HIPCHK(ctx, hipMemcpyAsync(W + o_logp, hp, sizeof(double) * 3 * n_eval, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(W + o_lat, hlat, sizeof(int) * n_eval, hipMemcpyHostToDevice, ctx->stream));
    HPrepArgs P;
    P.T = T; P.dt = dt; P.logp = W + o_logp; P.kinv = W + o_kinv; P.q = W + o_q; P.dk = W + o_dk;
    P.scal = W + o_scal;
    size_t lds_prep = (size_t)(T * (T | 1) + 3 * TT) * 8;
    P.tm = nullptr;
    P.km = nullptr;
    if (lds_prep > 150 * 1024) {  // large window: only the factor stays in LDS
        P.tm = W + o_tm;
        lds_prep = (size_t)(T * (T | 1)) * 8;
    }
    if (huge) {  // ... and beyond ~138 bins it works in global memory as well
        P.km = W + o_km;
        lds_prep = 0;
    }
    if (lds_prep > 64 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(hstep_prep_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_prep));
    const bool big = T > 64 && T <= 128 && !sw.generic_seg;  // hstep_prep_big / hstep_seg_big
    ctx->last_hstep_path = big ? VLGP_PATH_HSTEP_BIG : VLGP_PATH_HSTEP_GENERIC;
    if (big) {
        const size_t lds_pb = (size_t)(4 * 64 * 66 + HmGeom<64>::TASK + 2 * 128) * 8;
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(hstep_prep_big),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pb));
        hipLaunchKernelGGL(hstep_prep_big, dim3(n_eval), dim3(256), lds_pb, ctx->stream, P);
    } else {
        hipLaunchKernelGGL(hstep_prep_kernel, dim3(n_eval), dim3(256), lds_prep, ctx->stream, P);
    }
    HIPCHK(ctx, hipGetLastError());

    HSegArgs S;
    S.T = T; S.L = L; S.M = M; S.off = us.d_off; S.mu = us.mu; S.w = us.w; S.bmat = nullptr;
    S.latent = reinterpret_cast<const int*>(W + o_lat);
    S.kinv = W + o_kinv; S.q = W + o_q; S.dk = W + o_dk; S.scal = W + o_scal; S.out = W + o_out;
    if (big) {  // one workgroup per segment, matrix pipe (hstep_seg_big)
        HBigArgs B;
        B.S = S; B.logp = W + o_logp; B.dt = dt;
        const size_t lds_big = (size_t)(3 * 64 * 66 + HmGeom<64>::TASK + 5 * 128) * 8;
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(hstep_seg_big),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_big));
        vlgp_prof_begin(ctx, VLGP_PROF_HSTEP);
        hipLaunchKernelGGL(hstep_seg_big, dim3(M, n_eval), dim3(256), lds_big, ctx->stream, B);
        vlgp_prof_end(ctx, VLGP_PROF_HSTEP, (double)n_eval * M);
        HIPCHK(ctx, hipGetLastError());
    } else {
    S.bmat = huge ? W + o_bmat : nullptr;
    const int nw = huge ? 4 : ((size_t)2 * (T * (T | 1) + 2 * T) * 8 <= 160 * 1024 ? 2 : 1);
    const size_t lds_seg = huge ? (size_t)nw * 2 * T * 8 : (size_t)nw * (T * (T | 1) + 2 * T) * 8;
    if (lds_seg > 64 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(hstep_seg_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_seg));
    vlgp_prof_begin(ctx, VLGP_PROF_HSTEP);
    hipLaunchKernelGGL(hstep_seg_kernel, dim3((M + nw - 1) / nw, n_eval), dim3(64 * nw), lds_seg, ctx->stream, S);
    vlgp_prof_end(ctx, VLGP_PROF_HSTEP, (double)n_eval * M);
    HIPCHK(ctx, hipGetLastError());
    }
    hipLaunchKernelGGL(hstep_reduce_kernel, dim3(n_eval), dim3(256), 0, ctx->stream, M, W + o_out, W + o_red);
    HIPCHK(ctx, hipGetLastError());
    CHK(vlgp_allreduce(ctx, W + o_red, 2LL * n_eval));
    double* hres = hp + 4 * n_eval + 8;
    HIPCHK(ctx, hipMemcpyAsync(hres, W + o_red, sizeof(double) * 2 * n_eval, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    for (int e = 0; e < n_eval; ++e) {
        ll[e] = hres[2 * e + 0];
        dll[3 * e + 0] = 0.0;
        dll[3 * e + 1] = hres[2 * e + 1];
        dll[3 * e + 2] = 0.0;
    }
    return VLGP_OK;
}
