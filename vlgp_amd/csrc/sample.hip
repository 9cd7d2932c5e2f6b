// api.sample_posterior (vlgp/api.py:142-168) on the device: draws from the variational posterior of one
// trial, independent Gaussians per latent with mean mu_l and covariance (K_l^-1 + W_l)^-1, K_l = G_l G_l'.
// The reference forms the T x T matrices (two dense inverses per latent plus multivariate_normal's SVD,
// O(T^3)); with H = G'WG (r x r) the same covariance is G (I + H)^-1 G' (the reference's `reg` regulariser,
// which only exists to make K invertible, set to zero), so a draw is mu + G L^-T eps with L L' = I + H and
// eps ~ N(0, I_r): O(T r^2 + n T r) per latent.  eps comes from the host (the caller's NumPy generator).
//
// One workgroup per latent.  r = number of leading non-zero columns of G_l (ichol_gauss stops early).
#include "ctx.h"

#define SP_THREADS 256

__global__ void __launch_bounds__(SP_THREADS)
sample_posterior_kernel(int T, int L, int R, int n, const double* mu, const double* w, const double* G,
                        const double* eps, double* z, double* out, int* fail) {
    extern __shared__ __attribute__((aligned(16))) double sp_smem[];
    __shared__ int s_r;
    __shared__ double s_piv;
    const int l = blockIdx.x, tid = threadIdx.x;
    const double* Gl = G + (int64_t)l * T * R;
    double* Hm = sp_smem;  // R x R, lower triangle holds the Cholesky factor afterwards
    if (tid == 0) s_r = 0;
    __syncthreads();
    {
        int mine = 0;
        for (int i = tid; i < T * R; i += SP_THREADS)
            if (Gl[i] != 0.0) {
                const int c = i % R + 1;
                mine = c > mine ? c : mine;
            }
        atomicMax(&s_r, mine);
    }
    __syncthreads();
    const int r = s_r < 1 ? 1 : s_r;
    // H = I + G' diag(w) G
    for (int e = tid; e < r * r; e += SP_THREADS) {
        const int i = e / r, j = e - i * r;
        double s = i == j ? 1.0 : 0.0;
        if (j <= i)
            for (int t = 0; t < T; ++t) s = fma(w[(int64_t)t * L + l] * Gl[(int64_t)t * R + i], Gl[(int64_t)t * R + j], s);
        Hm[i * R + j] = s;
    }
    __syncthreads();
    // Cholesky (right-looking), lower triangle in place
    bool ok = true;
    for (int k = 0; k < r; ++k) {
        if (tid == 0) {
            const double d = Hm[k * R + k];
            s_piv = (d > 0.0 && d < 1e300) ? sqrt(d) : -1.0;
        }
        __syncthreads();
        const double pv = s_piv;
        if (!(pv > 0.0)) { ok = false; break; }
        for (int i = k + tid; i < r; i += SP_THREADS) Hm[i * R + k] = i == k ? pv : Hm[i * R + k] / pv;
        __syncthreads();
        const int m = r - k - 1;
        for (int e = tid; e < m * m; e += SP_THREADS) {
            const int i = k + 1 + e / m, j = k + 1 + e % m;
            if (j <= i) Hm[i * R + j] = fma(-Hm[i * R + k], Hm[j * R + k], Hm[i * R + j]);
        }
        __syncthreads();
    }
    if (!ok) {  // cannot happen for w >= 0; draws collapse to the mean, counted like every other failed factor
        if (tid == 0) atomicAdd(fail, 1);
        for (int64_t e = tid; e < (int64_t)n * T; e += SP_THREADS) {
            const int64_t s = e / T;
            const int t = (int)(e - s * T);
            out[(s * T + t) * L + l] = mu[(int64_t)t * L + l];
        }
        return;
    }
    // z_s = L^-T eps_s (back substitution), one sample per thread; eps, z: (L, R, n) with the sample index contiguous
    const double* el = eps + (int64_t)l * R * n;
    double* zl = z + (int64_t)l * R * n;
    for (int s = tid; s < n; s += SP_THREADS) {
        for (int i = r - 1; i >= 0; --i) {
            double acc = el[(int64_t)i * n + s];
            for (int j = i + 1; j < r; ++j) acc = fma(-Hm[j * R + i], zl[(int64_t)j * n + s], acc);
            zl[(int64_t)i * n + s] = acc / Hm[i * R + i];
        }
    }
    __syncthreads();
    // out[s, t, l] = mu[t, l] + G[t, :] . z_s
    for (int64_t e = tid; e < (int64_t)n * T; e += SP_THREADS) {
        const int t = (int)(e / n);
        const int64_t s = e - (int64_t)t * n;
        double acc = mu[(int64_t)t * L + l];
        for (int c = 0; c < r; ++c) acc = fma(Gl[(int64_t)t * R + c], zl[(int64_t)c * n + s], acc);
        out[(s * T + t) * L + l] = acc;
    }
}

int launch_sample_posterior(vlgp_ctx* ctx, int T, int n, const double* d_mu, const double* d_w, const double* d_G,
                            const double* d_eps, double* d_z, double* d_out) {
    const int L = ctx->L, R = ctx->R;
    hipLaunchKernelGGL(sample_posterior_kernel, dim3(L), dim3(SP_THREADS), (size_t)R * R * 8, ctx->stream, T, L, R, n, d_mu,
                       d_w, d_G, d_eps, d_z, d_out, ctx->d_fail);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}
