// Prior factor on the device: pivoted incomplete Cholesky of the squared-
// exponential kernel (math.ichol_gauss, vlgp/math.py:76-126) for every latent
// of one unit length, plus the compaction that drops the trailing zero columns.
//
// One workgroup per latent.  Each of the (at most R) steps needs the arg-max
// of the residual diagonal and its sum (two block reductions), then one new
// column; the residual diagonal is recomputed from scratch each step exactly as
// the reference does (math.py:119).  Ties in the arg-max go to the lowest row,
// as numpy.argmax does; the last bit of the residuals may differ from NumPy's
// (different exp/summation order), see DESIGN.md "pivot chaos".
#include "ctx.h"

#define ICH_THREADS 256

// work: per latent [ Gp (T*R) | d (T) ] doubles, piv: per latent T ints
__global__ void __launch_bounds__(ICH_THREADS)
ichol_kernel(int T, int R, const double* omega, const double* sigma, double* work, int* piv,
             double* G_out, int* rank_out) {
    __shared__ double red_v[ICH_THREADS];
    __shared__ int red_i[ICH_THREADS];
    __shared__ double s_piv;
    __shared__ int s_jast;
    const int l = blockIdx.x, tid = threadIdx.x;
    double* Gp = work + (int64_t)l * ((int64_t)T * R + T);
    double* d = Gp + (int64_t)T * R;
    int* pv = piv + (int64_t)l * T;
    const double om = omega[l];
    const double tol_n = 1e-6 * T;

    for (int j = tid; j < T; j += ICH_THREADS) {
        d[j] = 1.0;
        pv[j] = j;
        for (int c = 0; c < R; ++c) Gp[(int64_t)j * R + c] = 0.0;
    }
    __syncthreads();

    int k = 0;
    for (; k < R; ++k) {
        // sum and arg-max of d[k:]
        double s = 0.0, best = -1.0;
        int bi = 0x7fffffff;
        for (int j = k + tid; j < T; j += ICH_THREADS) {
            const double dj = d[j];
            s += dj;
            if (dj > best) { best = dj; bi = j; }
        }
        red_v[tid] = s;
        __syncthreads();
        for (int o = ICH_THREADS / 2; o > 0; o >>= 1) {
            if (tid < o) red_v[tid] += red_v[tid + o];
            __syncthreads();
        }
        const double total = red_v[0];
        __syncthreads();
        if (!(total > tol_n)) break;
        red_v[tid] = best;
        red_i[tid] = bi;
        __syncthreads();
        for (int o = ICH_THREADS / 2; o > 0; o >>= 1) {
            if (tid < o) {
                const double ov = red_v[tid + o];
                const int oi = red_i[tid + o];
                if (ov > red_v[tid] || (ov == red_v[tid] && oi < red_i[tid])) {
                    red_v[tid] = ov;
                    red_i[tid] = oi;
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            const int jast = (k == 0) ? 0 : red_i[0];  // first pivot is row 0 (math.py:112)
            s_jast = jast;
            s_piv = sqrt(d[jast]);
        }
        __syncthreads();
        const int jast = s_jast;
        const double pivot = s_piv;
        if (jast != k) {  // swap rows k <-> jast (columns 0..k-1) and the pivot vector
            for (int c = tid; c < k; c += ICH_THREADS) {
                const double tmp = Gp[(int64_t)k * R + c];
                Gp[(int64_t)k * R + c] = Gp[(int64_t)jast * R + c];
                Gp[(int64_t)jast * R + c] = tmp;
            }
            if (tid == 0) {
                const int tmp = pv[k];
                pv[k] = pv[jast];
                pv[jast] = tmp;
            }
        }
        __syncthreads();
        const double xk = (double)pv[k];
        if (tid == 0) Gp[(int64_t)k * R + k] = pivot;
        for (int j = k + 1 + tid; j < T; j += ICH_THREADS) {
            double* row = Gp + (int64_t)j * R;
            const double* rk = Gp + (int64_t)k * R;
            const double dx = (double)pv[j] - xk;
            double dot = 0.0;
            for (int c = 0; c < k; ++c) dot += row[c] * rk[c];
            const double g = (exp(-om * (dx * dx)) - dot) / pivot;
            row[k] = g;
            double ss = 0.0;
            for (int c = 0; c < k; ++c) ss += row[c] * row[c];
            ss += g * g;
            d[j] = 1.0 - ss;
        }
        __syncthreads();
    }
    // un-pivot, scale by sigma: G[l, pv[j], :] = Gp[j, :] * sigma_l
    const double sg = sigma[l];
    for (int i = tid; i < T * R; i += ICH_THREADS) {
        const int j = i / R, c = i - j * R;
        G_out[((int64_t)l * T + pv[j]) * R + c] = Gp[i] * sg;
    }
    if (tid == 0) rank_out[l] = k;
}

// number of leading columns up to the last non-zero one, per latent
__global__ void __launch_bounds__(256) prior_rank_kernel(int T, int R, const double* G, int* rank_out) {
    __shared__ int s_r;
    const int l = blockIdx.x;
    if (threadIdx.x == 0) s_r = 0;
    __syncthreads();
    const double* Gl = G + (int64_t)l * T * R;
    int mine = 0;
    for (int i = threadIdx.x; i < T * R; i += 256) {
        if (Gl[i] != 0.0) {
            const int c = i % R + 1;
            mine = c > mine ? c : mine;
        }
    }
    atomicMax(&s_r, mine);
    __syncthreads();
    if (threadIdx.x == 0) rank_out[l] = s_r < 1 ? 1 : s_r;
}

__global__ void __launch_bounds__(256)
prior_compact_kernel(int T, int R, int L, const double* G, const int* rl, const int64_t* goff,
                     double* out) {
    const int l = blockIdx.y;
    const int r = rl[l];
    const int64_t n = (int64_t)T * r;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / r;
        const int c = (int)(i - t * r);
        out[goff[l] + i] = G[((int64_t)l * T + t) * R + c];
    }
}

int launch_ichol(vlgp_ctx* ctx, Prior& pr, const double* d_omega, const double* d_sigma) {
    const int L = ctx->L, R = ctx->R, T = pr.T;
    const int64_t per = (int64_t)T * R + T;
    // workspace: doubles for (Gp, d) per latent, then ints for pivots and ranks
    const int64_t n_d = per * L + ((int64_t)L * T + L + 1) / 2 + 1;
    CHK(vlgp_ensure_work(ctx, n_d));
    double* work = ctx->d_work;
    int* piv = reinterpret_cast<int*>(work + per * L);
    int* rank = piv + (int64_t)L * T;
    vlgp_prof_begin(ctx, VLGP_PROF_PRIOR);
    hipLaunchKernelGGL(ichol_kernel, dim3(L), dim3(ICH_THREADS), 0, ctx->stream, T, R, d_omega,
                       d_sigma, work, piv, pr.d_full, rank);
    vlgp_prof_end(ctx, VLGP_PROF_PRIOR, (double)L);
    HIPCHK(ctx, hipGetLastError());
    return launch_compact_prior(ctx, pr);
}

int launch_compact_prior(vlgp_ctx* ctx, Prior& pr) {
    const int L = ctx->L, R = ctx->R, T = pr.T;
    CHK(vlgp_ensure_work(ctx, 4 * L + 8));
    CHK(vlgp_ensure_pinned(ctx, 4 * L + 8));
    int* d_rank = reinterpret_cast<int*>(ctx->d_work);
    hipLaunchKernelGGL(prior_rank_kernel, dim3(L), dim3(256), 0, ctx->stream, T, R, pr.d_full, d_rank);
    HIPCHK(ctx, hipGetLastError());
    int* h_rank = reinterpret_cast<int*>(ctx->h_pinned);
    HIPCHK(ctx, hipMemcpyAsync(h_rank, d_rank, sizeof(int) * L, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    pr.rl.assign(h_rank, h_rank + L);
    pr.goff.resize(L);
    int64_t off = 0;
    for (int l = 0; l < L; ++l) {
        pr.goff[l] = off;
        off += (int64_t)T * pr.rl[l];
    }
    pr.compact_len = off;  // d_compact was allocated at full (L, T, R) capacity
    // device copies of rl / goff for the compaction kernel (reuse workspace)
    int64_t* h_goff = reinterpret_cast<int64_t*>(ctx->h_pinned) + L;  // after the ints
    for (int l = 0; l < L; ++l) h_goff[l] = pr.goff[l];
    int64_t* d_goff = reinterpret_cast<int64_t*>(ctx->d_work) + L;
    HIPCHK(ctx, hipMemcpyAsync(d_goff, h_goff, sizeof(int64_t) * L, hipMemcpyHostToDevice, ctx->stream));
    const int gx = (int)((((int64_t)T * R) + 255) / 256);
    hipLaunchKernelGGL(prior_compact_kernel, dim3(gx > 64 ? 64 : gx, L), dim3(256), 0, ctx->stream, T, R, L,
                       pr.d_full, d_rank, d_goff, pr.d_compact);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return VLGP_OK;
}
