// Small streaming kernels around the hot path: x.b precompute, latent affine
// maps (constrain_loading / constrain_latent), segment gather / scatter.
#include "ctx.h"

// xb[t, n] = sum_p x[t, p, n] * b[p, n]    (core.py:66, einsum "ijk,jk->ik")
__global__ void __launch_bounds__(256)
xb_kernel(int64_t rows, int N, int P, const double* x, const double* b, double* xb) {
    const int64_t total = rows * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / N;
        const int n = (int)(i - t * N);
        double s = 0.0;
        for (int p = 0; p < P; ++p) s = fma(x[(t * P + p) * N + n], b[p * N + n], s);
        xb[i] = s;
    }
}

// mu[t, :] <- (mu[t, :] - shift) @ map        (map is L x L row-major)
__global__ void __launch_bounds__(256)
latent_map_kernel(int64_t rows, int L, const double* map, const double* shift, double* mu) {
    extern __shared__ double sm[];
    double* m_s = sm;          // L*L
    double* s_s = sm + L * L;  // L
    for (int i = threadIdx.x; i < L * L; i += 256) m_s[i] = map[i];
    for (int i = threadIdx.x; i < L; i += 256) s_s[i] = shift ? shift[i] : 0.0;
    __syncthreads();
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < rows; t += (int64_t)gridDim.x * 256) {
        double in[16], out[16];
        for (int l = 0; l < L; ++l) in[l] = mu[t * L + l] - s_s[l];
        for (int c = 0; c < L; ++c) {
            double s = 0.0;
            for (int l = 0; l < L; ++l) s = fma(in[l], m_s[l * L + c], s);
            out[c] = s;
        }
        for (int c = 0; c < L; ++c) mu[t * L + c] = out[c];
    }
}

// dst row (k*window + r) <- src row (start[k] + r), `width` doubles per row
__global__ void __launch_bounds__(256)
gather_rows_kernel(int M, int window, int64_t width, const int64_t* start, const double* src, double* dst) {
    const int64_t per = (int64_t)window * width;
    const int64_t total = (int64_t)M * per;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t k = i / per, rem = i - k * per;
        dst[i] = src[start[k] * width + rem];
    }
}

// inverse of the gather for one unit at a time (launch order = unit order, so
// with overlapping segments the later unit wins, as sequential NumPy views do)
__global__ void __launch_bounds__(256)
scatter_unit_kernel(int k, int window, int64_t width, const int64_t* start, const double* src, double* dst) {
    const int64_t per = (int64_t)window * width;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < per; i += (int64_t)gridDim.x * 256)
        dst[start[k] * width + i] = src[(int64_t)k * per + i];
}
__global__ void __launch_bounds__(256)
scatter_rows_kernel(int M, int window, int64_t width, const int64_t* start, const double* src, double* dst) {
    const int64_t per = (int64_t)window * width;
    const int64_t total = (int64_t)M * per;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t k = i / per, rem = i - k * per;
        dst[start[k] * width + rem] = src[i];
    }
}

static inline int grid_for(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

int launch_xb(vlgp_ctx* ctx, UnitSet& us) {
    hipLaunchKernelGGL(xb_kernel, dim3(grid_for(us.rows * ctx->N)), dim3(256), 0, ctx->stream, us.rows, ctx->N,
                       ctx->P, us.x, ctx->d_b, us.d_xb);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

int launch_latent_map(vlgp_ctx* ctx, UnitSet& us, const double* d_map, const double* d_shift) {
    const int L = ctx->L;
    if (L > 16) return vlgp_fail(ctx, VLGP_ERR_ARG, "latent map supports at most 16 latents");
    hipLaunchKernelGGL(latent_map_kernel, dim3(grid_for(us.rows)), dim3(256), (size_t)(L * L + L) * 8, ctx->stream,
                       us.rows, L, d_map, d_shift, us.mu);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

int launch_gather(vlgp_ctx* ctx, UnitSet& src, UnitSet& dst, int window) {
    const int N = ctx->N, L = ctx->L, P = ctx->P;
    const int M = dst.M;
    auto go = [&](const double* s, double* d, int64_t width) {
        hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for((int64_t)M * window * width)), dim3(256), 0,
                           ctx->stream, M, window, width, dst.d_src_start, s, d);
    };
    go(src.y, dst.y, N);
    if (!src.x_ones) go(src.x, dst.x, (int64_t)P * N);
    go(src.mu, dst.mu, L);
    go(src.v, dst.v, L);
    go(src.w, dst.w, L);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

int launch_scatter(vlgp_ctx* ctx, UnitSet& cut, UnitSet& dst, int window) {
    const int L = ctx->L;
    // detect overlap on the host copy of the starts: disjoint -> one launch
    bool disjoint = true;
    for (size_t k = 1; k < cut.src_start.size(); ++k)
        if (cut.src_start[k] < cut.src_start[k - 1] + window) { disjoint = false; break; }
    if (disjoint) {
        hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((int64_t)cut.M * window * L)), dim3(256), 0,
                           ctx->stream, cut.M, window, (int64_t)L, cut.d_src_start, cut.mu, dst.mu);
        hipLaunchKernelGGL(scatter_rows_kernel, dim3(grid_for((int64_t)cut.M * window * L)), dim3(256), 0,
                           ctx->stream, cut.M, window, (int64_t)L, cut.d_src_start, cut.v, dst.v);
    } else {
        for (int k = 0; k < cut.M; ++k) {
            hipLaunchKernelGGL(scatter_unit_kernel, dim3(1), dim3(256), 0, ctx->stream, k, window, (int64_t)L,
                               cut.d_src_start, cut.mu, dst.mu);
            hipLaunchKernelGGL(scatter_unit_kernel, dim3(1), dim3(256), 0, ctx->stream, k, window, (int64_t)L,
                               cut.d_src_start, cut.v, dst.v);
        }
    }
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}
