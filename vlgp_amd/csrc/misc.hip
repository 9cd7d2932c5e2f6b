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
        double in[VLGP_MAX_L], out[VLGP_MAX_L];
        for (int l = 0; l < L; ++l) in[l] = mu[t * L + l] - s_s[l];
        for (int c = 0; c < L; ++c) {
            double s = 0.0;
            for (int l = 0; l < L; ++l) s = fma(in[l], m_s[l * L + c], s);
            out[c] = s;
        }
        for (int c = 0; c < L; ++c) mu[t * L + c] = out[c];
    }
}

// The same for L <= LT with the row in registers: the loop form above indexes two VLGP_MAX_L-sized local arrays, 1040 bytes of
// scratch per lane, for a kernel that every EM iteration launches (constrain_loading, vlgp/core.py:392-416) -- a scratch
// frame of that size is allocated by the runtime per dispatch (above its single-allocation limit), DESIGN.md section 4.1.
template <int LT>
__global__ void __launch_bounds__(256)
latent_map_kernel_t(int64_t rows, int L, const double* map, const double* shift, double* mu) {
    extern __shared__ double sm[];
    double* m_s = sm;          // L*L
    double* s_s = sm + L * L;  // L
    for (int i = threadIdx.x; i < L * L; i += 256) m_s[i] = map[i];
    for (int i = threadIdx.x; i < L; i += 256) s_s[i] = shift ? shift[i] : 0.0;
    __syncthreads();
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < rows; t += (int64_t)gridDim.x * 256) {
        double in[LT];
#pragma unroll
        for (int l = 0; l < LT; ++l) in[l] = l < L ? mu[t * L + l] - s_s[l] : 0.0;
#pragma unroll
        for (int c = 0; c < LT; ++c) {
            if (c < L) {
                double s = 0.0;
#pragma unroll
                for (int l = 0; l < LT; ++l)
                    if (l < L) s = fma(in[l], m_s[l * L + c], s);  // (same order of the sum as the loop form)
                mu[t * L + c] = s;
            }
        }
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

// ---- parameter snapshot of the M-step lane: a | b | noise | da | db | failure count into mapped host memory, one launch
// (six device-to-host copies of a few KB each are six 5 us steps at the end of the lane) ----
__global__ void __launch_bounds__(256)
snapshot_params_kernel(int na, int nb, int nn, const double* __restrict__ a, const double* __restrict__ b,
                       const double* __restrict__ noise, const double* __restrict__ da, const double* __restrict__ db,
                       const int* __restrict__ fail, double* __restrict__ host) {
    const int total = 2 * na + 2 * nb + nn;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        double v;
        if (i < na) v = a[i];
        else if (i < na + nb) v = b[i - na];
        else if (i < na + nb + nn) v = noise[i - na - nb];
        else if (i < 2 * na + nb + nn) v = da[i - na - nb - nn];
        else v = db[i - 2 * na - nb - nn];
        host[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<int*>(host + total) = *fail;
}
int launch_snapshot_params(vlgp_ctx* ctx, hipStream_t st, double* d_host) {
    const int na = ctx->L * ctx->N, nb = ctx->P * ctx->N, nn = ctx->N;
    const int total = 2 * na + 2 * nb + nn;
    int g = (total + 255) / 256;
    if (g > 64) g = 64;
    hipLaunchKernelGGL(snapshot_params_kernel, dim3(g), dim3(256), 0, st, na, nb, nn, ctx->d_a, ctx->d_b, ctx->d_noise, ctx->d_da,
                       ctx->d_db, ctx->d_fail_m, d_host);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

// ---- |mu|^2 and |dmu|^2 over the set (the stopping rule of core.vem, vlgp/core.py:300-305,350-354) ----
// One launch on the main stream: per-block sums in a fixed order, the block that draws the last ticket adds the
// partials (fixed tree) and publishes to mapped host memory with a sequence word -- no copy, no second kernel, nothing
// the host has to wait for when it queues it behind an E-step.
__global__ void __launch_bounds__(256)
norms_kernel(int64_t n, const double* __restrict__ mu, const double* __restrict__ dmu, double* part, unsigned* ticket,
             double* host, unsigned long long seq) {
    __shared__ double red[2][256];
    __shared__ int s_last;
    const int tid = threadIdx.x;
    double s0 = 0.0, s1 = 0.0;
    {   // four strided elements per step: eight loads in flight per lane (one at a time: 19 us for 16 MB, bound by latency)
        const int64_t st = (int64_t)gridDim.x * 256;
        for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < n; i += 4 * st) {
            double m[4], d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t j = i + q * st;
                m[q] = j < n ? mu[j] : 0.0;
                d[q] = (dmu && j < n) ? dmu[j] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s0 = fma(m[q], m[q], s0);
                s1 = fma(d[q], d[q], s1);
            }
        }
    }
    red[0][tid] = s0; red[1][tid] = s1;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        part[2 * blockIdx.x] = red[0][0];
        part[2 * blockIdx.x + 1] = red[1][0];
        __threadfence();
        s_last = atomicAdd(ticket, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    {
        double p0 = 0.0, p1 = 0.0;
        for (int b = tid; b < (int)gridDim.x; b += 256) {  // (fixed order: block b of every 256)
            p0 += __hip_atomic_load(part + 2 * b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p1 += __hip_atomic_load(part + 2 * b + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        red[0][tid] = p0;
        red[1][tid] = p1;
    }
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; }
        __syncthreads();
    }
    if (tid == 0) {
        *ticket = 0u;
        host[0] = red[0][0];
        host[1] = red[1][0];
        __threadfence_system();
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(host + 2), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int launch_norms(vlgp_ctx* ctx, UnitSet& us, double* d_part, unsigned* d_ticket, double* d_host, unsigned long long seq) {
    const int64_t n = us.rows * ctx->L;
    // (at most 256 blocks: every block ends with an atomic on ONE ticket word, ~25 ns each -- 977 blocks: 31 us)
    static const int gmax = getenv("VLGP_NORMS_BLOCKS") ? atoi(getenv("VLGP_NORMS_BLOCKS")) : 256;
    int g = (int)((n + 1023) / 1024);
    if (g > gmax) g = gmax;
    if (g < 1) g = 1;
    hipLaunchKernelGGL(norms_kernel, dim3(g), dim3(256), 0, ctx->stream, n, us.mu, us.dmu, d_part, d_ticket, d_host, seq);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

int launch_latent_map(vlgp_ctx* ctx, UnitSet& us, const double* d_map, const double* d_shift) {
    const int L = ctx->L;
    const dim3 grid(grid_for(us.rows)), blk(256);
    const size_t lds = (size_t)(L * L + L) * 8;
    if (L <= 5) hipLaunchKernelGGL((latent_map_kernel_t<5>), grid, blk, lds, ctx->stream, us.rows, L, d_map, d_shift, us.mu);
    else if (L <= 10) hipLaunchKernelGGL((latent_map_kernel_t<10>), grid, blk, lds, ctx->stream, us.rows, L, d_map, d_shift, us.mu);
    else if (L <= 16) hipLaunchKernelGGL((latent_map_kernel_t<16>), grid, blk, lds, ctx->stream, us.rows, L, d_map, d_shift, us.mu);
    else hipLaunchKernelGGL(latent_map_kernel, grid, blk, lds, ctx->stream, us.rows, L, d_map, d_shift, us.mu);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

// ---- shared rows of overlapping segments (UnitSet::d_links) ----
__global__ void __launch_bounds__(256)
links_copy_kernel(const int* links, int l0, int l1, int window, int L, int dir, int do_mu, double* mu, double* v) {
    for (int k = l0 + blockIdx.x; k < l1; k += gridDim.x) {
        const int ua = links[3 * k], ub = links[3 * k + 1], o = links[3 * k + 2];
        const int64_t ra = ((int64_t)ua * window + window - o) * L, rb = (int64_t)ub * window * L;  // tail of a, head of b
        for (int i = threadIdx.x; i < o * L; i += 256) {
            if (dir == 0) {
                if (do_mu) mu[rb + i] = mu[ra + i];
                v[rb + i] = v[ra + i];
            } else {
                if (do_mu) mu[ra + i] = mu[rb + i];
                v[ra + i] = v[rb + i];
            }
        }
    }
}

// the latent map once more on both copies of every shared row (an in-place constraint of the reference visits a shared
// row once per segment that holds it: vlgp/core.py:374-388,413-416)
__global__ void __launch_bounds__(256)
links_map_kernel(const int* links, int n_links, int window, int L, const double* map, const double* shift, double* mu) {
    extern __shared__ double sm[];
    double* m_s = sm;
    double* s_s = sm + L * L;
    for (int i = threadIdx.x; i < L * L; i += 256) m_s[i] = map[i];
    for (int i = threadIdx.x; i < L; i += 256) s_s[i] = shift ? shift[i] : 0.0;
    __syncthreads();
    for (int k = blockIdx.x; k < n_links; k += gridDim.x) {
        const int ua = links[3 * k], ub = links[3 * k + 1], o = links[3 * k + 2];
        for (int t = threadIdx.x; t < 2 * o; t += 256) {
            const int64_t row = t < o ? (int64_t)ua * window + window - o + t : (int64_t)ub * window + (t - o);
            double in[VLGP_MAX_L], out[VLGP_MAX_L];
            for (int l = 0; l < L; ++l) in[l] = mu[row * L + l] - s_s[l];
            for (int c = 0; c < L; ++c) {
                double acc = 0.0;
                for (int l = 0; l < L; ++l) acc = fma(in[l], m_s[l * L + c], acc);
                out[c] = acc;
            }
            for (int c = 0; c < L; ++c) mu[row * L + c] = out[c];
        }
    }
}

int launch_links_copy(vlgp_ctx* ctx, UnitSet& us, int l0, int l1, int dir) {
    if (l1 <= l0) return VLGP_OK;
    const int window = us.Tmax;
    int g = l1 - l0;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(links_copy_kernel, dim3(g), dim3(256), 0, ctx->stream, us.d_links, l0, l1, window, ctx->L, dir,
                       us.share_mu ? 1 : 0, us.mu, us.v);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}

int launch_links_map(vlgp_ctx* ctx, UnitSet& us, const double* d_map, const double* d_shift) {
    if (us.n_links < 1 || !us.share_mu) return VLGP_OK;
    const int L = ctx->L;
    int g = us.n_links > 1024 ? 1024 : us.n_links;
    hipLaunchKernelGGL(links_map_kernel, dim3(g), dim3(256), (size_t)(L * L + L) * 8, ctx->stream, us.d_links, us.n_links,
                       us.Tmax, L, d_map, d_shift, us.mu);
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

// ---------------------------------------------------------------------------
// Initial latents (preprocess.initialize, vlgp/preprocess.py:30-41): mu = y P - shift with P the
// (N, L) posterior-mean map of the factor-analysis fit, plus the column sums of y (for b =
// log mean y).  One workgroup per 64 rows: the y tile goes through LDS once (coalesced), thread
// (row, latent) takes a dot product, thread n a column sum; per-workgroup column sums are added
// in a fixed order by project_colsum_kernel.
// ---------------------------------------------------------------------------
#define PROJ_ROWS 64
__global__ void __launch_bounds__(512)
project_kernel(int N, int L, int64_t rows, const double* y, const double* proj, const double* shift, double* mu,
               double* part) {
    extern __shared__ double ys[];  // PROJ_ROWS x (N | 1)
    const int ld = N | 1;
    const int64_t r0 = (int64_t)blockIdx.x * PROJ_ROWS;
    const int nr = (int)((rows - r0) < PROJ_ROWS ? (rows - r0) : PROJ_ROWS);
    for (int i = threadIdx.x; i < nr * N; i += blockDim.x) {
        const int r = i / N, n = i - r * N;
        ys[r * ld + n] = y[r0 * N + i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nr * L; i += blockDim.x) {
        const int r = i % nr, l = i / nr;  // consecutive threads take consecutive rows: conflict-free LDS (odd ld)
        double s0 = 0.0, s1 = 0.0;
        int n = 0;
        for (; n + 1 < N; n += 2) {
            s0 = fma(ys[r * ld + n], proj[n * L + l], s0);
            s1 = fma(ys[r * ld + n + 1], proj[(n + 1) * L + l], s1);
        }
        if (n < N) s0 = fma(ys[r * ld + n], proj[n * L + l], s0);
        mu[(r0 + r) * L + l] = (s0 + s1) - shift[l];
    }
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        double s = 0.0;
        for (int r = 0; r < nr; ++r) s += ys[r * ld + n];
        part[(int64_t)blockIdx.x * N + n] = s;
    }
}

__global__ void __launch_bounds__(256) project_colsum_kernel(int N, int G, const double* part, double* out) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int g = 0;
    for (; g + 3 < G; g += 4) {
        s0 += part[(int64_t)g * N + n];
        s1 += part[(int64_t)(g + 1) * N + n];
        s2 += part[(int64_t)(g + 2) * N + n];
        s3 += part[(int64_t)(g + 3) * N + n];
    }
    for (; g < G; ++g) s0 += part[(int64_t)g * N + n];
    out[n] = (s0 + s1) + (s2 + s3);
}

// d_in: proj (N*L) then shift (L) at the start of d_work; the column sums land in d_out (N doubles)
int launch_project(vlgp_ctx* ctx, UnitSet& us, const double* d_proj, const double* d_shift, double* d_part,
                   double* d_out) {
    const int N = ctx->N, L = ctx->L;
    const int G = (int)((us.rows + PROJ_ROWS - 1) / PROJ_ROWS);
    const size_t lds = (size_t)PROJ_ROWS * (N | 1) * 8;
    if (lds > 64 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(project_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(project_kernel, dim3(G), dim3(512), lds, ctx->stream, N, L, us.rows, us.y, d_proj, d_shift,
                       us.mu, d_part);
    hipLaunchKernelGGL(project_colsum_kernel, dim3((N + 255) / 256), dim3(256), 0, ctx->stream, N, G, d_part, d_out);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}
