// Prior factor on the device: pivoted incomplete Cholesky of the squared-exponential kernel
// (math.ichol_gauss, vlgp/math.py:76-126) for every latent of one unit length, its rank and the
// compact copy (trailing zero columns dropped) the E-step kernels read -- one launch, no copies.
//
// The factor reproduces the reference's pivot sequence and its G BIT FOR BIT: every residual,
// every new column entry and the stopping sum are computed with the operation order NumPy /
// OpenBLAS use on the reference's side (np_exact.h: np.exp, np.dot, np.sum restated), the arg-max
// takes the first maximum in permuted order like numpy.argmax, and nothing is contracted.
//
// One workgroup per latent.  Rows stay at their ORIGINAL index in G (the reference swaps rows and
// un-permutes at the end, math.py:110,126: same thing); `piv` maps permuted position -> row.
// Per step: (1) np.sum(d[i:]) -- the leaves of NumPy's pairwise recursion are summed by groups of
// eight lanes (one lane per accumulator), thread 0 walks the recursion tree over the leaf sums;
// (2) first arg-max of d[i:]; (3) every remaining row gets its new column entry and residual
// (npx_ichol_row), one row per thread.
#include "ctx.h"
#include "np_exact.h"

struct IcholArgs {
    int T, R, L;
    double omega[VLGP_MAX_L], sigma[VLGP_MAX_L];
    double* full;        // (L, T, R) as the reference lays it out
    double* compact;     // latent l at l*T*R, (T, rank_l) row-major
    int* rl_table;       // (L) row of the device prior table, or null
    int* rank_dev;       // (L) device slot of this launch
    int* rank_host;      // (L) mapped host slot of this launch
    unsigned long long* flag_host;  // mapped host sequence word
    unsigned long long seq;
    unsigned* ticket;    // device arrival counter (zero between launches)
    double* gwork;       // global scratch for d, kv (2T per latent) when they do not fit LDS, else null
    int* giwork;         // ... piv (T per latent)
};

#define ICH_LEAVES(T) ((T) / 64 + 2)

// one leaf (n <= 128 contiguous elements) of NumPy's pairwise sum, by the 8 lanes [base, base+8) of a wave;
// k = lane - base is the accumulator this lane owns.  Result valid on every lane of the group.
__device__ static inline double ich_leaf8(const double* a, int n, int k, int base) {
    if (n < 8) {
        double res = -0.0;
        for (int i = 0; i < n; ++i) res = res + a[i];
        return res;
    }
    double r = a[k];
    const int nm = n - (n % 8);
    for (int i = 8; i < nm; i += 8) r = r + a[i + k];
    const double r0 = __shfl(r, base + 0), r1 = __shfl(r, base + 1), r2 = __shfl(r, base + 2), r3 = __shfl(r, base + 3);
    const double r4 = __shfl(r, base + 4), r5 = __shfl(r, base + 5), r6 = __shfl(r, base + 6), r7 = __shfl(r, base + 7);
    double res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (int i = nm; i < n; ++i) res = res + a[i];
    return res;
}

template <int NT>
__global__ void __launch_bounds__(NT) ichol_exact_kernel(IcholArgs A) {
    extern __shared__ __attribute__((aligned(16))) char ich_smem[];
    const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = A.T, R = A.R, NL = ICH_LEAVES(T);
    constexpr int NW = NT / 64;
    // LDS: [red_v 16 | leafsum NL | (d T | kv T)] doubles, [red_i 16 | misc 8 | leaf_start NL | leaf_len NL | (piv T)] ints
    double* red_v = reinterpret_cast<double*>(ich_smem);
    double* leafsum = red_v + 16;
    double* dbig = leafsum + NL;
    const bool in_lds = A.gwork == nullptr;
    double* d = in_lds ? dbig : A.gwork + (int64_t)l * 2 * T;
    double* kv = d + T;
    int* ibase = reinterpret_cast<int*>(in_lds ? dbig + 2 * (int64_t)T : dbig);
    int* red_i = ibase;
    int* misc = red_i + 16;        // 0: leaf count, 1: jast, 2: row of the pivot
    int* leaf_start = misc + 8;
    int* leaf_len = leaf_start + NL;
    int* piv = in_lds ? leaf_len + NL : A.giwork + (int64_t)l * T;
    __shared__ double s_pivot, s_total;

    double* G = A.full + (int64_t)l * T * R;
    const double om = A.omega[l];
    const double tol_n = 1e-6 * (double)T;

    for (int j = tid; j < T; j += NT) {
        d[j] = 1.0;
        piv[j] = j;
        const double dx = (double)j;
        kv[j] = npx_exp(-om * (dx * dx));  // exp(-omega (x_j - x_p)^2), a function of |j - p| only (dt = 1, math.py:101)
    }
    for (int64_t e = tid; e < (int64_t)T * R; e += NT) G[e] = 0.0;
    __syncthreads();

    int i = 0;
    for (; i < R; ++i) {
        const int n = T - i;
        // ---- np.sum(d[i:]) > tol * n ? (math.py:105) ----
        if (tid == 0) {  // leaves of the pairwise recursion, left to right
            int ss[32], sl[32], sp = 0, nl = 0;
            ss[0] = i; sl[0] = n;
            while (sp >= 0) {
                const int s = ss[sp], len = sl[sp];
                --sp;
                if (len <= 128) {
                    leaf_start[nl] = s; leaf_len[nl] = len; ++nl;
                } else {
                    int n2 = len / 2;
                    n2 -= n2 % 8;
                    ++sp; ss[sp] = s + n2; sl[sp] = len - n2;
                    ++sp; ss[sp] = s; sl[sp] = n2;
                }
            }
            misc[0] = nl;
        }
        __syncthreads();
        {
            const int nl = misc[0];
            const int base = lane & ~7, k = lane & 7;
            for (int lf = tid >> 3; lf < nl; lf += NT / 8) {
                const double s = ich_leaf8(d + leaf_start[lf], leaf_len[lf], k, base);
                if (k == 0) leafsum[lf] = s;
            }
        }
        __syncthreads();
        if (tid == 0) {
            int next = 0;
            // same traversal as npx_pairwise, leaves taken from leafsum in order
            int sl[32], st[32], sp = 0;
            double left[32], ret = 0.0;
            sl[0] = n; st[0] = 0;
            while (sp >= 0) {
                const int len = sl[sp];
                if (len <= 128) { ret = leafsum[next++]; --sp; continue; }
                int n2 = len / 2;
                n2 -= n2 % 8;
                if (st[sp] == 0) { st[sp] = 1; ++sp; sl[sp] = n2; st[sp] = 0; }
                else if (st[sp] == 1) { left[sp] = ret; st[sp] = 2; ++sp; sl[sp] = len - n2; st[sp] = 0; }
                else { ret = left[sp] + ret; --sp; }
            }
            s_total = 0.0 + ret;
        }
        __syncthreads();
        if (!(s_total > tol_n)) break;

        // ---- jast = i + argmax(d[i:]) (first maximum), 0 at the first step (math.py:106-112) ----
        if (i > 0) {
            double best = -INFINITY;
            int bi = 0x7fffffff;
            for (int p = i + tid; p < T; p += NT) {
                const double dp = d[p];
                if (dp > best) { best = dp; bi = p; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(best, o);
                const int oi = __shfl_xor(bi, o);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            if (lane == 0) { red_v[wave] = best; red_i[wave] = bi; }
            __syncthreads();
            if (tid == 0) {
                for (int w = 1; w < NW; ++w)
                    if (red_v[w] > best || (red_v[w] == best && red_i[w] < bi)) { best = red_v[w]; bi = red_i[w]; }
                misc[1] = bi;
            }
        } else if (tid == 0) {
            misc[1] = 0;
        }
        __syncthreads();  // (the barrier inside the i > 0 branch is block-uniform)
        if (tid == 0) {
            const int jast = misc[1];
            const double pivot = sqrt(d[jast]);
            const int t = piv[i]; piv[i] = piv[jast]; piv[jast] = t;
            misc[2] = piv[i];
            s_pivot = pivot;
            G[(int64_t)piv[i] * R + i] = pivot;
        }
        __syncthreads();
        // ---- new column and residuals of the remaining rows (math.py:114-119) ----
        {
            const int rowi = misc[2], mo = n - 1;
            const double pivot = s_pivot;
            const double* prow = G + (int64_t)rowi * R;
            for (int jj = tid; jj < mo; jj += NT) {
                const int p = i + 1 + jj, row = piv[p];
                const int dist = row > rowi ? row - rowi : rowi - row;
                d[p] = npx_ichol_row(G + (int64_t)row * R, prow, i, jj, mo, kv[dist], pivot);
            }
        }
        __syncthreads();
    }
    const int rank = i;
    // ---- G * sigma (gp.py:161), compact copy, rank ----
    {
        const double sg = A.sigma[l];
        double* C = A.compact + (int64_t)l * T * R;
        for (int64_t e = tid; e < (int64_t)T * R; e += NT) {
            const int row = (int)(e / R), c = (int)(e - (int64_t)row * R);
            const double v = G[e] * sg;
            G[e] = v;
            if (c < rank) C[(int64_t)row * rank + c] = v;
        }
    }
    if (tid == 0) {
        if (A.rl_table) A.rl_table[l] = rank;
        __hip_atomic_store(A.rank_dev + l, rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)A.L - 1) {  // last latent of this launch: publish every rank to the host mailbox
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int q = 0; q < A.L; ++q)
                A.rank_host[q] = __hip_atomic_load(A.rank_dev + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(A.flag_host, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The same factorisation for unit lengths of at most 64 bins (the windows of vem: one call per EM iteration, on the
// critical path between the H-step's last round and the next E-step): ONE wave per latent, lane <-> permuted position,
// the factor, the residuals and the permutation in LDS, no workgroup barrier and no single-thread section -- the
// residual sum is one leaf of NumPy's pairwise recursion (ich_leaf8 by every group of eight lanes at once), the
// arg-max a butterfly that leaves the first maximum on every lane.  Same primitives in the same order as the kernel
// above: the same bits (tests/test_gpu_parity.py::test_ichol_*).  RS: row stride of the LDS copy (odd: no bank runs).
// One row of one step (npx_ichol_row) for a wave whose lanes are the remaining rows: OpenBLAS' dgemv_t gives the
// rows of a call three different summation orders (groups of four, a trailing pair, a trailing single row); taken as
// three loops the wave walks each of them in turn -- here the loads are shared and only the arithmetic is selected.
__device__ static inline double ich_row_wave(double* row, const double* prow, int i, int jj, int mo, double kvv, double piv) {
    double dot;
    if (i == 0) {
        dot = 0.0;
    } else if (mo == 1) {  // a single remaining row is cblas_ddot (wave-uniform branch)
        dot = npx_dot_row([&](int l) { return row[l]; }, [&](int l) { return prow[l]; }, i, jj, mo);
    } else {
        const int k4 = i & ~3, n4 = (mo >> 2) << 2;
        const int path = jj < n4 ? 0 : (((mo & 2) && jj < n4 + 2) ? 1 : 2);
        double c0 = 0.0, c1 = 0.0, c2 = 0.0, c3 = 0.0;
        for (int l = 0; l < k4; l += 4) {
            const double a0 = row[l], a1 = row[l + 1], a2 = row[l + 2], a3 = row[l + 3];
            const double x0 = prow[l], x1 = prow[l + 1], x2 = prow[l + 2], x3 = prow[l + 3];
            if (path == 0) {         // four outputs at a time: 4-lane FMA accumulator
                c0 = fma(a0, x0, c0); c1 = fma(a1, x1, c1); c2 = fma(a2, x2, c2); c3 = fma(a3, x3, c3);
            } else if (path == 1) {  // trailing pair: 2-lane multiply, add
                c0 = c0 + a0 * x0; c1 = c1 + a1 * x1; c0 = c0 + a2 * x2; c1 = c1 + a3 * x3;
            } else {                 // trailing single output: two 2-lane accumulators
                c0 = c0 + a0 * x0; c1 = c1 + a1 * x1; c2 = c2 + a2 * x2; c3 = c3 + a3 * x3;
            }
        }
        double t = 0.0;
        if (k4) t = path == 1 ? c0 + c1 : (c0 + c2) + (c1 + c3);
        double y = 0.0 + t;
        const int r = i - k4;
        if (r == 1) y = fma(row[k4], prow[k4], y);
        else if (r == 2) y = y + fma(row[k4], prow[k4], row[k4 + 1] * prow[k4 + 1]);
        else if (r == 3) y = y + fma(row[k4 + 2], prow[k4 + 2], fma(row[k4], prow[k4], row[k4 + 1] * prow[k4 + 1]));
        dot = y;
    }
    const double g = (kvv - dot) / piv;
    row[i] = g;
    const double ss = 0.0 + npx_pairwise_leaf([&](int l) { const double v = row[l]; return v * v; }, i + 1);
    return 1.0 - ss;
}

__global__ void __launch_bounds__(64) ichol_exact_wave_kernel(IcholArgs A, int RS) {
    extern __shared__ __attribute__((aligned(16))) char ich_smem[];
    const int l = blockIdx.x, lane = threadIdx.x;
    const int T = A.T, R = A.R;
    double* Gs = reinterpret_cast<double*>(ich_smem);   // (T, RS)
    double* d = Gs + (int64_t)T * RS;                   // 64
    double* kv = d + 64;                                // 64
    int* piv = reinterpret_cast<int*>(kv + 64);         // 64

    const double om = A.omega[l];
    const double tol_n = 1e-6 * (double)T;
    for (int e = lane; e < T * RS; e += 64) Gs[e] = 0.0;
    if (lane < T) {
        d[lane] = 1.0;
        piv[lane] = lane;
        const double dx = (double)lane;
        kv[lane] = npx_exp(-om * (dx * dx));
    }
    __syncthreads();

    int i = 0;
    for (; i < R; ++i) {
        const int n = T - i;
        if (n <= 0) break;  // (np.sum of nothing is 0.0: not above the tolerance)
        // (every lane forming the whole sum from 64 preloaded values and selects instead of eight lanes and shuffles:
        // measured 14 us SLOWER per launch at rank 11, same box -- tools/variant_ab.sh)
        const double total = 0.0 + ich_leaf8(d + i, n, lane & 7, lane & ~7);
        if (!(total > tol_n)) break;
        int jast = 0;
        if (i > 0) {
            double best = -INFINITY;
            int bi = 0x7fffffff;
            if (lane >= i && lane < T) { best = d[lane]; bi = lane; }
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(best, o);
                const int oi = __shfl_xor(bi, o);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            jast = bi;
        }
        const double pivot = sqrt(d[jast]);
        const int rowi = piv[jast], rowo = piv[i];
        __syncthreads();  // every lane has read d[jast], piv[] before they change
        if (lane == 0) {
            piv[i] = rowi;
            piv[jast] = rowo;
            Gs[rowi * RS + i] = pivot;
        }
        __syncthreads();
        const int mo = n - 1;
        if (lane < mo) {
            const int p = i + 1 + lane, row = piv[p];
            const int dist = row > rowi ? row - rowi : rowi - row;
            d[p] = ich_row_wave(Gs + row * RS, Gs + rowi * RS, i, lane, mo, kv[dist], pivot);
        }
        __syncthreads();
    }
    const int rank = i;
    {
        const double sg = A.sigma[l];
        double* G = A.full + (int64_t)l * T * R;
        double* C = A.compact + (int64_t)l * T * R;
        for (int e = lane; e < T * R; e += 64) {
            const int row = e / R, c = e - row * R;
            const double v = Gs[row * RS + c] * sg;
            G[e] = v;
            if (c < rank) C[row * rank + c] = v;
        }
    }
    if (lane == 0) {
        if (A.rl_table) A.rl_table[l] = rank;
        __hip_atomic_store(A.rank_dev + l, rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(A.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == (unsigned)A.L - 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            for (int q = 0; q < A.L; ++q)
                A.rank_host[q] = __hip_atomic_load(A.rank_dev + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence_system();
            __hip_atomic_store(A.flag_host, A.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// number of leading columns up to the last non-zero one, per latent (host-injected factors)
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

// compact[l*T*R + t*r_l + c] = G[l, t, c], c < r_l
__global__ void __launch_bounds__(256)
prior_compact_kernel(int T, int R, const double* G, const int* rl, double* out) {
    const int l = blockIdx.y;
    const int r = rl[l];
    const int64_t n = (int64_t)T * r;
    const int64_t base = (int64_t)l * T * R;
    for (int64_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const int64_t t = i / r;
        const int c = (int)(i - t * r);
        out[base + i] = G[base + t * R + c];
    }
}

static int ensure_mailbox(vlgp_ctx* ctx) {
    if (ctx->h_prior_mb) return VLGP_OK;
    const size_t bytes = sizeof(int) * VLGP_PRIOR_SLOTS * VLGP_MAX_L + 64;
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->h_prior_mb), bytes, hipHostMallocMapped));
    memset(ctx->h_prior_mb, 0, bytes);
    HIPCHK(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_prior_mb_host), ctx->h_prior_mb, 0));
    HIPCHK(ctx, hipMalloc(&ctx->d_prior_mb, sizeof(int) * VLGP_PRIOR_SLOTS * VLGP_MAX_L + 64));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_prior_mb, 0, sizeof(int) * VLGP_PRIOR_SLOTS * VLGP_MAX_L + 64, ctx->stream));
    return VLGP_OK;
}

template <int NT>
static int launch_ichol_t(vlgp_ctx* ctx, const IcholArgs& A, size_t lds) {
    auto fn = ichol_exact_kernel<NT>;
    if (lds > 64 * 1024)
        HIPCHK(ctx, hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)lds));
    hipLaunchKernelGGL(fn, dim3(A.L), dim3(NT), lds, ctx->stream, A);
    return VLGP_OK;
}

// Factor every listed prior (all already present in ctx->priors) with the given hyper-parameters and wait for
// the ranks.  in_table: the priors' rows of the device table are current, the kernel refreshes their ranks.
// the ranks of a launch whose wait was left for later (launch_ichol_all, lazy): wait for its sequence word, take them
static int prior_wait_seq(vlgp_ctx* ctx, unsigned long long last) {
    int* flag_words = ctx->h_prior_mb + VLGP_PRIOR_SLOTS * VLGP_MAX_L;
    volatile unsigned long long* h_flag = reinterpret_cast<volatile unsigned long long*>(flag_words);
    unsigned spins = 0;
    while (*h_flag != last) {
        if ((++spins & 0xfff) == 0) {  // a faulted kernel must not hang the host
            const hipError_t qe = hipStreamQuery(ctx->stream);
            if (qe == hipSuccess) {
                if (*h_flag == last) break;
                if ((spins >> 12) > 64) return vlgp_fail(ctx, VLGP_ERR_HIP, "prior kernel finished without publishing its ranks");
            } else if (qe != hipErrorNotReady) {
                return vlgp_fail(ctx, VLGP_ERR_HIP, "prior kernel failed: %s", hipGetErrorString(qe));
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return VLGP_OK;
}
int vlgp_prior_collect(vlgp_ctx* ctx) {
    if (ctx->prior_pending.empty()) return VLGP_OK;
    std::vector<Prior*> prs;
    prs.swap(ctx->prior_pending);
    CHK(prior_wait_seq(ctx, ctx->prior_pending_seq));
    for (size_t j = 0; j < prs.size(); ++j)
        prs[j]->rl.assign(ctx->h_prior_mb + j * VLGP_MAX_L, ctx->h_prior_mb + j * VLGP_MAX_L + ctx->L);
    return VLGP_OK;
}

// lazy: (one batch, priors in the table) return right behind the launches; the host-side ranks are taken by
// vlgp_prior_collect, which every consumer of Prior::rl calls first -- the rebuild of an EM iteration then runs under the
// host's way to the next E-step instead of in front of it.
int launch_ichol_all(vlgp_ctx* ctx, const std::vector<Prior*>& prs, const double* omega, const double* sigma,
                     bool in_table, bool lazy) {
    const int L = ctx->L, R = ctx->R;
    CHK(ensure_mailbox(ctx));
    CHK(vlgp_prior_collect(ctx));
    unsigned long long* d_flag = reinterpret_cast<unsigned long long*>(ctx->d_prior_mb_host + VLGP_PRIOR_SLOTS * VLGP_MAX_L);
    unsigned* d_ticket = reinterpret_cast<unsigned*>(ctx->d_prior_mb + VLGP_PRIOR_SLOTS * VLGP_MAX_L);
    // global scratch for the lengths whose residuals / pivots do not fit LDS
    int64_t gw = 0;
    for (Prior* pr : prs)
        if (pr->T > 4096) gw = std::max<int64_t>(gw, (int64_t)L * (2 * (int64_t)pr->T + (pr->T + 1) / 2 + 1));
    if (gw) CHK(vlgp_ensure_work(ctx, gw));
    for (size_t base = 0; base < prs.size(); base += VLGP_PRIOR_SLOTS) {
        const size_t cnt = std::min<size_t>(VLGP_PRIOR_SLOTS, prs.size() - base);
        unsigned long long last = 0;
        for (size_t j = 0; j < cnt; ++j) {
            Prior& pr = *prs[base + j];
            IcholArgs A;
            A.T = pr.T; A.R = R; A.L = L;
            for (int l = 0; l < L; ++l) { A.omega[l] = omega[l]; A.sigma[l] = sigma[l]; }
            A.full = pr.d_full; A.compact = pr.d_compact;
            A.rl_table = (in_table && ctx->d_prior_rl && pr.index >= 0) ? ctx->d_prior_rl + (int64_t)pr.index * L : nullptr;
            A.rank_dev = ctx->d_prior_mb + j * VLGP_MAX_L;
            A.rank_host = ctx->d_prior_mb_host + j * VLGP_MAX_L;
            A.flag_host = d_flag;
            A.seq = last = ++ctx->prior_seq;
            A.ticket = d_ticket;
            const bool in_lds = pr.T <= 4096;
            A.gwork = in_lds ? nullptr : ctx->d_work;
            A.giwork = in_lds ? nullptr : reinterpret_cast<int*>(ctx->d_work + (int64_t)L * 2 * pr.T);
            const int NL = ICH_LEAVES(pr.T);
            const size_t lds = 8 * (size_t)(16 + NL + (in_lds ? 2 * pr.T : 0)) + 4 * (size_t)(24 + 2 * NL + (in_lds ? pr.T : 0)) + 16;
            vlgp_prof_begin(ctx, VLGP_PROF_PRIOR);
            const int RS = R | 1;
            const size_t lds_wave = 8 * ((size_t)pr.T * RS + 128) + 4 * 64;
            if (pr.T <= 64 && lds_wave <= 60 * 1024 && !getenv("VLGP_ICHOL_BLOCK")) {
                hipLaunchKernelGGL(ichol_exact_wave_kernel, dim3(L), dim3(64), lds_wave, ctx->stream, A, RS);
            } else if (pr.T <= 64) CHK(launch_ichol_t<64>(ctx, A, lds));
            else if (pr.T <= 512) CHK(launch_ichol_t<256>(ctx, A, lds));
            else CHK(launch_ichol_t<1024>(ctx, A, lds));
            vlgp_prof_end(ctx, VLGP_PROF_PRIOR, (double)L);
            HIPCHK(ctx, hipGetLastError());
        }
        if (lazy && in_table && prs.size() <= (size_t)VLGP_PRIOR_SLOTS) {
            ctx->prior_pending.assign(prs.begin(), prs.end());
            ctx->prior_pending_seq = last;
            return VLGP_OK;
        }
        CHK(prior_wait_seq(ctx, last));
        for (size_t j = 0; j < cnt; ++j) {
            Prior& pr = *prs[base + j];
            pr.rl.assign(ctx->h_prior_mb + j * VLGP_MAX_L, ctx->h_prior_mb + j * VLGP_MAX_L + L);
        }
    }
    return VLGP_OK;
}

// host-injected factor (vlgp_set_prior): rank per latent and the compact copy
int launch_compact_prior(vlgp_ctx* ctx, Prior& pr) {
    const int L = ctx->L, R = ctx->R, T = pr.T;
    CHK(vlgp_ensure_work(ctx, L + 8));
    CHK(vlgp_ensure_pinned(ctx, L + 8));
    int* d_rank = reinterpret_cast<int*>(ctx->d_work);
    hipLaunchKernelGGL(prior_rank_kernel, dim3(L), dim3(256), 0, ctx->stream, T, R, pr.d_full, d_rank);
    HIPCHK(ctx, hipGetLastError());
    const int gx = (int)((((int64_t)T * R) + 255) / 256);
    hipLaunchKernelGGL(prior_compact_kernel, dim3(gx > 64 ? 64 : gx, L), dim3(256), 0, ctx->stream, T, R, pr.d_full,
                       d_rank, pr.d_compact);
    HIPCHK(ctx, hipGetLastError());
    int* h_rank = reinterpret_cast<int*>(ctx->h_pinned);
    HIPCHK(ctx, hipMemcpyAsync(h_rank, d_rank, sizeof(int) * L, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    pr.rl.assign(h_rank, h_rank + L);
    return VLGP_OK;
}

// ---- debug: the device build of np_exact.h, element-wise (tests compare it with the host NumPy) ----
__global__ void npx_probe_kernel(int kind, int64_t n, const double* a, const double* b, double* out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (kind == 0) out[i] = npx_exp(a[i]);
    else if (kind == 1) out[i] = sqrt(a[i]);
    else if (kind == 2) out[i] = a[i] / b[i];
    else out[i] = fma(a[i], b[i], out[i]);
}
int launch_npx_probe(vlgp_ctx* ctx, int kind, int64_t n, const double* d_a, const double* d_b, double* d_out) {
    hipLaunchKernelGGL(npx_probe_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, kind, n, d_a, d_b,
                       d_out);
    HIPCHK(ctx, hipGetLastError());
    return VLGP_OK;
}
