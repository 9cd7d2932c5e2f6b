// C ABI of libvlgp_hip.so (include/vlgp_hip.h): handle management, host<->device
// marshalling, RCCL plumbing and the thin wrappers around the kernel launchers.
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <time.h>

#include <algorithm>
#include <new>

#include "ctx.h"

static thread_local std::string g_create_err;

int vlgp_fail(vlgp_ctx* ctx, int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    else g_create_err = buf;
    return code;
}

#define NEED_CTX(ctx) \
    if (!(ctx)) return vlgp_fail(nullptr, VLGP_ERR_ARG, "null handle")

static int dev_alloc(vlgp_ctx* ctx, double** p, int64_t n, bool zero) {
    *p = nullptr;
    if (n <= 0) n = 1;
    HIPCHK(ctx, hipMalloc(p, (size_t)n * sizeof(double)));
    if (zero) HIPCHK(ctx, hipMemsetAsync(*p, 0, (size_t)n * sizeof(double), ctx->stream));
    return VLGP_OK;
}

int vlgp_ensure_work(vlgp_ctx* ctx, int64_t n) {
    if (n <= ctx->work_len) return VLGP_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_work) HIPCHK(ctx, hipFree(ctx->d_work));
    ctx->d_work = nullptr;
    ctx->work_len = 0;
    const int64_t cap = n + n / 4 + 1024;
    HIPCHK(ctx, hipMalloc(&ctx->d_work, (size_t)cap * sizeof(double)));
    ctx->work_len = cap;
    return VLGP_OK;
}

int vlgp_ensure_work_m(vlgp_ctx* ctx, int64_t n) {
    if (n <= ctx->work_m_len) return VLGP_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->mstream));
    if (ctx->d_work_m) HIPCHK(ctx, hipFree(ctx->d_work_m));
    ctx->d_work_m = nullptr;
    ctx->work_m_len = 0;
    const int64_t cap = n + n / 4 + 1024;
    HIPCHK(ctx, hipMalloc(&ctx->d_work_m, (size_t)cap * sizeof(double)));
    ctx->work_m_len = cap;
    return VLGP_OK;
}

// wait for a queued norms pass (vlgp_norms_begin): its sequence word in mapped host memory
static int wait_norms(vlgp_ctx* ctx) {
    if (ctx->x_pending != 1) return VLGP_OK;
    volatile unsigned long long* flag = reinterpret_cast<volatile unsigned long long*>(ctx->h_xres + 2);
    unsigned spins = 0;
    while (*flag != ctx->x_seq) {
        if ((++spins & 0xfff) == 0) {  // a faulted kernel must not hang the host
            const hipError_t qe = hipStreamQuery(ctx->stream);
            if (qe == hipSuccess) {
                if (*flag == ctx->x_seq) break;
                if ((spins >> 12) > 64) return vlgp_fail(ctx, VLGP_ERR_HIP, "norms kernel finished without publishing");
            } else if (qe != hipErrorNotReady) {
                return vlgp_fail(ctx, VLGP_ERR_HIP, "norms kernel failed: %s", hipGetErrorString(qe));
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return VLGP_OK;
}

int vlgp_join_m(vlgp_ctx* ctx) {
    ++ctx->write_epoch;  // the callers are the entry points that read or write parameters / unit state (vlgp_hstep_prepare)
    // a pending norms pass (vlgp_norms_begin) reads mu, v, dmu on its own stream: the same callers wait for it
    CHK(wait_norms(ctx));
    if (!ctx->m_pending) return VLGP_OK;
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_m_done));  // m_pending is cleared by vlgp_mstep_end
    return VLGP_OK;
}

int vlgp_ensure_pinned(vlgp_ctx* ctx, int64_t n) {
    if (n <= ctx->pinned_len) return VLGP_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->h_pinned) HIPCHK(ctx, hipHostFree(ctx->h_pinned));
    ctx->h_pinned = nullptr;
    ctx->pinned_len = 0;
    const int64_t cap = n + 1024;
    HIPCHK(ctx, hipHostMalloc(&ctx->h_pinned, (size_t)cap * sizeof(double), hipHostMallocDefault));
    ctx->pinned_len = cap;
    return VLGP_OK;
}

UnitSet* vlgp_get_set(vlgp_ctx* ctx, int set, bool must_be_valid) {
    if (set < 0 || set >= VLGP_MAX_SETS) {
        vlgp_fail(ctx, VLGP_ERR_ARG, "unit set index %d out of range [0, %d)", set, VLGP_MAX_SETS);
        return nullptr;
    }
    UnitSet* us = &ctx->sets[set];
    if (must_be_valid && !us->valid) {
        vlgp_fail(ctx, VLGP_ERR_STATE, "unit set %d is empty (upload or cut first)", set);
        return nullptr;
    }
    return us;
}

// ---- profiling -----------------------------------------------------------
void vlgp_prof_begin(vlgp_ctx* ctx, int kind, hipStream_t st) {
    if (!ctx->prof_on) return;
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
    (void)hipEventRecord(a, st ? st : ctx->stream);
    ctx->pending.push_back({kind, a, b, 0.0, false});
}
void vlgp_prof_end(vlgp_ctx* ctx, int kind, double units, hipStream_t st) {
    if (!ctx->prof_on) return;
    // brackets nest (a whole E-step around the sampled launches inside it): close the innermost open one of this kind
    for (auto it = ctx->pending.rbegin(); it != ctx->pending.rend(); ++it) {
        if (it->kind != kind || it->done) continue;
        it->units = units;
        it->done = true;
        (void)hipEventRecord(it->b, st ? st : ctx->stream);
        return;
    }
}
static void prof_drain(vlgp_ctx* ctx) {
    if (ctx->pending.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipStreamSynchronize(ctx->mstream);
    for (auto& p : ctx->pending) {
        float ms = 0.f;
        if (p.done && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->prof[p.kind].launches += 1;
            ctx->prof[p.kind].ms += ms;
            ctx->prof[p.kind].units += p.units;
        }
        (void)hipEventDestroy(p.a);
        (void)hipEventDestroy(p.b);
    }
    ctx->pending.clear();
}

// ---- RCCL (loaded lazily: single-GPU use never touches it) ---------------
typedef struct { char internal[128]; } rccl_uid;
typedef int (*fn_getuid)(rccl_uid*);
typedef int (*fn_initrank)(void**, int, rccl_uid, int);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, void*, hipStream_t);
typedef int (*fn_destroy)(void*);
typedef const char* (*fn_errstr)(int);
typedef int (*fn_count)(void*, int*);
static struct {
    void* lib = nullptr;
    fn_getuid get_uid = nullptr;
    fn_initrank init_rank = nullptr;
    fn_allreduce all_reduce = nullptr;
    fn_destroy destroy = nullptr;
    fn_errstr errstr = nullptr;
    fn_count count = nullptr;  // ncclCommCount: what RCCL itself says the communicator spans
} g_rccl;

static int rccl_load(vlgp_ctx* ctx) {
    if (g_rccl.lib) return VLGP_OK;
    void* lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return vlgp_fail(ctx, VLGP_ERR_COMM, "cannot load librccl.so: %s", dlerror());
    g_rccl.get_uid = (fn_getuid)dlsym(lib, "ncclGetUniqueId");
    g_rccl.init_rank = (fn_initrank)dlsym(lib, "ncclCommInitRank");
    g_rccl.all_reduce = (fn_allreduce)dlsym(lib, "ncclAllReduce");
    g_rccl.destroy = (fn_destroy)dlsym(lib, "ncclCommDestroy");
    g_rccl.errstr = (fn_errstr)dlsym(lib, "ncclGetErrorString");
    g_rccl.count = (fn_count)dlsym(lib, "ncclCommCount");
    if (!g_rccl.get_uid || !g_rccl.init_rank || !g_rccl.all_reduce || !g_rccl.destroy)
        return vlgp_fail(ctx, VLGP_ERR_COMM, "librccl.so lacks an expected symbol");
    g_rccl.lib = lib;
    return VLGP_OK;
}


// ---- shared-memory test transport --------------------------------------------------
// VLGP_COMM_TRANSPORT=shm replaces RCCL by an all-reduce through a POSIX shared-memory
// segment (device -> host, sum in rank order, host -> device).  It exists so that the
// full multi-rank protocol -- sharding, the per-Newton-iteration M-step reductions, the
// H-step rounds, the norms -- can be exercised by several processes that share ONE GPU
// (RCCL refuses two ranks on a device).  It is a test vehicle, not a data path.
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#define SHM_MAX_RANKS 16
#define SHM_SLOT_DOUBLES (1 << 17)
struct ShmRegion {
    volatile long long ready[SHM_MAX_RANKS];
    volatile long long done[SHM_MAX_RANKS];
    double slot[SHM_MAX_RANKS][SHM_SLOT_DOUBLES];
};
struct ShmComm {
    ShmRegion* reg = nullptr;
    long long seq = 0;
    int rank = 0, world = 1;
    char name[64];
};
static bool shm_mode() {
    const char* t = getenv("VLGP_COMM_TRANSPORT");
    return t && strcmp(t, "shm") == 0;
}
static ShmComm* shm_open_comm(const char id[VLGP_UNIQUE_ID_BYTES], int rank, int world) {
    if (world > SHM_MAX_RANKS) return nullptr;
    ShmComm* c = new ShmComm();
    c->rank = rank; c->world = world;
    unsigned long long h = 1469598103934665603ULL;
    for (int i = 0; i < VLGP_UNIQUE_ID_BYTES; ++i) h = (h ^ (unsigned char)id[i]) * 1099511628211ULL;
    snprintf(c->name, sizeof(c->name), "/vlgp_shm_%016llx", h);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return nullptr; }
    if (ftruncate(fd, sizeof(ShmRegion)) != 0) { close(fd); delete c; return nullptr; }
    void* p = mmap(nullptr, sizeof(ShmRegion), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return nullptr; }
    c->reg = (ShmRegion*)p;  // a fresh segment is zero-filled: ready/done start at 0, seq at 1
    return c;
}
static void shm_close_comm(ShmComm* c) {
    if (!c) return;
    if (c->reg) munmap((void*)c->reg, sizeof(ShmRegion));
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
}
static int shm_allreduce(vlgp_ctx* ctx, ShmComm* c, hipStream_t st, double* d_buf, int64_t n) {
    if (n > SHM_SLOT_DOUBLES) return vlgp_fail(ctx, VLGP_ERR_COMM, "shm transport: buffer of %lld doubles too large", (long long)n);
    const long long s = ++c->seq;
    ShmRegion* R = c->reg;
    // a rank that died must not hang the others: every wait gives up after VLGP_SHM_TIMEOUT_S (default 60 s)
    const char* tmo_s = getenv("VLGP_SHM_TIMEOUT_S");
    const double tmo = tmo_s ? atof(tmo_s) : 60.0;
    auto wait_ge = [&](volatile long long* p, long long want) {
        const auto t0 = std::chrono::steady_clock::now();
        while (*p < want) {
            usleep(20);
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > tmo) return false;
        }
        return true;
    };
    // nobody may still be reading my slot from the previous round
    for (int k = 0; k < c->world; ++k)
        if (!wait_ge(&R->done[k], s - 1)) return vlgp_fail(ctx, VLGP_ERR_COMM, "shm transport: rank %d is not responding", k);
    HIPCHK(ctx, hipMemcpyAsync((void*)R->slot[c->rank], d_buf, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    __sync_synchronize();
    R->ready[c->rank] = s;
    for (int k = 0; k < c->world; ++k)
        if (!wait_ge(&R->ready[k], s)) return vlgp_fail(ctx, VLGP_ERR_COMM, "shm transport: rank %d is not responding", k);
    __sync_synchronize();
    std::vector<double> sum((size_t)n, 0.0);
    for (int k = 0; k < c->world; ++k)  // fixed rank order: every rank gets the same bits
        for (int64_t i = 0; i < n; ++i) sum[(size_t)i] += R->slot[k][i];
    __sync_synchronize();
    R->done[c->rank] = s;
    HIPCHK(ctx, hipMemcpyAsync(d_buf, sum.data(), sizeof(double) * n, hipMemcpyHostToDevice, st));
    HIPCHK(ctx, hipStreamSynchronize(st));
    return VLGP_OK;
}

// ---- host-side exchange of the H-step round sums -----------------------------------------
// With several ranks every H-step round needs sum_ranks (ll, dll) for a handful of evaluations.
// Each rank already has its own sums on the host (the round kernel's mailbox), so the cross-rank
// sum is done there: one cache line per rank in a POSIX shared-memory segment, busy-polled, summed
// in rank order (every rank gets the same bits).  ~3 us instead of an RCCL all-reduce plus a
// device-to-host copy (~50 us) on the latency-critical path of ~40 dependent rounds per EM
// iteration.  Single node only -- as is the rendezvous of the RCCL ids.
#define HX_MAX_VALS 64
struct HxRegion {
    long long ready[SHM_MAX_RANKS][8];   // one cache line per counter
    long long done[SHM_MAX_RANKS][8];
    double slot[SHM_MAX_RANKS][HX_MAX_VALS];
};
struct HxComm {
    HxRegion* reg = nullptr;
    long long seq = 0;
    int rank = 0, world = 1;
    char name[64];
};
static HxComm* hx_open(const char id[VLGP_UNIQUE_ID_BYTES], int rank, int world) {
    if (world > SHM_MAX_RANKS) return nullptr;
    HxComm* c = new HxComm();
    c->rank = rank; c->world = world;
    unsigned long long h = 1469598103934665603ULL;
    for (int i = 0; i < VLGP_UNIQUE_ID_BYTES; ++i) h = (h ^ (unsigned char)id[i]) * 1099511628211ULL;
    snprintf(c->name, sizeof(c->name), "/vlgp_hx_%016llx", h);
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0) { delete c; return nullptr; }
    if (ftruncate(fd, sizeof(HxRegion)) != 0) { close(fd); delete c; return nullptr; }
    void* p = mmap(nullptr, sizeof(HxRegion), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { delete c; return nullptr; }
    c->reg = (HxRegion*)p;  // a fresh segment is zero-filled
    return c;
}
static void hx_close(HxComm* c) {
    if (!c) return;
    if (c->reg) munmap((void*)c->reg, sizeof(HxRegion));
    if (c->rank == 0) shm_unlink(c->name);
    delete c;
}
static bool hx_wait(const long long* p, long long want) {
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned spins = 0;; ++spins) {
        if (__atomic_load_n(p, __ATOMIC_ACQUIRE) >= want) return true;
        if ((spins & 0xffff) == 0xffff &&
            std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120)) return false;
    }
}
int vlgp_hx_allreduce(vlgp_ctx* ctx, double* vals, int n) {
    HxComm* c = (HxComm*)ctx->hx;
    if (!c) return vlgp_fail(ctx, VLGP_ERR_STATE, "no host exchange segment");
    if (n < 0 || n > HX_MAX_VALS) return vlgp_fail(ctx, VLGP_ERR_ARG, "host exchange of %d values", n);
    const long long s = ++c->seq;
    HxRegion* R = c->reg;
    for (int k = 0; k < c->world; ++k)  // nobody may still be reading my slot from the previous round
        if (!hx_wait(&R->done[k][0], s - 1)) return vlgp_fail(ctx, VLGP_ERR_COMM, "host exchange: rank %d is not responding", k);
    for (int i = 0; i < n; ++i) R->slot[c->rank][i] = vals[i];
    __atomic_store_n(&R->ready[c->rank][0], s, __ATOMIC_RELEASE);
    for (int k = 0; k < c->world; ++k)
        if (!hx_wait(&R->ready[k][0], s)) return vlgp_fail(ctx, VLGP_ERR_COMM, "host exchange: rank %d is not responding", k);
    double sum[HX_MAX_VALS];
    for (int i = 0; i < n; ++i) sum[i] = 0.0;
    for (int k = 0; k < c->world; ++k)  // fixed rank order
        for (int i = 0; i < n; ++i) sum[i] += R->slot[k][i];
    __atomic_store_n(&R->done[c->rank][0], s, __ATOMIC_RELEASE);
    for (int i = 0; i < n; ++i) vals[i] = sum[i];
    return VLGP_OK;
}

int vlgp_allreduce(vlgp_ctx* ctx, double* d_buf, int64_t n) {
    if (ctx->shm) return shm_allreduce(ctx, (ShmComm*)ctx->shm, ctx->stream, d_buf, n);
    if (!ctx->comm) return VLGP_OK;
    const int rc = g_rccl.all_reduce(d_buf, d_buf, (size_t)n, /*ncclDouble*/ 8, /*ncclSum*/ 0, ctx->comm, ctx->stream);
    if (rc != 0)
        return vlgp_fail(ctx, VLGP_ERR_COMM, "ncclAllReduce failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    return VLGP_OK;
}

int vlgp_allreduce_m(vlgp_ctx* ctx, double* d_buf, int64_t n) {
    if (ctx->shm_m) return shm_allreduce(ctx, (ShmComm*)ctx->shm_m, ctx->mstream, d_buf, n);
    if (!ctx->comm_m) {
        if (ctx->shm) return vlgp_fail(ctx, VLGP_ERR_STATE, "multi-rank M-step needs vlgp_comm_init_aux");
        if (ctx->comm)
            return vlgp_fail(ctx, VLGP_ERR_STATE, "multi-rank M-step needs the second communicator (vlgp_comm_init_aux)");
        return VLGP_OK;
    }
    const int rc = g_rccl.all_reduce(d_buf, d_buf, (size_t)n, /*ncclDouble*/ 8, /*ncclSum*/ 0, ctx->comm_m, ctx->mstream);
    if (rc != 0)
        return vlgp_fail(ctx, VLGP_ERR_COMM, "ncclAllReduce (M-step lane) failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    return VLGP_OK;
}

extern "C" int vlgp_comm_unique_id(char id[VLGP_UNIQUE_ID_BYTES]) {
    if (shm_mode()) {  // any bytes unique to this call will do
        memset(id, 0, VLGP_UNIQUE_ID_BYTES);
        static int counter = 0;
        snprintf(id, VLGP_UNIQUE_ID_BYTES, "shm-%d-%d-%ld", (int)getpid(), counter++, (long)time(nullptr));
        return VLGP_OK;
    }
    CHK(rccl_load(nullptr));
    rccl_uid u;
    const int rc = g_rccl.get_uid(&u);
    if (rc != 0) return vlgp_fail(nullptr, VLGP_ERR_COMM, "ncclGetUniqueId failed (%d)", rc);
    memcpy(id, u.internal, VLGP_UNIQUE_ID_BYTES);
    return VLGP_OK;
}

extern "C" int vlgp_comm_init(vlgp_ctx* ctx, const char id[VLGP_UNIQUE_ID_BYTES], int rank, int world) {
    NEED_CTX(ctx);
    if (world < 1 || rank < 0 || rank >= world) return vlgp_fail(ctx, VLGP_ERR_ARG, "bad rank/world %d/%d", rank, world);
    ctx->rank = rank;
    ctx->world = world;
    for (auto& us : ctx->sets) us.rows_all_ranks = 0.0;  // row totals exchanged under another communicator are stale
    // a single rank needs no communicator; VLGP_FORCE_RCCL=1 builds one anyway so
    // that the RCCL plumbing can be exercised on a one-GPU box (tests)
    if (world == 1 && !getenv("VLGP_FORCE_RCCL")) return VLGP_OK;
    if (world > 1 && !getenv("VLGP_NO_HOST_EXCHANGE")) {
        hx_close((HxComm*)ctx->hx);  // a second vlgp_comm_init (transport fallback) starts over
        ctx->hx = hx_open(id, rank, world);  // optional: without it the H-step rounds use the device all-reduce
    }
    if (shm_mode()) {
        ctx->shm = shm_open_comm(id, rank, world);
        if (!ctx->shm) return vlgp_fail(ctx, VLGP_ERR_COMM, "cannot open the shared-memory test transport");
        return VLGP_OK;
    }
    CHK(rccl_load(ctx));
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    rccl_uid u;
    memcpy(u.internal, id, VLGP_UNIQUE_ID_BYTES);
    const int rc = g_rccl.init_rank(&ctx->comm, world, u, rank);
    if (rc != 0)
        return vlgp_fail(ctx, VLGP_ERR_COMM, "ncclCommInitRank failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    return VLGP_OK;
}

extern "C" int vlgp_comm_init_aux(vlgp_ctx* ctx, const char id[VLGP_UNIQUE_ID_BYTES]) {
    NEED_CTX(ctx);
    if (ctx->world == 1 && !getenv("VLGP_FORCE_RCCL")) return VLGP_OK;
    if (shm_mode()) {
        if (!ctx->shm) return vlgp_fail(ctx, VLGP_ERR_STATE, "vlgp_comm_init must precede vlgp_comm_init_aux");
        ctx->shm_m = shm_open_comm(id, ctx->rank, ctx->world);
        if (!ctx->shm_m) return vlgp_fail(ctx, VLGP_ERR_COMM, "cannot open the shared-memory test transport (aux)");
        return VLGP_OK;
    }
    if (!ctx->comm) return vlgp_fail(ctx, VLGP_ERR_STATE, "vlgp_comm_init must precede vlgp_comm_init_aux");
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    rccl_uid u;
    memcpy(u.internal, id, VLGP_UNIQUE_ID_BYTES);
    const int rc = g_rccl.init_rank(&ctx->comm_m, ctx->world, u, ctx->rank);
    if (rc != 0)
        return vlgp_fail(ctx, VLGP_ERR_COMM, "ncclCommInitRank (aux) failed: %s", g_rccl.errstr ? g_rccl.errstr(rc) : "?");
    return VLGP_OK;
}

extern "C" int vlgp_comm_host_exchange(vlgp_ctx* ctx) { return ctx && ctx->hx ? 1 : 0; }
extern "C" int vlgp_comm_transport(vlgp_ctx* ctx) { return !ctx ? 0 : (ctx->comm ? 1 : (ctx->shm ? 2 : 0)); }

extern "C" int vlgp_comm_rccl_ranks(vlgp_ctx* ctx, int* main_lane, int* m_lane) {
    NEED_CTX(ctx);
    if (!main_lane || !m_lane) return vlgp_fail(ctx, VLGP_ERR_ARG, "null out");
    *main_lane = 0;
    *m_lane = 0;
    if (ctx->comm && g_rccl.count && g_rccl.count(ctx->comm, main_lane) != 0) *main_lane = -1;
    if (ctx->comm_m && g_rccl.count && g_rccl.count(ctx->comm_m, m_lane) != 0) *m_lane = -1;
    return VLGP_OK;
}

extern "C" int vlgp_comm_allreduce_host(vlgp_ctx* ctx, double* buf, int n) {
    NEED_CTX(ctx);
    if (n < 0 || (n > 0 && !buf)) return vlgp_fail(ctx, VLGP_ERR_ARG, "bad allreduce arguments");
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    if (!ctx->comm && !ctx->shm) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        return VLGP_OK;
    }
    const int m = n > 0 ? n : 1;
    CHK(vlgp_ensure_work(ctx, m + 8));
    CHK(vlgp_ensure_pinned(ctx, m + 8));
    if (n > 0) memcpy(ctx->h_pinned, buf, sizeof(double) * n);
    else ctx->h_pinned[0] = 0.0;
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_work, ctx->h_pinned, sizeof(double) * m, hipMemcpyHostToDevice, ctx->stream));
    CHK(vlgp_allreduce(ctx, ctx->d_work, m));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->d_work, sizeof(double) * m, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (n > 0) memcpy(buf, ctx->h_pinned, sizeof(double) * n);
    return VLGP_OK;
}

// ---- lifetime --------------------------------------------------------------
extern "C" int vlgp_abi_version(void) { return VLGP_ABI_VERSION; }

extern "C" int vlgp_device_count(int* count) {
    int n = 0;
    const hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return vlgp_fail(nullptr, VLGP_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return VLGP_OK;
}

extern "C" const char* vlgp_last_error(vlgp_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

static void free_set(vlgp_ctx* ctx, UnitSet& us) {
    if (!us.valid) return;
    (void)hipStreamSynchronize(ctx->stream);
    auto fr = [](void* p) { if (p) (void)hipFree(p); };
    if (!us.alias) { fr(us.y); fr(us.x); fr(us.mu); fr(us.v); fr(us.w); }
    fr(us.dmu); fr(us.d_off); fr(us.d_src_start); fr(us.d_unit_prior); fr(us.d_xb); fr(us.d_scratch); fr(us.d_mu_stash);
    fr(us.d_links);
    us = UnitSet();
}

extern "C" int vlgp_create(int device, int N, int L, int P, int R, const uint8_t* gauss_mask, vlgp_ctx** out) {
    if (!out) return vlgp_fail(nullptr, VLGP_ERR_ARG, "null output handle");
    *out = nullptr;
    if (N < 1 || L < 1 || P < 1 || R < 1) return vlgp_fail(nullptr, VLGP_ERR_ARG, "N, L, P, R must be positive");
    if (R > VLGP_MAX_RANK) return vlgp_fail(nullptr, VLGP_ERR_ARG, "rank %d exceeds VLGP_MAX_RANK=%d", R, VLGP_MAX_RANK);
    if (L > VLGP_MAX_L) return vlgp_fail(nullptr, VLGP_ERR_ARG, "at most %d latents supported, got %d", VLGP_MAX_L, L);
    if (P > VLGP_MAX_XDIM) return vlgp_fail(nullptr, VLGP_ERR_ARG, "at most %d regressors supported, got %d", VLGP_MAX_XDIM, P);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev < 1)
        return vlgp_fail(nullptr, VLGP_ERR_HIP, "no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    if (device < 0 || device >= ndev) return vlgp_fail(nullptr, VLGP_ERR_ARG, "device %d out of range (have %d)", device, ndev);
    vlgp_ctx* ctx = new (std::nothrow) vlgp_ctx();
    if (!ctx) return vlgp_fail(nullptr, VLGP_ERR_HIP, "out of host memory");
    ctx->dev = device; ctx->N = N; ctx->L = L; ctx->P = P; ctx->R = R;
    vlgp_read_switches(ctx);
#define CREATE_CHK(call)                                                                  \
    do {                                                                                  \
        hipError_t e2 = (call);                                                           \
        if (e2 != hipSuccess) {                                                           \
            vlgp_fail(nullptr, VLGP_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e2)); \
            delete ctx;                                                                   \
            return VLGP_ERR_HIP;                                                          \
        }                                                                                 \
    } while (0)
    CREATE_CHK(hipSetDevice(device));
    hipDeviceProp_t prop;
    CREATE_CHK(hipGetDeviceProperties(&prop, device));
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (prop.sharedMemPerBlock > 0) ctx->lds_max = (int)prop.sharedMemPerBlock;
    {   // the main stream carries the latency-critical H-step rounds: give it the highest
        // priority, the M-step lane the lowest, so that M kernels only fill what H leaves idle
        int prio_lo = 0, prio_hi = 0;
        CREATE_CHK(hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi));
        CREATE_CHK(hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, prio_hi));
        CREATE_CHK(hipStreamCreateWithPriority(&ctx->mstream, hipStreamNonBlocking, prio_lo));
    }
    CREATE_CHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    CREATE_CHK(hipEventCreate(&ctx->ev_m_start));
    CREATE_CHK(hipEventCreate(&ctx->ev_m_done));
    CREATE_CHK(hipMalloc(&ctx->d_fail_m, sizeof(int)));
    CREATE_CHK(hipMemset(ctx->d_fail_m, 0, sizeof(int)));
    ctx->gauss.assign(N, 0);
    std::vector<int> gi(N, 0);
    for (int n = 0; n < N; ++n) {
        const uint8_t g = gauss_mask ? gauss_mask[n] : 0;
        ctx->gauss[n] = g ? 1 : 0;
        gi[n] = g ? 1 : 0;
        ctx->n_gauss += g ? 1 : 0;
    }
    CREATE_CHK(hipMalloc(&ctx->d_gauss, sizeof(int) * N));
    CREATE_CHK(hipMemcpy(ctx->d_gauss, gi.data(), sizeof(int) * N, hipMemcpyHostToDevice));
    CREATE_CHK(hipMalloc(&ctx->d_a, sizeof(double) * L * N));
    CREATE_CHK(hipMalloc(&ctx->d_da, sizeof(double) * L * N));
    CREATE_CHK(hipMalloc(&ctx->d_b, sizeof(double) * P * N));
    CREATE_CHK(hipMalloc(&ctx->d_db, sizeof(double) * P * N));
    CREATE_CHK(hipMalloc(&ctx->d_noise, sizeof(double) * N));
    CREATE_CHK(hipMemset(ctx->d_da, 0, sizeof(double) * L * N));
    CREATE_CHK(hipMemset(ctx->d_db, 0, sizeof(double) * P * N));
    CREATE_CHK(hipMalloc(&ctx->d_fail, sizeof(int)));
    CREATE_CHK(hipMemset(ctx->d_fail, 0, sizeof(int)));
#undef CREATE_CHK
    *out = ctx;
    return VLGP_OK;
}

static void free_priors(vlgp_ctx* ctx) {
    ctx->prior_pending.clear();  // (every caller has synchronised the stream; the ranks go with the priors)
    for (auto& kv : ctx->priors) {
        if (kv.second.d_full) (void)hipFree(kv.second.d_full);
        if (kv.second.d_compact) (void)hipFree(kv.second.d_compact);
    }
    ctx->priors.clear();
}

extern "C" int vlgp_destroy(vlgp_ctx* ctx) {
    if (!ctx) return VLGP_OK;
    (void)hipSetDevice(ctx->dev);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->mstream) (void)hipStreamSynchronize(ctx->mstream);
    if (ctx->m_graph_exec) (void)hipGraphExecDestroy(static_cast<hipGraphExec_t>(ctx->m_graph_exec));
    ctx->m_graph_exec = nullptr;
    ctx->m_pending = false;
    prof_drain(ctx);
    hx_close((HxComm*)ctx->hx);
    shm_close_comm((ShmComm*)ctx->shm_m);
    shm_close_comm((ShmComm*)ctx->shm);
    if (ctx->comm_m && g_rccl.destroy) g_rccl.destroy(ctx->comm_m);
    if (ctx->comm && g_rccl.destroy) g_rccl.destroy(ctx->comm);
    // aliased sets first, then owners
    for (int pass = 0; pass < 2; ++pass)
        for (auto& us : ctx->sets)
            if (us.valid && (pass == 1 || us.alias)) free_set(ctx, us);
    free_priors(ctx);
    auto fr = [](void* p) { if (p) (void)hipFree(p); };
    fr(ctx->d_gauss); fr(ctx->d_a); fr(ctx->d_b); fr(ctx->d_noise); fr(ctx->d_da); fr(ctx->d_db);
    fr(ctx->d_hwlm); fr(ctx->d_ecols); fr(ctx->d_fail); fr(ctx->d_fail_m); fr(ctx->d_work_m); fr(ctx->d_clk); fr(ctx->d_work);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_m_start) (void)hipEventDestroy(ctx->ev_m_start);
    if (ctx->ev_m_done) (void)hipEventDestroy(ctx->ev_m_done);
    if (ctx->ev_e_fork) (void)hipEventDestroy(ctx->ev_e_fork);
    for (int i = 0; i < VLGP_E_LANES - 1; ++i) {
        if (ctx->ev_e_join[i]) (void)hipEventDestroy(ctx->ev_e_join[i]);
        if (ctx->elane[i]) (void)hipStreamDestroy(ctx->elane[i]);
    }
    if (ctx->mstream) (void)hipStreamDestroy(ctx->mstream); fr((void*)ctx->d_prior_base); fr(ctx->d_prior_rl); fr(ctx->d_prior_goff);
    if (ctx->ev_e_done) (void)hipEventDestroy(ctx->ev_e_done);
    if (ctx->ev_stage_par) (void)hipEventDestroy(ctx->ev_stage_par);
    if (ctx->ev_stage_map) (void)hipEventDestroy(ctx->ev_stage_map);
    fr(ctx->d_xwork); fr(ctx->d_stage_map);
    if (ctx->h_xres) (void)hipHostFree(ctx->h_xres);
    if (ctx->h_msnap) (void)hipHostFree(ctx->h_msnap);
    if (ctx->h_stage_par) (void)hipHostFree(ctx->h_stage_par);
    if (ctx->h_stage_map) (void)hipHostFree(ctx->h_stage_map);
    if (ctx->h_pinned) (void)hipHostFree(ctx->h_pinned);
    if (ctx->h_hres) (void)hipHostFree(ctx->h_hres);
    if (ctx->h_prior_mb) (void)hipHostFree(ctx->h_prior_mb);
    fr(ctx->d_prior_mb);
    fr(ctx->d_hsync); fr(ctx->d_hmom); fr(ctx->d_hmpart);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return VLGP_OK;
}

extern "C" int vlgp_synchronize(vlgp_ctx* ctx) {
    NEED_CTX(ctx);
    CHK(vlgp_join_m(ctx));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return VLGP_OK;
}

extern "C" int vlgp_synchronize_main(vlgp_ctx* ctx) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return VLGP_OK;
}

extern "C" int vlgp_estep_wait(vlgp_ctx* ctx) {
    NEED_CTX(ctx);
    if (ctx->e_done_valid) HIPCHK(ctx, hipEventSynchronize(ctx->ev_e_done));
    else HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return VLGP_OK;
}

// ---- unit sets -------------------------------------------------------------
static int set_offsets(vlgp_ctx* ctx, UnitSet& us, int M, const int64_t* off) {
    us.M = M;
    us.off.assign(off, off + M + 1);
    us.rows = off[M] - off[0];
    us.Tmax = 0;
    us.Tmin = 0x7fffffff;
    for (int m = 0; m < M; ++m) {
        const int64_t T = off[m + 1] - off[m];
        if (T < 1) return vlgp_fail(ctx, VLGP_ERR_ARG, "unit %d has %lld rows", m, (long long)T);
        us.Tmax = std::max<int>(us.Tmax, (int)T);
        us.Tmin = std::min<int>(us.Tmin, (int)T);
    }
    HIPCHK(ctx, hipMalloc(&us.d_off, sizeof(int64_t) * (M + 1)));
    HIPCHK(ctx, hipMemcpyAsync(us.d_off, us.off.data(), sizeof(int64_t) * (M + 1), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMalloc(&us.d_unit_prior, sizeof(int) * M));
    us.prior_epoch = 0;
    return VLGP_OK;
}

static int up(vlgp_ctx* ctx, double** dst, const double* src, int64_t n) {
    CHK(dev_alloc(ctx, dst, n, src == nullptr));
    if (src) HIPCHK(ctx, hipMemcpyAsync(*dst, src, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    return VLGP_OK;
}

extern "C" int vlgp_upload_units(vlgp_ctx* ctx, int set, int M, const int64_t* offsets, const double* y,
                                 const double* x, const double* mu, const double* v, const double* w) {
    NEED_CTX(ctx);
    ctx->hmom_us = nullptr;  // unit state changes: cached H-step moments are stale
    CHK(vlgp_join_m(ctx));
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, false);
    if (!us) return VLGP_ERR_ARG;
    if (M < 1 || !offsets || !y) return vlgp_fail(ctx, VLGP_ERR_ARG, "upload needs M >= 1, offsets and y");
    if (offsets[0] != 0) return vlgp_fail(ctx, VLGP_ERR_ARG, "offsets[0] must be 0");
    if (!x && ctx->P != 1) return vlgp_fail(ctx, VLGP_ERR_ARG, "x == NULL (all ones) requires xdim == 1");
    for (auto& other : ctx->sets)
        if (other.valid && other.alias && other.parent == set) free_set(ctx, other);
    free_set(ctx, *us);
    CHK(set_offsets(ctx, *us, M, offsets));
    const int64_t rows = us->rows;
    CHK(up(ctx, &us->y, y, rows * ctx->N));
    us->x_ones = (x == nullptr);
    if (x) CHK(up(ctx, &us->x, x, rows * ctx->P * ctx->N));
    CHK(up(ctx, &us->mu, mu, rows * ctx->L));
    CHK(up(ctx, &us->v, v, rows * ctx->L));
    CHK(up(ctx, &us->w, w, rows * ctx->L));
    CHK(dev_alloc(ctx, &us->dmu, rows * ctx->L, true));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // host buffers are the caller's again
    us->valid = true;
    return VLGP_OK;
}

extern "C" int vlgp_cut_units(vlgp_ctx* ctx, int src, int dst, int M_dst, const int64_t* start, int window) {
    NEED_CTX(ctx);
    ctx->hmom_us = nullptr;  // unit state changes: cached H-step moments are stale
    CHK(vlgp_join_m(ctx));
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* s = vlgp_get_set(ctx, src, true);
    UnitSet* d = vlgp_get_set(ctx, dst, false);
    if (!s || !d) return VLGP_ERR_ARG;
    if (src == dst || M_dst < 1 || window < 1 || !start) return vlgp_fail(ctx, VLGP_ERR_ARG, "bad cut arguments");
    if (s->alias) return vlgp_fail(ctx, VLGP_ERR_STATE, "cannot cut a set that is itself a cut");
    bool exact = (int64_t)M_dst * window == s->rows;
    for (int k = 0; k < M_dst; ++k) {
        if (start[k] < 0 || start[k] + window > s->rows)
            return vlgp_fail(ctx, VLGP_ERR_ARG, "segment %d [%lld, +%d) outside the source set", k, (long long)start[k], window);
        if (start[k] != (int64_t)k * window) exact = false;
    }
    free_set(ctx, *d);
    std::vector<int64_t> off(M_dst + 1);
    for (int k = 0; k <= M_dst; ++k) off[k] = (int64_t)k * window;
    CHK(set_offsets(ctx, *d, M_dst, off.data()));
    d->parent = src;
    d->x_ones = s->x_ones;
    d->src_start.assign(start, start + M_dst);
    HIPCHK(ctx, hipMalloc(&d->d_src_start, sizeof(int64_t) * M_dst));
    HIPCHK(ctx, hipMemcpyAsync(d->d_src_start, start, sizeof(int64_t) * M_dst, hipMemcpyHostToDevice, ctx->stream));
    const int64_t rows = d->rows;
    CHK(dev_alloc(ctx, &d->dmu, rows * ctx->L, true));
    if (exact) {
        d->alias = true;
        d->y = s->y; d->x = s->x; d->mu = s->mu; d->v = s->v; d->w = s->w;
    } else {
        d->alias = false;
        CHK(dev_alloc(ctx, &d->y, rows * ctx->N, false));
        if (!s->x_ones) CHK(dev_alloc(ctx, &d->x, rows * ctx->P * ctx->N, false));
        CHK(dev_alloc(ctx, &d->mu, rows * ctx->L, false));
        CHK(dev_alloc(ctx, &d->v, rows * ctx->L, false));
        CHK(dev_alloc(ctx, &d->w, rows * ctx->L, false));
        CHK(launch_gather(ctx, *s, *d, window));
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    d->valid = true;
    return VLGP_OK;
}

extern "C" int vlgp_merge_units(vlgp_ctx* ctx, int cut_set) {
    NEED_CTX(ctx);
    ctx->hmom_us = nullptr;  // unit state changes: cached H-step moments are stale
    CHK(vlgp_join_m(ctx));
    UnitSet* c = vlgp_get_set(ctx, cut_set, true);
    if (!c) return VLGP_ERR_ARG;
    if (c->parent < 0) return vlgp_fail(ctx, VLGP_ERR_STATE, "set %d is not a cut", cut_set);
    if (c->alias) return VLGP_OK;
    UnitSet* p = vlgp_get_set(ctx, c->parent, true);
    if (!p) return VLGP_ERR_STATE;
    return launch_scatter(ctx, *c, *p, (int)(c->off[1] - c->off[0]));
}

extern "C" int vlgp_stash_mu(vlgp_ctx* ctx, int set, int restore) {
    NEED_CTX(ctx);
    CHK(vlgp_join_m(ctx));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    const size_t nb = (size_t)us->rows * ctx->L * sizeof(double);
    if (!restore) {
        if (!us->d_mu_stash) HIPCHK(ctx, hipMalloc(&us->d_mu_stash, nb));
        HIPCHK(ctx, hipMemcpyAsync(us->d_mu_stash, us->mu, nb, hipMemcpyDeviceToDevice, ctx->stream));
        return VLGP_OK;
    }
    if (!us->d_mu_stash) return vlgp_fail(ctx, VLGP_ERR_STATE, "set %d has no stashed mu", set);
    ctx->hmom_us = nullptr;
    HIPCHK(ctx, hipMemcpyAsync(us->mu, us->d_mu_stash, nb, hipMemcpyDeviceToDevice, ctx->stream));
    return VLGP_OK;
}

extern "C" int vlgp_download_units(vlgp_ctx* ctx, int set, double* mu, double* v, double* w, double* dmu) {
    NEED_CTX(ctx);
    CHK(vlgp_join_m(ctx));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    const size_t nb = (size_t)us->rows * ctx->L * sizeof(double);
    if (mu) HIPCHK(ctx, hipMemcpyAsync(mu, us->mu, nb, hipMemcpyDeviceToHost, ctx->stream));
    if (v) HIPCHK(ctx, hipMemcpyAsync(v, us->v, nb, hipMemcpyDeviceToHost, ctx->stream));
    if (w) HIPCHK(ctx, hipMemcpyAsync(w, us->w, nb, hipMemcpyDeviceToHost, ctx->stream));
    if (dmu) HIPCHK(ctx, hipMemcpyAsync(dmu, us->dmu, nb, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return VLGP_OK;
}

extern "C" int vlgp_free_units(vlgp_ctx* ctx, int set) {
    NEED_CTX(ctx);
    ctx->hmom_us = nullptr;  // unit state changes: cached H-step moments are stale
    CHK(vlgp_join_m(ctx));
    UnitSet* us = vlgp_get_set(ctx, set, false);
    if (!us) return VLGP_ERR_ARG;
    for (auto& other : ctx->sets)
        if (other.valid && other.alias && other.parent == set) free_set(ctx, other);
    free_set(ctx, *us);
    return VLGP_OK;
}

// ---- parameters ------------------------------------------------------------
extern "C" int vlgp_set_params(vlgp_ctx* ctx, const double* a, const double* b, const double* noise) {
    NEED_CTX(ctx);
    CHK(vlgp_join_m(ctx));
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    const int N = ctx->N, L = ctx->L, P = ctx->P;
    // through a pinned buffer of its own: the copies are truly asynchronous and nothing waits for them here (the E-step's
    // launches follow on the same stream); the buffer is reused only after the event behind the last copy has fired
    if (!ctx->h_stage_par) {
        HIPCHK(ctx, hipHostMalloc(&ctx->h_stage_par, sizeof(double) * ((size_t)(L + P) * N + N), hipHostMallocDefault));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_stage_par, hipEventDisableTiming));
    }
    if (ctx->stage_par_busy) HIPCHK(ctx, hipEventSynchronize(ctx->ev_stage_par));
    double* ha = ctx->h_stage_par;
    double* hb = ha + (size_t)L * N;
    double* hn = hb + (size_t)P * N;
    if (a) {
        memcpy(ha, a, sizeof(double) * L * N);
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_a, ha, sizeof(double) * L * N, hipMemcpyHostToDevice, ctx->stream));
    }
    if (b) {
        memcpy(hb, b, sizeof(double) * P * N);
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_b, hb, sizeof(double) * P * N, hipMemcpyHostToDevice, ctx->stream));
    }
    if (noise) {
        memcpy(hn, noise, sizeof(double) * N);
        HIPCHK(ctx, hipMemcpyAsync(ctx->d_noise, hn, sizeof(double) * N, hipMemcpyHostToDevice, ctx->stream));
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_stage_par, ctx->stream));
    ctx->stage_par_busy = true;
    ctx->msnap_valid = false;
    if (a && b && noise) ctx->have_params = true;
    return VLGP_OK;
}

extern "C" int vlgp_get_params(vlgp_ctx* ctx, double* a, double* b, double* noise, double* da, double* db) {
    NEED_CTX(ctx);
    CHK(vlgp_join_m(ctx));
    const int N = ctx->N, L = ctx->L, P = ctx->P;
    if (ctx->msnap_valid) {  // the M-step lane left them in pinned memory behind its last kernel (vlgp_mstep_begin)
        const double* h = ctx->h_msnap;
        const size_t ln[5] = {(size_t)L * N, (size_t)P * N, (size_t)N, (size_t)L * N, (size_t)P * N};
        double* out[5] = {a, b, noise, da, db};
        for (int i = 0; i < 5; ++i) {
            if (out[i]) memcpy(out[i], h, sizeof(double) * ln[i]);
            h += ln[i];
        }
        return VLGP_OK;
    }
    // five small copies into pageable memory are five synchronous staged transfers (~90 us per EM iteration): gather
    // them in the pinned buffer with truly asynchronous copies, one synchronisation, then hand them out
    double* dst[5] = {a, b, noise, da, db};
    const double* src[5] = {ctx->d_a, ctx->d_b, ctx->d_noise, ctx->d_da, ctx->d_db};
    const size_t len[5] = {(size_t)L * N, (size_t)P * N, (size_t)N, (size_t)L * N, (size_t)P * N};
    size_t total = 0;
    for (int i = 0; i < 5; ++i) total += dst[i] ? len[i] : 0;
    CHK(vlgp_ensure_pinned(ctx, (int64_t)total + 8));
    size_t o = 0;
    for (int i = 0; i < 5; ++i) {
        if (!dst[i]) continue;
        HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned + o, src[i], sizeof(double) * len[i], hipMemcpyDeviceToHost, ctx->stream));
        o += len[i];
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    o = 0;
    for (int i = 0; i < 5; ++i) {
        if (!dst[i]) continue;
        memcpy(dst[i], ctx->h_pinned + o, sizeof(double) * len[i]);
        o += len[i];
    }
    return VLGP_OK;
}

// ---- prior -----------------------------------------------------------------
static int rebuild_prior_table(vlgp_ctx* ctx) {
    CHK(vlgp_prior_collect(ctx));
    const int L = ctx->L;
    const int rows = (int)ctx->priors.size();
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_prior_base) (void)hipFree((void*)ctx->d_prior_base);
    if (ctx->d_prior_rl) (void)hipFree(ctx->d_prior_rl);
    if (ctx->d_prior_goff) (void)hipFree(ctx->d_prior_goff);
    ctx->d_prior_base = nullptr; ctx->d_prior_rl = nullptr; ctx->d_prior_goff = nullptr;
    ctx->prior_rows = rows;
    ctx->prior_epoch++;
    if (rows == 0) return VLGP_OK;
    std::vector<const double*> base(rows);
    std::vector<int> rl(rows * L);
    std::vector<int64_t> goff(rows * L);
    int i = 0;
    for (auto& kv : ctx->priors) {
        Prior& pr = kv.second;
        pr.index = i;
        base[i] = pr.d_compact;
        for (int l = 0; l < L; ++l) {
            rl[i * L + l] = pr.rl[l];
            goff[i * L + l] = pr.goff[l];
        }
        ++i;
    }
    HIPCHK(ctx, hipMalloc((void**)&ctx->d_prior_base, sizeof(double*) * rows));
    HIPCHK(ctx, hipMalloc(&ctx->d_prior_rl, sizeof(int) * rows * L));
    HIPCHK(ctx, hipMalloc(&ctx->d_prior_goff, sizeof(int64_t) * rows * L));
    HIPCHK(ctx, hipMemcpy((void*)ctx->d_prior_base, base.data(), sizeof(double*) * rows, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_prior_rl, rl.data(), sizeof(int) * rows * L, hipMemcpyHostToDevice));
    HIPCHK(ctx, hipMemcpy(ctx->d_prior_goff, goff.data(), sizeof(int64_t) * rows * L, hipMemcpyHostToDevice));
    return VLGP_OK;
}

static int new_prior(vlgp_ctx* ctx, int T, Prior** out) {
    if (T < 1) return vlgp_fail(ctx, VLGP_ERR_ARG, "prior length must be positive");
    auto it = ctx->priors.find(T);
    if (it != ctx->priors.end()) {
        *out = &it->second;
        return VLGP_OK;
    }
    Prior pr;
    pr.T = T;
    const int64_t n = (int64_t)ctx->L * T * ctx->R;
    HIPCHK(ctx, hipMalloc(&pr.d_full, (size_t)n * sizeof(double)));
    HIPCHK(ctx, hipMalloc(&pr.d_compact, (size_t)n * sizeof(double)));
    // capacity layout of the compact copy: latent l at l*T*R whatever the ranks turn out to be, so that the
    // factorisation kernel can write it (and the table never needs the ranks of the other latents)
    pr.goff.resize(ctx->L);
    for (int l = 0; l < ctx->L; ++l) pr.goff[l] = (int64_t)l * T * ctx->R;
    pr.compact_len = n;
    pr.rl.assign(ctx->L, 1);
    auto res = ctx->priors.emplace(T, pr);
    *out = &res.first->second;
    return VLGP_OK;
}

extern "C" int vlgp_clear_prior(vlgp_ctx* ctx) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    free_priors(ctx);
    return rebuild_prior_table(ctx);
}

extern "C" int vlgp_build_prior(vlgp_ctx* ctx, int n_lengths, const int* lengths, const double* omega,
                                const double* sigma) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    if (n_lengths < 1 || !lengths || !omega || !sigma) return vlgp_fail(ctx, VLGP_ERR_ARG, "bad build_prior arguments");
    CHK(vlgp_prior_collect(ctx));
    // the reference replaces the whole dict on every call (gp.py:158).  Same set of lengths as the table holds
    // (every H-step): the factors are rebuilt in place and the kernel refreshes the ranks of the table rows --
    // no allocation, no copy.  Otherwise: drop the lengths not listed, add the new ones, rebuild the table.
    bool same = (int)ctx->priors.size() > 0 && ctx->d_prior_rl != nullptr;
    for (int i = 0; same && i < n_lengths; ++i) same = ctx->priors.count(lengths[i]) > 0;
    for (auto it = ctx->priors.begin(); same && it != ctx->priors.end(); ++it)
        same = std::find(lengths, lengths + n_lengths, it->first) != lengths + n_lengths;
    if (!same) {
        for (auto it = ctx->priors.begin(); it != ctx->priors.end();) {
            if (std::find(lengths, lengths + n_lengths, it->first) == lengths + n_lengths) {
                HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
                if (it->second.d_full) (void)hipFree(it->second.d_full);
                if (it->second.d_compact) (void)hipFree(it->second.d_compact);
                it = ctx->priors.erase(it);
            } else {
                ++it;
            }
        }
    }
    std::vector<Prior*> prs;
    for (int i = 0; i < n_lengths; ++i) {
        Prior* pr = nullptr;
        CHK(new_prior(ctx, lengths[i], &pr));
        if (std::find(prs.begin(), prs.end(), pr) == prs.end()) prs.push_back(pr);
    }
    // same lengths as the table holds (the rebuild of every EM iteration): nothing here needs the ranks on the host --
    // the next consumer of Prior::rl (an E-step dispatch, vlgp_get_prior) takes them
    CHK(launch_ichol_all(ctx, prs, omega, sigma, same, same));
    return same ? VLGP_OK : rebuild_prior_table(ctx);
}

extern "C" int vlgp_set_prior(vlgp_ctx* ctx, int T, const double* G) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    if (!G) return vlgp_fail(ctx, VLGP_ERR_ARG, "null G");
    CHK(vlgp_prior_collect(ctx));
    Prior* pr = nullptr;
    CHK(new_prior(ctx, T, &pr));
    const int64_t n = (int64_t)ctx->L * T * ctx->R;
    HIPCHK(ctx, hipMemcpyAsync(pr->d_full, G, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    CHK(launch_compact_prior(ctx, *pr));
    return rebuild_prior_table(ctx);
}

extern "C" int vlgp_get_prior(vlgp_ctx* ctx, int T, double* G, int* rank_out) {
    NEED_CTX(ctx);
    CHK(vlgp_prior_collect(ctx));
    auto it = ctx->priors.find(T);
    if (it == ctx->priors.end()) return vlgp_fail(ctx, VLGP_ERR_STATE, "no prior factor for length %d", T);
    const int64_t n = (int64_t)ctx->L * T * ctx->R;
    if (G) {
        HIPCHK(ctx, hipMemcpyAsync(G, it->second.d_full, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (rank_out)
        for (int l = 0; l < ctx->L; ++l) rank_out[l] = it->second.rl[l];
    return VLGP_OK;
}

int vlgp_bind_priors(vlgp_ctx* ctx, UnitSet& us) {
    if (us.prior_epoch == ctx->prior_epoch) return VLGP_OK;
    std::vector<int> idx(us.M);
    for (int m = 0; m < us.M; ++m) {
        const int T = (int)(us.off[m + 1] - us.off[m]);
        auto it = ctx->priors.find(T);
        if (it == ctx->priors.end())
            return vlgp_fail(ctx, VLGP_ERR_STATE, "no prior factor for unit length %d (call build_prior/set_prior)", T);
        idx[m] = it->second.index;
    }
    HIPCHK(ctx, hipMemcpy(us.d_unit_prior, idx.data(), sizeof(int) * us.M, hipMemcpyHostToDevice));
    us.prior_epoch = ctx->prior_epoch;
    return VLGP_OK;
}

int vlgp_refresh_xb(vlgp_ctx* ctx, UnitSet& us) {
    if (us.x_ones) return VLGP_OK;
    if (!us.d_xb) HIPCHK(ctx, hipMalloc(&us.d_xb, (size_t)us.rows * ctx->N * sizeof(double)));
    return launch_xb(ctx, us);
}

// ---- E / M / H ---------------------------------------------------------------
static int begin_count(vlgp_ctx* ctx) {
    HIPCHK(ctx, hipMemsetAsync(ctx->d_fail, 0, sizeof(int), ctx->stream));
    return VLGP_OK;
}
static int end_count(vlgp_ctx* ctx, int* n_failed) {
    if (!n_failed) return VLGP_OK;
    HIPCHK(ctx, hipMemcpyAsync(n_failed, ctx->d_fail, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return VLGP_OK;
}
#define NEED_PARAMS(ctx) \
    if (!(ctx)->have_params) return vlgp_fail(ctx, VLGP_ERR_STATE, "parameters not set (vlgp_set_params)")

extern "C" int vlgp_update_w(vlgp_ctx* ctx, int set) {
    NEED_CTX(ctx);
    CHK(vlgp_prior_collect(ctx));
    ctx->hmom_us = nullptr;  // unit state changes: cached H-step moments are stale
    CHK(vlgp_join_m(ctx));
    NEED_PARAMS(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    return launch_estep(ctx, *us, EM_W, 0, 0.0, 0);
}

extern "C" int vlgp_update_v(vlgp_ctx* ctx, int set, int vb, int* n_failed) {
    NEED_CTX(ctx);
    CHK(vlgp_prior_collect(ctx));
    CHK(vlgp_join_m(ctx));
    NEED_PARAMS(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    if (n_failed) *n_failed = 0;
    if (!vb) return VLGP_OK;
    CHK(begin_count(ctx));
    CHK(launch_estep(ctx, *us, EM_FACTOR0 | EM_V, 0, 0.0, 1));
    return end_count(ctx, n_failed);
}

// core.estep over overlapping segments, as the reference's sequential loop over VIEWS does it (vlgp/core.py:123-126 with
// vlgp/util.py:482-496): stage by stage; before a stage the shared rows (mu while it is still shared, v) come over from
// the previous stage's units, after it they go back, so that both copies of a shared row always hold what the
// reference's single row holds.  A stage is a contiguous range of equal-length units: a view of the set.
static int estep_staged(vlgp_ctx* ctx, UnitSet& us, int mode, int n_iter, double dmu_bound, int vb) {
    const int W = us.Tmax, N = ctx->N, L = ctx->L, P = ctx->P;
    CHK(vlgp_bind_priors(ctx, us));
    const int n_stages = (int)us.stage_start.size() - 1;
    for (int s = 0; s < n_stages; ++s) {
        const int u0 = us.stage_start[s], cnt = us.stage_start[s + 1] - u0;
        if (cnt < 1) continue;
        CHK(launch_links_copy(ctx, us, us.link_start[s], us.link_start[s + 1], 0));
        UnitSet v;
        v.valid = true; v.M = cnt; v.rows = (int64_t)cnt * W; v.Tmax = v.Tmin = W;
        v.off.resize(cnt + 1);
        for (int m = 0; m <= cnt; ++m) v.off[m] = (int64_t)m * W;
        v.d_off = us.d_off;  // 0, W, 2 W, ...: the first cnt + 1 entries serve any range of equal-length units
        const int64_t r0 = (int64_t)u0 * W;
        v.y = us.y + r0 * N; v.x = us.x ? us.x + r0 * P * N : nullptr;
        v.mu = us.mu + r0 * L; v.v = us.v + r0 * L; v.w = us.w + r0 * L; v.dmu = us.dmu + r0 * L;
        v.x_ones = us.x_ones; v.alias = true;  // (owns nothing)
        v.d_unit_prior = us.d_unit_prior + u0; v.prior_epoch = us.prior_epoch;
        if (!us.x_ones) {
            if (!us.d_xb) HIPCHK(ctx, hipMalloc(&us.d_xb, (size_t)us.rows * N * sizeof(double)));
            v.d_xb = us.d_xb + r0 * N;
        }
        v.d_scratch = us.d_scratch; v.scratch_len = us.scratch_len;  // the stages run one after the other
        const int rc = launch_estep(ctx, v, mode, n_iter, dmu_bound, vb);
        us.d_scratch = v.d_scratch; us.scratch_len = v.scratch_len;  // (re)allocated inside: stays with the set
        if (rc != VLGP_OK) return rc;
        CHK(launch_links_copy(ctx, us, us.link_start[s], us.link_start[s + 1], 1));
    }
    return VLGP_OK;
}

extern "C" int vlgp_estep(vlgp_ctx* ctx, int set, int n_iter, double dmu_bound, int vb, int* n_failed) {
    NEED_CTX(ctx);
    CHK(vlgp_prior_collect(ctx));
    ctx->hmom_us = nullptr;  // unit state changes: cached H-step moments are stale
    CHK(vlgp_join_m(ctx));
    NEED_PARAMS(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    if (n_failed) *n_failed = 0;
    if (n_iter < 1) return VLGP_OK;  // core.py:24-25
    if (!(dmu_bound > 0)) return vlgp_fail(ctx, VLGP_ERR_ARG, "dmu_bound must be positive");
    CHK(begin_count(ctx));
    const int mode = EM_FACTOR0 | EM_MEAN | EM_W | (vb ? EM_V : 0);
    if (us->stage_start.size() > 2) CHK(estep_staged(ctx, *us, mode, n_iter, dmu_bound, vb ? 1 : 0));
    else CHK(launch_estep(ctx, *us, mode, n_iter, dmu_bound, vb ? 1 : 0));
    if (!ctx->ev_e_done) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_e_done, hipEventDisableTiming));
    HIPCHK(ctx, hipEventRecord(ctx->ev_e_done, ctx->stream));  // (vlgp_estep_wait: this call's launches, not what is queued behind)
    ctx->e_done_valid = true;
    return end_count(ctx, n_failed);
}

extern "C" int vlgp_set_overlaps(vlgp_ctx* ctx, int set, int n_stages, const int* stage_start, int n_links,
                                 const int* links, const int* link_start) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    if (us->alias || us->Tmin != us->Tmax)
        return vlgp_fail(ctx, VLGP_ERR_STATE, "overlaps belong to a copied cut of equal-length units");
    if (n_stages < 1 || !stage_start || stage_start[0] != 0 || stage_start[n_stages] != us->M || n_links < 0 ||
        (n_links > 0 && (!links || !link_start)))
        return vlgp_fail(ctx, VLGP_ERR_ARG, "bad overlap description");
    for (int s = 0; s < n_stages; ++s)
        if (stage_start[s + 1] < stage_start[s]) return vlgp_fail(ctx, VLGP_ERR_ARG, "stages must be ordered");
    for (int k = 0; k < n_links; ++k) {
        const int a = links[3 * k], b = links[3 * k + 1], o = links[3 * k + 2];
        if (a < 0 || a >= us->M || b < 0 || b >= us->M || a == b || o < 1 || o >= us->Tmax)
            return vlgp_fail(ctx, VLGP_ERR_ARG, "bad overlap link %d", k);
    }
    if (n_links > 0) {
        // link_start[s] .. link_start[s + 1] index `links` for stage s: they go to the copy kernels unchecked afterwards
        if (link_start[0] != 0 || link_start[n_stages] != n_links)
            return vlgp_fail(ctx, VLGP_ERR_ARG, "link_start must run from 0 to n_links");
        for (int s = 0; s < n_stages; ++s) {
            if (link_start[s + 1] < link_start[s]) return vlgp_fail(ctx, VLGP_ERR_ARG, "link_start must be non-decreasing");
            for (int k = link_start[s]; k < link_start[s + 1]; ++k) {
                const int b = links[3 * k + 1];  // the second unit of a link lies in the stage the link is filed under
                if (b < stage_start[s] || b >= stage_start[s + 1])
                    return vlgp_fail(ctx, VLGP_ERR_ARG, "overlap link %d is filed under stage %d but its second unit is not in it", k, s);
            }
        }
    }
    us->stage_start.assign(stage_start, stage_start + n_stages + 1);
    us->link_start.assign(n_links > 0 ? link_start : stage_start, (n_links > 0 ? link_start : stage_start) + n_stages + 1);
    if (n_links == 0) us->link_start.assign(n_stages + 1, 0);
    if (us->d_links) HIPCHK(ctx, hipFree(us->d_links));
    us->d_links = nullptr;
    us->n_links = n_links;
    us->share_mu = true;
    if (n_links > 0) {
        HIPCHK(ctx, hipMalloc(&us->d_links, sizeof(int) * 3 * n_links));
        HIPCHK(ctx, hipMemcpy(us->d_links, links, sizeof(int) * 3 * n_links, hipMemcpyHostToDevice));
    }
    return VLGP_OK;
}

extern "C" int vlgp_unshare_mu(vlgp_ctx* ctx, int set) {
    NEED_CTX(ctx);
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    us->share_mu = false;
    return VLGP_OK;
}

extern "C" int vlgp_mstep_begin(vlgp_ctx* ctx, int set, int n_iter, int use_hessian, double eps, double lr,
                                double da_bound, double db_bound) {
    NEED_CTX(ctx);
    NEED_PARAMS(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    if (ctx->m_pending) return vlgp_fail(ctx, VLGP_ERR_STATE, "an M-step is already in flight (call vlgp_mstep_end)");
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    if (!(da_bound > 0) || !(db_bound > 0)) return vlgp_fail(ctx, VLGP_ERR_ARG, "da_bound/db_bound must be positive");
    // fork: the M-step lane starts after everything already queued on the main stream
    HIPCHK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->mstream, ctx->ev_fork, 0));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_fail_m, 0, sizeof(int), ctx->mstream));
    HIPCHK(ctx, hipEventRecord(ctx->ev_m_start, ctx->mstream));
    ctx->msnap_valid = false;
    if (n_iter >= 1)  // core.py:131-133
        CHK(launch_mstep(ctx, *us, n_iter, use_hessian, eps, lr, da_bound, db_bound));
    {   // what the caller pulls afterwards (vlgp_get_params, the failure count), written out by the lane itself
        const int N = ctx->N, L = ctx->L, P = ctx->P;
        if (!ctx->h_msnap) {
            HIPCHK(ctx, hipHostMalloc(&ctx->h_msnap, sizeof(double) * (2 * (size_t)(L + P) * N + N + 8), hipHostMallocMapped));
            HIPCHK(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_msnap), ctx->h_msnap, 0));
        }
        CHK(launch_snapshot_params(ctx, ctx->mstream, ctx->d_msnap));
        ctx->msnap_valid = true;  // (read only after vlgp_join_m; any vlgp_set_params in between clears it)
    }
    HIPCHK(ctx, hipEventRecord(ctx->ev_m_done, ctx->mstream));
    // NO device-side join here: the H-step rounds and the prior rebuild queued on the main stream meanwhile
    // touch neither a, b nor anything the M-step writes, and every entry point that does joins on the host
    // first (vlgp_join_m waits for ev_m_done) -- a stream wait here would serialise the H-step behind the M-step
    ctx->m_pending = true;
    return VLGP_OK;
}

extern "C" int vlgp_mstep_end(vlgp_ctx* ctx, int* n_failed, double* device_ms) {
    NEED_CTX(ctx);
    if (!ctx->m_pending) return vlgp_fail(ctx, VLGP_ERR_STATE, "no M-step in flight");
    CHK(vlgp_join_m(ctx));
    ctx->m_pending = false;
    {
        float ms = 0.f;
        HIPCHK(ctx, hipEventElapsedTime(&ms, ctx->ev_m_start, ctx->ev_m_done));
        ctx->last_m_ms = ms;  // (vlgp_hstep_begin weighs it against the H-step bracket's duration)
        if (device_ms) *device_ms = ms;
    }
    if (n_failed) {
        if (ctx->msnap_valid) {
            const int N = ctx->N, L = ctx->L, P = ctx->P;
            memcpy(n_failed, ctx->h_msnap + 2 * (size_t)(L + P) * N + N, sizeof(int));
        } else {
            HIPCHK(ctx, hipMemcpy(n_failed, ctx->d_fail_m, sizeof(int), hipMemcpyDeviceToHost));
        }
    }
    return VLGP_OK;
}

extern "C" int vlgp_mstep(vlgp_ctx* ctx, int set, int n_iter, int use_hessian, double eps, double lr,
                          double da_bound, double db_bound, int* n_failed) {
    NEED_CTX(ctx);
    if (n_failed) *n_failed = 0;
    if (n_iter < 1) return VLGP_OK;  // core.py:131-133
    CHK(vlgp_mstep_begin(ctx, set, n_iter, use_hessian, eps, lr, da_bound, db_bound));
    return vlgp_mstep_end(ctx, n_failed, nullptr);
}

extern "C" int vlgp_hstep_objective(vlgp_ctx* ctx, int set, int window, double dt, int n_eval, const int* latent,
                                    const double* logp, double* ll, double* dll) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    if (n_eval < 1 || !latent || !logp || !ll || !dll) return vlgp_fail(ctx, VLGP_ERR_ARG, "bad hstep arguments");
    return launch_hstep(ctx, *us, window, dt, n_eval, latent, logp, ll, dll);
}

extern "C" int vlgp_hstep_prepare(vlgp_ctx* ctx, int set, int window) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    ctx->hprep = false;
    if (ctx->hmom_bracket || window < 1) return VLGP_OK;
    // On the MAIN stream, behind everything queued so far -- the call may come while the E-step still runs (the host
    // then waits for the E-step alone: vlgp_estep_wait).  (Measured: the same work on the side stream, enqueued during the
    // E-step, makes a fourth busy queue beside the E-step's two lanes and the M-step lane -- E-step 2.2 -> 4.0 ms, the
    // queue pathology of DESIGN 4.1.)
    CHK(hstep_prepare(ctx, *us, window, ctx->stream));
    ctx->hprep = ctx->hmom_us == us;
    ctx->hprep_epoch = ctx->write_epoch;
    return VLGP_OK;
}

extern "C" int vlgp_hstep_begin(vlgp_ctx* ctx, int set, int window) {
    NEED_CTX(ctx);
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    ctx->hmom_bracket = true;
    // the first objective call inside the bracket builds the moments and the w copy -- unless vlgp_hstep_prepare did,
    // for this set and window, and no entry point that may change the units ran since
    if (!(ctx->hprep && ctx->hprep_epoch == ctx->write_epoch && ctx->hmom_us == us && ctx->hmom_T == window)) {
        ctx->hmom_us = nullptr;
        ctx->hwlm_valid = false;
    }
    ctx->hprep = false;
    // the rounds' waves take the high instruction priority unless the M-step lane was the longer one last time
    ctx->h_prio = (ctx->last_m_ms > 0.0 && ctx->last_h_ms > 0.0 && ctx->last_m_ms > ctx->last_h_ms) ? 0 : 1;
    ctx->h_t0 = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
    return VLGP_OK;
}

extern "C" int vlgp_hstep_end(vlgp_ctx* ctx) {
    NEED_CTX(ctx);
    if (ctx->hmom_bracket && ctx->h_t0 > 0.0)
        ctx->last_h_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count() - ctx->h_t0;
    ctx->hmom_bracket = false;
    ctx->hmom_us = nullptr;
    ctx->hwlm_valid = false;
    return VLGP_OK;
}

// ---- constraints / norms -----------------------------------------------------
extern "C" int vlgp_apply_latent_map(vlgp_ctx* ctx, int set, const double* map, const double* shift) {
    NEED_CTX(ctx);
    ctx->hmom_us = nullptr;  // unit state changes: cached H-step moments are stale
    CHK(vlgp_join_m(ctx));
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us || !map) return vlgp_fail(ctx, VLGP_ERR_ARG, "bad latent map arguments");
    const int L = ctx->L;
    // staged through a pinned and a device buffer of its own, reused after the event behind the last kernel that read
    // them: nothing waits here (this is the first call of every EM iteration, constrain_loading)
    if (!ctx->h_stage_map) {
        HIPCHK(ctx, hipHostMalloc(&ctx->h_stage_map, sizeof(double) * (L * L + L + 8), hipHostMallocDefault));
        HIPCHK(ctx, hipMalloc(&ctx->d_stage_map, sizeof(double) * (L * L + L + 8)));
        HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ev_stage_map, hipEventDisableTiming));
    }
    if (ctx->stage_map_busy) HIPCHK(ctx, hipEventSynchronize(ctx->ev_stage_map));
    memcpy(ctx->h_stage_map, map, sizeof(double) * L * L);
    if (shift) memcpy(ctx->h_stage_map + L * L, shift, sizeof(double) * L);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_stage_map, ctx->h_stage_map, sizeof(double) * (L * L + L), hipMemcpyHostToDevice, ctx->stream));
    const double* d_map = ctx->d_stage_map;
    const double* d_shift = shift ? ctx->d_stage_map + L * L : nullptr;
    CHK(launch_latent_map(ctx, *us, d_map, d_shift));
    // the set's units are independent copies: an IN-PLACE constraint of the reference (everything but "svd", which
    // rebinds mu) visits a row shared by two overlapping segments twice -- the caller says so with vlgp_unshare_mu
    CHK(launch_links_map(ctx, *us, d_map, d_shift));
    HIPCHK(ctx, hipEventRecord(ctx->ev_stage_map, ctx->stream));
    ctx->stage_map_busy = true;
    return VLGP_OK;
}

static int moments_host(vlgp_ctx* ctx, UnitSet& us, std::vector<double>& out) {
    const int L = ctx->L;
    const int K = L * (L + 1) / 2 + 3 * L + 1;
    CHK(vlgp_ensure_pinned(ctx, K + 8));
    CHK(launch_moments(ctx, us));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned, ctx->d_work, sizeof(double) * K, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    out.assign(ctx->h_pinned, ctx->h_pinned + K);
    return VLGP_OK;
}

extern "C" int vlgp_norms_begin(vlgp_ctx* ctx, int set);
extern "C" int vlgp_norms_end(vlgp_ctx* ctx, double out[2]);
extern "C" int vlgp_norms(vlgp_ctx* ctx, int set, double out[2]) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    if (ctx->world == 1) {  // one rank: the kernel of the two-halves form (the same sums, bit for bit)
        CHK(vlgp_norms_begin(ctx, set));
        return vlgp_norms_end(ctx, out);
    }
    std::vector<double> m;
    CHK(moments_host(ctx, *us, m));
    const int L = ctx->L, t = L * (L + 1) / 2;
    double s = 0.0;
    for (int l = 0; l < L; ++l) s += m[t + 2 * L + l];
    out[0] = s;
    out[1] = m[t + 3 * L];
    return VLGP_OK;
}

extern "C" int vlgp_norms_begin(vlgp_ctx* ctx, int set) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    CHK(wait_norms(ctx));  // (a pass never collected: dropped)
    ctx->x_pending = 0;
    ctx->x_set = set;
    if (ctx->world > 1) {  // the sums cross ranks on the main lane's communicator: computed when they are collected
        ctx->x_pending = 2;
        return VLGP_OK;
    }
    if (!ctx->h_xres) {
        HIPCHK(ctx, hipMalloc(&ctx->d_xwork, sizeof(double) * (2 * 1024 + 8)));
        HIPCHK(ctx, hipMemsetAsync(ctx->d_xwork, 0, sizeof(double) * (2 * 1024 + 8), ctx->stream));
        HIPCHK(ctx, hipHostMalloc(&ctx->h_xres, sizeof(double) * 8, hipHostMallocMapped));
        memset(ctx->h_xres, 0, sizeof(double) * 8);
        HIPCHK(ctx, hipHostGetDevicePointer(reinterpret_cast<void**>(&ctx->d_xres), ctx->h_xres, 0));
    }
    ++ctx->x_seq;
    CHK(launch_norms(ctx, *us, ctx->d_xwork, reinterpret_cast<unsigned*>(ctx->d_xwork + 2 * 1024), ctx->d_xres, ctx->x_seq));
    ctx->x_pending = 1;
    return VLGP_OK;
}

extern "C" int vlgp_norms_end(vlgp_ctx* ctx, double out[2]) {
    NEED_CTX(ctx);
    if (!ctx->x_pending) return vlgp_fail(ctx, VLGP_ERR_STATE, "no norms pass pending (vlgp_norms_begin)");
    if (ctx->x_pending == 2) {
        ctx->x_pending = 0;
        return vlgp_norms(ctx, ctx->x_set, out);
    }
    CHK(wait_norms(ctx));
    ctx->x_pending = 0;
    out[0] = ctx->h_xres[0];
    out[1] = ctx->h_xres[1];
    return VLGP_OK;
}

extern "C" int vlgp_latent_moments(vlgp_ctx* ctx, int set, double* sum1, double* sum2, double* count) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    std::vector<double> m;
    CHK(moments_host(ctx, *us, m));
    const int L = ctx->L, t = L * (L + 1) / 2;
    for (int l = 0; l < L; ++l) {
        if (sum1) sum1[l] = m[t + l];
        if (sum2) sum2[l] = m[t + 2 * L + l];
    }
    if (count) {
        double c = (double)us->rows;
        if (ctx->world > 1) {
            CHK(vlgp_ensure_work(ctx, 8));
            HIPCHK(ctx, hipMemcpy(ctx->d_work, &c, sizeof(double), hipMemcpyHostToDevice));
            CHK(vlgp_allreduce(ctx, ctx->d_work, 1));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            HIPCHK(ctx, hipMemcpy(&c, ctx->d_work, sizeof(double), hipMemcpyDeviceToHost));
        }
        *count = c;
    }
    return VLGP_OK;
}

extern "C" int vlgp_project_units(vlgp_ctx* ctx, int set, const double* proj, const double* shift, double* colsum) {
    NEED_CTX(ctx);
    ctx->hmom_us = nullptr;  // mu changes
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    CHK(vlgp_join_m(ctx));
    UnitSet* us = vlgp_get_set(ctx, set, true);
    if (!us) return VLGP_ERR_ARG;
    if (!proj || !shift) return vlgp_fail(ctx, VLGP_ERR_ARG, "null projection");
    if (us->alias) return vlgp_fail(ctx, VLGP_ERR_STATE, "project the owning set, not an aliased cut");
    const int N = ctx->N, L = ctx->L;
    const int64_t G = (us->rows + 63) / 64;
    const int64_t o_proj = 0, o_shift = (int64_t)N * L, o_out = o_shift + L + (L & 1), o_part = o_out + N + (N & 1);
    CHK(vlgp_ensure_work(ctx, o_part + G * N));
    CHK(vlgp_ensure_pinned(ctx, o_out + N + 8));
    memcpy(ctx->h_pinned, proj, sizeof(double) * N * L);
    memcpy(ctx->h_pinned + o_shift, shift, sizeof(double) * L);
    HIPCHK(ctx, hipMemcpyAsync(ctx->d_work, ctx->h_pinned, sizeof(double) * (o_shift + L), hipMemcpyHostToDevice,
                               ctx->stream));
    CHK(launch_project(ctx, *us, ctx->d_work + o_proj, ctx->d_work + o_shift, ctx->d_work + o_part,
                       ctx->d_work + o_out));
    HIPCHK(ctx, hipMemcpyAsync(ctx->h_pinned + o_out, ctx->d_work + o_out, sizeof(double) * N, hipMemcpyDeviceToHost,
                               ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (colsum) memcpy(colsum, ctx->h_pinned + o_out, sizeof(double) * N);
    return VLGP_OK;
}

// ---- posterior draws -----------------------------------------------------------
extern "C" int vlgp_sample_posterior(vlgp_ctx* ctx, int T, const double* mu, const double* w, const double* G, int nsamples,
                                     const double* eps, double* out, int* n_failed) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    if (T < 1 || nsamples < 1 || !mu || !w || !G || !eps || !out) return vlgp_fail(ctx, VLGP_ERR_ARG, "bad sample_posterior arguments");
    const int L = ctx->L, R = ctx->R;
    const int64_t n_mu = (int64_t)T * L, n_G = (int64_t)L * T * R, n_eps = (int64_t)L * R * nsamples;
    const int64_t n_out = (int64_t)nsamples * T * L;
    const int64_t o_mu = 0, o_w = o_mu + n_mu, o_G = o_w + n_mu, o_eps = o_G + n_G, o_z = o_eps + n_eps, o_out = o_z + n_eps;
    CHK(vlgp_ensure_work(ctx, o_out + n_out));
    double* W = ctx->d_work;
    if (n_failed) *n_failed = 0;
    CHK(begin_count(ctx));
    HIPCHK(ctx, hipMemcpyAsync(W + o_mu, mu, sizeof(double) * n_mu, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(W + o_w, w, sizeof(double) * n_mu, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(W + o_G, G, sizeof(double) * n_G, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(W + o_eps, eps, sizeof(double) * n_eps, hipMemcpyHostToDevice, ctx->stream));
    CHK(launch_sample_posterior(ctx, T, nsamples, W + o_mu, W + o_w, W + o_G, W + o_eps, W + o_z, W + o_out));
    HIPCHK(ctx, hipMemcpyAsync(out, W + o_out, sizeof(double) * n_out, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return end_count(ctx, n_failed);
}

// ---- measurement -------------------------------------------------------------
extern "C" int vlgp_debug_last_estep_path(vlgp_ctx* ctx, int* path) {
    NEED_CTX(ctx);
    if (!path) return vlgp_fail(ctx, VLGP_ERR_ARG, "null path");
    *path = ctx->last_estep_path;
    return VLGP_OK;
}

void vlgp_read_switches(vlgp_ctx* ctx) {
    HstepSwitches& w = ctx->hsw;
    w.dense = getenv("VLGP_HSTEP_DENSE") != nullptr;
    w.generic = getenv("VLGP_HSTEP_GENERIC") != nullptr;
    w.lowrank = getenv("VLGP_HSTEP_LOWRANK") != nullptr;
    w.generic_seg = getenv("VLGP_HSTEP_GENERIC_SEG") != nullptr;
    w.debug_occ = getenv("VLGP_DEBUG_OCC") != nullptr;
    w.fuse_tables = getenv("VLGP_HSTEP_FUSE_TABLES") != nullptr;
    w.lr_tol = getenv("VLGP_HSTEP_LR_TOL") ? atof(getenv("VLGP_HSTEP_LR_TOL")) : 1e-12;
}

extern "C" int vlgp_debug_reload_switches(vlgp_ctx* ctx) {
    NEED_CTX(ctx);
    vlgp_read_switches(ctx);
    return VLGP_OK;
}

extern "C" int vlgp_debug_last_hstep_path(vlgp_ctx* ctx, int* path) {
    NEED_CTX(ctx);
    if (!path) return vlgp_fail(ctx, VLGP_ERR_ARG, "null path");
    *path = ctx->last_hstep_path;
    return VLGP_OK;
}

extern "C" int vlgp_debug_hstep_stats(vlgp_ctx* ctx, double out[4]) {
    NEED_CTX(ctx);
    if (!out) return vlgp_fail(ctx, VLGP_ERR_ARG, "null out");
    for (int i = 0; i < 4; ++i) out[i] = ctx->hstat[i];
    return VLGP_OK;
}

extern "C" int vlgp_debug_npx(vlgp_ctx* ctx, int kind, int64_t n, const double* a, const double* b, double* out) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipSetDevice(ctx->dev));
    if (n < 1 || !a || !out || kind < 0 || kind > 3 || (kind >= 2 && !b)) return vlgp_fail(ctx, VLGP_ERR_ARG, "bad probe arguments");
    CHK(vlgp_ensure_work(ctx, 3 * n));
    double* W = ctx->d_work;
    HIPCHK(ctx, hipMemcpyAsync(W, a, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    if (b) HIPCHK(ctx, hipMemcpyAsync(W + n, b, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(W + 2 * n, out, sizeof(double) * n, hipMemcpyHostToDevice, ctx->stream));
    CHK(launch_npx_probe(ctx, kind, n, W, W + n, W + 2 * n));
    HIPCHK(ctx, hipMemcpyAsync(out, W + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return VLGP_OK;
}

extern "C" int vlgp_debug_phase_clock(vlgp_ctx* ctx, int on, uint64_t out[8]) {
    NEED_CTX(ctx);
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (out) {
        for (int i = 0; i < 8; ++i) out[i] = 0;
        if (ctx->d_clk) HIPCHK(ctx, hipMemcpy(out, ctx->d_clk, 8 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    }
    if (on) {
        if (!ctx->d_clk) HIPCHK(ctx, hipMalloc(&ctx->d_clk, 8 * sizeof(uint64_t)));
        HIPCHK(ctx, hipMemset(ctx->d_clk, 0, 8 * sizeof(uint64_t)));
    } else if (ctx->d_clk) {
        (void)hipFree(ctx->d_clk);
        ctx->d_clk = nullptr;
    }
    return VLGP_OK;
}

extern "C" int vlgp_profile_enable(vlgp_ctx* ctx, int on) {
    NEED_CTX(ctx);
    if (!on) prof_drain(ctx);
    ctx->prof_on = on != 0;
    return VLGP_OK;
}
extern "C" int vlgp_profile_reset(vlgp_ctx* ctx) {
    NEED_CTX(ctx);
    prof_drain(ctx);
    for (auto& p : ctx->prof) p = ProfSlot();
    return VLGP_OK;
}
extern "C" int vlgp_profile_get(vlgp_ctx* ctx, int kind, int64_t* launches, double* total_ms, double* units) {
    NEED_CTX(ctx);
    if (kind < 0 || kind >= VLGP_PROF_KINDS) return vlgp_fail(ctx, VLGP_ERR_ARG, "bad profile kind %d", kind);
    prof_drain(ctx);
    ProfSlot tot = ctx->prof[kind];
    if (kind == VLGP_PROF_ESTEP)  // every E-step kernel, whichever instantiation ran
        for (int k = VLGP_PROF_ESTEP_RA16; k <= VLGP_PROF_ESTEP_GENERIC; ++k) {
            tot.launches += ctx->prof[k].launches;
            tot.ms += ctx->prof[k].ms;
            tot.units += ctx->prof[k].units;
        }
    if (launches) *launches = tot.launches;
    if (total_ms) *total_ms = tot.ms;
    if (units) *units = tot.units;
    return VLGP_OK;
}
