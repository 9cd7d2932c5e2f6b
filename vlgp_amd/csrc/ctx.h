// Internal state behind the opaque vlgp_ctx handle (host side, C++17).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "../../include/vlgp_hip.h"

#define VLGP_WAVE 64
#define VLGP_E_LANES 4          // split E-step: the unit set runs as up to this many independent lanes (streams)
#define VLGP_PRIOR_SLOTS 64   // prior lengths factored per mailbox round

// One low-rank prior factor per distinct unit length (gp.make_cholesky).
struct Prior {
    int T = 0;
    double* d_full = nullptr;     // (L, T, R) as the reference lays it out
    double* d_compact = nullptr;  // per latent: (T, rl) row-major, zero columns dropped
    std::vector<int> rl;          // effective rank per latent
    std::vector<int64_t> goff;    // offset of latent l inside d_compact (doubles)
    int64_t compact_len = 0;
    int index = -1;               // row in the device prior table
};

struct UnitSet {
    bool valid = false;
    int M = 0;
    int64_t rows = 0;
    int Tmax = 0, Tmin = 0;
    std::vector<int64_t> off;     // host copy of offsets (M+1)
    int64_t* d_off = nullptr;
    double *y = nullptr, *x = nullptr, *mu = nullptr, *v = nullptr, *w = nullptr, *dmu = nullptr;
    bool x_ones = true;
    bool alias = false;           // shares y/x/mu/v/w with `parent` (exact tiling cut)
    int parent = -1;
    std::vector<int64_t> src_start;  // for non-aliased cuts: source row of each unit
    int64_t* d_src_start = nullptr;
    int* d_unit_prior = nullptr;  // prior-table row per unit
    uint64_t prior_epoch = 0;     // epoch of the table d_unit_prior was built against
    double* d_xb = nullptr;       // (rows, N) x.b, only when !x_ones
    double* d_mu_stash = nullptr; // copy of mu taken by vlgp_stash_mu (rows x L)
    double* d_scratch = nullptr;  // long-unit E-step scratch
    int64_t scratch_len = 0;
    double rows_all_ranks = 0.0;  // sum of `rows` over the ranks (M-step noise), exchanged on first use; 0 = unknown
    // Overlapping segments of a non-aliased cut (trial lengths that are not multiples of the window, vlgp/util.py:482-496).
    // In the reference they are VIEWS of the same trial rows: segment k + 1 starts its E-step from the mu, v that segment k
    // left in the shared rows, and an in-place constraint touches a shared row once per segment that holds it.  Here the
    // units are independent copies stored STAGE-MAJOR (stage = position in a chain of overlapping neighbours), the E-step
    // runs stage by stage and the shared rows are copied forward before and back after each stage (vlgp_set_overlaps).
    std::vector<int> stage_start;   // unit index of the first unit of each stage, + M at the end; empty: no overlaps
    std::vector<int> link_start;    // per stage s: links [link_start[s], link_start[s + 1]) have their second unit in s
    int* d_links = nullptr;         // (n_links, 3): first unit, second unit, shared rows
    int n_links = 0;
    bool share_mu = true;           // false once the segments' mu was rebound (constrain_loading "svd")
};

// Debug / test switches of the H-step dispatch (environment VLGP_HSTEP_*, VLGP_DEBUG_OCC), read ONCE when the handle is
// created and again on vlgp_debug_reload_switches: the objective call itself -- ~40 times per EM iteration, on the
// optimiser's critical path -- reads this struct, not the environment.
struct HstepSwitches {
    bool dense = false, generic = false, lowrank = false, generic_seg = false, debug_occ = false, fuse_tables = false;
    double lr_tol = 1e-12;
};
void vlgp_read_switches(struct vlgp_ctx* ctx);

struct ProfSlot {
    int64_t launches = 0;
    double ms = 0.0;
    double units = 0.0;  // work units covered by the timed launches (kernel specific)
};

struct vlgp_ctx {
    int dev = 0;
    hipStream_t stream = nullptr;
    int N = 0, L = 0, P = 0, R = 0;
    int n_cu = 256;
    std::vector<uint8_t> gauss;   // host copy
    int n_gauss = 0;
    int* d_gauss = nullptr;       // (N) int flags
    double *d_a = nullptr, *d_b = nullptr, *d_noise = nullptr, *d_da = nullptr, *d_db = nullptr;
    bool have_params = false;

    std::map<int, Prior> priors;
    uint64_t prior_epoch = 1;
    const double** d_prior_base = nullptr;  // table row -> compact base pointer
    int* d_prior_rl = nullptr;              // (rows, L)
    int64_t* d_prior_goff = nullptr;        // (rows, L)
    int prior_rows = 0;
    // mailbox of the prior kernel: per launch slot the L ranks, then a sequence word (mapped host memory)
    int* h_prior_mb = nullptr;
    int* d_prior_mb_host = nullptr;         // device view of h_prior_mb
    int* d_prior_mb = nullptr;              // device slots + arrival counter
    unsigned long long prior_seq = 0;
    std::vector<struct Prior*> prior_pending;   // launched, host-side ranks not taken yet (vlgp_prior_collect)
    unsigned long long prior_pending_seq = 0;

    UnitSet sets[VLGP_MAX_SETS];

    // second execution lane: the M-step runs here, concurrently with the H-step
    // rounds on the main stream (they touch disjoint data: a, b vs omega)
    hipStream_t mstream = nullptr;
    double* d_work_m = nullptr;
    int64_t work_m_len = 0;
    int* d_fail_m = nullptr;
    void* comm_m = nullptr;       // its own ncclComm_t (collectives of one communicator must not interleave)
    hipEvent_t ev_fork = nullptr, ev_m_start = nullptr, ev_m_done = nullptr;
    bool m_pending = false;
    // single rank: the M-step's launch sequence (52 launches at Mniter = 25) as an instantiated hipGraph, replayed while
    // its key (buffers, sizes, options) is unchanged: one enqueue instead of ~0.25 ms of host launch calls in front of
    // the H-step's first round
    void* m_graph_exec = nullptr;
    std::vector<double> m_graph_key;

    double* d_ecols = nullptr;    // E-step per-channel records + wconst (fast kernel), rebuilt per launch
    int* d_fail = nullptr;        // device failure counter
    unsigned long long* d_clk = nullptr;  // E-step per-phase cycle counters (debug), 8 slots
    double* d_work = nullptr;     // general workspace (M-step partials, H-step, reductions)
    int64_t work_len = 0;
    double* h_pinned = nullptr;   // small pinned staging buffer
    int64_t pinned_len = 0;
    // H-step round kernel: device flags/counter and the mapped host mailbox it publishes to
    unsigned* d_hsync = nullptr;
    double* h_hres = nullptr;     // 48 results + sequence word
    double* d_hres = nullptr;     // device view of h_hres
    unsigned h_seq = 0;
    double* d_hmpart = nullptr;   // their partial sums by chunks of segments
    bool hprep = false;           // vlgp_hstep_prepare built moments and the w copy for write_epoch hprep_epoch
    hipEvent_t ev_e_done = nullptr;  // behind the last launch of the most recent vlgp_estep call (vlgp_estep_wait)
    bool e_done_valid = false;
    uint64_t hprep_epoch = 0, write_epoch = 0;  // write_epoch: bumped by every entry point that may change unit state
    double* d_hmom = nullptr;     // (L, T, T) second moments of mu for the quadratic terms
    int64_t hmom_len = 0;
    const UnitSet* hmom_us = nullptr;
    int hmom_T = 0;
    bool hmom_bracket = false;    // inside vlgp_hstep_begin/end: d_hmom stays valid across objective calls
    double* d_hwlm = nullptr;     // inside the bracket: w of the set latent-major, (L, rows), for the round kernels
    int64_t hwlm_len = 0;
    bool hwlm_valid = false;
    // low-rank H-step round (hstep_lr.h): per (window, dt, tol) the largest omega whose folded kernel blocks have rank <= r
    struct LrThr { int T; double dt, tol; std::vector<double> om; };
    std::vector<LrThr> lr_thr;
    HstepSwitches hsw;
    std::vector<const void*> lds_attr_done;  // kernels whose dynamic-LDS ceiling was raised on this handle's device
    // instruction priority of the round kernels' waves beside the M-step lane: high while the H-step bracket is the longer
    // of the two (durations of the previous EM iteration; hstep.hip, hstep_wave_prio)
    int h_prio = 1;
    double last_h_ms = 0.0, last_m_ms = 0.0;  // 0 = unknown
    double h_t0 = 0.0;
    int last_hstep_path = 0;      // VLGP_PATH_HSTEP_* of the most recent H-step objective call
    double hstat[4] = {0.0, 0.0, 0.0, 0.0};  // vlgp_debug_hstep_stats

    // profiling
    bool prof_on = false;
    ProfSlot prof[VLGP_PROF_KINDS];
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    struct PendingProf { int kind; hipEvent_t a, b; double units; bool done; };
    std::vector<PendingProf> pending;

    // RCCL
    void* comm = nullptr;         // ncclComm_t
    void* shm = nullptr;          // shared-memory test transport (VLGP_COMM_TRANSPORT=shm), main lane
    void* shm_m = nullptr;        // ... M-step lane
    void* hx = nullptr;           // host-side exchange segment for the H-step round sums (single node)
    int rank = 0, world = 1;

    // second lane of the split E-step (estep_split.hip): half of the unit set runs its sweeps here
    hipStream_t elane[VLGP_E_LANES - 1] = {};
    hipEvent_t ev_e_fork = nullptr, ev_e_join[VLGP_E_LANES - 1] = {};
    int last_estep_path = 0;      // VLGP_PATH_ESTEP_* of the most recent E-step / update_w / update_v launch
    int last_estep_mix = 0;       // that call ran mixed lane-per-task / wave-per-task launches (esplit_mix)
    int lds_max = 64 * 1024;      // hipDeviceAttributeMaxSharedMemoryPerBlock (gfx950: 160 KB)

    // Pieces of an EM iteration's tail taken off its critical path (api.hip):
    // (1) the norms of mu, dmu for the stopping rule: one kernel queued behind the E-step, results in mapped host memory
    double* d_xwork = nullptr;    // 1024 x 2 partial sums | ticket
    double* h_xres = nullptr;     // mapped: |mu|^2, |dmu|^2, sequence word
    double* d_xres = nullptr;     // its device view
    unsigned long long x_seq = 0;
    int x_pending = 0;            // 1: queued (sequence x_seq), 2: deferred to vlgp_norms_end (several ranks)
    int x_set = -1;
    // (2) the M-step lane leaves a, b, noise, da, db and its failure count in pinned memory behind its last kernel
    double* h_msnap = nullptr;    // mapped host memory
    double* d_msnap = nullptr;    // its device view
    bool msnap_valid = false;
    // (3) vlgp_set_params / vlgp_apply_latent_map stage through their own pinned (and device) buffers, reuse guarded by
    // an event: no stream synchronisation in front of the E-step's first launch
    double* h_stage_par = nullptr;
    double* h_stage_map = nullptr;
    double* d_stage_map = nullptr;
    hipEvent_t ev_stage_par = nullptr, ev_stage_map = nullptr;
    bool stage_par_busy = false, stage_map_busy = false;

    std::string err;
};

// ---- error plumbing ------------------------------------------------------
int vlgp_fail(vlgp_ctx* ctx, int code, const char* fmt, ...);
#define HIPCHK(ctx, call)                                                          \
    do {                                                                           \
        hipError_t e__ = (call);                                                   \
        if (e__ != hipSuccess)                                                     \
            return vlgp_fail(ctx, VLGP_ERR_HIP, "%s failed: %s (%s:%d)", #call,    \
                             hipGetErrorString(e__), __FILE__, __LINE__);          \
    } while (0)
#define CHK(expr)                 \
    do {                          \
        int rc__ = (expr);        \
        if (rc__ != VLGP_OK) return rc__; \
    } while (0)

int vlgp_ensure_work(vlgp_ctx* ctx, int64_t n_doubles);
int vlgp_ensure_pinned(vlgp_ctx* ctx, int64_t n_doubles);
int vlgp_allreduce(vlgp_ctx* ctx, double* d_buf, int64_t n);  // in place, sum, on ctx->stream
int vlgp_hx_allreduce(vlgp_ctx* ctx, double* h_vals, int n);   // host values, n <= 64, rank-order sum; needs ctx->hx
int vlgp_allreduce_m(vlgp_ctx* ctx, double* d_buf, int64_t n);  // same on the M-step lane (comm_m, mstream)
int vlgp_ensure_work_m(vlgp_ctx* ctx, int64_t n_doubles);
int vlgp_join_m(vlgp_ctx* ctx);  // wait for a pending asynchronous M-step
int vlgp_bind_priors(vlgp_ctx* ctx, UnitSet& us);             // (re)build d_unit_prior
int vlgp_refresh_xb(vlgp_ctx* ctx, UnitSet& us);              // xb = x.b when x is general
UnitSet* vlgp_get_set(vlgp_ctx* ctx, int set, bool must_be_valid);

// profiling brackets (HIP events on ctx->stream)
void vlgp_prof_begin(vlgp_ctx* ctx, int kind, hipStream_t st = nullptr);
void vlgp_prof_end(vlgp_ctx* ctx, int kind, double units = 0.0, hipStream_t st = nullptr);

// ---- kernel launchers (one per translation unit) -------------------------
// mode bits for the E-step kernel
#define EM_FACTOR0 1   // factor I + G'WG from the incoming w before the first sweep
#define EM_MEAN 2      // run mean-update sweeps (n_iter of them)
#define EM_W 4         // recompute w
#define EM_V 8         // update v from the factor
int launch_estep(vlgp_ctx* ctx, UnitSet& us, int mode, int n_iter, double dmu_bound, int vb);
// runs entirely on ctx->mstream with the M-step lane's workspace / communicator
int launch_mstep(vlgp_ctx* ctx, UnitSet& us, int n_iter, int use_hessian, double eps, double lr,
                 double da_bound, double db_bound);
int launch_hstep(vlgp_ctx* ctx, UnitSet& us, int window, double dt, int n_eval, const int* latent,
                 const double* logp, double* ll, double* dll);
// factor the listed priors (bit-exact ichol_gauss), ranks and compact copies included; returns with pr.rl set
int launch_ichol_all(vlgp_ctx* ctx, const std::vector<Prior*>& prs, const double* omega, const double* sigma,
                     bool in_table, bool lazy = false);
int vlgp_prior_collect(vlgp_ctx* ctx);  // host-side ranks of a lazy launch_ichol_all (call before reading Prior::rl)
int launch_compact_prior(vlgp_ctx* ctx, Prior& pr);  // host-injected d_full -> rl, d_compact
int launch_sample_posterior(vlgp_ctx* ctx, int T, int n, const double* d_mu, const double* d_w, const double* d_G,
                            const double* d_eps, double* d_z, double* d_out);
int launch_npx_probe(vlgp_ctx* ctx, int kind, int64_t n, const double* d_a, const double* d_b, double* d_out);
int launch_xb(vlgp_ctx* ctx, UnitSet& us);
int launch_latent_map(vlgp_ctx* ctx, UnitSet& us, const double* d_map, const double* d_shift);
int launch_snapshot_params(vlgp_ctx* ctx, hipStream_t st, double* d_host);  // a | b | noise | da | db | failure count
int launch_norms(vlgp_ctx* ctx, UnitSet& us, double* d_part, unsigned* d_ticket, double* d_host, unsigned long long seq);
// shared rows of overlapping segments: dir 0 copies first unit's tail -> second unit's head (mu if share_mu, v), dir 1 back,
// for the links [l0, l1); launch_links_map applies the latent map once more to both copies of every shared row
int launch_links_copy(vlgp_ctx* ctx, UnitSet& us, int l0, int l1, int dir);
int launch_links_map(vlgp_ctx* ctx, UnitSet& us, const double* d_map, const double* d_shift);
int hstep_prepare(vlgp_ctx* ctx, UnitSet& us, int T, hipStream_t st);  // moments of mu + w latent-major for the rounds (hstep.hip)
int launch_moments(vlgp_ctx* ctx, UnitSet& us);  // tri(L) gram | sum mu | sum v | sum mu^2 | |dmu|^2 at ctx->d_work
int launch_project(vlgp_ctx* ctx, UnitSet& us, const double* d_proj, const double* d_shift, double* d_part,
                   double* d_out);  // mu = y proj - shift; d_out = column sums of y
int launch_gather(vlgp_ctx* ctx, UnitSet& src, UnitSet& dst, int window);
int launch_scatter(vlgp_ctx* ctx, UnitSet& cut, UnitSet& dst, int window);
