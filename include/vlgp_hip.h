/*
 * vlgp_hip.h -- C ABI of libvlgp_hip.so, the MI355X (gfx950) engine behind the
 * variational-EM hot path of catniplab/vlgp.
 *
 * The reference has no FFI: its seam is the Python contract
 *     estep|mstep|hstep|update_w|update_v|make_cholesky|infer(trials, params, config) -> None
 * (vlgp/core.py:22-471, vlgp/gp.py:65-162).  Each entry point below names the
 * reference function it replaces.  The Python host (vlgp_amd/engine.py) binds
 * these with ctypes; see INTEGRATION.md for the stub a maintainer of the
 * reference would add.
 *
 * Conventions
 *   - plain pointers and sizes only; all floating point is IEEE double
 *   - every function returns an int status: 0 = ok, <0 = error (text via
 *     vlgp_last_error); nothing throws across the boundary
 *   - caller owns host buffers; the library owns device buffers behind the
 *     opaque handle; one host thread per handle; one handle per GPU
 *   - all work is enqueued on the handle's own HIP stream; calls that return
 *     host data synchronise that stream, the others are asynchronous
 *   - numerical failures never return an error: a non-positive pivot in a
 *     Cholesky zeroes that latent's update for that unit (reference:
 *     vlgp/core.py:92-94,112-113,194-196) and is counted in *n_failed
 *
 * Data model
 *   unit      one trial or one fixed-length segment: T_m time bins
 *   unit set  M units packed back to back in row-major arrays
 *               y   (rows, N)      observations          rows = sum T_m
 *               x   (rows, P, N)   regressors (NULL = all ones, P must be 1)
 *               mu, v, w, dmu (rows, L)
 *             addressed through a CSR-style offsets[M+1]; VLGP_MAX_SETS slots
 *   params    a (L, N) loading, b (P, N) bias/regression, noise (N)
 *   prior     per distinct unit length T: G (L, T, R), K_l ~= G_l G_l^T
 */
#ifndef VLGP_HIP_H
#define VLGP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VLGP_ABI_VERSION 1
#define VLGP_MAX_SETS 4
#define VLGP_MAX_L 64             /* latents per handle.  The reference has no bound (vlgp/core.py:76,106; gp.py:82);
                                     up to 10 the specialised kernels run, up to 16 the register-resident generic
                                     ones, above that loop-based fallbacks (slow, same results) */
#define VLGP_MAX_XDIM 64          /* regressors per channel (1 + history, vlgp/preprocess.py:53): up to 8 specialised */
#define VLGP_MAX_RANK 64          /* R <= 64 (the reference hard-codes 50, preprocess.py:75) */
#define VLGP_UNIQUE_ID_BYTES 128

#define VLGP_OK 0
#define VLGP_ERR_ARG (-1)         /* bad argument / shape */
#define VLGP_ERR_HIP (-2)         /* HIP runtime error (no GPU, OOM, launch failure) */
#define VLGP_ERR_STATE (-3)       /* call sequence error (missing prior, empty set ...) */
#define VLGP_ERR_COMM (-4)        /* RCCL error */

typedef struct vlgp_ctx vlgp_ctx;

/* ---- lifetime --------------------------------------------------------- */
int vlgp_abi_version(void);
int vlgp_device_count(int* count);
/* gauss_mask[n] != 0 marks a Gaussian channel, 0 a Poisson one
 * (params["likelihood"], vlgp/preprocess.py:59-64). */
int vlgp_create(int device, int N, int L, int P, int R, const uint8_t* gauss_mask,
                vlgp_ctx** out);
int vlgp_destroy(vlgp_ctx* ctx);
/* Last error text of this handle (or of the failed vlgp_create when ctx is NULL). */
const char* vlgp_last_error(vlgp_ctx* ctx);
int vlgp_synchronize(vlgp_ctx* ctx);
/* Drains the main stream only: an M-step begun with vlgp_mstep_begin keeps running on its lane (vlgp_synchronize
 * waits for it too).  The EM loop uses it to time the E-step (core.py:307-315) with the M-step already enqueued. */
int vlgp_synchronize_main(vlgp_ctx* ctx);

/* ---- unit sets -------------------------------------------------------- */
/* Replaces the list-of-dicts the reference passes around (trial dict keys
 * y, x, mu, w, v; vlgp/preprocess.py:38-46).  x == NULL means x == 1, P == 1
 * (what preprocess.initialize builds when the user gives no regressors);
 * mu, v, w may be NULL (zero-filled). */
int vlgp_upload_units(vlgp_ctx* ctx, int set, int M, const int64_t* offsets,
                      const double* y, const double* x, const double* mu,
                      const double* v, const double* w);
/* util.cut_trials (vlgp/util.py:457-499): build set `dst` of M_dst units of
 * `window` rows each, unit k starting at row src_row_start[k] of set `src`.
 * When the segments tile `src` exactly (no overlap) `dst` aliases the same
 * device arrays, as the reference's NumPy views do; otherwise rows are copied
 * and segments are independent ("Hard parts"). */
int vlgp_cut_units(vlgp_ctx* ctx, int src, int dst, int M_dst,
                   const int64_t* src_row_start, int window);
/* Write mu and v of a cut set back into its source set (no-op when aliased;
 * with overlapping segments later segments win, in order). */
int vlgp_merge_units(vlgp_ctx* ctx, int cut_set);
/* Overlapping segments of a copied cut (trial lengths that are not multiples of the window).  In the reference they are
 * NumPy VIEWS of the same trial rows (vlgp/util.py:482-496): core.estep runs them one after the other, segment k + 1
 * starting from the mu, v that segment k left in the shared rows (vlgp/core.py:123-126), and an in-place constraint
 * (vlgp/core.py:374-388,413-416) touches a shared row once per segment that holds it.  The caller orders the units of
 * the cut STAGE-MAJOR (stage = position in a chain of overlapping neighbours; stage_start[n_stages + 1] unit indices)
 * and lists the links (first unit, second unit, shared rows), sorted by the stage of their second unit
 * (link_start[n_stages + 1]).  vlgp_estep then runs stage by stage, copying the shared rows forward before and back
 * after each stage; vlgp_apply_latent_map applies its map a second time to the shared rows.  vlgp_unshare_mu: mu stops
 * being shared (constrain_loading "svd" rebinds every segment's mu, vlgp/core.py:407-408); v stays shared. */
int vlgp_set_overlaps(vlgp_ctx* ctx, int set, int n_stages, const int* stage_start, int n_links,
                      const int* links, const int* link_start);
int vlgp_unshare_mu(vlgp_ctx* ctx, int set);
/* restore == 0: keep a device-side copy of the set's mu; restore != 0: write it back.  For the one place where
 * the reference DETACHES segments from their trials: constrain_loading == "svd" rebinds every segment's mu
 * (vlgp/core.py:407-408, `trial["mu"] = trial["mu"] @ us`), after which the parent trials keep the values they
 * had at that moment -- the final inference of fit starts from those. */
int vlgp_stash_mu(vlgp_ctx* ctx, int set, int restore);
/* Any of the output pointers may be NULL. */
int vlgp_download_units(vlgp_ctx* ctx, int set, double* mu, double* v, double* w,
                        double* dmu);
int vlgp_free_units(vlgp_ctx* ctx, int set);

/* ---- parameters ------------------------------------------------------- */
int vlgp_set_params(vlgp_ctx* ctx, const double* a, const double* b, const double* noise);
/* Any pointer may be NULL.  da, db are the last M-step increments
 * (params["da"], params["db"], vlgp/core.py:201,219). */
int vlgp_get_params(vlgp_ctx* ctx, double* a, double* b, double* noise, double* da,
                    double* db);

/* ---- prior factor ----------------------------------------------------- */
/* gp.make_cholesky (vlgp/gp.py:150-162) + math.ichol_gauss (vlgp/math.py:76-126)
 * on the device: REPLACES the whole prior table with one factor per listed
 * length, G_l = ichol_gauss(T, omega_l, R) * sigma_l. */
int vlgp_build_prior(vlgp_ctx* ctx, int n_lengths, const int* lengths,
                     const double* omega, const double* sigma);
/* Inject a host-made factor G (L, T, R) for length T (adds/replaces one entry). */
int vlgp_set_prior(vlgp_ctx* ctx, int T, const double* G);
int vlgp_clear_prior(vlgp_ctx* ctx);
/* G (L, T, R) out; rank_out (L) = number of non-zero columns per latent (may be NULL). */
int vlgp_get_prior(vlgp_ctx* ctx, int T, double* G, int* rank_out);

/* ---- E-step ----------------------------------------------------------- */
/* core.update_w (vlgp/core.py:419-442) */
int vlgp_update_w(vlgp_ctx* ctx, int set);
/* core.update_v (vlgp/core.py:445-471); vb == 0 is the reference's early
 * return for method != "VB". */
int vlgp_update_v(vlgp_ctx* ctx, int set, int vb, int* n_failed);
/* core.estep -> core.infer_single_trial (vlgp/core.py:22-126): n_iter inner
 * iterations for every unit of the set.  n_failed may be NULL (no sync then). */
int vlgp_estep(vlgp_ctx* ctx, int set, int n_iter, double dmu_bound, int vb,
               int* n_failed);
/* Wait for the launches of the most recent vlgp_estep call -- NOT for what has been queued behind them on the
 * same stream since (vlgp_hstep_prepare: the timers of core.vem bracket the E-step alone, vlgp/core.py:308-312). */
int vlgp_estep_wait(vlgp_ctx* ctx);

/* ---- M-step ----------------------------------------------------------- */
/* core.mstep (vlgp/core.py:129-249) over the concatenation of all units of
 * the set.  With a communicator attached the sufficient statistics are
 * all-reduced over ranks once per Newton iteration. */
int vlgp_mstep(vlgp_ctx* ctx, int set, int n_iter, int use_hessian, double eps,
               double learning_rate, double da_bound, double db_bound, int* n_failed);

/* Asynchronous form: vlgp_mstep_begin enqueues the whole M-step on the handle's
 * second stream (after everything already queued) and returns; the H-step
 * objective, vlgp_build_prior, vlgp_norms may run meanwhile (they touch neither a, b
 * nor the rows the M-step reads for writing).  vlgp_mstep_end waits, reports the
 * failure count and the device time in ms.  Every other entry point that reads or
 * writes parameters or unit state joins a pending M-step first. */
int vlgp_mstep_begin(vlgp_ctx* ctx, int set, int n_iter, int use_hessian, double eps,
                     double learning_rate, double da_bound, double db_bound);
int vlgp_mstep_end(vlgp_ctx* ctx, int* n_failed, double* device_ms);

/* ---- H-step ----------------------------------------------------------- */
/* The objective scipy's L-BFGS-B minimises in gp.optimze1d (vlgp/gp.py:100-123):
 * construct_posterior_cov (gp.py:126-147) + elbo (gp.py:12-43), mask [0,1,0].
 * n_eval independent evaluations in one call: evaluation e uses latent
 * latent[e] and log-parameters logp[3e..3e+2] = log(sigma^2, omega, eps).
 * Outputs the UN-negated ll[e] and dll[3e..3e+2] summed over all units of the
 * set (and over ranks).  Every unit must have exactly `window` rows, window <= 1024 (4 ... 64: low-rank
 * or dense round kernels; 65 ... 128: workgroup-per-segment kernels; above: generic kernels, matrices in global memory)
 * (windows <= 50 -- 50 is the reference's default -- share a dedicated kernel). */
int vlgp_hstep_objective(vlgp_ctx* ctx, int set, int window, double dt, int n_eval,
                         const int* latent, const double* logp, double* ll,
                         double* dll);

/* Brackets the objective calls of one gp.optimize run (vlgp/gp.py:65-97), during
 * which the set's mu and w do not change: the per-latent second-moment matrices
 * sum_i mu_i mu_i' that the quadratic terms of every evaluation share are then built
 * once (by the first objective call) instead of on every call.  The caller must not
 * modify the set between begin and end.  Optional: without it every
 * vlgp_hstep_objective call is self-contained. */
int vlgp_hstep_begin(vlgp_ctx* ctx, int set, int window);
int vlgp_hstep_end(vlgp_ctx* ctx);
/* Optional, before vlgp_hstep_begin: enqueue what the bracket builds from the units alone (those second moments and
 * the latent-major copy of w the round kernels read) right away -- core.vem runs the H-step after the E-step
 * (vlgp/core.py:320-327), and both are final once the E-step is done, so called there they are ready before the
 * first objective call instead of in front of its round.  Dropped if any entry point that may change the units runs
 * before vlgp_hstep_begin; a no-op for sets the round kernels do not take. */
int vlgp_hstep_prepare(vlgp_ctx* ctx, int set, int window);

/* ---- constraints / norms --------------------------------------------- */
/* mu <- (mu - shift) @ map for every unit (map (L, L) row-major, shift (L) or
 * NULL).  Covers core.constrain_loading (vlgp/core.py:392-416: map = s I, or
 * the SVD factor) and core.constrain_latent (vlgp/core.py:366-389). */
int vlgp_apply_latent_map(vlgp_ctx* ctx, int set, const double* map, const double* shift);
/* out[0] = sum mu^2, out[1] = sum dmu^2 over the set (and over ranks):
 * the convergence test of core.vem (vlgp/core.py:300-305,350-354). */
int vlgp_norms(vlgp_ctx* ctx, int set, double out[2]);
/* The same in two halves: vlgp_norms_begin enqueues the sums (one kernel, results in mapped host memory) behind
 * everything already queued and returns; vlgp_norms_end waits for them.  core.vem takes the norms after the M- and
 * H-step (vlgp/core.py:350-354), but mu and dmu are final once the E-step (and constrain_latent) are done: begun
 * there -- even while the E-step still runs -- nothing is left to wait for at the end of the iteration.  Every entry
 * point that writes unit state waits for a pending pass first. */
int vlgp_norms_begin(vlgp_ctx* ctx, int set);
int vlgp_norms_end(vlgp_ctx* ctx, double out[2]);
/* Initial latents of preprocess.initialize (vlgp/preprocess.py:30-41) on the device: for every row of the
 * set mu = y . proj - shift, proj (N, L) row-major the posterior-mean map of the factor-analysis fit
 * (FactorAnalysis.transform: (y - mean) W'Psi^-1 (I + W Psi^-1 W')^-1), shift = mean . proj (L).
 * colsum (N, may be NULL) receives the column sums of y over THIS rank's rows (b = log mean y). */
int vlgp_project_units(vlgp_ctx* ctx, int set, const double* proj, const double* shift, double* colsum);
/* Column sums over the set (and ranks): sum1[l] = sum mu[:,l], sum2[l] = sum mu[:,l]^2,
 * *count = number of rows.  For core.constrain_latent. */
int vlgp_latent_moments(vlgp_ctx* ctx, int set, double* sum1, double* sum2, double* count);

/* ---- posterior draws --------------------------------------------------- */
/* api.sample_posterior (vlgp/api.py:142-168) for one trial of T bins: out (nsamples, T, L) row-major,
 * out[s, :, l] = mu[:, l] + G_l Lc^-T eps[l, :, s] with Lc Lc' = I + G_l' diag(w[:, l]) G_l -- a draw from
 * N(mu_l, (K_l^-1 + W_l)^-1), K_l = G_l G_l', without any T x T matrix.  Host arrays: mu, w (T, L);
 * G (L, T, R) = params["cholesky"][T]; eps (L, R, nsamples) standard normal draws (rows beyond a latent's
 * effective rank are ignored). */
int vlgp_sample_posterior(vlgp_ctx* ctx, int T, const double* mu, const double* w, const double* G,
                          int nsamples, const double* eps, double* out, int* n_failed);

/* ---- multi-GPU (RCCL over xGMI) --------------------------------------- */
/* Rank 0 makes the id, every rank passes the same id to vlgp_comm_init. */
int vlgp_comm_unique_id(char id[VLGP_UNIQUE_ID_BYTES]);
int vlgp_comm_init(vlgp_ctx* ctx, const char id[VLGP_UNIQUE_ID_BYTES], int rank, int world);
/* Second communicator (its own unique id) for the M-step lane, so that its
 * all-reduces may overlap the H-step's; without it a multi-rank M-step runs
 * its collectives un-overlapped on the first communicator's stream order. */
int vlgp_comm_init_aux(vlgp_ctx* ctx, const char id[VLGP_UNIQUE_ID_BYTES]);
/* 1 when the ranks share the host-side exchange segment for the H-step round sums (single node,
 * world > 1): the H-step then issues no RCCL collective, so the M-step lane may run concurrently with
 * it across ranks (its communicator is the only one in flight). */
int vlgp_comm_host_exchange(vlgp_ctx* ctx);
/* Transport behind the handle's all-reduces: 0 none (single rank), 1 RCCL, 2 host shared memory
 * (VLGP_COMM_TRANSPORT=shm, a test vehicle: several ranks on one GPU). */
int vlgp_comm_transport(vlgp_ctx* ctx);
/* Ranks RCCL itself reports (ncclCommCount) for the handle's two communicators -- main lane and M-step lane; 0 when no
 * RCCL communicator is attached (single rank, or the shared-memory test transport), -1 if the query failed. */
int vlgp_comm_rccl_ranks(vlgp_ctx* ctx, int* main_lane, int* m_lane);
/* In-place sum over ranks of n host doubles (staged through the device, on the
 * handle's stream, synchronous).  With no communicator attached it is a no-op.
 * n == 0 is a pure barrier. */
int vlgp_comm_allreduce_host(vlgp_ctx* ctx, double* buf, int n);

/* ---- measurement ------------------------------------------------------ */
/* HIP-event timing of the kernels an entry point launches, on the handle's
 * stream.  kind: 0 = E-step kernel, 1 = M-step statistics kernel,
 * 2 = H-step objective kernel, 3 = prior (ichol) kernel. */
#define VLGP_PROF_ESTEP 0
#define VLGP_PROF_MSTEP 1
#define VLGP_PROF_HSTEP 2
#define VLGP_PROF_PRIOR 3
/* the fast E-step kernel by instantiation (register-array size 16 / 24 / 32), the long-unit kernel and the
 * generic kernel; kind 0 reports the sum of all E-step kernels */
#define VLGP_PROF_ESTEP_RA16 4
#define VLGP_PROF_ESTEP_RA24 5
#define VLGP_PROF_ESTEP_RA32 6
#define VLGP_PROF_ESTEP_LONG 7
#define VLGP_PROF_ESTEP_GENERIC 8
#define VLGP_PROF_ESTEP_PASS 9     /* split E-step: one (T x N) pass over all rows (sampled: one sweep per call) */
#define VLGP_PROF_ESTEP_FACTOR 10  /* split E-step: factor + variance launch, units = (unit, latent) tasks */
#define VLGP_PROF_ESTEP_MEAN 11    /* split E-step: mean-update launch, units = (unit, latent) tasks */
#define VLGP_PROF_HSTEP_LR 12      /* H-step: the low-rank round kernel (hstep_round_lr), units = segment-evaluations;
                                    * kind 2 then counts only the dense round kernel */
#define VLGP_PROF_HSTEP_TAB 13     /* H-step: the tables kernel in front of a low-rank round, units = evaluations */
#define VLGP_PROF_KINDS 14
int vlgp_profile_enable(vlgp_ctx* ctx, int on);
int vlgp_profile_reset(vlgp_ctx* ctx);
/* launches, total milliseconds and work units recorded for `kind` since the last
 * reset.  Units: E-step = unit-sweeps (units x inner iterations), M-step = rows
 * streamed, H-step = segment-evaluations, prior = factors built. */
int vlgp_profile_get(vlgp_ctx* ctx, int kind, int64_t* launches, double* total_ms, double* units);
/* E-step phase anatomy: when enabled, thread 0 of every workgroup adds its
 * shader-clock cycles per phase into 8 counters (0 staging, 1 y.a pass + first
 * factor, 2 residual pass, 3 mean update, 4 curvature pass, 5 factor + variance).
 * `out` may be NULL; on != 0 also zeroes the counters. */
int vlgp_debug_phase_clock(vlgp_ctx* ctx, int on, uint64_t out[8]);
/* Element-wise probe of the device arithmetic the prior kernel relies on being bit-identical to the
 * reference's NumPy (vlgp/math.py:113-119): kind 0 out = exp(a) as np.exp computes it, 1 out = sqrt(a),
 * 2 out = a / b, 3 out = fma(a, b, out).  Host arrays of n doubles; b may be NULL for kinds 0, 1. */
int vlgp_debug_npx(vlgp_ctx* ctx, int kind, int64_t n, const double* a, const double* b, double* out);
/* Which implementation ran the most recent vlgp_estep / vlgp_update_w / vlgp_update_v of this handle (the
 * reference has one serial loop, vlgp/core.py:123-126; the library picks a kernel family by set shape, and the
 * parity tests assert that the family they mean to check is the one that ran). */
#define VLGP_PATH_ESTEP_NONE 0
#define VLGP_PATH_ESTEP_SPLIT 1    /* chip-wide launches per phase (many short units) */
#define VLGP_PATH_ESTEP_FAST 2     /* persistent workgroup per unit, register-resident factors */
#define VLGP_PATH_ESTEP_LONG 3     /* long units (T > 64), one workgroup per trial */
#define VLGP_PATH_ESTEP_GENERIC 4  /* generic kernels (rank > 50 slots, L > 10, ...) */
#define VLGP_PATH_ESTEP_LSPLIT 5   /* long units as chip-wide launches: one workgroup per (unit, latent) task */
#define VLGP_PATH_ESTEP_SPLIT_MIXED 6 /* SPLIT whose per-latent launches carried lane-per-task and wave-per-task blocks in one grid */
int vlgp_debug_last_estep_path(vlgp_ctx* ctx, int* path);
/* Same for the most recent vlgp_hstep_objective call: which kernel family evaluated the per-segment terms. */
#define VLGP_PATH_HSTEP_NONE 0
#define VLGP_PATH_HSTEP_LOWRANK 1  /* exact low-rank (Woodbury) round, sixteen segments per workgroup (hstep_lr.h) */
#define VLGP_PATH_HSTEP_DENSE 2    /* one wave per segment, blocked elimination on the matrix pipe (hstep_mfma.h) */
#define VLGP_PATH_HSTEP_BIG 3      /* windows 65 ... 128: one workgroup per segment */
#define VLGP_PATH_HSTEP_GENERIC 4  /* generic kernels (any window; the reference's omega retry) */
#define VLGP_PATH_HSTEP_OLD 5      /* round-1 kernels behind their debug switches */
#define VLGP_PATH_HSTEP_MIXED 6    /* one round, two launches: low-rank kernel up to rank 32, dense kernel for the rougher evaluations */
int vlgp_debug_last_hstep_path(vlgp_ctx* ctx, int* path);
/* The H-step's debug switches (environment VLGP_HSTEP_DENSE / _GENERIC / _LOWRANK / _GENERIC_SEG / _LR_TOL,
 * VLGP_DEBUG_OCC) are read when the handle is created; this reads them again (tests that switch kernels on a live
 * handle).  Nothing in production needs it. */
int vlgp_debug_reload_switches(vlgp_ctx* ctx);
/* Counters of the H-step objective calls since the handle was created: out[0] evaluations that took the low-rank round,
 * out[1] the sum of their predicted ranks (even + odd block), out[2] evaluations that took the dense round,
 * out[3] low-rank rounds re-run densely because a rank exceeded the prediction. */
int vlgp_debug_hstep_stats(vlgp_ctx* ctx, double out[4]);

#ifdef __cplusplus
}
#endif
#endif /* VLGP_HIP_H */
