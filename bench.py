#!/usr/bin/env python
"""Headline benchmark: EM iterations/sec of the vLGP variational-EM loop.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[2], "C3"): 200 trials x 1000 bins x 100 Poisson
channels, 5 latents, fp64, synthetic Lorenz/Poisson data (seed 0), defaults of
the reference (Eniter = Mniter = 25, window = 50, rank = 50, VB, Hstep on).
A "step" is ONE full EM iteration (E-step + M-step + H-step + convergence
norms) over all 4000 fifty-bin segments -- the body of core.vem
(vlgp/core.py:298-357), driven through exactly the code ``vlgp_amd.fit`` runs.
With N > 1 the 200 trials are block-partitioned over the ranks (strong
scaling: total work fixed) and the M/H-step statistics are all-reduced by RCCL.

Prints ONE JSON line on rank 0 (see the contract in the task statement); the
extra objects are ``roofline`` (dominant kernel, HIP-event timed on the
engine's stream inside the timed region) and ``cpu_baseline`` (the NumPy/SciPy
oracle timed on a bounded sample of the same workload on this host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector = matrix peak (AMD spec)
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec

# C3s8 (C3s4, C3s2) = the shard of C3 one rank holds at 8 (4, 2) GPUs (25 of the 200 trials): its time per EM iteration on one MI355X is the
# compute a rank has left at N = 8, i.e. an upper bound on the strong-scaling speed-up before any exchange
WORKLOADS = {"C1": (10, 200, 20, 3), "C2": (50, 500, 50, 3), "C3": (200, 1000, 100, 5), "C3s8": (25, 1000, 100, 5),
             "C3s2": (100, 1000, 100, 5), "C3s4": (50, 1000, 100, 5),  # the shards of C3 at 2 and 4 GPUs
             # BASELINE.json configs[4] on ONE GPU: 500 ragged trials (500 ... 2000 bins, multiples of the window),
             # 150 Poisson + 50 Gaussian channels, ten latents (SURVEY 8(d) recipe; n_bins here = the longest trial)
             "C5": (500, 2000, 200, 10)}
C5_GAUSS = 50


def workload_extras(name, n_trials):
    """(trial lengths or None, per-channel likelihood list or None) of a workload."""
    if name != "C5":
        return None, None
    lengths = (50 * np.random.default_rng(0).integers(10, 41, n_trials)).tolist()
    n = WORKLOADS[name][2]
    return lengths, ["poisson"] * (n - C5_GAUSS) + ["gaussian"] * C5_GAUSS


def build_inputs(name):
    """Full synthetic trial list with the initialisation api.fit would compute
    (FactorAnalysis on a 10 % subsample, preprocess.initialize), done once on
    the whole data so that every rank starts from identical parameters."""
    from vlgp_amd import synth
    from vlgp_amd.preprocess import get_config, get_params, initialize

    n_trials, n_bins, N, L = WORKLOADS[name]
    lengths, lik = workload_extras(name, n_trials)
    if lengths is None:
        trials = synth.make_trials(n_trials, n_bins, N, L, seed=0)
    else:
        trials = synth.make_trials(n_trials, n_bins, N, L, seed=0, n_gauss=C5_GAUSS, lengths=lengths)
    cfg = get_config()
    params = get_params(trials, L, omega_bound=cfg["omega_bound"], **({"lik": lik} if lik else {}))
    np.random.seed(0)
    initialize(trials, params, cfg)
    for tr in trials:
        del tr["x"], tr["w"], tr["v"]  # x == 1 is the default; w, v are rebuilt by fit
    return trials, params["a"], params["b"], (n_trials, n_bins, N, L)


def estep_flops(T, N, L, ranks):
    """SURVEY.md section 8(d) count for one unit x one inner sweep of the E-step: the (T x N) passes plus, per
    latent of rank r, the build of I + G'WG, its factorisation and the mean / variance updates."""
    return 12.0 * T * L * N + sum(5.0 * T * r * r + 2.0 / 3.0 * r ** 3 + 8.0 * T * r for r in ranks)


def algorithmic_work(T, N, L, P):
    """SURVEY.md section 8(d) minimal-algorithm counts per work unit (the E-step one is estep_flops)."""
    return {
        # one row of one Newton iteration of the M-step
        "mstep_flops_per_row": 4.0 * L * N + N * (2.0 * L * L + 9.0 * L + 4.0 * P * P),
        "mstep_bytes_per_row_survey": 8.0 * (N * (1 + P) + 2 * L),   # SURVEY's count: y and x re-read every iteration
        "mstep_bytes_per_row_kernel": 8.0 * 2 * L,                   # what the kernel streams: mu, v (y hoisted, x == 1)
        # one segment of one H-step objective evaluation
        "hstep_flops_per_seg_eval": T ** 3 + 4.0 * T * T,
        "hstep_bytes_per_seg_eval": 16.0 * T,
    }


_PMC_CACHE = {}


def pmc_traffic(kernel_key):
    """Per-launch HBM bytes of a kernel from the committed rocprofv3 --pmc passes (profiles/r2/pmc_summary.json,
    else profiles/r1: FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM, plus WRITE_SIZE; separate
    passes).  None when no measurement is on file for that exact kernel name."""
    if not _PMC_CACHE:
        for rnd in ("r1", "r2", "r3", "r4", "r5", "r6"):  # later rounds override
            try:
                with open(os.path.join(ROOT, "profiles", rnd, "pmc_summary.json")) as f:
                    _PMC_CACHE.update(json.load(f))
            except Exception:
                pass
        _PMC_CACHE.setdefault("_", {})
    hit = _PMC_CACHE.get(kernel_key)
    if hit is None and kernel_key.endswith("*"):  # "name<5, 1, false*": the first kernel whose name starts like that
        cands = [k for k in sorted(_PMC_CACHE) if k.startswith(kernel_key[:-1])]
        if cands:  # the instantiation the passes saw most often
            hit = _PMC_CACHE[max(cands, key=lambda k: _PMC_CACHE[k].get("launches_fetch_pass", 0))]
    return (hit or {}).get("hbm_bytes_per_launch")


_STATS_CACHE = {}


def rocprof_avg_ms(kernel_key):
    """Average duration (ms) of a kernel in the newest committed `rocprofv3 --kernel-trace --stats` summary of this command
    (profiles/r*/*kernel_stats*.csv), same name matching as pmc_traffic; (None, None) when the kernel is not on file.
    The live HIP-event figure brackets the launch on its stream -- with the second E-step lane or the M-step lane busy it
    also contains the wait for the other lane's workgroups to drain -- while rocprofv3 reports the kernel's own span."""
    import csv, glob

    if not _STATS_CACHE:
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "*kernel_stats*.csv")))
        newest = {}
        for fpath in files:  # later rounds / later tags override
            rnd = os.path.basename(os.path.dirname(fpath))
            newest[rnd] = fpath
        if newest:
            fpath = newest[sorted(newest)[-1]]
            try:
                with open(fpath) as f:
                    for row in csv.DictReader(f):
                        name = row["Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
                        _STATS_CACHE[name] = (float(row["AverageNs"]) * 1e-6, int(row["Calls"]), os.path.relpath(fpath, ROOT))
            except Exception:
                pass
        _STATS_CACHE.setdefault("_", (None, 0, None))
    hit = _STATS_CACHE.get(kernel_key)
    if hit is None and kernel_key.endswith("*"):
        cands = [k for k in sorted(_STATS_CACHE) if k.startswith(kernel_key[:-1])]
        if cands:
            hit = _STATS_CACHE[max(cands, key=lambda k: _STATS_CACHE[k][1])]
    return (hit[0], hit[2]) if hit else (None, None)


def host_info():
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count() or 1


def _oracle_em_iteration(name, budget_trials, threads):
    """One timed EM iteration (the 2nd of two) of the oracle on the first ``budget_trials`` trials."""
    from oracle import vlgp_oracle as O

    from threadpoolctl import threadpool_limits

    with threadpool_limits(limits=threads):
        trials, a0, b0, (n_trials, n_bins, N, L) = build_inputs(name)
        trials = trials[:budget_trials]
        _, lik = workload_extras(name, n_trials)
        for tr in trials:
            nb = tr["y"].shape[0]
            tr["x"] = np.ones((nb, 1, N))
            tr["w"] = np.zeros((nb, L))
            tr["v"] = np.zeros((nb, L))
        cfg = O.make_config(max_iter=2, min_iter=2)
        params = O.make_params(trials, L, a=a0.copy(), b=b0.copy(), **({"lik": lik} if lik else {}))
        params["da"], params["db"] = np.zeros_like(a0), np.zeros_like(b0)
        O.fill_trials(trials)
        O.make_cholesky(trials, params)
        O.update_w(trials, params)
        O.update_v(trials, params, cfg)
        segs = O.cut_trials(trials, cfg["window"])
        O.make_cholesky(segs, params)
        O.fill_trials(segs)
        t0 = time.perf_counter()
        O.vem(segs, params, cfg)
        wall = time.perf_counter() - t0
    rt = cfg["runtime"]
    return rt["em_elapsed"][-1], (rt["e_elapsed"][-1], rt["m_elapsed"][-1], rt["h_elapsed"][-1]), wall, len(segs), n_trials


def cpu_baseline(name, budget_trials):
    """The oracle's EM iteration on a bounded sample of the same workload, on this host, scaled to the full
    trial count: with one BLAS thread (the reference is effectively single-core) and
    with OPENBLAS threads = nproc (50x50 matrices do not thread -- reported, not the baseline)."""
    model, nproc = host_info()
    per1, (e1, m1, h1), wall1, nseg, n_trials = _oracle_em_iteration(name, budget_trials, 1)
    scale = n_trials / float(budget_trials)
    out = {
        "value": 1.0 / (per1 * scale), "unit": "EM it/s", "cores": 1, "kind": "port",
        "cpu_model": model, "nproc": nproc,
        "sample": "oracle/vlgp_oracle.py vem on the first %d of %d trials (%d segments), 2 EM iterations, "
                  "2nd timed (E %.1fs, M %.1fs, H %.1fs), scaled x%g to the full workload; %.0fs wall"
                  % (budget_trials, n_trials, nseg, e1, m1, h1, scale, wall1),
        "e_step_ms_full": 1e3 * e1 * scale,
    }
    if nproc > 1:
        half = max(min(budget_trials // 2, 6), 1)   # a side note (50 x 50 matrices do not thread): kept short
        pern, _, walln, _, _ = _oracle_em_iteration(name, half, nproc)
        out["all_threads"] = {"value": 1.0 / (pern * n_trials / float(half)), "unit": "EM it/s", "cores": nproc,
                              "sample": "same on the first %d trials with %d BLAS threads; %.0fs wall" % (half, nproc, walln)}
    return out


def self_launch(n):
    """`python bench.py --gpus N` without torchrun: start the N ranks ourselves, one process per GPU, with the
    environment torchrun would export (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); rank 0 prints the one line.  Returns
    the exit status: non-zero as soon as a rank fails (the others are stopped), so a line is either carried by all N ranks
    or absent.  Plain subprocesses: no torch on the data path, none needed for the launch either."""
    import socket
    import subprocess
    import tempfile

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory(prefix="vlgp_bench_") as rdv:
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VLGP_RENDEZVOUS_DIR=rdv)
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                          stdout=None if r == 0 else subprocess.DEVNULL))
        status = 0
        live = list(procs)
        while live and status == 0:
            time.sleep(0.05)
            for p in list(live):
                rc = p.poll()
                if rc is not None:
                    live.remove(p)
                    status = status or rc
        for p in live:  # a rank failed: the others would wait for it in a collective
            p.terminate()
        for p in live:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
        return status if status >= 0 else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5,
                    help="untimed EM iterations first (the first four run the rank 17-32 classes of the E-step: "
                         "omega starts at its upper bound)")
    ap.add_argument("--workload", default="C3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--allow-shm", action="store_true",
                    help="tests only: accept the host shared-memory transport (several ranks on ONE GPU); a multi-GPU "
                         "line is otherwise printed only when RCCL carries every rank")
    ap.add_argument("--cpu-trials", type=int, default=25,
                    help="trials of the workload the CPU baseline (oracle) runs: 25 of 200 = one eighth of C3, ~25 s")
    ap.add_argument("--cpu-full", action="store_true",
                    help="CPU baseline on the WHOLE workload (one full-size oracle EM iteration, about four minutes "
                         "for C3 on one core) instead of the bounded sample; quoted once in DESIGN.md")
    ap.add_argument("--kernel-steps", type=int, default=6,
                    help="extra EM iterations after the timed region, M-step serialised, for the per-kernel timings")
    args = ap.parse_args()

    import vlgp_amd
    from vlgp_amd import _lib
    from vlgp_amd.api import FitSession
    from vlgp_amd.dist import Comm

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))  # no launcher around us: be our own (one process per GPU)
    comm = Comm.from_env()
    if comm.world != args.gpus:  # (a bare `--gpus 8` can never print an n_gpus: 1 line)
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, comm.world))
    rank, world = comm.rank, comm.world
    device = getattr(comm, "local_rank", 0) if world > 1 else 0
    device = int(os.environ.get("VLGP_DEVICE", device))  # tests: several ranks on one GPU (shm transport)

    if world > 1 and os.environ.get("VLGP_COMM_TRANSPORT", "") == "shm" and not args.allow_shm:
        raise SystemExit("bench.py --gpus %d: VLGP_COMM_TRANSPORT=shm is the one-GPU test transport; a multi-GPU number "
                         "must come from RCCL (pass --allow-shm in tests)" % args.gpus)
    trials, a0, b0, (n_trials, n_bins, N, L) = build_inputs(args.workload)
    mine = comm.shard(trials)
    total_iters = args.warmup + args.steps
    _, lik = workload_extras(args.workload, n_trials)
    sess = FitSession(mine, L, device=device, comm=comm if world > 1 else None, verbose=False,
                      a=a0.copy(), b=b0.copy(), max_iter=total_iters, min_iter=total_iters, **({"lik": lik} if lik else {}))
    eng = sess.eng
    cfg = sess.config
    n_seg_local = len(sess.segs)
    n_seg = sum(int(tr["y"].shape[0]) // cfg["window"] for tr in trials)

    for _ in range(args.warmup):
        sess.em_iteration()
    eng.profile(True)
    eng.profile_reset()
    eng.barrier()
    t0 = time.perf_counter()
    ranks_per_step = []
    for _ in range(args.steps):
        ranks_per_step.append([int(r) for r in eng.prior_ranks(cfg["window"])])  # the factor this E-step uses
        sess.em_iteration()
    eng.barrier()
    elapsed = time.perf_counter() - t0
    times = np.zeros(world)
    times[rank] = elapsed
    eng.allreduce_host(times)
    elapsed = float(times.max())
    per_rank_ms = (1e3 * times / args.steps).tolist()   # a straggler shows here
    if world > 1 and not (eng.transport == "rccl" or args.allow_shm):
        raise SystemExit("bench.py --gpus %d: transport is %r, not rccl: no line printed" % (args.gpus, eng.transport))

    kinds = (("estep", _lib.PROF_ESTEP), ("mstep", _lib.PROF_MSTEP), ("hstep", _lib.PROF_HSTEP),
             ("prior", _lib.PROF_PRIOR), ("estep_ra16", _lib.PROF_ESTEP_RA16), ("estep_ra24", _lib.PROF_ESTEP_RA24),
             ("estep_ra32", _lib.PROF_ESTEP_RA32), ("estep_long", _lib.PROF_ESTEP_LONG),
             ("estep_generic", _lib.PROF_ESTEP_GENERIC), ("estep_pass", _lib.PROF_ESTEP_PASS),
             ("estep_factor", _lib.PROF_ESTEP_FACTOR), ("estep_mean", _lib.PROF_ESTEP_MEAN),
             ("hstep_lr", _lib.PROF_HSTEP_LR), ("hstep_tab", _lib.PROF_HSTEP_TAB))
    prof_live = {k: eng.profile_get(i) for k, i in kinds}
    # Kernel-timing pass.  In the timed region the M-step lane runs beside the H-step rounds: a HIP-event pair
    # then brackets the dispatch arbitration between the two lanes as well as the kernel (both lanes are
    # throughput-bound, each kernel appears ~1.3-2x longer than it runs alone; rocprofv3 serialises the lanes and
    # reports the stand-alone durations).  The per-kernel figures below therefore come from a second session on the
    # same inputs, run after the timed region with the M-step serialised (VLGP_M_SEQUENTIAL) through the same EM
    # iterations as the timed region's first ones; the overlapped averages of the timed region are kept as
    # avg_ms_overlapped.  `value` is untouched by this.
    rt = sess.runtime
    n_timed_iters = len(rt["em_elapsed"])
    prof = prof_live
    k_steps, k_ranks = args.steps, ranks_per_step
    kernel_timing = "HIP events inside the timed region"
    omega = np.array(sess.params["omega"]).tolist()
    ranks_used = [int(r) for r in eng.prior_ranks(cfg["window"])]
    transport = eng.transport
    rccl_ranks = eng.rccl_ranks  # from ncclCommCount, not from WORLD_SIZE: what RCCL saw
    hstat = [float(v) for v in eng.hstep_stats()]
    if world == 1 and args.kernel_steps > 0:
        # (round 6) a SECOND session, warmed up like the first: its kernel_steps iterations are EM iterations
        # warmup + 1 ... warmup + kernel_steps of the same fit, i.e. the first iterations of the timed region -- same
        # ranks, same rank classes of every kernel (until round 5 they were the iterations AFTER the timed region,
        # where one latent has drifted to rank 18)
        eng.profile(False)
        sess.close()
        sess = FitSession(mine, L, device=device, comm=None, verbose=False, a=a0.copy(), b=b0.copy(),
                          max_iter=args.warmup + args.kernel_steps, min_iter=args.warmup + args.kernel_steps,
                          **({"lik": lik} if lik else {}))
        eng = sess.eng
        for _ in range(args.warmup):
            sess.em_iteration()
        os.environ["VLGP_M_SEQUENTIAL"] = "1"
        eng.profile(True)
        eng.profile_reset()
        h0 = [float(v) for v in eng.hstep_stats()]
        ranks_kernel = []
        for _ in range(args.kernel_steps):
            ranks_kernel.append([int(r) for r in eng.prior_ranks(cfg["window"])])
            sess.em_iteration()
        eng.synchronize()
        prof = {k: eng.profile_get(i) for k, i in kinds}
        hstat = [float(v) - h for v, h in zip(eng.hstep_stats(), h0)]
        k_steps, k_ranks = args.kernel_steps, ranks_kernel
        kernel_timing = ("HIP events over EM iterations %d ... %d of a second, identically warmed-up session (= the first %d "
                         "iterations of the timed region: same ranks) with the M-step lane serialised (in the timed region the "
                         "two lanes overlap and an event pair also brackets their dispatch arbitration; that figure is "
                         "avg_launch_ms_overlapped_in_timed_region)" % (args.warmup + 1, args.warmup + args.kernel_steps,
                                                                         args.kernel_steps))
        os.environ.pop("VLGP_M_SEQUENTIAL", None)
    eng.profile(False)
    timed = slice(args.warmup, args.warmup + args.steps)
    # SURVEY 8(d)'s protocol: a fit from cold with max_iter = min_iter = 10, iteration 1 dropped -- the session above
    # started cold, so these are its iterations 2 ... 10 (warm-up and timed ones alike; vlgp/core.py:307-331 timers)
    survey = None
    if len(rt["em_elapsed"]) >= 10:
        survey = {"value": 1.0 / float(np.mean(rt["em_elapsed"][1:10])), "unit": "EM it/s",
                  "ms_per_e_step": 1e3 * float(np.mean(rt["e_elapsed"][1:10])),
                  "ms_per_iteration": [round(1e3 * float(t), 3) for t in rt["em_elapsed"][:10]],
                  "protocol": "SURVEY 8(d): EM iterations 2 ... 10 of a fit from cold (max_iter = min_iter = 10, the first "
                              "dropped), 1 / mean(runtime['em_elapsed'][1:10]); `value` is the steady state after "
                              "--warmup iterations"}
    phase_ms = {k: 1e3 * float(np.mean(rt[k + "_elapsed"][timed])) for k in ("e", "m", "h", "em")}
    sess.close()

    if rank != 0:
        return

    T = cfg["window"]
    work = algorithmic_work(T, N, L, 1)
    LT = 3 if L <= 3 else (5 if L <= 5 else (8 if L <= 8 else 10))

    def entry(n, ms, per_launch, unit, bound, peak, **extra):
        d = {"launches": n, "avg_ms": ms / n, "total_ms": ms, "unit": unit, "bound": bound,
             "achieved": per_launch / (ms / n * 1e-3) / (1e12 if unit == "TFLOP/s" else 1e9), "peak": peak}
        d["frac"] = d["achieved"] / peak
        d.update(extra)
        return d

    kernels = {}
    # E-step: one entry per instantiation that ran.  "achieved" charges what the kernel executed -- the SURVEY
    # count at the EFFECTIVE ranks of the factor each launch used; "nominal" is the same count at the
    # reference's fixed rank 50 (what a dense rank-50 implementation would have to do).
    seg_local = n_seg_local
    by_ra = {16: [], 24: [], 32: []}
    for rk in k_ranks:
        rmax = max(rk)
        by_ra[16 if rmax <= 16 else (24 if rmax <= 24 else 32)].append(rk)
    split = prof["estep_pass"][0] > 0   # the E-step ran as a sequence of chip-wide launches (estep_split.hip)
    for ra, key in ((16, "estep_ra16"), (24, "estep_ra24"), (32, "estep_ra32")):
        n_e, ms_e, u_e = prof[key]
        if not n_e:
            continue
        rks = by_ra[ra] or k_ranks
        exe = float(np.mean([estep_flops(T, N, L, rk) for rk in rks])) * u_e / n_e
        nom = estep_flops(T, N, L, [50] * L) * u_e / n_e
        if split:
            name = "E-step call, max rank <= %d (sequence of esplit_* launches, one HIP-event pair around all of them)" % ra
        else:
            name = "estep_fast_kernel<%d, %d, %d>" % (LT, 16 if ra == 16 else 32, ra)
        kernels[name] = entry(n_e, ms_e, exe, "TFLOP/s", "fp64", FP64_PEAK_TFLOPS, units_per_launch=u_e / n_e,
                              flops_per_launch_executed=exe, flops_per_launch_nominal_rank50=nom,
                              achieved_nominal_rank50=nom / (ms_e / n_e * 1e-3) / 1e12,
                              frac_nominal_rank50=nom / (ms_e / n_e * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
                              mean_effective_ranks=np.mean(np.array(rks, dtype=float), axis=0).round(2).tolist(),
                              pmc_key=name, per_step_ms=ms_e / k_steps)
        if split:
            kernels[name]["sequence"] = True   # not a single kernel: never the "dominant kernel"
    if split:
        # the launches of one sweep per E-step call carry their own event pairs (sampled inside the timed region)
        sweeps = cfg["Eniter"] if "Eniter" in cfg else 25
        rk_mean = np.mean(np.array(k_ranks, dtype=float), axis=0)
        ra_typ = max(by_ra, key=lambda k: len(by_ra[k]))  # the instantiation most E-step calls used
        n_p, ms_p, u_p = prof["estep_pass"]
        fl = 6.0 * L * N * u_p / n_p  # 12 T L N per unit-sweep for the two passes
        kernels["esplit_pass<%d>" % LT] = entry(
            n_p, ms_p, fl, "TFLOP/s", "fp64", FP64_PEAK_TFLOPS, units_per_launch=u_p / n_p,
            flops_per_launch_executed=fl, sampled="one residual and one curvature pass per E-step call",
            per_step_ms=ms_p / n_p * 2 * sweeps, pmc_key="esplit_pass<%d, 1, false*" % LT,
            note="lane <-> row, the channels of a 64-row group split over four waves; the count leaves out the exp "
                 "(12 fp64 operations per (row, channel) with the table, about as many flops again)")
        n_f, ms_f, u_f = prof["estep_factor"]
        if n_f:
            fl = float(np.mean([5.0 * T * r * r + 2.0 / 3.0 * r ** 3 for r in rk_mean])) * u_f / n_f
            kernels["esplit_latent<factor>"] = entry(
                n_f, ms_f, fl, "TFLOP/s", "fp64", FP64_PEAK_TFLOPS, units_per_launch=u_f / n_f,
                flops_per_launch_executed=fl, sampled="one launch per E-step call",
                per_step_ms=ms_f / n_f * (sweeps + 1), pmc_key="esplit_lane<0*",
                note="factor + variance launches of one lane (half of the units): esplit_lane<0, *> (one LANE per (unit, "
                     "latent), ranks <= 16: a few hundred long workgroups, bound by one workgroup's latency) plus the "
                     "wave-per-task esplit_latent<*, false> launch of the latents above rank 16; the event pair also "
                     "brackets the other lane's row passes running beside it")
        n_u, ms_u, u_u = prof["estep_mean"]
        if n_u:
            fl = float(np.mean([8.0 * T * r for r in rk_mean])) * u_u / n_u
            kernels["esplit_latent<mean>"] = entry(
                n_u, ms_u, fl, "TFLOP/s", "fp64", FP64_PEAK_TFLOPS, units_per_launch=u_u / n_u,
                flops_per_launch_executed=fl, sampled="one launch per E-step call",
                per_step_ms=ms_u / n_u * sweeps, pmc_key="esplit_lane<1*")
    n_m, ms_m, u_m = prof["mstep"]
    if n_m:
        fl = work["mstep_flops_per_row"] * u_m / n_m
        kernels["mstep_accum<NEWTON>"] = entry(
            n_m, ms_m, fl, "TFLOP/s", "fp64", FP64_PEAK_TFLOPS, units_per_launch=u_m / n_m,
            flops_per_launch_executed=fl,
            bytes_per_launch_streamed=work["mstep_bytes_per_row_kernel"] * u_m / n_m,
            bytes_per_launch_survey_count=work["mstep_bytes_per_row_survey"] * u_m / n_m,
            pmc_key="mstep_accum<%d, 1, 1*" % LT, per_step_ms=ms_m / k_steps,
            avg_ms_overlapped=(prof_live["mstep"][1] / prof_live["mstep"][0]) if prof_live["mstep"][0] else None,
            note="compute-bound (exp + FMA per (row, channel)); y is read once per M-step by the PREP pass, "
                 "each Newton launch streams only mu, v")
    n_h, ms_h, u_h = prof["hstep"]
    if n_h:
        fl = work["hstep_flops_per_seg_eval"] * u_h / n_h
        hname = "hstep_round_mfma<50, 4, true>" if T <= 50 else "hstep_round_mfma<64, 4>"
        kernels[hname] = entry(
            n_h, ms_h, fl, "TFLOP/s", "fp64", FP64_PEAK_TFLOPS, units_per_launch=u_h / n_h,
            flops_per_launch_executed=fl, bytes_per_launch_algorithmic=work["hstep_bytes_per_seg_eval"] * u_h / n_h,
            pmc_key=hname, per_step_ms=ms_h / k_steps,
            avg_ms_overlapped=(prof_live["hstep"][1] / prof_live["hstep"][0]) if prof_live["hstep"][0] else None,
            note="dense round (evaluations whose kernel matrix has numerical rank > 32): algorithmic count "
                 "M (T^3 + 4 T^2) per evaluation (SURVEY 8d); the matrix-pipe kernel issues 78 "
                 "v_mfma_f64_16x16x4 = 160 kflop per segment plus the panel eliminations")
    n_l, ms_l, u_l = prof["hstep_lr"]
    if n_l:
        # the low-rank round: SURVEY 8(d)'s count M (T^3 + 4 T^2) is what the REFERENCE's algorithm needs per evaluation
        # and stays the "algorithmic" figure; what the kernel executes is the Woodbury form at the numerical rank r of
        # the kernel matrix: three pair-GEMMs (r (r + 1) / 2 columns x ceil(T / 2) depth, 2 flop per multiply-add) and
        # a symmetric Gauss-Jordan inverse (r^3 multiply-adds in the lane-per-row layout) per segment
        hs = hstat
        r_mean = hs[1] / hs[0] if hs[0] else 0.0
        nt = (T + 1) // 2
        exe_seg = 3 * (r_mean * (r_mean + 1) / 2) * nt * 2 + 2 * r_mean ** 3
        fl = work["hstep_flops_per_seg_eval"] * u_l / n_l
        lname = "hstep_round_lr<%d, 4, RC>" % (50 if T <= 50 else 64)
        kernels[lname] = entry(
            n_l, ms_l, fl, "TFLOP/s", "fp64", FP64_PEAK_TFLOPS, units_per_launch=u_l / n_l,
            flops_per_launch_executed=exe_seg * u_l / n_l, flops_per_launch_survey_count=fl,
            achieved_executed=exe_seg * u_l / n_l / (ms_l / n_l * 1e-3) / 1e12,
            frac_executed=exe_seg * u_l / n_l / (ms_l / n_l * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            mean_predicted_rank=r_mean, low_rank_evaluations=hs[0], dense_evaluations=hs[2], reruns=hs[3],
            bytes_per_launch_algorithmic=work["hstep_bytes_per_seg_eval"] * u_l / n_l,
            pmc_key="hstep_round_lr<*", per_step_ms=ms_l / k_steps,
            avg_ms_overlapped=(prof_live["hstep_lr"][1] / prof_live["hstep_lr"][0]) if prof_live["hstep_lr"][0] else None,
            frac_dense_equivalent=fl / (ms_l / n_l * 1e-3) / 1e12 / FP64_PEAK_TFLOPS,
            frac_basis="`achieved` / `frac` (= frac_dense_equivalent): the launch priced at SURVEY 8(d)'s count "
                       "M (T^3 + 4 T^2) per evaluation, the work of the reference's dense algorithm, which this kernel "
                       "does NOT execute; `achieved_executed` / `frac_executed`: the flops the low-rank form issues at the "
                       "mean rank -- that one is the utilisation of the fp64 pipes",
            note="exact low-rank (Woodbury) round, hstep_lr.h")
        n_t, ms_t, u_t = prof["hstep_tab"]
        if n_t:
            kernels["hstep_lr_tables"] = {"launches": n_t, "avg_ms": ms_t / n_t, "total_ms": ms_t,
                                          "per_step_ms": ms_t / k_steps,
                                          "note": "pivoted Cholesky + omega-tangent of the folded kernel blocks, one block "
                                                  "per evaluation, in front of every low-rank round"}
    n_p, ms_p, u_p = prof["prior"]
    if n_p:
        kernels["ichol_exact_kernel"] = {"launches": n_p, "avg_ms": ms_p / n_p, "total_ms": ms_p,
                                         "note": "bit-exact math.ichol_gauss, one launch per H-step"}
    # north_star also asks for the HBM side of the factorisation kernels: measured bytes (PMC passes on
    # file, keyed by the exact kernel name) over the live launch time, against the 8 TB/s peak
    for kd in kernels.values():
        ra, rsrc = rocprof_avg_ms(kd.get("pmc_key", ""))
        if ra:
            kd["avg_ms_rocprofv3"] = ra
            kd["avg_ms_rocprofv3_source"] = rsrc
        tb = pmc_traffic(kd.get("pmc_key", ""))
        if tb:
            kd["hbm_bytes_per_launch_pmc"] = tb
            kd["hbm_gbs"] = tb / (kd["avg_ms"] * 1e-3) / 1e9
            kd["hbm_frac"] = kd["hbm_gbs"] / HBM_PEAK_GBS
    timed_kernels = {k: v for k, v in kernels.items() if "achieved" in v and not v.get("sequence")}
    dominant = max(timed_kernels, key=lambda k: timed_kernels[k].get("per_step_ms", 0.0)) if timed_kernels else None
    roofline = None
    if dominant:
        kd = kernels[dominant]
        # the contract's "mfma" = the compute roof: on CDNA4 the fp64 matrix and vector peaks coincide (78.6 TFLOP/s)
        roofline = {"kernel": dominant, "bound": "mfma", "achieved": kd["achieved"], "peak": kd["peak"],
                    "unit": kd["unit"], "frac": kd["frac"], "traffic": kd.get("hbm_bytes_per_launch_pmc"),
                    "hbm_gbs": kd.get("hbm_gbs"), "hbm_frac": kd.get("hbm_frac"),
                    "avg_launch_ms": kd["avg_ms"], "launches": kd["launches"],
                    "avg_launch_ms_rocprofv3": kd.get("avg_ms_rocprofv3"),
                    "avg_launch_ms_rocprofv3_source": kd.get("avg_ms_rocprofv3_source"),
                    "units_per_launch": kd["units_per_launch"],
                    "algorithmic_flops_per_launch": kd.get("flops_per_launch_survey_count", kd["flops_per_launch_executed"]),
                    "executed_flops_per_launch": kd["flops_per_launch_executed"],
                    "frac_executed": kd.get("frac_executed"),
                    "frac_dense_equivalent": kd.get("frac_dense_equivalent"),
                    "frac_basis": kd.get("frac_basis"),
                    "not_live": ["traffic", "hbm_gbs", "hbm_frac", "avg_launch_ms_rocprofv3"],
                    "avg_launch_ms_overlapped_in_timed_region": kd.get("avg_ms_overlapped"),
                    "timing": kernel_timing,
                    "traffic_source": "replayed from the committed rocprofv3 --pmc passes of this command "
                                      "(profiles/r*/pmc_summary.json: FETCH_SIZE x 2 + WRITE_SIZE, separate passes); "
                                      "not collected in this run"}

    out = {
        "metric": "EM iterations/sec", "value": args.steps / elapsed, "unit": "EM it/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": ("%s: %d trials x %d bins x %d Poisson channels, %d latents, window %d -> %d segments; "
                                "Eniter=Mniter=25, rank 50, VB, Hstep on" % (args.workload, n_trials, n_bins, N, L,
                                                                              cfg["window"], n_seg)) if lik is None else
                               ("%s: %d ragged trials of 500 ... %d bins (%d bins in all) x %d channels (%d Poisson + %d "
                                "Gaussian), %d latents, window %d -> %d segments; Eniter=Mniter=25, rank 50, VB, Hstep on"
                                % (args.workload, n_trials, n_bins, sum(int(tr["y"].shape[0]) for tr in trials), N,
                                   N - C5_GAUSS, C5_GAUSS, L, cfg["window"], n_seg)),
                   "parallelism": "trials sharded over %d rank(s)%s" % (
                       world, "" if world == 1 else "; all-reduce of the M-step statistics and norms over " + transport +
                       ", H-step round sums added on the host (shared memory)"),
                   "transport": transport, "rccl_ranks": int(rccl_ranks[0]), "rccl_ranks_m_lane": int(rccl_ranks[1]),
                   "rccl_ranks_source": "ncclCommCount of the handle's communicators"},
        "ms_per_e_step": phase_ms["e"], "ms_per_m_step": phase_ms["m"], "ms_per_h_step": phase_ms["h"],
        "ms_per_step_per_rank": per_rank_ms,
        # every timed EM iteration's own timers (runtime lists of vlgp/core.py:307-331), rank 0
        "phase_ms_per_step": {k: [round(1e3 * float(t), 3) for t in rt[k + "_elapsed"][timed]] for k in ("e", "m", "h", "em")},
        "value_survey_protocol": survey["value"] if survey else None, "survey_protocol": survey,
        "roofline": roofline, "kernels": kernels,
        # the H-step beyond its kernel: dependent L-BFGS-B rounds of the timed region, their kernel time at the stand-alone
        # launch average, and what is left (launch + mailbox + host optimiser step per round, prior rebuild, M-step lane)
        "h_step": (lambda n_live, ms_live, ms_alone: {
            "rounds_per_step": n_live / args.steps,
            "kernel_ms_per_step_at_standalone_avg": n_live / args.steps * ms_alone,
            # event-bracketed in the timed region, i.e. sharing the chip with the M-step lane's launches
            "kernel_ms_per_step_beside_m_step": ms_live / args.steps,
            "non_kernel_ms_per_step": phase_ms["h"] - ms_live / args.steps,
            "non_kernel_us_per_round": 1e3 * (phase_ms["h"] - ms_live / args.steps) / max(n_live / args.steps, 1e-9),
        })(prof_live["hstep"][0] + prof_live["hstep_lr"][0],
           prof_live["hstep"][1] + prof_live["hstep_lr"][1] + prof_live["hstep_tab"][1],
           ((prof["hstep"][1] + prof["hstep_lr"][1] + prof["hstep_tab"][1]) / (prof["hstep"][0] + prof["hstep_lr"][0]))
           if prof["hstep"][0] + prof["hstep_lr"][0] else 0.0),
        "effective_rank": ranks_used, "effective_rank_per_step": ranks_per_step, "omega_final": omega,
        "kernel_pass_rank_per_step": k_ranks,  # the ranks the `kernels` / `roofline` timings were taken at
    }
    if not args.no_cpu_baseline and world == 1:
        cb = cpu_baseline(args.workload, n_trials if args.cpu_full else min(args.cpu_trials, n_trials))
        out["cpu_baseline"] = cb
        out["speedup_vs_cpu_baseline"] = out["value"] / cb["value"]
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
