"""The N > 1 path on ONE GPU: two processes share device 0 and exchange their
M-step / H-step / norm sums through the shared-memory test transport
(VLGP_COMM_TRANSPORT=shm; RCCL itself refuses two ranks on one device and is
covered single-rank in test_gpu_parity.py).  Everything else -- sharding, the
collective call sequence, replicated solves and L-BFGS-B decisions, the bench
launch line the driver uses -- is the production code."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

from conftest import ROOT, relerr

pytestmark = pytest.mark.gpu


def _worker(rank, world, tmp, q, inject, workload="C1", iters=3):
    os.environ.update({"VLGP_COMM_TRANSPORT": "shm", "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": "0", "MASTER_PORT": "29999", "VLGP_RENDEZVOUS_DIR": tmp})
    sys.path.insert(0, ROOT)
    import bench
    from vlgp_amd import synth
    from vlgp_amd.api import FitSession
    from vlgp_amd.dist import Comm

    comm = Comm.from_env() if world > 1 else None
    if inject:
        trials, a0, b0, dims = bench.build_inputs(workload)
        kw = dict(a=a0.copy(), b=b0.copy())
    else:  # nothing injected: the pooled factor-analysis initialisation (ranks seed differently on purpose)
        trials = synth.make_trials(10, 200, 20, 3, seed=0)
        dims, kw = (10, 200, 20, 3), {}
        np.random.seed(4 + 3 * rank)
    mine = comm.shard(trials) if comm else trials
    sess = FitSession(mine, dims[3], device=0, comm=comm, verbose=False, max_iter=iters, min_iter=iters, **kw)
    assert sess.eng.transport == ("shm" if world > 1 else "none")
    sess.run()
    # The final inference builds math.ichol_gauss's factor from the fitted omega on the FULL trial length, and on
    # rank-exhausted lengths (1000 bins at the reference's fixed rank 50) its greedy pivot order is chaotic in omega
    # ("pivot hazard"): two omegas 1e-11 apart -- sharded sums associate differently -- can pick other
    # pivots and give another, equally valid incomplete factor whose posterior differs at O(0.1) in EVERY latent (the
    # latents are coupled through the rates).  The fitted omega is therefore reported and compared as it is, and the
    # final inference of a sharded run then starts from the unsharded run's omega (handed over through a file), so that
    # the posterior means compare the inference itself.
    omega_fit = np.array(sess.params["omega"])
    ref = os.path.join(os.environ["VLGP_TEST_SHARED"], "omega_%s.npy" % workload)
    if world == 1:
        np.save(ref, omega_fit)
    else:
        assert os.path.exists(ref), "the unsharded run comes first and leaves its omega at %s" % ref
        sess.params["omega"] = np.load(ref)
    res = sess.finish()
    p = res["params"]
    if world > 1:
        # ... and the run's OWN omega through the prior build and an inference of its own (api.transform: factor from
        # params["omega"] for every trial length, then core.infer from w = v = 0 -- not the state `finish` infers from, so
        # the posterior is checked for sanity, not against the hand-over's): finite, non-negative variances, the factor
        # of the fitted omega on every length of the shard
        import vlgp_amd as V

        own = dict(p, omega=omega_fit, cholesky={})
        mine2 = [{"ID": t["ID"], "y": t["y"], "mu": t["mu"].copy()} for t in mine[:4]]
        V.transform(mine2, own, res["config"])
        assert sorted(own["cholesky"]) == sorted({t["y"].shape[0] for t in mine2})
        for t2 in mine2:
            G = own["cholesky"][t2["y"].shape[0]]
            assert G.shape == p["cholesky"][t2["y"].shape[0]].shape and np.all(np.isfinite(G)) and np.any(G != 0.0)
            assert np.all(np.isfinite(t2["mu"])) and np.all(np.isfinite(t2["v"])) and np.all(t2["v"] >= 0.0)
    q.put((rank, p["a"], p["b"], p["noise"], omega_fit, [t["ID"] for t in mine],
           np.stack([t["mu"] for t in mine]), res["config"]["runtime"]["it"]))


def _collect(q, procs, limit=600.0):
    """One result per process; fails at once when a worker dies instead of waiting out the limit."""
    import queue
    import time

    got, t0 = [], time.time()
    while len(got) < len(procs):
        try:
            got.append(q.get(timeout=2))
        except queue.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or time.time() - t0 > limit:
                for p in procs:
                    if p.is_alive():
                        p.kill()
                raise AssertionError("worker exit codes %r after %.0f s" % ([p.exitcode for p in procs], time.time() - t0))
    return got


PARAM_TOL = 1e-5  # a, b, noise, posterior means: sharded against unsharded fit
OMEGA_TOL = 1e-4  # omega: limited by L-BFGS-B's own stopping rule (ftol = 2.2e-9), see the first test


def _run_worlds(worlds, inject, workload="C1", iters=3):
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = {}
    assert worlds[0] == 1, "the unsharded run leaves the omega the sharded ones infer from"
    with tempfile.TemporaryDirectory() as shared:
        os.environ["VLGP_TEST_SHARED"] = shared  # (inherited by the spawned workers)
        for world in worlds:
            q = ctx.Queue()
            with tempfile.TemporaryDirectory() as tmp:
                procs = [ctx.Process(target=_worker, args=(r, world, tmp, q, inject, workload, iters)) for r in range(world)]
                for p in procs:
                    p.start()
                res = sorted(_collect(q, procs), key=lambda r: r[0])
                for p in procs:
                    p.join(timeout=120)
                    assert p.exitcode == 0
            out[world] = res
    return out


def test_c4_full_size_two_ranks_match_single_process():
    """BASELINE configs[3] at full size (200 x 1000 x 100, 5 latents, 4000 segments) split over two ranks -- on one GPU,
    through the shared-memory transport -- against the single-process fit: two EM iterations with the H-step on
    (sharded M-step statistics, host-side exchange of every H-step round's sums, replicated solves)."""
    out = _run_worlds((1, 2), inject=True, workload="C3", iters=2)
    one, rs = out[1][0], out[2]
    for i in (1, 2, 3, 4):
        assert np.array_equal(rs[0][i], rs[1][i])           # replicated parameters: bit-identical on both ranks
    for i in (1, 2, 3):
        assert relerr(rs[0][i], one[i]) < PARAM_TOL, i
    assert relerr(rs[0][4], one[4]) < OMEGA_TOL
    assert [len(r[5]) for r in rs] == [100, 100] and sum((r[5] for r in rs), []) == one[5]
    assert relerr(np.concatenate([r[6] for r in rs]), one[6]) < PARAM_TOL
    assert all(r[7] == 2 for r in rs)


@pytest.mark.parametrize("world", [8, 7])
def test_c4_eight_way_and_uneven_seven_way_on_one_gpu(world):
    """The 8-GPU configuration's code path (BASELINE configs[3]: 25 trials per rank; and an uneven 7-way split,
    29 / 28 trials) with every rank on ONE GPU over the shared-memory transport: mailbox and exchange-segment
    sizing at eight ranks, sharded M-step statistics, the H-step round sums added on the host, against the
    single-process fit.  Two EM iterations, H-step on."""
    out = _run_worlds((1, world), inject=True, workload="C3", iters=2)
    one, rs = out[1][0], out[world]
    for r in rs[1:]:
        for i in (1, 2, 3, 4):
            assert np.array_equal(rs[0][i], r[i])            # replicated parameters: bit-identical on every rank
    for i in (1, 2, 3):
        assert relerr(rs[0][i], one[i]) < PARAM_TOL, i
    assert relerr(rs[0][4], one[4]) < OMEGA_TOL
    sizes = [len(r[5]) for r in rs]
    assert sum(sizes) == 200 and max(sizes) - min(sizes) <= 1 and sum((r[5] for r in rs), []) == one[5]
    assert relerr(np.concatenate([r[6] for r in rs]), one[6]) < PARAM_TOL
    assert all(r[7] == 2 for r in rs)


def test_ranks_on_one_gpu_match_single_process():
    """2 and 4 ranks (uneven shards of the 10 trials at 4) against the single-process fit."""
    out = _run_worlds((1, 2, 4), inject=True)
    one = out[1][0]
    for world in (2, 4):
        rs = out[world]
        # parameters are replicated: bit-identical on every rank
        for r in rs[1:]:
            for i in (1, 2, 3, 4):
                assert np.array_equal(rs[0][i], r[i])
        # and equal to the single-process fit up to the order of the row sums.  The sharded sums differ from the
        # unsharded ones in the last bit; the only amplifier left is L-BFGS-B itself (gp.py:114): SciPy stops on a
        # relative decrease of 2.2e-9 in the objective, which pins the minimiser omega to a few 1e-5 at best (a
        # last-bit change in (ll, dll) can end a line search one evaluation earlier or later), so omega is held
        # to OMEGA_TOL and what depends on it to PARAM_TOL
        for i in (1, 2, 3):
            assert relerr(rs[0][i], one[i]) < PARAM_TOL, (world, i)
        assert relerr(rs[0][4], one[4]) < OMEGA_TOL, world
        assert sum((r[5] for r in rs), []) == one[5]  # contiguous shards cover the trials in order
        assert relerr(np.concatenate([r[6] for r in rs]), one[6]) < PARAM_TOL  # full-length posterior means
        assert all(r[7] == 3 for r in rs) and one[7] == 3


def test_two_ranks_default_initialisation_matches_single_process():
    """No a, b, mu handed in: the factor-analysis initialisation is pooled over the ranks (rank 0's subsample
    draw, pooled second moments, global channel means), so the sharded fit starts where the unsharded one does."""
    out = _run_worlds((1, 2), inject=False)
    one, (r0, r1) = out[1][0], out[2]
    for i in (1, 2, 3, 4):
        assert np.array_equal(r0[i], r1[i])
        assert relerr(r0[i], one[i]) < (OMEGA_TOL if i == 4 else PARAM_TOL), i
    assert relerr(np.concatenate([r0[6], r1[6]]), one[6]) < PARAM_TOL


def test_bench_launch_line_two_ranks_one_gpu():
    env = dict(os.environ, VLGP_COMM_TRANSPORT="shm", VLGP_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "C1"]
    done = subprocess.run(cmd + ["--allow-shm"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 prints exactly one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["config"]["transport"] == "shm" and d["config"]["rccl_ranks"] == 0 and len(d["ms_per_step_per_rank"]) == 2
    # without the explicit flag a line that RCCL did not carry is refused
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert done.returncode != 0 and "must come from RCCL" in done.stderr
    assert not [ln for ln in done.stdout.splitlines() if ln.startswith("{")]


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the way the driver runs the one-GPU bench): bench.py starts
    the two ranks itself and rank 0 prints the one line -- never an n_gpus: 1 line for a --gpus 2 request."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(VLGP_COMM_TRANSPORT="shm", VLGP_DEVICE="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--workload", "C1", "--no-cpu-baseline"]
    done = subprocess.run(cmd + ["--allow-shm"], env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["transport"] == "shm" and len(d["ms_per_step_per_rank"]) == 2
    # the ranks it started refuse the test transport without the flag: non-zero status, no line
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert done.returncode != 0 and not [ln for ln in done.stdout.splitlines() if ln.startswith("{")]


def test_rccl_failure_is_loud():
    """Two ranks on ONE device make ncclCommInitRank fail ("duplicate GPU"): the job must stop with the
    reason on every rank -- a silent move to the host shared-memory transport would make a multi-GPU number
    meaningless.  The shared-memory transport is opt-in (VLGP_COMM_TRANSPORT=shm, the tests above)."""
    env = {k: v for k, v in os.environ.items() if k != "VLGP_COMM_TRANSPORT"}
    env["VLGP_DEVICE"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29534", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "C1"]
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert done.returncode != 0
    assert "communicator initialisation failed" in done.stderr and "VLGP_COMM_TRANSPORT=shm" in done.stderr
    assert not [ln for ln in done.stdout.splitlines() if ln.startswith("{")]  # no result line
