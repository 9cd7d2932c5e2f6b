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


def _worker(rank, world, tmp, q):
    os.environ.update({"VLGP_COMM_TRANSPORT": "shm", "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": "0", "MASTER_PORT": "29999", "VLGP_RENDEZVOUS_DIR": tmp})
    sys.path.insert(0, ROOT)
    import bench
    from vlgp_amd.api import FitSession
    from vlgp_amd.dist import Comm

    trials, a0, b0, dims = bench.build_inputs("C1")
    comm = Comm.from_env() if world > 1 else None
    mine = comm.shard(trials) if comm else trials
    sess = FitSession(mine, dims[3], device=0, comm=comm, verbose=False, a=a0.copy(), b=b0.copy(),
                      max_iter=3, min_iter=3)
    sess.run()
    res = sess.finish()
    p = res["params"]
    q.put((rank, p["a"], p["b"], p["noise"], np.array(p["omega"]), [t["ID"] for t in mine],
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


def test_two_ranks_one_gpu_match_single_process():
    import multiprocessing as mp

    ctx = mp.get_context("spawn")
    out = {}
    for world in (1, 2):
        q = ctx.Queue()
        with tempfile.TemporaryDirectory() as tmp:
            procs = [ctx.Process(target=_worker, args=(r, world, tmp, q)) for r in range(world)]
            for p in procs:
                p.start()
            res = sorted(_collect(q, procs), key=lambda r: r[0])
            for p in procs:
                p.join(timeout=120)
                assert p.exitcode == 0
        out[world] = res
    one = out[1][0]
    r0, r1 = out[2]
    # parameters are replicated: bit-identical on both ranks
    for i in (1, 2, 3, 4):
        assert np.array_equal(r0[i], r1[i])
    # and equal to the single-process fit up to the order of the row sums (the prior factor is a bit-exact
    # function of omega, so nothing amplifies the last-bit differences of the M/H-step sums)
    assert relerr(r0[1], one[1]) < 1e-6 and relerr(r0[2], one[2]) < 1e-6
    assert relerr(r0[3], one[3]) < 1e-6 and relerr(r0[4], one[4]) < 1e-6
    assert r0[5] + r1[5] == one[5]  # contiguous shards cover the trials in order
    assert r0[7] == r1[7] == one[7] == 3


def test_bench_launch_line_two_ranks_one_gpu():
    env = dict(os.environ, VLGP_COMM_TRANSPORT="shm", VLGP_DEVICE="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "C1"]
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stderr[-2000:]
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # rank 0 prints exactly one JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "strong"


def test_rccl_failure_falls_back_to_shared_memory():
    """Two ranks on ONE device make ncclCommInitRank fail on both ranks ("duplicate GPU"): the job must
    carry on through the shared-memory all-reduce (Comm.attach) and say so, not abort."""
    env = {k: v for k, v in os.environ.items() if k != "VLGP_COMM_TRANSPORT"}
    env["VLGP_DEVICE"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29534", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--steps", "2", "--warmup", "1", "--workload", "C1"]
    done = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert done.returncode == 0, done.stderr[-2000:]
    assert "using the shared-memory all-reduce" in done.stderr
    lines = [ln for ln in done.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
