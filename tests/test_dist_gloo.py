"""world_size = 2 on CPU (gloo): the N > 1 path shards units and exchanges only
sums.  Checks (a) the sufficient-statistics M-step protocol reproduces the
single-process M-step when the rows are split over two ranks, (b) the RCCL
unique-id rendezvous used by vlgp_amd.dist, (c) the block partition."""
import os
import sys
import tempfile

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def _worker(rank, world, port, tmp, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as td

    from oracle import vlgp_oracle as O
    from vlgp_amd.dist import exchange_unique_id, shard_bounds

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime

    td.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    try:
        uid = exchange_unique_id(rank, world, lambda: bytes(range(128)), path=os.path.join(tmp, "uid"))
        assert uid == bytes(range(128))

        g = dict(np.load(os.path.join(GOLDEN, "mstep_mixed.npz")))
        cat = lambda k: np.concatenate(list(g[k]), axis=0)
        rows = cat("y").shape[0]
        lo, hi = shard_bounds(rows, rank, world)

        def allreduce(buf):
            t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64))
            td.all_reduce(t)
            return t.numpy()

        got = O.mstep_sharded(cat("y")[lo:hi], cat("x")[lo:hi], cat("mu")[lo:hi], cat("v")[lo:hi],
                              g["a"], g["b"], g["gauss"], 25, allreduce=allreduce)
        err = max(float(np.abs(arr - g[k + "_H_25"]).max() / np.abs(g[k + "_H_25"]).max())
                  for k, arr in zip(("a", "b", "noise"), (got[0], got[1], got[4])))
        # replicated solves must agree bit for bit across ranks
        chk = torch.from_numpy(np.concatenate([got[0].ravel(), got[1].ravel()]).copy())
        both = [torch.zeros_like(chk) for _ in range(world)]
        td.all_gather(both, chk)
        same = all(torch.equal(both[0], t) for t in both)
        q.put((rank, err, same, (lo, hi)))
    finally:
        td.destroy_process_group()


def _free_port():
    """A port nobody listens on right now (a pid-derived one can collide with a leftover of a killed run)."""
    import socket

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _collect(q, procs, limit=900.0):
    """One result per worker.  The first `import torch` after the image's pages left the cache takes minutes, so the
    limit is generous; a worker that DIED fails the test at once instead of waiting it out."""
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



def test_two_rank_mstep_protocol_and_rendezvous():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import torch  # noqa: F401  (pages the libraries in ONCE, here, before two workers import them side by side)

    port = _free_port()
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=_worker, args=(r, 2, port, tmp, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = _collect(q, procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    res.sort()
    assert res[0][3][1] == res[1][3][0]  # contiguous shards
    for rank, err, same, _ in res:
        assert err < 1e-9, (rank, err)
        assert same


def _init_worker(rank, world, port, q):
    """preprocess.initialize on a shard with the pooled factor analysis, sums through gloo."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as td

    from vlgp_amd import synth
    from vlgp_amd.dist import shard
    from vlgp_amd.preprocess import get_config, get_params, initialize

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import datetime

    td.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    try:
        class Pool:  # what an attached Engine offers: in-place sum over ranks of a float64 array
            def __init__(self):
                self.rank, self.world = rank, world

            def allreduce_host(self, arr):
                t = torch.from_numpy(arr.reshape(-1))
                td.all_reduce(t)
                return arr

        trials = synth.make_trials(7, 150, 12, 3, seed=2)  # 7 trials over 2 ranks: uneven shards
        mine = shard(trials, rank, world)
        cfg = get_config()
        params = get_params(mine, 3, omega_bound=cfg["omega_bound"])
        np.random.seed(11 + 5 * rank)  # ranks do NOT share an RNG state: rank 0's draw must be the one used
        initialize(mine, params, cfg, pool=Pool() if world > 1 else None)
        q.put((rank, params["a"], params["b"], params["noise"], np.concatenate([t["mu"] for t in mine])))
    finally:
        td.destroy_process_group()


def test_two_rank_pooled_initialisation_matches_single_process():
    """A sharded fit must start from the initialisation of the unsharded one (every rank used to
    run its own factor analysis and keep its own latents)."""
    import torch.multiprocessing as mp

    from vlgp_amd import synth
    from vlgp_amd.preprocess import get_config, get_params, initialize

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    import torch  # noqa: F401  (see above)

    port = _free_port()
    procs = [ctx.Process(target=_init_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(_collect(q, procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    trials = synth.make_trials(7, 150, 12, 3, seed=2)
    cfg = get_config()
    params = get_params(trials, 3, omega_bound=cfg["omega_bound"])
    np.random.seed(11)  # rank 0's stream
    initialize(trials, params, cfg)
    for k, i in (("a", 1), ("b", 2), ("noise", 3)):
        assert np.array_equal(res[0][i], res[1][i])  # replicated bit for bit
        assert np.abs(res[0][i] - params[k]).max() < 1e-9 * max(np.abs(params[k]).max(), 1.0), k
    mu = np.concatenate([res[0][4], res[1][4]])
    want = np.concatenate([t["mu"] for t in trials])
    assert np.abs(mu - want).max() < 1e-9 * np.abs(want).max()
