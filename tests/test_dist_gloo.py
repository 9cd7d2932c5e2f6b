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
    td.init_process_group("gloo", rank=rank, world_size=world)
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


def test_two_rank_mstep_protocol_and_rendezvous():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as tmp:
        procs = [ctx.Process(target=_worker, args=(r, 2, port, tmp, q)) for r in range(2)]
        for p in procs:
            p.start()
        res = [q.get(timeout=240) for _ in procs]
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    res.sort()
    assert res[0][3][1] == res[1][3][0]  # contiguous shards
    for rank, err, same, _ in res:
        assert err < 1e-9, (rank, err)
        assert same
