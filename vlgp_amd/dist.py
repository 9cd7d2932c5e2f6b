"""Multi-GPU plumbing: one process per GPU, units block-partitioned over ranks,
parameters replicated.

The data path has exactly one kind of exchange: RCCL ``allReduce(sum, fp64)``
over xGMI of small fused buffers -- the M-step sufficient statistics (once per
Newton iteration), the H-step (ll, dll) pairs (once per objective evaluation)
and the convergence norms -- all issued by ``libvlgp_hip.so`` on the engine's
own stream.  E-step, update_w/v and the final inference need no communication.

This module only (a) splits the trial list, (b) gets the 128-byte RCCL unique
id from rank 0 to the other ranks of the node, (c) attaches the communicator.
"""
import os
import time

__all__ = ["shard_bounds", "shard", "Comm"]


def shard_bounds(n_items, rank, world):
    """Contiguous block [lo, hi) of ``n_items`` for ``rank``; sizes differ by at most one."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(items, rank, world):
    lo, hi = shard_bounds(len(items), rank, world)
    return items[lo:hi]


def _rendezvous_path():
    # every rank of one torchrun launch shares the agent as parent process and
    # the master port; together they name the launch on this node
    tag = "%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
    return os.path.join(os.environ.get("VLGP_RENDEZVOUS_DIR", "/tmp"), "vlgp_rccl_%s.id" % tag)


def exchange_unique_id(rank, world, make_id, path=None, timeout=120.0):
    """Rank 0 calls ``make_id()`` and publishes the bytes through an atomically
    renamed file; the other ranks of the node poll for it (single-node only)."""
    path = path or _rendezvous_path()
    if world == 1:
        return make_id()
    if rank == 0:
        uid = make_id()
        tmp = "%s.%d.tmp" % (path, os.getpid())
        with open(tmp, "wb") as f:
            f.write(uid)
            f.flush()
            os.fsync(f.fileno())
        os.replace(tmp, path)
        return uid
    deadline = time.time() + timeout
    while time.time() < deadline:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if uid:
                return uid
        except FileNotFoundError:
            pass
        time.sleep(0.01)
    raise TimeoutError("rank %d: no RCCL unique id at %s after %.0fs" % (rank, path, timeout))


class Comm:
    """Rank/world of this process plus the RCCL id; ``attach`` binds an Engine."""

    def __init__(self, rank=0, world=1, uid=None, path=None):
        self.rank, self.world, self.uid, self.path = int(rank), int(world), uid, path

    @classmethod
    def from_env(cls):
        """RANK / WORLD_SIZE / LOCAL_RANK as torchrun exports them."""
        rank = int(os.environ.get("RANK", "0"))
        world = int(os.environ.get("WORLD_SIZE", "1"))
        c = cls(rank, world)
        c.local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
        return c

    def attach(self, engine):
        if self.world == 1 and not os.environ.get("VLGP_FORCE_RCCL"):
            return
        from .engine import unique_id

        base = self.path or _rendezvous_path()
        if self.uid is None:
            self.uid = exchange_unique_id(self.rank, self.world, unique_id, base)
        if getattr(self, "uid_aux", None) is None:  # second communicator: the M-step lane
            self.uid_aux = exchange_unique_id(self.rank, self.world, unique_id, base + ".aux")
        engine.comm_init(self.uid, self.rank, self.world, self.uid_aux)
        engine.barrier()
        if self.rank == 0:
            for path in (base, base + ".aux"):
                try:
                    os.remove(path)
                except OSError:
                    pass

    def shard(self, items):
        return shard(items, self.rank, self.world)
