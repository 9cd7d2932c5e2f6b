"""Multi-GPU plumbing: one process per GPU, units block-partitioned over ranks,
parameters replicated.

The data path has two exchanges, both issued by ``libvlgp_hip.so``: RCCL
``allReduce(sum, fp64)`` over xGMI of small fused buffers -- the M-step
sufficient statistics (once per Newton iteration) and the convergence norms
-- and a host-side rank-order sum of the H-step (ll, dll) pairs (once per
round; every rank's round kernel already publishes them to its host).  E-step,
update_w/v and the final inference need no communication.

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


def shard_bounds_weighted(weights, rank, world):
    """Contiguous block [lo, hi) for ``rank`` such that the blocks' total weights are as even as a contiguous cut
    allows: cut k sits where the running sum first reaches k / world of the total (ragged trials: weight = bins, so
    that every GPU holds about the same number of rows; BASELINE configs[4]).  Every rank computes the same cuts; a
    block is never empty while there are at least ``world`` items."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world %d" % (rank, world))
    n = len(weights)
    if n < world:
        return shard_bounds(n, rank, world)
    total = float(sum(weights))
    cuts, run = [0], 0.0
    for i, w in enumerate(weights):
        run += float(w)
        while len(cuts) < world and run >= total * len(cuts) / world:
            cuts.append(i + 1)
    while len(cuts) < world:
        cuts.append(n)
    cuts.append(n)
    for k in range(1, world):                 # no empty block: at least one item each, front to back ...
        cuts[k] = max(cuts[k], cuts[k - 1] + 1)
    for k in range(world - 1, 0, -1):         # ... and enough items left for the blocks behind
        cuts[k] = min(cuts[k], n - (world - k))
    return cuts[rank], cuts[rank + 1]


def shard(items, rank, world):
    """Block of ``items`` for ``rank``: equal counts, or -- for trial dicts of unequal length -- about equal rows."""
    try:
        lengths = [int(it["y"].shape[0]) for it in items]
    except (TypeError, KeyError, IndexError, AttributeError):
        lengths = None
    if lengths and len(set(lengths)) > 1:
        lo, hi = shard_bounds_weighted(lengths, rank, world)
    else:
        lo, hi = shard_bounds(len(items), rank, world)
    return items[lo:hi]


def _rendezvous_path():
    """Where rank 0 leaves the communicator ids for the other ranks of this launch (single node).

    Every rank of one torchrun launch shares the agent as parent process and the master port; the
    parent's start time (``/proc/<ppid>/stat``) tells a recycled pid from the original, so a file left
    behind by a crashed earlier launch is never mistaken for this one's.  The files live in a per-user
    directory of mode 0700 that this uid owns (checked: a directory somebody else pre-created is refused)."""
    ppid = os.getppid()
    born = "0"
    try:
        with open("/proc/%d/stat" % ppid) as f:
            born = f.read().rsplit(")", 1)[1].split()[19]  # field 22: start time in clock ticks since boot
    except (OSError, IndexError):
        pass
    tag = "%s_%s_%s_%s" % (os.environ.get("MASTER_PORT", "0"), ppid, born, os.environ.get("TORCHELASTIC_RUN_ID", "x"))
    base = os.environ.get("VLGP_RENDEZVOUS_DIR")
    if not base:
        base = os.path.join("/tmp", "vlgp_%d" % os.getuid())
        os.makedirs(base, mode=0o700, exist_ok=True)
        st = os.lstat(base)  # somebody else may have created it first: refuse anything that is not ours alone
        import stat as _stat
        if not _stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid():
            raise RuntimeError("rendezvous directory %s is not a directory owned by uid %d; set VLGP_RENDEZVOUS_DIR"
                               % (base, os.getuid()))
        if _stat.S_IMODE(st.st_mode) != 0o700:
            os.chmod(base, 0o700)
            if _stat.S_IMODE(os.lstat(base).st_mode) != 0o700:
                raise RuntimeError("rendezvous directory %s must have mode 0700" % base)
    return os.path.join(base, "vlgp_rccl_%s.id" % tag)


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
        """Bind ``engine`` to this launch's communicators (main lane + M-step lane).

        RCCL is the transport.  If it cannot be used -- ``librccl.so`` missing on some rank, or
        ``ncclCommInitRank`` failing (on this pool e.g. ``hipIpcGetMemHandle: invalid argument`` when
        ``HSA_ENABLE_IPC_MODE_LEGACY=0`` is not exported) -- the job FAILS, on every rank, with the
        reason: a multi-GPU run that silently moved its collectives to host memory would report
        numbers that say nothing about xGMI.  The host shared-memory all-reduce exists for tests that
        put several ranks on one GPU and is strictly opt-in: ``VLGP_COMM_TRANSPORT=shm``.
        The ranks agree on "RCCL loads everywhere" through per-rank status files before anyone enters
        ``ncclCommInitRank`` (a rank that went ahead alone would block forever)."""
        if self.world == 1 and not os.environ.get("VLGP_FORCE_RCCL"):
            return
        from .engine import VlgpError, unique_id

        base = self.path or _rendezvous_path()
        if os.environ.get("VLGP_COMM_TRANSPORT") != "shm" and self.world > 1 and self.uid is None:
            try:
                unique_id()  # probes dlopen(librccl) + ncclGetUniqueId on this rank
                mine = b"ok"
            except VlgpError as err:
                mine = ("no: %s" % err).encode()
            votes = [exchange_unique_id(0 if r == self.rank else 1, 2, lambda: mine, "%s.vote%d" % (base, r))
                     for r in range(self.world)]
            bad = [(r, v) for r, v in enumerate(votes) if v != b"ok"]
            if bad:
                raise VlgpError("RCCL is unavailable on rank(s) %s (%s); set VLGP_COMM_TRANSPORT=shm to run the "
                                "ranks over the host shared-memory test transport instead"
                                % ([r for r, _ in bad], bad[0][1].decode(errors="replace")))
        if self.uid is not None:  # ids handed in by the caller
            if getattr(self, "uid_aux", None) is None:
                self.uid_aux = exchange_unique_id(self.rank, self.world, unique_id, base + ".aux")
        else:
            self.uid = exchange_unique_id(self.rank, self.world, unique_id, base)
            self.uid_aux = exchange_unique_id(self.rank, self.world, unique_id, base + ".aux")
        try:
            engine.comm_init(self.uid, self.rank, self.world, self.uid_aux)
        except VlgpError as err:
            raise VlgpError("rank %d: communicator initialisation failed (%s); no fallback is taken -- "
                            "VLGP_COMM_TRANSPORT=shm selects the host shared-memory test transport explicitly"
                            % (self.rank, err)) from err
        engine.barrier()
        if self.rank == 0:
            for path in (base, base + ".aux"):
                try:
                    os.remove(path)
                except OSError:
                    pass
        try:
            os.remove("%s.vote%d" % (base, self.rank))
        except OSError:
            pass

    def shard(self, items):
        return shard(items, self.rank, self.world)
