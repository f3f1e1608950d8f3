"""Sample-parallel execution across the GPUs of one node.

The reference's only parallelism is "N independent OS processes / cluster jobs, each on a contiguous slice of the
scenario signal, results meeting on the file system" (lib/linearMPC.py:786-825, lib/controller_evaluation.py:273-295).
Here: one process per GPU, contiguous shards of the sample batch, no communication during the solves and ONE gather of
the first moves at the end -- RCCL over xGMI through the library itself (``Comm`` = nnmpc_comm_*; no other GPU runtime
binding is needed).  The same sharding contract runs over a ``torch.distributed`` group (gloo) in the CPU tests.
"""
import ctypes as C
import os
import time

import numpy as np

from . import _lib


def shard_bounds(total, rank, world):
    """Contiguous [lo, hi) slice of ``total`` samples owned by ``rank`` (sizes differ by <= 1),
    the same cut _split_scenarios makes when the division is exact."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


_T_START = time.time()            # this process's start, near enough (the module is imported before any communicator exists)


def job_key():
    """Key of the rendezvous file, the same on every rank of ONE launch and different between launches: a nonce the launcher
    generated.  ``NNMPC_JOB_KEY`` (bench.py's self-launch exports a fresh uuid4 per launch) is taken as it is.
    ``TORCHELASTIC_RUN_ID`` (torch.distributed.run, when not the default "none") is NOT fresh by itself -- the id survives elastic
    restarts and a user-set --rdzv-id survives whole runs -- so TORCHELASTIC_RESTART_COUNT and the pid of the elastic agent (the
    common parent of the ranks of a node) go into the key with it: a restarted or repeated job looks at another path.  Without
    either -- ranks started by hand -- MASTER_PORT + the parent's pid has to do (weak).  Whatever the key, a reader only accepts a
    file written after it started itself (exchange_unique_id)."""
    port = os.environ.get("MASTER_PORT", "0")
    clean = lambda t: "".join(ch if ch.isalnum() else "_" for ch in t)[:64]
    nonce = os.environ.get("NNMPC_JOB_KEY", "")
    if nonce:
        return clean(nonce) + "_" + port, True
    run_id = os.environ.get("TORCHELASTIC_RUN_ID", "")
    if run_id and run_id != "none":
        return f"{clean(run_id)}_r{clean(os.environ.get('TORCHELASTIC_RESTART_COUNT', '0'))}_p{os.getppid()}_{port}", True
    return f"{port}_{os.getppid()}", False


def _uid_path(key):
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), f"nnmpc_uid_{os.getuid()}_{key}")


_CLEANUP = set()                  # rendezvous files this process (rank 0) wrote and has not removed yet


def _cleanup_files(*_a):
    for path in list(_CLEANUP):
        try:
            os.unlink(path)
        except OSError:
            pass
        _CLEANUP.discard(path)


def _install_cleanup():
    """Rank 0 removes its rendezvous file on any exit it can see (normal exit, SIGTERM): a file left behind by a rank 0 that died
    inside ncclCommInitRank is what a later launch with a recycled key would otherwise trip over."""
    import atexit
    import signal
    if getattr(_install_cleanup, "done", False):
        return
    _install_cleanup.done = True
    atexit.register(_cleanup_files)
    try:
        prev = signal.getsignal(signal.SIGTERM)

        def on_term(signum, frame):
            _cleanup_files()
            if callable(prev):
                prev(signum, frame)
            else:
                signal.signal(signal.SIGTERM, signal.SIG_DFL)
                os.kill(os.getpid(), signal.SIGTERM)
        signal.signal(signal.SIGTERM, on_term)
    except (ValueError, OSError):                      # not the main thread: atexit has to do
        pass


FRESH_SLACK_S = 120.0             # a usable id was written no earlier than this long before the reader started


def exchange_unique_id(rank, world, key, make_id, timeout_s=120.0, strong_key=True, t_start=None):
    """The 128-byte RCCL unique id of rank 0 reaches the other ranks through a file under $TMPDIR (all ranks of a job are
    on ONE node).  ``key`` must be the same on every rank and unique to the launch (``job_key()``).

    Rank 0 removes whatever sits at the path (a leftover of a killed job with a recycled key), writes a private temporary
    (O_EXCL | O_NOFOLLOW, mode 0600) and renames it into place: readers see all 128 bytes or no file, and a pre-created
    symlink is replaced, not followed; it removes the file again at exit / on SIGTERM (and, normally, as soon as every rank has
    joined: Comm.__init__).  Readers OPEN first (O_NOFOLLOW) and check the descriptor they will read from -- a regular file of
    128 bytes that this user owns with mode 0600, written no earlier than FRESH_SLACK_S before this reader started, for
    strong and weak keys alike: a stale id would send ncclCommInitRank into a rendezvous nobody else attends, and RCCL has no
    timeout.  (``strong_key`` is kept for callers; it no longer relaxes any check.)"""
    import stat as _stat
    path = _uid_path(key)
    if rank == 0:
        uid = make_id()
        try:
            os.unlink(path)
        except FileNotFoundError:
            pass
        tmp = f"{path}.{os.getpid()}.{int.from_bytes(os.urandom(4), 'little'):08x}"
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL | getattr(os, "O_NOFOLLOW", 0), 0o600)
        try:
            os.write(fd, uid)
        finally:
            os.close(fd)
        _install_cleanup()
        _CLEANUP.add(path)
        os.replace(tmp, path)
        return uid
    t_ref = _T_START if t_start is None else t_start
    t0 = time.time()
    while time.time() - t0 < timeout_s:
        fd = -1
        try:
            fd = os.open(path, os.O_RDONLY | getattr(os, "O_NOFOLLOW", 0) | getattr(os, "O_NONBLOCK", 0))
            st = os.fstat(fd)                          # the object that is read, not whatever the path names a moment later
            ok = (_stat.S_ISREG(st.st_mode) and st.st_uid == os.getuid() and (st.st_mode & 0o077) == 0 and st.st_size == 128
                  and st.st_mtime >= t_ref - FRESH_SLACK_S)
            if ok:
                uid = os.read(fd, 129)
                if len(uid) == 128:
                    return uid
        except OSError:                                # not there yet, a symlink (ELOOP), not ours to open
            pass
        finally:
            if fd >= 0:
                os.close(fd)
        time.sleep(0.01)
    raise TimeoutError(f"rank {rank}: no usable RCCL unique id at {path} after {timeout_s} s")


class Comm:
    """RCCL communicator of one rank (nnmpc_comm_*).  The HIP device of this rank must be current (_lib.set_device)."""

    def __init__(self, rank, world, key=None, uid=None):
        lib = _lib.load()
        self._lib, self.rank, self.world = lib, int(rank), int(world)
        self._path = None
        if uid is None:
            def make():
                buf = C.create_string_buffer(128)
                _lib.check(lib.nnmpc_comm_unique_id(buf), "nnmpc_comm_unique_id")
                return buf.raw
            if world == 1:
                uid = make()
            else:
                strong = True
                if key is None:
                    key, strong = job_key()
                uid = exchange_unique_id(self.rank, self.world, key, make, strong_key=strong)
                self._path = _uid_path(key)
        self._h = C.c_void_p()
        _lib.check(lib.nnmpc_comm_init(C.byref(self._h), C.c_char_p(uid), self.rank, self.world), "nnmpc_comm_init")
        if self._path and self.rank == 0:
            self.barrier()                          # everybody has read the id: the file can go
            _cleanup_files()
        elif self._path:
            self.barrier()

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nnmpc_comm_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def barrier(self):
        _lib.check(self._lib.nnmpc_comm_barrier(self._h), "nnmpc_comm_barrier")

    def allreduce_max(self, value):
        v = C.c_double(float(value))
        _lib.check(self._lib.nnmpc_comm_allreduce_max(self._h, C.byref(v)), "nnmpc_comm_allreduce_max")
        return v.value

    def gather_rows(self, local, rows, row_doubles, recv=None, root=0):
        """``local``: device buffer (data_ptr()) with rows[rank] x row_doubles f64; ``recv``: device buffer with
        sum(rows) x row_doubles on ``root``.  One gather over xGMI; ragged shards allowed."""
        arr = (C.c_int64 * self.world)(*[int(r) for r in rows])
        q = lambda a: None if a is None else C.c_void_p(a.data_ptr())
        _lib.check(self._lib.nnmpc_comm_gather_rows(self._h, q(local), arr, int(row_doubles), q(recv), int(root)),
                   "nnmpc_comm_gather_rows")


def gather_rows(local, total, dst=0, group=None):
    """Gather per-rank row blocks (torch tensors, shard_bounds order) on ``dst`` over a torch.distributed group.

    Returns the (total, ...) tensor on dst, None elsewhere.  One collective.
    """
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    sizes = [shard_bounds(total, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, bufs, dst=dst, group=group)
        return torch.cat([bufs[r][:hi - lo] for r, (lo, hi) in enumerate(sizes)], dim=0)
    dist.gather(pad, None, dst=dst, group=group)
    return None


def solve_sharded(solve_local, X0, lb, ub, nu, dst=0, group=None, device=None, comm=None):
    """Shard (X0, lb, ub) rows over the ranks, solve locally, gather the first moves on dst.

    ``solve_local(x0, lb, ub) -> u (b, n) or (b, nu)`` numpy in/out (e.g. BatchedBoxQP.solve_batch(...)['u']).
    With ``comm`` (a ``Comm``) the gather is the library's RCCL one and the result is a numpy array on dst;
    otherwise a torch.distributed group (gloo in the CPU tests) carries it and a tensor comes back.
    """
    if comm is not None:
        world, rank = comm.world, comm.rank
    else:
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    total = X0.shape[0]
    lo, hi = shard_bounds(total, rank, world)
    u = solve_local(X0[lo:hi], lb[lo:hi], ub[lo:hi])
    first = np.ascontiguousarray(u[:, :nu])
    if comm is not None:
        rows = [shard_bounds(total, r, world)[1] - shard_bounds(total, r, world)[0] for r in range(world)]
        send = _lib.DeviceArray.from_host(first) if first.size else None
        recv = _lib.DeviceArray((total, nu), np.float64) if rank == dst else None
        comm.gather_rows(send, rows, nu, recv, root=dst)
        return recv.to_host() if rank == dst else None
    first = torch.from_numpy(first)
    if device is not None:
        first = first.to(device)
    return gather_rows(first, total, dst=dst, group=group)
