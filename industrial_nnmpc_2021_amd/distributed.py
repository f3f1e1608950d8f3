"""Sample-parallel execution across the GPUs of one node.

The reference's only parallelism is "N independent OS processes / cluster jobs,
each on a contiguous slice of the scenario signal, results meeting on the file
system" (lib/linearMPC.py:786-825, lib/controller_evaluation.py:273-295).  Here:
one process per GPU (torch.distributed; backend "nccl" = RCCL on ROCm, "gloo" in
the CPU tests), contiguous shards of the sample batch, no communication during
the solves and ONE gather of the first moves at the end.
"""
import numpy as np


def shard_bounds(total, rank, world):
    """Contiguous [lo, hi) slice of ``total`` samples owned by ``rank`` (sizes differ by <= 1),
    the same cut _split_scenarios makes when the division is exact."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_rows(local, total, dst=0, group=None):
    """Gather per-rank row blocks (torch tensors, shard_bounds order) on ``dst``.

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


def solve_sharded(solve_local, X0, lb, ub, nu, dst=0, group=None, device=None):
    """Shard (X0, lb, ub) rows over the ranks, solve locally, gather the first moves on dst.

    ``solve_local(x0, lb, ub) -> u (b, n)`` numpy in/out (e.g. BatchedBoxQP.solve_batch(...)['u']).
    """
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    total = X0.shape[0]
    lo, hi = shard_bounds(total, rank, world)
    u = solve_local(X0[lo:hi], lb[lo:hi], ub[lo:hi])
    first = torch.from_numpy(np.ascontiguousarray(u[:, :nu]))
    if device is not None:
        first = first.to(device)
    return gather_rows(first, total, dst=dst, group=group)
