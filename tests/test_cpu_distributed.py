"""N > 1 path on CPU: world_size-2 gloo processes shard a batch and gather the first moves."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from industrial_nnmpc_2021_amd import distributed as dd


def test_shard_bounds_cover_everything():
    for total in (0, 1, 7, 16, 1001):
        for world in (1, 2, 3, 8):
            cuts = [dd.shard_bounds(total, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1


def _fake_solver(x0, lb, ub):
    # deterministic stand-in for the GPU solve (no GPU in this test): clip of a linear map
    n = 6
    W = np.arange(x0.shape[1] * n).reshape(x0.shape[1], n) / 10.0
    return np.clip(x0 @ W, np.tile(lb, 3), np.tile(ub, 3))


def _worker(rank, world, port, total, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rng = np.random.default_rng(5)                     # same stream on every rank
    X0 = rng.standard_normal((total, 4)); lb = -np.ones((total, 2)); ub = np.ones((total, 2))
    res = dd.solve_sharded(_fake_solver, X0, lb, ub, nu=2, dst=0)
    if rank == 0:
        full = _fake_solver(X0, lb, ub)[:, :2]
        out.put(bool(np.array_equal(res.numpy(), full)))
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_solve_and_gather():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 11, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True
