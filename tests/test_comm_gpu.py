"""The RCCL path of the library with one rank (what a 1-GPU box can run): communicator from a unique id, the gather,
max-reduce and barrier the multi-GPU bench uses, solve_sharded over it, and bench.py's own rank code (NNMPC_FORCE_DIST)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_one_rank_comm_gather_and_sharded_solve():
    from industrial_nnmpc_2021_amd import _lib, distributed as dd, synthetic
    from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    _lib.set_device(0)
    comm = dd.Comm(0, 1)
    assert comm.allreduce_max(3.25) == 3.25
    comm.barrier()
    a = np.arange(35, dtype=np.float64).reshape(7, 5)
    send, recv = _lib.DeviceArray.from_host(a), _lib.DeviceArray((7, 5), np.float64)
    comm.gather_rows(send, [7], 5, recv, root=0)
    assert np.array_equal(recv.to_host(), a)
    pl = synthetic.plant("mini_cdu", seed=1)
    P, tq, nu = build_regulator_matrices(pl)
    s = synthetic.samples(pl, 37, seed=2, sx=2.5)
    x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), axis=1)
    lb, ub = pl["ulb"].T - s["us"], pl["uub"].T - s["us"]
    qp = BatchedBoxQP(P, tq, nu, max_batch=128)
    first = dd.solve_sharded(lambda a_, b_, c_: qp.solve_batch(a_, b_, c_, first_move_only=True)["u"], x0, lb, ub, nu, comm=comm)
    assert np.array_equal(first, qp.solve_batch(x0, lb, ub)["u"][:, :nu])
    comm.close(); qp.close()


def test_bench_rank_code_with_forced_communicator():
    env = dict(os.environ, NNMPC_FORCE_DIST="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--batch", "4096", "--steps", "1", "--warmup", "1", "--no-extras",
                        "--no-pdip", "--no-host-io", "--no-cpu-baseline", "--parity-rows", "2"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 1 and line["value"] > 0 and line["status_hist"][0] == 4096
    assert line["parity"]["active_set_hamming"] == 0 and line["parity"]["max_rel_err_vs_fp64_oracle"] < 1e-8
    assert line["roofline"]["frac"] > 0 and "cpu_baseline" not in line
    assert len(r.stdout.strip().splitlines()[-1]) < 4096
