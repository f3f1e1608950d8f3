"""N > 1 path on CPU: world_size-2 gloo processes shard a batch and gather the first moves."""
import os
import socket

import numpy as np
import pytest
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
    _run_world(2, 11)


def _run_world(world, total, timeout=300):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, out)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        assert p.exitcode == 0
    assert out.get(timeout=5) is True


def test_three_and_eight_rank_sharded_solve_ragged_million():
    """The real solve_sharded contract at world sizes 3 and 8 with 1 000 001 samples: ragged shards (sizes differ by one),
    results in rank order on the root."""
    _run_world(3, 1000001)
    _run_world(8, 1000001)


def test_eight_gpu_batch_rule_is_config_4():
    """bench.py --gpus 8: 125 000 problems per rank = BASELINE.json configs[3] "1M sampled x0 sharded across 8 x MI355X"."""
    import bench
    assert bench.default_batch("cdu", 8) * 8 == 1000000
    assert bench.default_batch("cdu", 1) == 100000 and bench.default_batch("cdu", 2) == 100000
    assert bench.default_batch("cstrs", 1) == 10000
    cuts = [dd.shard_bounds(1000000, r, 8) for r in range(8)]
    assert all(hi - lo == 125000 for lo, hi in cuts)


_CHILD = r'''
import os, sys, time
r = int(os.environ["RANK"])
print("rank", r, os.environ["LOCAL_RANK"], os.environ["WORLD_SIZE"], os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"], os.environ["NNMPC_JOB_KEY"], flush=True)
mode = sys.argv[1]
if mode == "fail" and r == 1:
    sys.stderr.write("rank 1 has no device\n"); sys.exit(3)
if mode in ("fail", "hang") and r != 1:
    time.sleep(120)                                   # a rank stuck in a collective nobody else attends
if mode == "hang" and r == 1:
    time.sleep(120)
'''


def test_launcher_sets_rank_env_and_stops_everything_when_a_rank_dies(tmp_path, capfd):
    """bench.launch_ranks: every child gets RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* / one NNMPC_JOB_KEY, only rank 0's stdout is
    passed through; a rank that exits non-zero (or a wall-clock limit) stops the others at once instead of leaving them in a
    collective for ever, and the failed rank is named."""
    import time
    import bench
    script = tmp_path / "child.py"
    script.write_text(_CHILD)
    rc = bench.launch_ranks(["ok"], 3, script=str(script))
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [l.split() for l in out.strip().splitlines()]
    assert len(lines) == 1 and lines[0][:4] == ["rank", "0", "0", "3"] and lines[0][4] == "127.0.0.1"      # rank 0's stdout only
    assert len(lines[0][6]) == 32                                                                         # uuid4 hex
    t0 = time.time()
    rc = bench.launch_ranks(["fail"], 3, script=str(script))
    out, err = capfd.readouterr()
    assert rc != 0 and time.time() - t0 < 30
    assert "rank 1 of 3 failed (exit code 3)" in err and "rank 1 has no device" in err
    t0 = time.time()
    rc = bench.launch_ranks(["hang"], 2, script=str(script), timeout_s=2.0)
    out, err = capfd.readouterr()
    assert rc != 0 and time.time() - t0 < 30 and "still running after 2 s" in err


def test_rendezvous_file_is_private_fresh_and_not_followed(tmp_path, monkeypatch):
    """The RCCL unique id's side channel: mode 0600, created with O_EXCL | O_NOFOLLOW and renamed into place (a pre-created
    symlink is replaced, not written through); a reader ignores a stale file under a weak key, a foreign-mode file, a short file."""
    import stat
    import time
    monkeypatch.setenv("TMPDIR", str(tmp_path))
    uid = bytes(range(128))
    key = "k1"
    path = dd._uid_path(key)
    victim = tmp_path / "victim"
    victim.write_bytes(b"precious")
    os.symlink(victim, path)                                          # somebody pre-created the path as a symlink
    assert dd.exchange_unique_id(0, 2, key, lambda: uid) == uid
    assert victim.read_bytes() == b"precious" and not os.path.islink(path)
    st = os.lstat(path)
    assert stat.S_ISREG(st.st_mode) and (st.st_mode & 0o777) == 0o600 and st.st_size == 128
    assert dd.exchange_unique_id(1, 2, key, None, timeout_s=2.0) == uid
    old = time.time() - 3600
    os.utime(path, (old, old))                                        # a leftover of a killed job that recycled the key
    for strong in (False, True):                                      # stale is stale, whatever the key: a torchrun id is reused across
        with pytest.raises(TimeoutError):                             # elastic restarts and with a user-set --rdzv-id
            dd.exchange_unique_id(1, 2, key, None, timeout_s=0.3, strong_key=strong)
    assert dd.exchange_unique_id(1, 2, key, None, timeout_s=2.0, t_start=old) == uid   # (a reader that started back then takes it)
    os.utime(path, None)
    os.chmod(path, 0o644)
    with pytest.raises(TimeoutError):
        dd.exchange_unique_id(1, 2, key, None, timeout_s=0.3)
    os.chmod(path, 0o600)
    # the checks run on the descriptor that is read: a path swapped for a symlink to a same-sized file of another kind is not followed
    other = tmp_path / "other"
    other.write_bytes(bytes(128))
    os.chmod(other, 0o600)
    os.remove(path)
    os.symlink(other, path)
    with pytest.raises(TimeoutError):
        dd.exchange_unique_id(1, 2, key, None, timeout_s=0.3)
    os.remove(path)
    # rank 0 takes its file away at exit / SIGTERM (and Comm.__init__ as soon as every rank has joined)
    assert dd.exchange_unique_id(0, 2, key, lambda: uid) == uid and os.path.exists(path) and path in dd._CLEANUP
    dd._cleanup_files()
    assert not os.path.exists(path) and not dd._CLEANUP
    monkeypatch.setenv("NNMPC_JOB_KEY", "abc-123")
    monkeypatch.setenv("MASTER_PORT", "29511")
    assert dd.job_key() == ("abc_123_29511", True)
    monkeypatch.delenv("NNMPC_JOB_KEY")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "none")
    k, strong = dd.job_key()
    assert not strong and k.startswith("29511_")
    # a torchrun id alone is not fresh: restart count and the agent's pid are part of the key
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "my-job")
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "0")
    k0, strong = dd.job_key()
    monkeypatch.setenv("TORCHELASTIC_RESTART_COUNT", "1")
    k1, _ = dd.job_key()
    assert strong and k0 != k1 and f"_p{os.getppid()}_" in k0 and k0.startswith("my_job_r0_")


def _dataset_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sim, ref = _toy_simulator()

    def gather(rows, counts):
        t = torch.from_numpy(np.ascontiguousarray(rows))
        mx = max(counts)
        pad = torch.zeros((mx, t.shape[1]), dtype=t.dtype)
        pad[:t.shape[0]] = t
        if rank == 0:
            bufs = [torch.empty_like(pad) for _ in range(world)]
            dist.gather(pad, bufs, dst=0)
            return torch.cat([bufs[r][:counts[r]] for r in range(world)], dim=0).numpy()
        dist.gather(pad, None, dst=0)
        return None
    data = sim.generate_dataset(data_filename="d.h5py", rank=rank, world=world, gather=gather, write_files=False)
    if rank == 0:
        out.put(all(np.array_equal(data[k], ref[k]) for k in ("x", "uprev", "xs", "us", "u", "status")))
    else:
        assert data is None
    dist.barrier()
    dist.destroy_process_group()


def _toy_simulator():
    """OfflineSimulator with 5 tasks x 2 chains whose simulate_chains is a deterministic stand-in (no GPU here): the records
    are functions of the chain's own setpoint / disturbance slice, so any mix-up of tasks, ranks or row order shows."""
    from industrial_nnmpc_2021_amd import linearMPC as lm
    rng = np.random.default_rng(3)
    Nx, Nu, Ny, Nd, nt, npp, T = 3, 2, 2, 1, 5, 2, 4
    sim = lm.OfflineSimulator.__new__(lm.OfflineSimulator)
    sim.num_data_gen_task, sim.num_process_per_task, sim.Nx, sim.Nu = nt, npp, Nx, Nu
    sim.x0 = sim.uprev0 = sim.A = sim.B = sim.Bd = sim.regulator = sim.ulb = sim.uub = None
    sim.target_selectors = [None] * npp
    sp = rng.standard_normal((nt * npp * T, Ny)); ds = rng.standard_normal((nt * npp * T, Nd))
    sim.setpoints, sim.disturbances = sim._split_scenarios(setpoints=sp, disturbances=ds)

    def fake(x0, uprev0, A, B, Bd, regulator, ulb, uub, selectors, setpoints, disturbances, **kw):
        nc = len(setpoints)
        f = lambda w, s: np.stack([np.tile(setpoints[c][:, :1] * s + disturbances[c][:, :1], (1, w)) for c in range(nc)])
        return dict(x=f(Nx, 1.0), uprev=f(Nu, 2.0), xs=f(Nx, 3.0), us=f(Nu, 4.0), u=f(Nu, 5.0), status=np.zeros((nc, T), np.int32))
    lm.simulate_chains, keep = fake, lm.simulate_chains
    try:
        ref = sim.generate_dataset(data_filename="d.h5py", write_files=False)           # single process: every task in one batch
    finally:
        pass
    return sim, ref


def test_chain_tasks_are_sharded_over_ranks_and_gathered_in_task_order():
    """OfflineSimulator.generate_dataset at world sizes 2 and 3 (5 tasks: ragged blocks) over a gloo gather == the single-process run."""
    for world in (2, 3):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        ctx = mp.get_context("spawn")
        out = ctx.Queue()
        procs = [ctx.Process(target=_dataset_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert out.get(timeout=5) is True


def test_bench_sizes_the_blas_pools_below_the_cpu_quota():
    """bench.py, before it imports numpy: OPENBLAS / OMP / MKL thread counts default to min(8, usable CPUs / 2) -- OpenBLAS's own
    default (one thread per logical CPU of the host) overruns a container's CPU quota and gets the GPU-polling thread throttled
    (DESIGN.md section 5); an explicit setting in the environment is left alone."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); os.environ.pop('OPENBLAS_NUM_THREADS', None); os.environ['OMP_NUM_THREADS'] = '3'; "
            "import bench; n = bench.usable_cpus(); assert n >= 1; "
            "assert os.environ['OPENBLAS_NUM_THREADS'] == str(max(1, min(8, n // 2))), os.environ['OPENBLAS_NUM_THREADS']; "
            "assert os.environ['OMP_NUM_THREADS'] == '3'; print('ok')") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr
