"""CPU suite for the pieces either side of the regulator path: the target-selector problem (reduction to the inputs, the
oracle pinned by an independent solver), the data-set layout helpers and the rank rendezvous of the multi-GPU path."""
import multiprocessing as mp
import os

import numpy as np
import pytest
import scipy.linalg
from scipy.optimize import minimize, LinearConstraint, Bounds

from oracle import qp as oqp
from industrial_nnmpc_2021_amd import controller_evaluation as ce, distributed as dd, linearMPC as lm
from industrial_nnmpc_2021_amd.target import ReducedTargetProblem


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _full_space(g):
    ts = lm.TargetSelector(A=g["A"], B=g["B"], C=g["C"], H=g["H"], Bd=g["Bd"], Cd=g["Cd"], usp=g["usp"], Rs=g["Rs"], Qs=g["Qs"],
                           ulb=g["ulb"], uub=g["uub"], backend="host")
    return ts


def test_target_fixture_oracle_pinned_by_an_independent_solver(golden_dir):
    """The fixture pairs (reference TargetSelector matrices + oracle.qp.solve_exact_eq at the qp seam) against scipy's SLSQP
    on the same full-space problem -- a different algorithm (sequential least-squares QP), not the oracle typed twice."""
    g = _load(golden_dir, "target.npz")
    ts = _full_space(g)
    Nx, Nu = g["B"].shape
    for i in range(0, g["ysp"].shape[0], 3):
        q, h, b = ts._setup_changing_matrices(g["ysp"][i][:, None], g["dhat"][i][:, None])
        z_ref = np.concatenate((g["xs"][i], g["us"][i]))
        res = minimize(lambda z: 0.5 * z @ (ts.P @ z) + q[:, 0] @ z, np.zeros(Nx + Nu), jac=lambda z: ts.P @ z + q[:, 0],
                       method="SLSQP", constraints=[{"type": "eq", "fun": lambda z: ts.tA @ z - b[:, 0], "jac": lambda z: ts.tA}],
                       bounds=[(None, None)] * Nx + [(-1.0, 1.0)] * Nu, options=dict(ftol=1e-15, maxiter=500))
        assert res.success and np.abs(res.x - z_ref).max() < 1e-5, (i, np.abs(res.x - z_ref).max())
        # and the oracle itself on the reference's matrices reproduces the stored optimum
        z = oqp.solve_exact_eq(ts.P, q, ts.G, h, ts.tA, b)
        assert np.abs(z - z_ref).max() < 1e-9


def test_reduction_to_the_inputs_is_exact(golden_dir):
    """xs = Xb b + Xu us, E us = Eb b, reduced Hessian / linear term: solving the REDUCED problem with the oracle gives the
    full-space optimum of every fixture pair (what nnmpc_ts_solve_batch is handed is the same problem)."""
    g = _load(golden_dir, "target.npz")
    red = ReducedTargetProblem(g["A"], g["B"], g["C"], g["H"], g["Bd"], g["Cd"], g["Qs"], g["Rs"], g["usp"])
    Nu = g["B"].shape[1]
    q, e, b = red.reduce(g["ysp"], g["dhat"])
    E_ = np.vstack((np.eye(Nu), -np.eye(Nu)))
    for i in range(g["ysp"].shape[0]):
        us = oqp.solve_exact_eq(red.Pr, q[i], E_, np.ones(2 * Nu), red.E, e[i])
        assert np.abs(us - g["us"][i]).max() < 1e-9
        assert np.abs(red.expand(b[i:i + 1], us[None, :])[0] - g["xs"][i]).max() < 1e-9
    assert np.linalg.eigvalsh(red.Pr).min() > 0
    # a plant with an integrator that H C does not see: the reduction refuses (the full-space host path stays available)
    A = g["A"].copy()
    w, V = np.linalg.eig(A)
    A2 = scipy.linalg.block_diag(A, np.eye(1))
    B2 = np.vstack((g["B"], np.zeros((1, Nu)))); C2 = np.hstack((g["C"], np.zeros((g["C"].shape[0], 1))))
    Bd2 = np.vstack((g["Bd"], np.zeros((1, g["Bd"].shape[1]))))
    with pytest.raises(ValueError, match="rank deficient"):
        ReducedTargetProblem(A2, B2, C2, g["H"], Bd2, g["Cd"], g["Qs"], g["Rs"], g["usp"])


def test_host_backend_solve_batch_deduplicates(golden_dir):
    g = _load(golden_dir, "target.npz")
    ts = _full_space(g)
    idx = np.array([0, 3, 0, 3, 3, 1])
    Xs, Us = ts.solve_batch(g["ysp"][idx], g["dhat"][idx])
    assert np.abs(Us - g["us"][idx]).max() < 1e-7 and np.abs(Xs - g["xs"][idx]).max() < 1e-7
    with pytest.raises(NotImplementedError):
        lm.TargetSelector(A=g["A"], B=g["B"], C=g["C"], H=g["H"], Bd=g["Bd"], Cd=g["Cd"], usp=g["usp"], Rs=g["Rs"], Qs=g["Qs"],
                          ulb=g["ulb"], uub=g["uub"], ylb=-np.ones((6, 1)), yub=np.ones((6, 1)))     # hip backend: input box only


def test_data_set_layout_helpers(tmp_path, monkeypatch):
    """_post_process_data concatenates '<task>-<proc>-<name>' task-major and averages data_gen_time (reference
    lib/controller_evaluation.py:273-295); _get_data_for_training returns (data, xscale) or the dict alone (:254-271)."""
    monkeypatch.chdir(tmp_path)
    rng = np.random.default_rng(0)
    parts = {}
    for task in range(2):
        for proc in range(2):
            d = dict(x=rng.standard_normal((5, 3)), uprev=rng.standard_normal((5, 2)), xs=rng.standard_normal((5, 3)),
                     us=rng.standard_normal((5, 2)), u=rng.standard_normal((5, 2)), data_gen_time=float(task + proc))
            parts[(task, proc)] = d
            lm._save_training_data(d, f"{task}-{proc}-set.h5py")
    data = ce._post_process_data(data_filename="set.h5py", num_data_gen_task=2, num_process_per_task=2)
    order = [(0, 0), (0, 1), (1, 0), (1, 1)]
    for k in ("x", "uprev", "xs", "us", "u"):
        assert np.array_equal(data[k], np.concatenate([parts[o][k] for o in order], axis=0))
    assert float(data["data_gen_time"]) == 1.0
    again = ce._load_training_data("set.h5py")
    assert np.array_equal(again["x"], data["x"])
    tr, xscale = ce._get_data_for_training(data=data, num_samples=12)
    assert np.allclose(xscale, 0.5 * (data["x"][:12].max(0) - data["x"][:12].min(0)))
    assert np.allclose(tr["xs"] * xscale, data["xs"][:12]) and np.array_equal(tr["uprev"], data["uprev"][:12])
    assert isinstance(ce._get_data_for_training(data=data, num_samples=7, scale=False), dict)
    assert ce.get_data_for_training is ce._get_data_for_training


def _uid_rank(rank, key, q):
    uid = dd.exchange_unique_id(rank, 2, key, lambda: bytes(range(128)), timeout_s=20.0)
    q.put((rank, uid))


def test_unique_id_reaches_every_rank_through_tmp():
    """The side channel of nnmpc_comm_init: rank 0 publishes 128 bytes atomically, the others wait for the file."""
    key = f"test_{os.getpid()}"
    ctx = mp.get_context("fork")
    q = ctx.Queue()
    late = ctx.Process(target=_uid_rank, args=(1, key, q)); late.start()      # the reader starts first and has to wait
    first = ctx.Process(target=_uid_rank, args=(0, key, q)); first.start()
    got = dict(q.get(timeout=30) for _ in range(2))
    late.join(30); first.join(30)
    assert got[0] == got[1] == bytes(range(128))
    os.remove(dd._uid_path(key))


def test_bench_launches_its_own_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 2` without WORLD_SIZE starts 2 children with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set
    (here, without GPUs, every child refuses loudly and the parent reports failure)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the launcher is exercised by the RCCL tests")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "needs a GPU per rank" in r.stderr
