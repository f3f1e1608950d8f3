"""Shared builders for the parity tests (oracle side)."""
import numpy as np

from industrial_nnmpc_2021_amd import condense as pc, synthetic
from oracle import condense as oc, qp as oqp


def regulator_problem(name, seed=0, rho=0.97):
    """(plant dict, oracle DenseRegulator) for a synthetic plant."""
    pl = synthetic.plant(name, seed, rho)
    reg = oc.setup_regulator(pl["A"], pl["B"], pl["Q"], pl["R"], pl["S"], pl["N"], pl["ulb"], pl["uub"])
    return pl, reg


def batch_inputs(pl, B, seed=1, sx=1.0):
    """x0 (B, n_aug), lb, ub (B, nu) exactly as get_control_sequence forms them
    (reference lib/linearMPC.py:682-689)."""
    s = synthetic.samples(pl, B, seed, sx)
    x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), axis=1)
    lb = pl["ulb"].T - s["us"]
    ub = pl["uub"].T - s["us"]
    return s, x0, lb, ub


def oracle_solve(reg, x0, lb, ub):
    """Exact optimum + active rows (reference G row order) for every sample."""
    B = x0.shape[0]
    n = reg.N * reg.nu
    U = np.empty((B, n))
    act = np.zeros((B, 2 * n), bool)
    Ps = np.tril(reg.P) + np.tril(reg.P, -1).T
    for b in range(B):
        G, h = oqp.box_as_Gh(reg.nu, reg.N, lb[b], ub[b])
        info = {}
        U[b] = oqp.solve_exact(Ps, reg.tq @ x0[b], G, h, info=info)
        act[b, info["active"]] = True
        assert info["kkt"][0] < 1e-6 * max(1.0, np.abs(reg.tq @ x0[b]).max()) and info["kkt"][1] < 1e-9
    return U, act


def oracle_box_rows(Ps, tq, nu, N, x0, lb, ub, rows):
    """[(u*, active rows of G)] of the listed problems by oracle.qp.solve_exact_box.  At the CDU size a solve takes seconds, so
    the rows are shared out over worker PROCESSES -- started as fresh interpreters (`python -m tests.oracle_worker`), never
    forked: a child forked from a process that has touched the GPU inherits the KFD descriptors and objects whose __del__ calls
    into the runtime (undefined behaviour on ROCm).  Matrices and results travel as .npz files under /dev/shm."""
    import os
    import shutil
    import subprocess
    import sys
    import tempfile
    rows = [int(r) for r in rows]
    nw = max(1, min(16, len(rows), (os.cpu_count() or 1) // 4))
    if nw == 1 or Ps.shape[0] < 1024:
        out = []
        for r in rows:
            info = {"nu": nu}
            xe = oqp.solve_exact_box(Ps, tq @ x0[r], np.tile(lb[r], N), np.tile(ub[r], N), info=info)
            out.append((xe, info["active"]))
        return out
    keep = np.array(sorted(set(rows)), dtype=int)           # only the listed rows travel (the batch can be hundreds of MB)
    sub = {int(r): i for i, r in enumerate(keep)}
    d = tempfile.mkdtemp(prefix="nnmpc_test_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        np.savez(os.path.join(d, "job.npz"), Ps=Ps, tq=tq, nu=nu, N=N, x0=x0[keep], lb=lb[keep], ub=ub[keep])
        threads = max(1, (os.cpu_count() or 1) // (2 * nw))
        env = dict(os.environ, OMP_NUM_THREADS=str(threads), OPENBLAS_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads))
        procs = [subprocess.Popen([sys.executable, "-m", "tests.oracle_worker", d, str(w), str(nw)], cwd=root, env=env) for w in range(nw)]
        codes = [p.wait() for p in procs]
        if any(codes):
            raise RuntimeError(f"oracle worker exit codes {codes}")
        res = {}
        for w in range(nw):
            with np.load(os.path.join(d, f"out{w}.npz"), allow_pickle=False) as f:
                for i in f["rows"]:
                    res[int(i)] = (f[f"x{i}"], f[f"a{i}"])
        return [res[sub[r]] for r in rows]
    finally:
        shutil.rmtree(d, ignore_errors=True)
