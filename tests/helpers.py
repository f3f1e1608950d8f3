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


_JOB = None


def _oracle_box_row(arg):
    r, threads = arg
    Ps, tq, nu, N, x0, lb, ub = _JOB

    def run():
        info = {"nu": nu}
        xe = oqp.solve_exact_box(Ps, tq @ x0[r], np.tile(lb[r], N), np.tile(ub[r], N), info=info)
        return xe, info["active"]
    if threads:
        from threadpoolctl import threadpool_limits
        with threadpool_limits(limits=threads):
            return run()
    return run()


def oracle_box_rows(Ps, tq, nu, N, x0, lb, ub, rows):
    """[(u*, active rows of G)] of the listed problems by oracle.qp.solve_exact_box, in forked workers (the solves are
    independent and take seconds each at the CDU size; the workers inherit the matrices and never touch the GPU)."""
    import multiprocessing as mp
    import os
    global _JOB
    _JOB = (Ps, tq, nu, N, x0, lb, ub)
    nw = max(1, min(16, len(rows), (os.cpu_count() or 1) // 4))
    if nw == 1 or Ps.shape[0] < 1024:
        return [_oracle_box_row((int(r), 0)) for r in rows]
    with mp.get_context("fork").Pool(nw) as pool:
        return pool.map(_oracle_box_row, [(int(r), max(1, (os.cpu_count() or 1) // (2 * nw))) for r in rows])
