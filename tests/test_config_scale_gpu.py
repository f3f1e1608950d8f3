"""BASELINE.json's configurations at their real batch sizes on the GPU, against the oracle:

* configs[2]  CDU-size plant (n = 4480, m = 8960), 100 000 sampled x0 in ONE call of the default method: every result's
  status, >= 2000 random rows through independent fp64 KKT conditions (numpy), >= 32 rows -- the largest active sets of
  the batch among them -- against oracle.qp.solve_exact_box (u* to 1e-8 relative, active sets bit-exact);
* the same plant at other operating points (0.5-5 % of the bounds active);
* configs[4]  structured-NN forward, CDU architecture, 1 M states, f32 and bf16: 4096 random rows against oracle.nn and the
  exact steady-state row.
(configs[1], 10k CSTRs-size problems, is test_fullsize_gpu.test_cstrs_config_10k_batch_kkt; configs[3] needs 8 GPUs.)
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cdu():
    from industrial_nnmpc_2021_amd import synthetic
    from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
    pl = synthetic.plant("cdu", 0)
    P, tq, nu = build_regulator_matrices(pl)
    return pl, P, tq, nu


def _samples(pl, B, seed, sx):
    from industrial_nnmpc_2021_amd import synthetic
    s = synthetic.samples(pl, B, seed, sx)
    x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), axis=1)
    return x0, np.ascontiguousarray(pl["ulb"].T - s["us"]), np.ascontiguousarray(pl["uub"].T - s["us"])


def _rows(dev, rows, width):
    """rows of an HBM-resident (B, width) f64 array (B x n doubles is 3.6 GB at the CDU size: pull only what is checked)"""
    from industrial_nnmpc_2021_amd import _lib
    out = np.empty((len(rows), width))
    for i, r in enumerate(rows):
        _lib.check(_lib.load().nnmpc_memcpy_d2h(out[i].ctypes.data_as(C.c_void_p), dev.row_ptr(int(r)), width * 8), "d2h")
    return out


def _solve_on_device(qp, x0, lb, ub):
    from industrial_nnmpc_2021_amd import _lib
    D = _lib.DeviceArray
    B = x0.shape[0]
    dx, dl, du = D.from_host(x0), D.from_host(lb), D.from_host(ub)
    u, act, st, it = D((B, qp.n), np.float64), D((B, qp.words), np.uint32), D((B,), np.int32), D((B, 2), np.int32)
    qp.solve_batch_device(B, dx, dl, du, u, act, st, it)
    for a in (dx, dl, du):
        a.free()
    return u, act.to_host(), st.to_host(), it.to_host()


def _check(P, tq, nu, N, x0, lb, ub, u_dev, act, rows_kkt, rows_oracle):
    from oracle import qp as oqp
    n = P.shape[0]
    Ps = np.tril(P) + np.tril(P, -1).T
    rows = np.unique(np.concatenate((rows_kkt, rows_oracle)))
    U = _rows(u_dev, rows, n)
    bits = np.unpackbits(act[rows].view(np.uint8), axis=1, bitorder="little")[:, :2 * n].astype(bool)
    k, c = np.arange(n) // nu, np.arange(n) % nu
    au, al = bits[:, k * 2 * nu + c], bits[:, k * 2 * nu + nu + c]
    G = U @ Ps + x0[rows] @ tq.T
    LB, UB = np.tile(lb[rows], (1, N)), np.tile(ub[rows], (1, N))
    assert not (au & al).any()
    assert (U <= UB + 1e-9).all() and (U >= LB - 1e-9).all()
    assert np.abs(np.where(au, U - UB, 0)).max() == 0 and np.abs(np.where(al, U - LB, 0)).max() == 0
    scale = np.maximum(1.0, np.abs(x0[rows] @ tq.T).max(axis=1, keepdims=True))
    assert (np.abs(np.where(~(au | al), G, 0)) <= 1e-7 * scale).all()
    assert (np.where(au, -G, 1) > 0).all() and (np.where(al, G, 1) > 0).all()
    pos = {int(r): i for i, r in enumerate(rows)}
    from tests.helpers import oracle_box_rows
    for r, (xe, active) in zip(rows_oracle, oracle_box_rows(Ps, tq, nu, N, x0, lb, ub, rows_oracle)):
        ref = np.zeros(2 * n, bool)
        ref[active] = True
        i = pos[int(r)]
        assert np.abs(U[i] - xe).max() <= 1e-8 * max(1.0, np.abs(xe).max()), r
        assert np.array_equal(bits[i], ref), (r, int((bits[i] != ref).sum()))


def test_cdu_config_100k_batch_against_oracle_and_kkt():
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    pl, P, tq, nu = _cdu()
    B, N = 100000, pl["N"]
    x0, lb, ub = _samples(pl, B, 1000, 2.0)                 # the bench's own batch (rank 0)
    qp = BatchedBoxQP(P, tq, nu, max_batch=1024)
    qp.stats(reset=True)
    u, act, st, it = _solve_on_device(qp, x0, lb, ub)
    assert (st == 0).all(), np.bincount(st, minlength=3)
    s = qp.stats()
    assert s["asm_solved"] == B and s["factorizations"] == 0     # the fast path solved them, none went to the PDIP fallback
    nact = np.unpackbits(act.view(np.uint8), axis=1).sum(axis=1)
    assert 0.005 * 8960 < nact.mean() < 0.05 * 8960               # 0.5 - 5 % of the bounds active
    rng = np.random.default_rng(0)
    largest = np.argsort(-nact)[:16]                              # the rows the side-stream kernels / the tail finished
    rows_oracle = np.unique(np.concatenate((largest, rng.choice(B, 16, replace=False))))
    assert rows_oracle.size >= 32 - 2
    _check(P, tq, nu, N, x0, lb, ub, u, act, rng.choice(B, 2000, replace=False), rows_oracle)
    u.free(); qp.close()


@pytest.mark.parametrize("sx", [1.0, 3.0, 4.0])
def test_cdu_operating_points(sx):
    """Other state spreads: from a handful to several hundred active bounds per problem."""
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    pl, P, tq, nu = _cdu()
    B, N = 16384, pl["N"]
    x0, lb, ub = _samples(pl, B, 40 + int(sx), sx)
    qp = BatchedBoxQP(P, tq, nu, max_batch=1024, seg_max=16384)
    u, act, st, it = _solve_on_device(qp, x0, lb, ub)
    assert (st == 0).all(), np.bincount(st, minlength=3)
    nact = np.unpackbits(act.view(np.uint8), axis=1).sum(axis=1)
    rng = np.random.default_rng(1)
    rows_oracle = np.unique(np.concatenate((np.argsort(-nact)[:2], rng.choice(B, 2, replace=False))))
    _check(P, tq, nu, N, x0, lb, ub, u, act, rng.choice(B, 500, replace=False), rows_oracle)
    u.free(); qp.close()


@pytest.mark.parametrize("mode,tol", [("f32", 1e-4), ("bf16", 3e-2), ("bf16x3", 1e-4)])
def test_nn_config_1m_states(mode, tol):
    """[536, 832, 832, 832, 32] WithoutUprev (cdu_train.py:33, :77-80), 1 048 576 states through the device entry point."""
    from industrial_nnmpc_2021_amd import _lib
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    from oracle import nn as onn
    rng = np.random.default_rng(0)
    nx, nu, hid, B = 252, 32, 832, 1 << 20
    dims = [2 * nx + nu, hid, hid, hid, nu]
    W = []
    for i in range(4):
        W.append(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i]))
        if i < 3:
            W.append(0.05 * rng.standard_normal(dims[i + 1]))
    xscale = rng.uniform(0.5, 2.0, nx)
    x = rng.standard_normal((B, nx)); xs = 0.3 * rng.standard_normal((B, nx)); us = rng.uniform(-0.5, 0.5, (B, nu))
    x[7] = xs[7]                                            # steady-state row
    D = _lib.DeviceArray
    dx, dxs, dus, du = D.from_host(x), D.from_host(xs), D.from_host(us), D((B, nu), np.float64)
    net = StructuredNN(W, nx, nu, nnwithuprev=False, xscale=xscale, ulb=-np.ones(nu), uub=np.ones(nu), max_batch=262144,
                       use_bf16={"f32": False, "bf16": True, "bf16x3": "split"}[mode])
    net.forward_device(B, dx, None, dxs, dus, du)
    u = du.to_host()
    rows = np.concatenate(([7, 0, B - 1, 262143, 262144], rng.choice(B, 4091, replace=False)))
    ref = onn.control_input(W, x[rows], None, xs[rows], us[rows], xscale, -np.ones(nu), np.ones(nu), False)
    assert np.abs(u[rows] - ref).max() <= tol * max(1.0, np.abs(ref).max())
    assert np.array_equal(u[7], np.clip(us[7], -1, 1))       # both passes identical: exact
    assert np.isfinite(u).all() and (np.abs(u) <= 1.0).all()
    net.close()
    for a in (dx, dxs, dus, du):
        a.free()
