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


@pytest.mark.parametrize("sx", [1.0, 3.0, 4.0, 6.0])
def test_cdu_operating_points(sx):
    """Other state spreads: from a handful to several hundred active bounds per problem (sx = 6: 3.2 % of the 8960 bounds on
    average, sets beyond 400 -- the upper end of SURVEY 8d's 0.5-5 % regime; four oracle rows there, the largest set among them)."""
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


def test_unstable_family_at_cdu_size():
    """SURVEY 8(d)'s second synthetic family at n = 4480: spectral radius 1.03, so the reference re-parameterises u = Kx + v and
    hands cvxopt a DENSE G = tE (I + tK tB) (lib/linearMPC.py:366-382, :476-479).  The GPU solves the input-space box form
    (DenseQPRegulator._box_form); checked here against (a) the exact optimum of the ORIGINAL dense-G problem
    (oracle.qp.solve_exact on P, tq x0, G, h(x0) in v-space, mapped back the way the reference does, :507-509) on 4 rows, the
    largest active set among them -- u* to 1e-7 relative (cond(P_box) = 1.6e6), active rows of G bit for bit -- and (b) independent
    fp64 KKT conditions of the box form on 500 rows."""
    import bench
    from industrial_nnmpc_2021_amd import condense
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    from oracle import qp as oqp
    pl, reg, Pw, tqw = bench.make_unstable_problem()
    nu, n, N = reg.Nu, Pw.shape[0], pl["N"]
    assert n == 4480 and reg.reparameterize
    B = 16384
    x0, lb, ub = _samples(pl, B, 4000, 1.5)
    qp = BatchedBoxQP(Pw, tqw, nu, max_batch=1024, seg_max=16384)
    qp.stats(reset=True)
    u, act, st, it = _solve_on_device(qp, x0, lb, ub)
    assert (st == 0).all(), np.bincount(st, minlength=4)
    nact = np.unpackbits(act.view(np.uint8), axis=1).sum(axis=1)
    assert 0.005 * 8960 < nact.mean() < 0.05 * 8960
    rng = np.random.default_rng(2)
    # (b) KKT conditions of the box form
    rows = np.sort(rng.choice(B, 500, replace=False))
    U = _rows(u, rows, n)
    bits = np.unpackbits(act[rows].view(np.uint8), axis=1, bitorder="little")[:, :2 * n].astype(bool)
    k, c = np.arange(n) // nu, np.arange(n) % nu
    au, al = bits[:, k * 2 * nu + c], bits[:, k * 2 * nu + nu + c]
    Gr = U @ Pw + x0[rows] @ tqw.T
    LB, UB = np.tile(lb[rows], (1, N)), np.tile(ub[rows], (1, N))
    assert (U <= UB + 1e-9).all() and (U >= LB - 1e-9).all()
    scale = np.maximum(1.0, np.abs(x0[rows] @ tqw.T).max(axis=1, keepdims=True))
    assert (np.abs(np.where(~(au | al), Gr, 0)) <= 1e-7 * scale).all()
    assert (np.where(au, -Gr, 1) > 0).all() and (np.where(al, Gr, 1) > 0).all()
    # (a) the original problem: min 1/2 v'Pv + (tq x0)'v  s.t.  G v <= h(x0), then u = Mg v + tK tA x0
    Mg, KA = condense.constraint_map(reg.A, reg.B, reg.Krep, N)
    G = reg.tE @ Mg
    rows_o = np.unique(np.concatenate((np.argsort(-nact)[:1], rng.choice(B, 3, replace=False))))
    Uo = _rows(u, rows_o, n)
    for i, r in enumerate(rows_o):
        te = np.tile(np.concatenate((ub[r], -lb[r])), N)                  # _get_h with this sample's shifted bounds (:484-493)
        h = te - reg.tE @ (KA @ x0[r])
        info = {}
        v = oqp.solve_exact(reg.P, reg.tq @ x0[r], G, h, tol=1e-6, info=info)
        assert info["kkt"][0] < 1e-8 and info["kkt"][1] < 1e-9, info["kkt"]
        ue = Mg @ v + KA @ x0[r]
        assert np.abs(Uo[i] - ue).max() <= 1e-7 * max(1.0, np.abs(ue).max()), (r, np.abs(Uo[i] - ue).max())
        ref = np.zeros(2 * n, bool)
        ref[info["active"]] = True
        got = np.unpackbits(act[r].view(np.uint8), bitorder="little")[:2 * n].astype(bool)
        assert np.array_equal(got, ref), (r, int((got != ref).sum()))
    u.free(); qp.close()


@pytest.mark.parametrize("mode,tol,withuprev", [("f32", 1e-4, False), ("bf16", 3e-2, False), ("bf16x3", 1e-4, False),
                                                ("f32", 1e-4, True), ("bf16", 3e-2, True)])
def test_nn_config_1m_states(mode, tol, withuprev):
    """[536, 832, 832, 832, 32] WithoutUprev (what the reference uses for the CDU: cdu_train.py:33, :77-80) and the 568-input
    RegulatorLayerWithUprev BASELINE.json's config 5 names (lib/LinearMPCLayers.py:40-61), 1 048 576 states through the device
    entry point."""
    from industrial_nnmpc_2021_amd import _lib
    from industrial_nnmpc_2021_amd.nn import StructuredNN
    from oracle import nn as onn
    rng = np.random.default_rng(0)
    nx, nu, hid, B = 252, 32, 832, 1 << 20
    dims = [2 * nx + (2 if withuprev else 1) * nu, hid, hid, hid, nu]
    W = []
    for i in range(4):
        W.append(rng.standard_normal((dims[i], dims[i + 1])) * np.sqrt(2.0 / dims[i]))
        if i < 3:
            W.append(0.05 * rng.standard_normal(dims[i + 1]))
    xscale = rng.uniform(0.5, 2.0, nx)
    x = rng.standard_normal((B, nx)); xs = 0.3 * rng.standard_normal((B, nx)); us = rng.uniform(-0.5, 0.5, (B, nu))
    up = us + rng.uniform(-0.3, 0.3, (B, nu)) if withuprev else None
    x[7] = xs[7]                                            # steady-state row
    if withuprev:
        up[7] = us[7]
    D = _lib.DeviceArray
    dx, dxs, dus, du = D.from_host(x), D.from_host(xs), D.from_host(us), D((B, nu), np.float64)
    dup = D.from_host(up) if withuprev else None
    net = StructuredNN(W, nx, nu, nnwithuprev=withuprev, xscale=xscale, ulb=-np.ones(nu), uub=np.ones(nu), max_batch=262144,
                       use_bf16={"f32": False, "bf16": True, "bf16x3": "split"}[mode])
    net.forward_device(B, dx, dup, dxs, dus, du)
    u = du.to_host()
    rows = np.concatenate(([7, 0, B - 1, 262143, 262144], rng.choice(B, 4091, replace=False)))
    ref = onn.control_input(W, x[rows], up[rows] if withuprev else None, xs[rows], us[rows], xscale, -np.ones(nu), np.ones(nu), withuprev)
    assert np.abs(u[rows] - ref).max() <= tol * max(1.0, np.abs(ref).max())
    assert np.array_equal(u[7], np.clip(us[7], -1, 1))       # both passes identical: exact
    assert np.isfinite(u).all() and (np.abs(u) <= 1.0).all()
    net.close()
    for a in (dx, dxs, dus, du, dup):
        if a is not None:
            a.free()


def test_cdu_config_100k_far_field_and_first_move_calls():
    """The 100 000-problem CDU batch again after the far-field factors of its window exist (the first call of a handle runs the
    dense form and factors the window it met): same active sets and status, u* to rounding, the oracle on a few rows; then the
    first-move call (NNMPC_OUT_FIRST_MOVE: column tiles beyond the window certified by |U_j| |T_p| are skipped): active sets
    and status bit for bit, first moves to the last bit."""
    from industrial_nnmpc_2021_amd import _lib
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    pl, P, tq, nu = _cdu()
    B, N = 100000, pl["N"]
    x0, lb, ub = _samples(pl, B, 1000, 2.0)
    qp = BatchedBoxQP(P, tq, nu, max_batch=1024)
    u1, act1, st1, _ = _solve_on_device(qp, x0, lb, ub)     # dense form; the wrapper factors the window afterwards
    assert qp.stats()["asm_far_passes"] == 0
    rng = np.random.default_rng(4)
    rows = np.sort(rng.choice(B, 64, replace=False))
    U1 = _rows(u1, rows, qp.n)
    u1.free()
    qp.stats(reset=True)
    u2, act2, st2, _ = _solve_on_device(qp, x0, lb, ub)
    assert qp.stats()["asm_far_passes"] >= 1
    assert (st2 == 0).all() and np.array_equal(st1, st2) and np.array_equal(act1, act2)
    U2 = _rows(u2, rows, qp.n)
    assert np.abs(U1 - U2).max() < 1e-11
    nact = np.unpackbits(act2.view(np.uint8), axis=1).sum(axis=1)
    rows_oracle = np.unique(np.concatenate((np.argsort(-nact)[:3], rng.choice(B, 3, replace=False))))
    _check(P, tq, nu, N, x0, lb, ub, u2, act2, rng.choice(B, 1000, replace=False), rows_oracle)
    # first-move call
    D = _lib.DeviceArray
    dx, dl, du = D.from_host(x0), D.from_host(lb), D.from_host(ub)
    first, act, st, it = D((B, nu), np.float64), D((B, qp.words), np.uint32), D((B,), np.int32), D((B, 2), np.int32)
    qp.solve_batch_device(B, dx, dl, du, first, act, st, it, first_move_only=True)
    assert np.array_equal(act.to_host(), act2) and np.array_equal(st.to_host(), st2)
    F = first.to_host()
    assert np.array_equal(F[rows], U2[:, :nu])
    assert np.array_equal(F[:4096], _rows(u2, np.arange(4096), qp.n)[:, :nu])
    for a in (dx, dl, du, first, act, st, it, u2):
        a.free()
    qp.close()


def test_cdu_size_chains_against_the_oracle_step_by_step():
    """4 lock-step chains x 6 closed-loop steps at n = 4480 (simulate_offline's loop, lib/linearMPC.py:845-866): at every step
    the recorded move equals the first move of the exact optimum (oracle.qp.solve_exact_box) of the QP the recorded state
    defines, and the next recorded state is A x + B u + Bd d."""
    from industrial_nnmpc_2021_amd.chain import DeviceChains
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    from tests.helpers import oracle_box_rows
    pl, P, tq, nu = _cdu()
    N, Nx = pl["N"], pl["A"].shape[0]
    nc, T, Nd = 4, 6, 5
    rng = np.random.default_rng(12)
    Bd = rng.standard_normal((Nx, Nd)) / np.sqrt(Nx)
    qp = BatchedBoxQP(P, tq, nu, max_batch=128, seg_max=128)
    ch = DeviceChains(qp, nc, pl["A"], pl["B"], Bd, pl["ulb"], pl["uub"], 2.0 * rng.standard_normal(Nx), np.zeros(nu))
    Xs = np.repeat(0.5 * rng.standard_normal((2, nc, Nx)), 3, axis=0)
    Us = np.repeat(rng.uniform(-0.5, 0.5, (2, nc, nu)), 3, axis=0)
    Dd = np.repeat(rng.standard_normal((3, nc, Nd)), 2, axis=0)
    rec = ch.run(Xs, Us, Dd)
    assert (rec["status"] == 0).all()
    Ps = np.tril(P) + np.tril(P, -1).T
    # every (step, chain) is one QP of the regulator: x0 = [x - xs; uprev - us], bounds shifted by us (:682-689)
    X0 = np.concatenate((rec["x"] - Xs, rec["uprev"] - Us), axis=2).reshape(T * nc, -1)
    LB = (pl["ulb"].T - Us).reshape(T * nc, nu); UB = (pl["uub"].T - Us).reshape(T * nc, nu)
    sols = oracle_box_rows(Ps, tq, nu, N, X0, LB, UB, list(range(T * nc)))
    first = np.stack([xe[:nu] for xe, _ in sols]).reshape(T, nc, nu) + Us
    assert np.abs(rec["u"] - first).max() < 1e-8
    assert np.abs(rec["u"]).max() > 0.999                    # the moves do hit their bounds
    nxt = rec["x"][:-1] @ pl["A"].T + rec["u"][:-1] @ pl["B"].T + Dd[:-1] @ Bd.T
    assert np.abs(rec["x"][1:] - nxt).max() < 1e-10 and np.abs(rec["uprev"][1:] - rec["u"][:-1]).max() == 0
    ch.close(); qp.close()
