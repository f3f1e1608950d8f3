"""Constructed edge cases of the batched box QP on the GPU (reference problem: lib/linearMPC.py:484-512), every one in a
batch next to ordinary problems: a tie (x_unc exactly on a bound), lb == ub, a fully saturated horizon, the empty set,
NaN / Inf inputs and lb > ub (rejected: status NUMERIC, u = NaN, neighbours unaffected), first-move output."""
import numpy as np
import pytest

from tests.helpers import regulator_problem, batch_inputs

pytestmark = pytest.mark.gpu


def _qp(reg, **kw):
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    return BatchedBoxQP(reg.P, reg.tq, reg.nu, max_batch=128, **kw)


# "auto" finishes a call of <= 256 problems in the device tail kernel alone; "auto-rounds" forces the lock-step round kernels
METHODS = {"auto": dict(method="auto"), "auto-rounds": dict(method="auto", asm_tail_batch=-1), "pdip": dict(method="pdip"),
           "asm": dict(method="asm"), "asm-rounds": dict(method="asm", asm_tail_batch=-1)}


def _exact(reg, x0, lb, ub):
    from oracle import qp as oqp
    n = reg.N * reg.nu
    Ps = np.tril(reg.P) + np.tril(reg.P, -1).T
    info = {"nu": reg.nu}
    x = oqp.solve_exact_box(Ps, reg.tq @ x0, np.tile(lb, reg.N), np.tile(ub, reg.N), info=info)
    rows = np.zeros(2 * n, bool)
    rows[info["active"]] = True
    return x, rows


@pytest.mark.parametrize("method", ["auto", "auto-rounds", "pdip"])
def test_tie_empty_and_saturated(method):
    pl, reg = regulator_problem("mini_cdu", seed=3)
    n, nu, N = reg.N * reg.nu, reg.nu, reg.N
    s, x0, lb, ub = batch_inputs(pl, 12, 5, 2.0)
    Ps = np.tril(reg.P) + np.tril(reg.P, -1).T
    # row 0: empty set (x0 = 0 -> u* = 0)
    x0[0] = 0.0; lb[0], ub[0] = -1.0, 1.0
    # rows 1, 4: ties -- small x0 so that nothing is active, then the upper bound of input 2 is put ON the largest
    # unconstrained value of that input over the horizon.  Row 1: a hair above it (1e-12, inside the feasibility slack
    # bound_tol = 1e-9): feasible with zero multiplier => NOT active (tie rule: active iff the multiplier is > 0).
    # Row 4: 5e-10 below it (violated, but by less than the slack): the bound must come out ACTIVE with a tiny positive
    # multiplier, on the device as in the oracle.
    x0[1] *= 1e-3; lb[1], ub[1] = -1.0, 1.0
    xunc = -np.linalg.solve(Ps, reg.tq @ x0[1])
    ub[1, 2] = xunc[2::nu].max() + 1e-12
    x0[4] = x0[1]; lb[4], ub[4] = lb[1], ub[1].copy()
    ub[4, 2] = xunc[2::nu].max() - 5e-10
    # rows 2, 3: the whole horizon saturated (huge initial state)
    x0[2] *= 200.0; x0[3] *= -300.0
    qp = _qp(reg, **METHODS[method])
    out = qp.solve_batch(x0, lb, ub)
    assert (out["status"] == 0).all(), out["status"]
    assert np.abs(out["u"][0]).max() < 1e-12 and not out["active"][0].any()
    assert not out["active"][1].any() and np.abs(out["u"][1] - xunc).max() < 1e-11
    # row 4: every method the ABI exposes returns the oracle's set -- the PDIP path's polish makes a bound violated by less than
    # bound_tol active once before it accepts (kkt_check's tie round), the default path starts from the bounds x_unc violates
    assert out["active"][4].sum() == 1
    for b in range(12):
        if b == 1:
            continue
        xe, rows = _exact(reg, x0[b], lb[b], ub[b])
        assert np.abs(out["u"][b] - xe).max() <= 1e-8 * max(1.0, np.abs(xe).max()), b
        assert np.array_equal(out["active"][b], rows), b
    assert out["active"][2].sum() >= 0.9 * n and out["active"][3].sum() >= 0.9 * n     # (nearly) every variable at a bound
    qp.close()


@pytest.mark.parametrize("method", ["auto", "auto-rounds", "pdip"])
def test_equal_bounds_fix_a_variable(method):
    """ulb == uub for one input: it is fixed over the whole horizon; the rest must be the optimum of the reduced problem."""
    from oracle import qp as oqp
    pl, reg = regulator_problem("mini_cdu", seed=4)
    n, nu, N = reg.N * reg.nu, reg.nu, reg.N
    s, x0, lb, ub = batch_inputs(pl, 6, 6, 2.5)
    fixed_val = 0.25
    lb[:, 1] = fixed_val; ub[:, 1] = fixed_val
    qp = _qp(reg, **METHODS[method])
    out = qp.solve_batch(x0, lb, ub)
    assert (out["status"] == 0).all(), out["status"]
    Ps = np.tril(reg.P) + np.tril(reg.P, -1).T
    fx = np.zeros(n, bool); fx[1::nu] = True
    k, c = np.arange(n) // nu, np.arange(n) % nu
    for b in range(6):
        q = reg.tq @ x0[b]
        qr = q[~fx] + Ps[np.ix_(~fx, fx)] @ np.full(fx.sum(), fixed_val)
        lbr, ubr = np.tile(lb[b], N)[~fx], np.tile(ub[b], N)[~fx]
        info = {}
        xr = oqp.solve_exact_box(Ps[np.ix_(~fx, ~fx)], qr, lbr, ubr, info=info)
        u = out["u"][b]
        assert np.abs(u[fx] - fixed_val).max() == 0.0
        assert np.abs(u[~fx] - xr).max() <= 1e-8 * max(1.0, np.abs(xr).max())
        au, al = out["active"][b][k * 2 * nu + c], out["active"][b][k * 2 * nu + nu + c]
        assert np.array_equal(au[~fx], info["au"]) and np.array_equal(al[~fx], info["al"])
        g = Ps @ u + q                                       # a fixed variable sits on the side its multiplier is positive on
        assert ((au[fx] & (g[fx] < 0)) | (al[fx] & (g[fx] > 0)) | (~au[fx] & ~al[fx] & (np.abs(g[fx]) < 1e-9))).all()
    qp.close()


@pytest.mark.parametrize("method", ["auto", "auto-rounds", "pdip", "asm", "asm-rounds"])
def test_invalid_inputs_are_rejected_not_certified(method):
    pl, reg = regulator_problem("mini_cdu", seed=5)
    n = reg.N * reg.nu
    s, x0, lb, ub = batch_inputs(pl, 10, 7, 2.0)
    good = _qp(reg, **METHODS[method]).solve_batch(x0, lb, ub)
    x0b, lbb, ubb = x0.copy(), lb.copy(), ub.copy()
    x0b[1, 3] = np.nan
    x0b[4, 0] = np.inf
    lbb[6, 2], ubb[6, 2] = 0.5, -0.5                        # lb > ub
    ubb[8, 1] = np.nan
    qp = _qp(reg, **METHODS[method])
    out = qp.solve_batch(x0b, lbb, ubb)
    bad = [1, 4, 6, 8]
    assert (out["status"][bad] == 2).all(), out["status"]
    assert np.isnan(out["u"][bad]).all() and not out["active"][bad].any()
    ok = [b for b in range(10) if b not in bad]
    assert (out["status"][ok] == 0).all()
    assert np.abs(out["u"][ok] - good["u"][ok]).max() < 1e-9 and np.array_equal(out["active"][ok], good["active"][ok])
    qp.close()


@pytest.mark.parametrize("method", ["auto", "auto-rounds", "pdip"])
def test_first_move_output_equals_head_of_sequence(method):
    pl, reg = regulator_problem("mini_cdu", seed=6)
    s, x0, lb, ub = batch_inputs(pl, 300, 8, 2.5)           # more problems than resident slots
    qp = _qp(reg, **METHODS[method])
    full = qp.solve_batch(x0, lb, ub)
    first = qp.solve_batch(x0, lb, ub, first_move_only=True)
    assert first["u"].shape == (300, reg.nu)
    assert np.array_equal(first["u"], full["u"][:, :reg.nu])
    assert np.array_equal(first["active"], full["active"]) and np.array_equal(first["status"], full["status"])
    qp.close()


def test_auto_method_survives_an_unusable_inverse():
    """nnmpc_qp_set_inverse rejects an inverse that misses its |P Pinv - I| check; method 'auto' then runs on the PDIP path."""
    import ctypes as C
    from industrial_nnmpc_2021_amd import _lib
    pl, reg = regulator_problem("mini_cdu", seed=7)
    s, x0, lb, ub = batch_inputs(pl, 8, 9, 2.0)
    qp = _qp(reg, method="pdip")
    n = reg.N * reg.nu
    junk = np.eye(n)
    kunc = np.zeros((n, reg.tq.shape[1]))
    rc = qp._lib.nnmpc_qp_set_inverse(qp._h, junk.ctypes.data_as(C.c_void_p), kunc.ctypes.data_as(C.c_void_p))
    assert rc == _lib.EINVAL and b"not usable" in qp._lib.nnmpc_last_error()
    out = qp.solve_batch(x0, lb, ub)
    assert (out["status"] == 0).all()
    xe, rows = _exact(reg, x0[0], lb[0], ub[0])
    assert np.abs(out["u"][0] - xe).max() < 1e-8 and np.array_equal(out["active"][0], rows)
    qp.close()
