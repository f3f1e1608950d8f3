"""Far-field form of the full-width pass (qp_wide.h, nnmpc_qp_set_farfield) against the dense form and the oracle.

x[W:] = U (Vx x0 + Vl lam) with U [Vx | Vl] = [Kunc[W:] | -Pinv[W:, 0:W]] -- verified on the device -- must give the dense
form's results (sequence calls), and first-move calls, which skip the column tiles |U_j| |T_p| <= min(ub, -lb) certifies,
must return the same active sets, status and first moves as sequence calls.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _mid(seed=3):
    from industrial_nnmpc_2021_amd import synthetic
    from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
    pl = synthetic.plant("mid_cdu", seed)
    P, tq, nu = build_regulator_matrices(pl)
    return pl, P, tq, nu


def _batch(pl, B, seed, sx, bound_scale=None):
    from industrial_nnmpc_2021_amd import synthetic
    s = synthetic.samples(pl, B, seed, sx)
    x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), axis=1)
    lb, ub = pl["ulb"].T - s["us"], pl["uub"].T - s["us"]
    if bound_scale is not None:                       # per-problem boxes of very different widths around zero
        w = np.random.default_rng(seed + 1).uniform(*bound_scale, (B, 1))
        lb, ub = lb * w, ub * w
    return x0, np.ascontiguousarray(lb), np.ascontiguousarray(ub)


def test_factors_are_verified_and_bad_ones_refused():
    from industrial_nnmpc_2021_amd import _lib
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    import ctypes as C
    pl, P, tq, nu = _mid()
    n = P.shape[0]
    qp = BatchedBoxQP(P, tq, nu, max_batch=512, farfield="auto")
    assert n % 128 == 0
    r = qp.prepare_farfield(256)
    assert 0 < r <= tq.shape[1] + 8                   # numerical rank ~ the (augmented) state dimension, far below n_aug + W
    assert qp.prepare_farfield(200) == 0 and qp.prepare_farfield(n) == 0     # not a window the pass can use
    rng = np.random.default_rng(0)
    U, Vx, Vl = rng.standard_normal((n - 128, 4)), rng.standard_normal((4, tq.shape[1])), rng.standard_normal((4, 128))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = _lib.load().nnmpc_qp_set_farfield(qp._h, 128, 4, p(U), p(Vx), p(Vl))
    assert rc == _lib.EINVAL and b"not usable" in _lib.load().nnmpc_last_error()
    qp.close()


@pytest.mark.parametrize("sx,bound_scale", [(1.5, None), (3.0, None), (2.0, (0.05, 1.0))])
def test_far_field_equals_dense_form_and_first_move_calls_equal_sequence_calls(sx, bound_scale):
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    from tests.helpers import oracle_box_rows
    pl, P, tq, nu = _mid()
    N, n = pl["N"], P.shape[0]
    B = 3000                                          # > 256: the lock-step rounds, not the device tail
    x0, lb, ub = _batch(pl, B, 11, sx, bound_scale)
    dense = BatchedBoxQP(P, tq, nu, max_batch=512, farfield=None)
    far = BatchedBoxQP(P, tq, nu, max_batch=512, farfield="auto")
    a = dense.solve_batch(x0, lb, ub)
    far.solve_batch(x0, lb, ub)                       # runs in the dense form and factors the windows it met
    far.stats(reset=True)
    b = far.solve_batch(x0, lb, ub)
    assert far.stats()["asm_far_passes"] > 0
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    assert np.array_equal(a["active"], b["active"])
    assert np.abs(a["u"] - b["u"]).max() < 1e-11
    c = far.solve_batch(x0, lb, ub, first_move_only=True)
    assert np.array_equal(c["active"], b["active"]) and np.array_equal(c["status"], b["status"])
    assert np.array_equal(c["u"], b["u"][:, :nu])
    d = dense.solve_batch(x0, lb, ub, first_move_only=True)
    assert np.array_equal(d["active"], a["active"]) and np.array_equal(d["u"], a["u"][:, :nu])
    # a few rows against the exact optimum
    Ps = np.tril(P) + np.tril(P, -1).T
    rows = [0, 1, int(np.argmax(b["active"].sum(axis=1)))]
    for r, (xe, active) in zip(rows, oracle_box_rows(Ps, tq, nu, N, x0, lb, ub, rows)):
        ref = np.zeros(2 * n, bool); ref[active] = True
        assert np.abs(b["u"][r] - xe).max() <= 1e-8 * max(1.0, np.abs(xe).max())
        assert np.array_equal(b["active"][r], ref)
    dense.close(); far.close()


def test_steady_state_input_on_a_bound_is_never_skipped():
    """us = uub: the shifted upper bounds are 0, min(ub, -lb) = 0 -- the certificate covers nothing and every tile is evaluated;
    results equal the dense form's."""
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    pl, P, tq, nu = _mid()
    B = 1500
    x0, lb, ub = _batch(pl, B, 5, 1.0)
    ub[:, ::2] = 0.0                                  # every second input sits on its upper bound at steady state
    lb[:, 1::4] = 0.0
    dense = BatchedBoxQP(P, tq, nu, max_batch=512, farfield=None)
    far = BatchedBoxQP(P, tq, nu, max_batch=512, farfield="auto")
    a = dense.solve_batch(x0, lb, ub, first_move_only=True)
    far.solve_batch(x0, lb, ub)
    b = far.solve_batch(x0, lb, ub, first_move_only=True)
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["active"], b["active"])
    assert np.abs(a["u"] - b["u"]).max() < 1e-11
    dense.close(); far.close()


def test_bound_violated_beyond_the_window_is_found_in_first_move_calls():
    """Re-entry: problems whose bounds tighten along the horizon cannot be built with per-stage-constant bounds, so the far
    violation is provoked the other way -- a tiny box (everything saturates early, the window grows) next to wide ones in one
    batch: windows differ between rounds, several factorisations are used, results equal the dense form's."""
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    pl, P, tq, nu = _mid(seed=7)
    B = 2000
    x0, lb, ub = _batch(pl, B, 21, 4.0, (0.02, 0.6))
    dense = BatchedBoxQP(P, tq, nu, max_batch=512, farfield=None)
    far = BatchedBoxQP(P, tq, nu, max_batch=512, farfield="auto")
    a = dense.solve_batch(x0, lb, ub)
    for _ in range(3):                                # every call may meet (and then factor) one more window
        b = far.solve_batch(x0, lb, ub)
    c = far.solve_batch(x0, lb, ub, first_move_only=True)
    assert np.array_equal(a["status"], b["status"]) and np.array_equal(a["active"], b["active"]) and np.array_equal(a["active"], c["active"])
    ok = a["status"] == 0
    assert ok.mean() > 0.95
    assert np.abs(a["u"][ok] - b["u"][ok]).max() < 1e-10 and np.array_equal(c["u"][ok], b["u"][ok][:, :nu])
    dense.close(); far.close()
