"""BASELINE.json's full sizes (CDU n = 4480, CSTRs n = 540, batch 10k) through size-independent
properties: fp64 KKT conditions evaluated independently in numpy, odd symmetry, batch
permutation invariance / determinism, trivial problems, ragged batch sizes."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _setup(name, B, seed, sx, **kw):
    from industrial_nnmpc_2021_amd import synthetic
    from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    pl = synthetic.plant(name, 0)
    P, tq, nu = build_regulator_matrices(pl)
    s = synthetic.samples(pl, B, seed, sx)
    x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), axis=1)
    return pl, P, tq, nu, x0, pl["ulb"].T - s["us"], pl["uub"].T - s["us"], BatchedBoxQP(P, tq, nu, **kw)


def _kkt(P, tq, nu, N, x0, lb, ub, out, stat_tol):
    """Independent fp64 check of every returned solution."""
    Ps = np.tril(P) + np.tril(P, -1).T
    U = out["u"]
    G = U @ Ps + x0 @ tq.T                     # gradient rows
    LB, UB = np.tile(lb, (1, N)), np.tile(ub, (1, N))
    n = P.shape[0]
    k, c = np.arange(n) // nu, np.arange(n) % nu
    au, al = out["active"][:, k * 2 * nu + c], out["active"][:, k * 2 * nu + nu + c]
    assert not (au & al).any()
    assert (U <= UB + 1e-9).all() and (U >= LB - 1e-9).all()                 # primal feasibility
    assert np.abs(np.where(au, U - UB, 0)).max() == 0 and np.abs(np.where(al, U - LB, 0)).max() == 0
    scale = np.maximum(1.0, np.abs(x0 @ tq.T).max(axis=1, keepdims=True))
    free = ~(au | al)
    assert (np.abs(np.where(free, G, 0)) <= stat_tol * scale).all()            # stationarity on the free set
    assert (np.where(au, -G, 1) > 0).all() and (np.where(al, G, 1) > 0).all()  # multiplier signs


def test_cdu_size_kkt_symmetry_permutation():
    B = 96
    pl, P, tq, nu, x0, lb, ub, qp = _setup("cdu", B, 21, 2.0, max_batch=128)
    N = pl["N"]
    x0[0] = 0.0                                            # trivial problem: u* = 0, nothing active
    lb[0], ub[0] = -1.0, 1.0
    qp.stats(reset=True)
    out = qp.solve_batch(x0, lb, ub)
    assert (out["status"] == 0).all()
    st = qp.stats()
    # default method: the shared-inverse active-set pass must be what solved them (no silent fallback)
    assert st["asm_solved"] == B and st["factorizations"] == 0
    assert st["asm_e2max"] < 1e-9 and st["asm_e1max"] < 1e-9 * max(1.0, np.abs(tq).max())
    _kkt(P, tq, nu, N, x0, lb, ub, out, 1e-7)
    assert np.abs(out["u"][0]).max() < 1e-12 and not out["active"][0].any()
    assert out["active"].any(axis=1).sum() > B // 2          # the batch does exercise the bounds
    # odd symmetry: (x0, lb, ub) -> (-x0, -ub, -lb)  =>  u -> -u, upper/lower rows swap
    o2 = qp.solve_batch(-x0, -ub, -lb)
    assert np.abs(o2["u"] + out["u"]).max() < 1e-8
    n = P.shape[0]
    k, c = np.arange(n) // nu, np.arange(n) % nu
    assert np.array_equal(o2["active"][:, k * 2 * nu + c], out["active"][:, k * 2 * nu + nu + c])
    # permutation invariance and ragged sizes (continuous batching must not mix problems up)
    perm = np.random.default_rng(0).permutation(B)[:77]
    o3 = qp.solve_batch(x0[perm], lb[perm], ub[perm])
    assert np.abs(o3["u"] - out["u"][perm]).max() < 1e-9
    assert np.array_equal(o3["active"], out["active"][perm])
    assert qp.solve_batch(x0[:0], lb[:0], ub[:0])["u"].shape == (0, n)


def test_cstrs_config_10k_batch_kkt():
    """configs[1]: 10k sampled x0 at the CSTRs size (n = 540); more problems than resident slots."""
    B = 10000
    pl, P, tq, nu, x0, lb, ub, qp = _setup("cstrs", B, 22, 2.0, max_batch=4096)
    out = qp.solve_batch(x0, lb, ub)
    hist = np.bincount(out["status"], minlength=3)
    assert hist[2] == 0 and hist[0] >= 0.999 * B, hist      # this synthetic plant has cond(P) = 4e7
    ok = out["status"] == 0
    sub = {k: v[ok] for k, v in out.items()}
    _kkt(P, tq, nu, pl["N"], x0[ok], lb[ok], ub[ok], sub, 1e-7)


def test_cdu_size_pdip_path_and_fallback_agree_with_active_set_pass():
    """The two device paths are independent implementations (f32 MFMA Cholesky PDIP + f64 polish vs
    f64 shared-inverse active set): same optimum, same active sets.  With asm_max_active tiny the
    auto method must hand everything to the PDIP path and still return certified results."""
    B = 48
    pl, P, tq, nu, x0, lb, ub, qp = _setup("cdu", B, 23, 2.0, max_batch=128)
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    fast = qp.solve_batch(x0, lb, ub)
    qp2 = BatchedBoxQP(P, tq, nu, max_batch=128, method="pdip")
    slow = qp2.solve_batch(x0, lb, ub)
    assert (fast["status"] == 0).all() and (slow["status"] == 0).all()
    assert (slow["factorizations"] >= 1).all() and (fast["factorizations"] == 0).all()
    assert np.abs(fast["u"] - slow["u"]).max() < 1e-7
    assert np.array_equal(fast["active"], slow["active"])
    qp3 = BatchedBoxQP(P, tq, nu, max_batch=128, method="auto", asm_max_active=16)
    qp3.stats(reset=True)
    fb = qp3.solve_batch(x0, lb, ub)
    st = qp3.stats()
    assert (fb["status"] == 0).all() and st["asm_solved"] < B and st["factorizations"] > 0
    assert np.abs(fb["u"] - fast["u"]).max() < 1e-7 and np.array_equal(fb["active"], fast["active"])


@pytest.mark.parametrize("method", ["auto", "pdip"])
def test_batches_larger_than_a_segment(method):
    """seg_max caps the problems per lock-step pass: a batch of several segments (with a ragged last one) must give
    exactly what one big segment gives, problem by problem."""
    B = 1000
    pl, P, tq, nu, x0, lb, ub, qp1 = _setup("mini_cdu", B, 33, 2.5, max_batch=128, method=method)
    ref = qp1.solve_batch(x0, lb, ub)
    assert (ref["status"] == 0).all()
    from industrial_nnmpc_2021_amd.qp import BatchedBoxQP
    qp2 = BatchedBoxQP(P, tq, nu, max_batch=128, method=method, seg_max=384)
    out = qp2.solve_batch(x0, lb, ub)
    assert (out["status"] == 0).all()
    assert np.array_equal(out["active"], ref["active"])
    assert np.abs(out["u"] - ref["u"]).max() <= 1e-9
    _kkt(P, tq, nu, pl["N"], x0, lb, ub, out, 1e-7)
