"""The numerical claim behind the corrected f32 solves of the rounds (csrc/qp_asm.h: asm_reg_core<..., REF>),
emulated in numpy on CPU: on the plants of the bench the active-set systems S = (P^-1)_AA are well conditioned (cond(S) of the order of
ten), so ONE fp64 correction of a solve through an f32 Cholesky factor -- r = b - S lam in fp64, L L' dl = r in f32, lam + dl in fp64 --
leaves a residual at the level the library's acceptance test asks for (1e-12 max |b|, checked there against the window GEMM's row),
and a second one reaches the fp64 floor.  No GPU, no library call: the test pins the assumption, the GPU tests pin the kernels."""
import numpy as np
import pytest


def _sets(name, nsamp, sx):
    from industrial_nnmpc_2021_amd import synthetic
    from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
    pl = synthetic.plant(name, seed=0)
    P, tq, nu = build_regulator_matrices(pl)
    Ps = np.tril(P) + np.tril(P, -1).T
    H = np.linalg.inv(Ps)
    s = synthetic.samples(pl, nsamp, seed=3, sx=sx)
    x0 = np.concatenate((s["x"] - s["xs"], s["uprev"] - s["us"]), axis=1)
    lb, ub = pl["ulb"].T - s["us"], pl["uub"].T - s["us"]
    N = P.shape[0] // nu
    xunc = -(H @ (tq @ x0.T)).T
    for b in range(nsamp):
        lbf, ubf = np.tile(lb[b], N), np.tile(ub[b], N)
        A = np.where((xunc[b] > ubf) | (xunc[b] < lbf))[0]          # the first set of the rounds: the bounds x_unc violates
        if len(A) < 4:
            continue
        rhs = xunc[b][A] - np.where(xunc[b][A] > ubf[A], ubf[A], lbf[A])
        yield H[np.ix_(A, A)], rhs


def _solve_f32(L32, r):
    y = np.linalg.solve(L32.astype(np.float64), r).astype(np.float32)
    return np.linalg.solve(L32.T.astype(np.float64), y.astype(np.float64)).astype(np.float32).astype(np.float64)


@pytest.mark.parametrize("name,sx", [("mini_cdu", 2.0), ("mini_cdu", 4.0), ("mini_cstrs", 1.0)])
def test_one_fp64_correction_of_an_f32_solve_reaches_the_acceptance_level(name, sx):
    seen = 0
    for S, b in _sets(name, 12, sx):
        if np.linalg.cond(S) > 1e4:          # (the library would send such a set to the fp64 kernel: its test fails)
            continue
        seen += 1
        L32 = np.linalg.cholesky(S.astype(np.float32).astype(np.float64)).astype(np.float32)   # factor of the f32-rounded system
        lam = _solve_f32(L32, b.astype(np.float32).astype(np.float64))
        exact = np.linalg.solve(S, b)
        scale = np.abs(b).max()
        r0 = np.abs(b - S @ lam).max() / scale
        lam = lam + _solve_f32(L32, b - S @ lam)
        r1 = np.abs(b - S @ lam).max() / scale
        assert r0 < 1e-4 and r1 <= 1e-12, (r0, r1, np.linalg.cond(S))
        assert np.abs(lam - exact).max() <= 1e-11 * np.abs(exact).max()
        lam = lam + _solve_f32(L32, b - S @ lam)
        assert np.abs(b - S @ lam).max() / scale <= 1e-14
    assert seen >= 3
