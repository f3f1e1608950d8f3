"""Host side of the far-field factors: the staircase rotation of qp._staircase (no GPU).

M = [Kunc[W:] | -Pinv[W:, :W]] of an MPC regulator has numerical rank ~Nx and its suffixes M[128 j:] lose rank as j grows (the closed
loop forgets its fast modes first).  _staircase rotates the basis of the truncated SVD so that a 128-row block uses only the leading
k_j coordinates; the library's GEMM stops there.  Checked here: the rotated factors still reproduce M to the tolerance, G is
orthonormal, the zero pattern is a staircase in whole K chunks of 16, and it is markedly shorter than the rank."""
import numpy as np
import scipy.linalg as sla

from industrial_nnmpc_2021_amd import synthetic
from industrial_nnmpc_2021_amd.linearMPC_build import build_regulator_matrices
from industrial_nnmpc_2021_amd.qp import _staircase


def _far_block(name, W):
    pl = synthetic.plant(name, seed=0)
    P, tq, nu = build_regulator_matrices(pl)
    c = sla.cho_factor(P)
    n = P.shape[0]
    Hinv, Kunc = sla.cho_solve(c, np.eye(n)), -sla.cho_solve(c, tq)
    return np.hstack((Kunc[W:], -Hinv[W:, :W])), pl


def test_staircase_factors_of_the_mid_size_regulator():
    W = 256
    M, pl = _far_block("mid_cdu", W)                                  # n = 1024: (768, n_aug + 256)
    U, s, Vt = sla.svd(M, full_matrices=False)
    tol = 1e-11 * s[0]                                                # (this plant's inverse carries rounding noise at 3e-13 of sigma_1)
    r = int((s > tol).sum())
    assert r == pl["A"].shape[0]                                      # Nx: beyond the window everything follows the terminal-cost LQ recursion
    U1, G = _staircase(U[:, :r] * s[:r], tol)
    assert np.abs(G @ G.T - np.eye(G.shape[0])).max() < 1e-12
    V1 = G @ Vt[:r]
    assert np.abs(U1 @ V1 - M).max() < 50 * tol                       # what the device check of nnmpc_qp_set_farfield bounds by 1e-9
    nb = M.shape[0] // 128
    k = []
    for j in range(nb):
        blk = U1[128 * j:128 * (j + 1)]
        nz = np.flatnonzero(np.abs(blk).max(axis=0) > 0)
        kj = int(nz.max()) + 1 if nz.size else 0
        assert (blk[:, kj:] == 0).all()
        k.append(kj)
    assert all(a >= b for a, b in zip(k, k[1:]))                      # a staircase: later column tiles use fewer coordinates
    assert all(kj % 16 == 0 or kj == G.shape[0] for kj in k)          # whole K chunks of the GEMM
    assert k[-1] < k[0] and sum(k) < 0.8 * nb * G.shape[0]            # ... and markedly fewer than the rank on average


def test_staircase_of_a_generic_matrix_keeps_everything():
    """No decay along the rows: nothing can be dropped, the factors come back as a rotation of themselves."""
    rng = np.random.default_rng(0)
    U0 = rng.standard_normal((384, 40))
    U1, G = _staircase(U0, 1e-12)
    assert G.shape == (40, 40) and np.abs(G @ G.T - np.eye(40)).max() < 1e-12
    assert np.abs(U1 @ G - U0).max() < 1e-11
    assert (np.abs(U1) > 0).any(axis=0).all()


def test_staircase_of_an_all_zero_tail():
    rng = np.random.default_rng(1)
    U0 = np.zeros((512, 24))
    U0[:128] = rng.standard_normal((128, 24))
    U1, G = _staircase(U0, 1e-12)
    assert np.abs(U1 @ G - U0).max() < 1e-11
    assert (U1[128:] == 0).all()
