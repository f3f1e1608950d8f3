"""Condensing of the regulator problem into (P, tq[, M]) without the dense stacks.

Same matrices as the reference's DenseQPRegulator._setup_fixed_matrices
(lib/linearMPC.py:384-482):

    P  = tB' tQ tB + tR + tB' tM + tM' tB            (:472)
    tq = (tB' tQ + tM') tA                           (:473)
    G  = tE (tK tB[:N nx] + I)  when re-parameterised (:479)

but the reference materialises tB ((N+1)nx x N nu), tQ (((N+1)nx)^2 -- 12.8 GB
at the CDU size) and calls matrix_power O(N^2) times.  Here the block-Toeplitz
structure tB[i, j] = A^(i-j-1) B =: G_(i-j) is used instead:

    P[j, l] = sum_{t=1}^{a-1} G_t' Q G_(t+d) + G_a' Pf G_(a+d) + (M-terms) + [j==l] R,
              a = N - j,  d = j - l >= 0,

i.e. cumulative sums along the block diagonals of Gs' Q Gs plus one block of
Gs' Pf Gs (two (N nu) x nx x (N nu) GEMMs), and tq by the backward recursion
Y_j = Q A^j + A' Y_(j+1),  tq[j] = B' Y_(j+1) + M' A^j.   fp64, host, one-time.
"""
import numpy as np


def _gs(A, B, N):
    nx, nu = B.shape
    Gs = np.empty((N + 1, nx, nu))
    Gs[0] = 0.0
    Gs[1] = B
    for t in range(2, N + 1):
        Gs[t] = A @ Gs[t - 1]
    return Gs


def condense(A, B, Q, R, M, Pf, N):
    """Return (P (N nu, N nu), tq (N nu, nx)) for stage cost 1/2(x'Qx + u'Ru + 2x'Mu),
    terminal cost 1/2 x'Pf x and dynamics x+ = Ax + Bu."""
    nx, nu = B.shape
    n = N * nu
    Gs = _gs(A, B, N)                                   # Gs[t] = A^(t-1) B
    Gm = np.concatenate(list(Gs[1:]), axis=1)           # nx x (N nu), block t-1 = G_t
    X = Gm.T @ (Q @ Gm)                                 # X[t-1, s-1] = G_t' Q G_s
    Z = Gm.T @ (Pf @ Gm)
    X4 = X.reshape(N, nu, N, nu)
    Z4 = Z.reshape(N, nu, N, nu)
    P = np.zeros((N, nu, N, nu))
    for d in range(N):                                  # block diagonal d = j - l
        L = N - d                                       # a runs 1..L
        idx = np.arange(L)
        xd = X4[idx, :, idx + d, :]                     # (L, nu, nu): G_t' Q G_(t+d), t = 1..L
        zd = Z4[idx, :, idx + d, :]
        cum = np.cumsum(xd, axis=0) - xd                # sum_{t<a}
        S = cum + zd                                    # S(a, a+d), a = 1..L
        # (j, l) = (N - a, N - a - d)
        a = idx + 1
        j = N - a
        l = j - d
        if d == 0:
            blk = S + R
        else:
            blk = S + (M.T @ Gs[d])                     # j > l : M' G_(j-l)
        P[j, :, l, :] = blk
        if d:
            P[l, :, j, :] = np.transpose(blk, (0, 2, 1))
    P = P.reshape(n, n)
    # tq
    tq = np.empty((N, nu, nx))
    Apow = [np.eye(nx)]
    for _ in range(N):
        Apow.append(A @ Apow[-1])
    Y = Pf @ Apow[N]
    for j in range(N - 1, -1, -1):
        tq[j] = B.T @ Y + M.T @ Apow[j]
        Y = Q @ Apow[j] + A.T @ Y
    return P, tq.reshape(n, nx)


def constraint_map(A, B, K, N):
    """M_G = I + tK tB[:N nx]  (block lower triangular Toeplitz, blocks K G_(i-j))
    and tK tA[:N nx] ((N nu) x nx) of the re-parameterised problem (:479, :490)."""
    nx, nu = B.shape
    Gs = _gs(A, B, N)
    KG = np.stack([K @ Gs[t] for t in range(N)])        # KG[t] = K G_t (KG[0] = 0)
    Mg = np.zeros((N, nu, N, nu))
    for i in range(N):
        for j in range(i):
            Mg[i, :, j, :] = KG[i - j]
    Mg = Mg.reshape(N * nu, N * nu) + np.eye(N * nu)
    KA = np.empty((N, nu, nx))
    Ap = np.eye(nx)
    for i in range(N):
        KA[i] = K @ Ap
        Ap = A @ Ap
    return Mg, KA.reshape(N * nu, nx)
