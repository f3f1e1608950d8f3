"""Batched steady-state target problems on the GPU (host wrapper over nnmpc_ts_*).

The reference solves one target QP per simulation step (TargetSelector.solve -> cvxopt.solvers.qp(P, q, G, h, A, b),
lib/linearMPC.py:298-311, called at :851 and :654) in z = (xs, us), n = Nx + Nu unknowns, Nx + Nz equalities

    [I - A, -B; H C, 0] z = tb [ysp; dhat]            (:262-266)

and the input box (:253-260).  All of that but q and b is shared by every call, and the setpoint / disturbance
signals of the offline simulation are piecewise constant (sample_prbs_like), so

* the equalities are eliminated ONCE on the host (fp64): with F = [I - A; H C] of full column rank, xs = Xb b + Xu us and
  Nz equalities E us = Eb b remain on the inputs -- each problem is a QP in Nu unknowns (see include/nnmpc.h);
* only the DISTINCT (ysp, dhat) rows of a batch are solved (one wave per problem, nnmpc_ts_solve_batch) and the results
  are broadcast back.
"""
import ctypes as C

import numpy as np

from . import _lib


class ReducedTargetProblem:
    """Shared matrices of the reduced problem; raises ValueError when F = [I - A; H C] is rank deficient (the
    caller then keeps the full-space formulation)."""

    def __init__(self, A, B, C_, H, Bd, Cd, Qs, Rs, usp):
        Nx, Nu = B.shape
        Ny, Nz, Nd = C_.shape[0], H.shape[0], Bd.shape[1]
        F = np.vstack((np.eye(Nx) - A, H @ C_))                        # (Nx + Nz) x Nx
        U, sv, Vt = np.linalg.svd(F, full_matrices=True)
        if sv.min() <= 1e-10 * sv.max():
            raise ValueError("target selector: [I - A; H C] is rank deficient (integrating modes invisible to H C)")
        Fp = (Vt.T / sv) @ U[:, :Nx].T                                 # left inverse (pseudo-inverse) of F
        Nl = U[:, Nx:]                                                  # basis of the left null space, (Nx + Nz) x Nz
        Bt = np.vstack((B, np.zeros((Nz, Nu))))
        self.tb = np.block([[np.zeros((Nx, Ny)), Bd], [H, -(H @ Cd)]])  # b = tb [ysp; dhat]     (:264-266)
        self.Xb, self.Xu = Fp, Fp @ Bt                                  # xs = Xb b + Xu us
        self.E, self.Eb = Nl.T @ Bt, -Nl.T                              # E us = Eb b
        CQC = C_.T @ (Qs @ C_)
        Pr = Rs + self.Xu.T @ CQC @ self.Xu
        self.Pr = 0.5 * (Pr + Pr.T)
        self.Qb = self.Xu.T @ CQC @ self.Xb                             # q = Qb b + Qy y + q0,  y = ysp - Cd dhat
        self.Qy = -self.Xu.T @ (C_.T @ Qs)
        self.q0 = -(Rs @ usp).ravel()
        self.Cd = Cd
        self.Nx, self.Nu, self.Ny, self.Nz, self.Nd = Nx, Nu, Ny, Nz, Nd

    def reduce(self, Ysp, Dhat):
        """rows (ysp, dhat) -> (q (M, Nu), e (M, Nz), b (M, Nx + Nz))."""
        b = np.concatenate((Ysp, Dhat), axis=1) @ self.tb.T
        y = Ysp - Dhat @ self.Cd.T
        return b @ self.Qb.T + y @ self.Qy.T + self.q0, b @ self.Eb.T, b

    def expand(self, b, Us):
        return b @ self.Xb.T + Us @ self.Xu.T


def unique_rows_piecewise(key):
    """(distinct rows, index of every row's distinct row) like np.unique(key, axis=0, return_inverse=True), for signals that
    are piecewise constant along the rows (sample_prbs_like: one change per ~200 steps): the runs are found by ONE comparison
    pass and only their first rows are sorted -- np.unique on all 357 600 x 95 rows of a CDU task list took 7.5 of the 10.5 s
    the whole data set needs."""
    M = key.shape[0]
    if M == 0:
        return key, np.zeros(0, np.intp)
    first = np.ones(M, bool)
    np.any(key[1:] != key[:-1], axis=1, out=first[1:])
    run = np.cumsum(first) - 1
    uniq, inv = np.unique(key[first], axis=0, return_inverse=True)
    return uniq, np.ravel(inv)[run]


class BatchedTargetSelector:
    """Distinct (ysp, dhat) pairs of a batch, solved on the GPU; results broadcast to the rows they came from."""

    def __init__(self, A, B, C_, H, Bd, Cd, Qs, Rs, usp, ulb, uub):
        self.red = ReducedTargetProblem(A, B, C_, H, Bd, Cd, Qs, Rs, usp)
        lib = _lib.load()
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        self._keep = (f(self.red.Pr), f(self.red.E), f(ulb).ravel(), f(uub).ravel())
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self._h = C.c_void_p()
        _lib.check(lib.nnmpc_ts_create(C.byref(self._h), self.red.Nu, self.red.Nz, *[p(a) for a in self._keep]),
                   "nnmpc_ts_create")
        self._lib = lib
        self.last_distinct = 0

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nnmpc_ts_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def solve_reduced(self, q, e):
        """q (M, Nu), e (M, Nz) -> (us (M, Nu), lam_eq (M, Nz), active (M, Nu) uint8, status (M,))."""
        q = np.ascontiguousarray(q, dtype=np.float64)
        e = np.ascontiguousarray(e, dtype=np.float64)
        M = q.shape[0]
        us, lam = np.empty((M, self.red.Nu)), np.empty((M, self.red.Nz))
        act, st = np.empty((M, self.red.Nu), np.uint8), np.empty(M, np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.nnmpc_ts_solve_batch(self._h, M, p(q), p(e), p(us), p(lam), p(act), p(st), _lib.HOST),
                   "nnmpc_ts_solve_batch")
        return us, lam, act, st

    def solve_batch(self, Ysp, Dhat):
        """Ysp (M, Ny), Dhat (M, Nd) -> (Xs (M, Nx), Us (M, Nu), status (M,)); duplicates are solved once."""
        Ysp = np.asarray(Ysp, float).reshape(-1, self.red.Ny)
        Dhat = np.asarray(Dhat, float).reshape(-1, self.red.Nd)
        key = np.ascontiguousarray(np.concatenate((Ysp, Dhat), axis=1))
        uniq, inv = unique_rows_piecewise(key)
        self.last_distinct = uniq.shape[0]
        q, e, b = self.red.reduce(uniq[:, :self.red.Ny], uniq[:, self.red.Ny:])
        us, _, _, st = self.solve_reduced(q, e)
        xs = self.red.expand(b, us)
        return xs[inv], us[inv], st[inv]
