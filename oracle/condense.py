"""fp64 restatement of the reference's regulator condensing (TEST ORACLE).

Follows lib/linearMPC.py of the reference:
  dlqr                         :22-40
  regulator augmentation       :626-644  (get_augmented_matrices_for_regulator)
  DenseQPRegulator.__init__    :339-364
  _reparameterize              :366-382
  _get_tA_tB                   :397-428
  _get_tQ_tR_tM_tE_tK          :430-465
  _get_P_tq / _getG / _get_h   :467-493
  solve                        :495-512
  get_control_sequence         :682-689
The dense stacked matrices are built explicitly, exactly like the reference
does, so this is only meant for small sizes (tests); the product code builds
P/tq by recursion instead and is checked against this.
"""
import numpy as np
import scipy.linalg


def dlqr(A, B, Q, R, M=None):
    """Discrete LQR with cross term, stage cost x'Qx + 2x'Mu + u'Ru.

    Reference: lib/linearMPC.py:22-40.
    """
    if M is None:
        At, Qt, M = A, Q, np.zeros(B.shape)
    else:
        RiMt = scipy.linalg.solve(R, M.T)
        At = A - B @ RiMt
        Qt = Q - M @ RiMt
    Pi = scipy.linalg.solve_discrete_are(At, B, Qt, R)
    K = -scipy.linalg.solve(B.T @ Pi @ B + R, B.T @ Pi @ A + M.T)
    return K, Pi


def augment_for_regulator(A, B, Q, R, S):
    """Rate-of-change augmentation.  Reference: lib/linearMPC.py:626-644."""
    nx, nu = B.shape
    Aaug = np.zeros((nx + nu, nx + nu))
    Aaug[:nx, :nx] = A
    Baug = np.vstack((B, np.eye(nu)))
    Qaug = scipy.linalg.block_diag(Q, S)
    Raug = R + S
    Maug = np.vstack((np.zeros((nx, nu)), -S))
    return Aaug, Baug, Qaug, Raug, Maug


class DenseRegulator:
    """Condensed regulator QP data (P, tq, G, h(x0)).

    Reference: DenseQPRegulator, lib/linearMPC.py:321-517.  ``qp`` is any
    callable ``qp(P, q, G, h) -> x`` (the cvxopt seam, :503-504).
    """

    def __init__(self, A, B, Q, R, M, N, ulb, uub):
        self.N, self.ulb, self.uub = N, ulb, uub
        self.nx, self.nu = B.shape
        self.Krep, self.Pf = dlqr(A, B, Q, R, M)
        # :366-382 -- re-parameterise u = K x + v when A is not stable.
        self.reparameterize = bool(np.any(np.abs(np.linalg.eigvals(A)) >= 1.0))
        if self.reparameterize:
            K = self.Krep
            Q = Q + K.T @ R @ K + M @ K + K.T @ M.T
            M = K.T @ R + M
            A = A + B @ K
        self.A, self.B, self.Q, self.R, self.M = A, B, Q, R, M
        self._stack()

    def _stack(self):
        N, nx, nu = self.N, self.nx, self.nu
        A, B = self.A, self.B
        pw = [np.eye(nx)]
        for _ in range(N):
            pw.append(A @ pw[-1])
        self.tA = np.vstack(pw)                                   # :416-417
        tB = np.zeros(((N + 1) * nx, N * nu))                     # :418-428
        for i in range(1, N + 1):
            for j in range(i):
                tB[i * nx:(i + 1) * nx, j * nu:(j + 1) * nu] = pw[i - j - 1] @ B
        self.tB = tB
        tQ = scipy.linalg.block_diag(*([self.Q] * N + [self.Pf]))  # :454-455
        tR = scipy.linalg.block_diag(*([self.R] * N))
        tM = np.vstack((scipy.linalg.block_diag(*([self.M] * N)),
                        np.zeros((nx, N * nu))))
        E = np.vstack((np.eye(nu), -np.eye(nu)))
        self.tE = scipy.linalg.block_diag(*([E] * N))
        self.P = tB.T @ (tQ @ tB) + tR + tB.T @ tM + tM.T @ tB      # :472
        self.tq = (tB.T @ tQ + tM.T) @ self.tA                     # :473
        if self.reparameterize:
            self.tK = scipy.linalg.block_diag(*([self.Krep] * N))
            self.G = self.tE @ (self.tK @ tB[:N * nx]) + self.tE   # :479
        else:
            self.tK = None
            self.G = self.tE                                       # :481

    def h(self, x0):
        """:484-493"""
        te = np.tile(np.vstack((self.uub, -self.ulb)), (self.N, 1))
        if self.reparameterize:
            n = self.N * self.nx
            return te - self.tE @ (self.tK @ (self.tA[:n] @ x0))
        return te

    def solve(self, x0, qp):
        """:495-512"""
        v = np.asarray(qp(self.P, self.tq @ x0, self.G, self.h(x0))).reshape(-1, 1)
        if self.reparameterize:
            n = self.N * self.nx
            v = self.tK @ (self.tA[:n] @ x0 + self.tB[:n] @ v) + v
        return v


def setup_regulator(A, B, Q, R, S, N, ulb, uub):
    """Reference: LinearMPCController.setup_regulator, lib/linearMPC.py:596-604."""
    Aa, Ba, Qa, Ra, Ma = augment_for_regulator(A, B, Q, R, S)
    return DenseRegulator(Aa, Ba, Qa, Ra, Ma, N, ulb, uub)


def control_sequence(reg, x, uprev, xs, us, ulb, uub, qp):
    """Reference: LinearMPCController.get_control_sequence, :682-689."""
    reg.ulb = ulb - us
    reg.uub = uub - us
    x0 = np.vstack((x - xs, uprev - us))
    return reg.solve(x0, qp) + np.tile(us, (reg.N, 1))
