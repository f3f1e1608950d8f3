"""One-time host setup of the condensed regulator matrices for a plant dict
(augmentation -> DARE -> re-parameterisation test -> condensing), following
DenseQPRegulator.__init__ of the reference (lib/linearMPC.py:339-395) but
through the structured recursion of ``condense.py``."""
import numpy as np
import scipy.linalg

from . import condense


def dlqr(A, B, Q, R, M=None):
    """Discrete LQR for stage cost x'Qx + 2x'Mu + u'Ru (reference lib/linearMPC.py:22-40)."""
    if M is not None:
        RinvMT = scipy.linalg.solve(R, M.T)
        Atilde = A - B @ RinvMT
        Qtilde = Q - M @ RinvMT
    else:
        Atilde, Qtilde, M = A, Q, np.zeros(B.shape)
    Pi = scipy.linalg.solve_discrete_are(Atilde, B, Qtilde, R)
    K = -scipy.linalg.solve(B.T @ Pi @ B + R, B.T @ Pi @ A + M.T)
    return (K, Pi)


def augmented_matrices_for_regulator(A, B, Q, R, S):
    """Rate-of-change augmentation (reference lib/linearMPC.py:626-644)."""
    Nx, Nu = B.shape
    Aaug = np.block([[A, np.zeros((Nx, Nu))], [np.zeros((Nu, Nx + Nu))]])
    Baug = np.concatenate((B, np.eye(Nu)), axis=0)
    Qaug = scipy.linalg.block_diag(Q, S)
    Raug = R + S
    Maug = np.concatenate((np.zeros((Nx, Nu)), -S), axis=0)
    return (Aaug, Baug, Qaug, Raug, Maug)


def build_regulator_matrices(pl):
    """(P, tq, nu) of the box-constrained (stable-plant) regulator QP."""
    Aa, Ba, Qa, Ra, Ma = augmented_matrices_for_regulator(pl["A"], pl["B"], pl["Q"], pl["R"], pl["S"])
    _, Pf = dlqr(Aa, Ba, Qa, Ra, Ma)
    if np.any(np.abs(np.linalg.eigvals(Aa)) >= 1.0):
        raise NotImplementedError("unstable plant: re-parameterised dense-G path")
    P, tq = condense.condense(Aa, Ba, Qa, Ra, Ma, Pf, pl["N"])
    return P, tq, Ba.shape[1]
