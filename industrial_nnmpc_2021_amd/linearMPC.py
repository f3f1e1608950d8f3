"""Host-side mirror of the reference's ``lib/linearMPC.py`` interface.

Same class / function names, keyword-only constructors, argument meaning and
column-vector ``(k, 1)`` conventions as the reference, so the module drops in
under ``lib/controller_evaluation.py`` (which does ``from linearMPC import
DenseQPRegulator, online_simulation, LinearMPCController, dlqr``).  What
changes is where the arithmetic of the hot path runs:

* ``DenseQPRegulator.solve`` (reference :495-512 -> cvxopt.solvers.qp) hands
  the condensed QP to the HIP library (``qp.BatchedBoxQP``); the additive
  ``solve_batch`` entry point solves B independent (x0, ulb, uub) per call.
* ``OfflineSimulator.generate_data`` / ``simulate_offline`` (:803-880) advance
  all chains of a task in lock-step on the device (``chain.DeviceChains``): chain
  state, target pairs and recorded trajectories stay in HBM, one batched regulator
  solve per step, instead of one OS process per chain.
* ``TargetSelector.solve`` (:298-311 -> cvxopt.solvers.qp with equalities): the
  distinct (ysp, dhat) pairs of a task are solved once, batched, on the GPU
  (``target.BatchedTargetSelector``).
* the condensed matrices are built by recursion (``condense.py``) instead of
  the dense block-diagonal stacks (:397-474).

Kalman filter and plant simulator stay on the host in fp64: they are not on the
hot path.
"""
import sys
import time
import warnings

import numpy as np
import scipy.linalg

from . import condense
from .linearMPC_build import dlqr, augmented_matrices_for_regulator
from .host_qp import solve_small_qp

__all__ = ["dlqr", "dlqe", "c2d", "assert_detectable", "assert_stabilizable",
           "LinearPlantSimulator", "KalmanFilter", "TargetSelector", "DenseQPRegulator",
           "LinearMPCController", "online_simulation", "OfflineSimulator", "simulate_offline"]


def dlqe(A, C, Q, R):
    """Discrete-time Kalman filter gain (reference :42-48)."""
    P = scipy.linalg.solve_discrete_are(A.T, C.T, Q, R)
    L = scipy.linalg.solve(C @ P @ C.T + R, C @ P).T
    return (L, P)


def c2d(A, B, sample_time):
    """Zero-order-hold discretisation through one matrix exponential (reference :50-64)."""
    Nx, Nu = B.shape
    M = np.zeros((Nx + Nu, Nx + Nu))
    M[:Nx, :Nx], M[:Nx, Nx:] = A, B
    E = scipy.linalg.expm(M * sample_time)
    return (E[:Nx, :Nx], E[:Nx, Nx:])


def _unstable_modes_visible(X, Y):
    w, V = np.linalg.eig(X)
    for v in V[:, np.abs(w) >= 1.0].T:
        if np.linalg.norm(Y @ v) <= 1e-8:
            return False
    return True


def assert_detectable(A, C):
    """(reference :79-81)"""
    assert _unstable_modes_visible(A, C)


def assert_stabilizable(A, B):
    """(reference :83-85)"""
    assert _unstable_modes_visible(A.T, B.T)


class LinearPlantSimulator:
    """Linear plant with measurement noise (reference :87-131)."""

    def __init__(self, *, A, B, C, Bp, Rv, sample_time, x0):
        self.A, self.B, self.C, self.Bp = A, B, C, Bp
        self.Nx, self.Nu, self.Ny = A.shape[0], B.shape[1], C.shape[0]
        self.measurement_noise_std = np.sqrt(np.diag(Rv)[:, np.newaxis])
        self.sample_time = sample_time
        self.x, self.u, self.p = [x0], [], []
        self.v = [self.measurement_noise_std * np.random.randn(self.Ny, 1)]
        self.y = [self.C @ x0 + self.v[-1]]
        self.t = [0.]

    def step(self, u, p):
        x = self.A @ self.x[-1] + self.B @ u + self.Bp @ p
        v = self.measurement_noise_std * np.random.randn(self.Ny, 1)
        y = self.C @ x + v
        self._append_data(x, u, p, v, y)
        return y

    def _append_data(self, x, u, p, v, y):
        self.x.append(x); self.u.append(u); self.p.append(p); self.v.append(v); self.y.append(y)
        self.t.append(self.t[-1] + self.sample_time)


class KalmanFilter:
    """Steady-state Kalman filter (reference :133-176)."""

    def __init__(self, *, A, B, C, Qw, Rv, xprior):
        self.A, self.B, self.C, self.Qw, self.Rv = A, B, C, Qw, Rv
        (self.L, _) = dlqe(A, C, Qw, Rv)
        self.xhat, self.xhat_pred, self.y, self.uprev = [xprior], [], [], []

    def solve(self, y, uprev):
        xhat_pred = self.A @ self.xhat[-1] + self.B @ uprev
        xhat = xhat_pred + self.L @ (y - self.C @ xhat_pred)
        self.xhat.append(xhat); self.xhat_pred.append(xhat_pred); self.y.append(y); self.uprev.append(uprev)
        return xhat


class TargetSelector:
    """Steady-state target QP (reference :178-319): same (P, q, G, h, A, b) as the reference builds.

    ``backend="hip"`` (default): equalities eliminated once on the host, the distinct (ysp, dhat) pairs of a
    batch solved on the GPU (``target.BatchedTargetSelector``); needs [I - A; H C] of full column rank and the
    input box only.  ``backend="host"``: the full-space problem in the small fp64 host solver (``host_qp``), one
    at a time -- explicit opt-in (output constraints ylb/yub, rank-deficient plants), never a silent fallback.
    """

    def __init__(self, *, A, B, C, H, Bd, Cd, usp, Rs, Qs, ulb, uub, ylb=None, yub=None, backend="hip"):
        self.A, self.B, self.C, self.H, self.Bd, self.Cd, self.Rs, self.Qs = A, B, C, H, Bd, Cd, Rs, Qs
        self.Nx, self.Nu, self.Ny = A.shape[0], B.shape[1], C.shape[0]
        self.Nd, self.Nz = Bd.shape[1], H.shape[0]
        self.usp = usp
        self.ysp, self.dhats, self.xs, self.us = [], [], [], []
        self.ulb, self.uub, self.ylb, self.yub = ulb, uub, ylb, yub
        if backend not in ("hip", "host"):
            raise ValueError("backend must be 'hip' or 'host'")
        if backend == "hip" and (ylb is not None or yub is not None):
            raise NotImplementedError("output constraints (ylb, yub) are only handled by backend='host': pass backend='host'")
        self.backend = backend
        if backend == "hip":
            # the reduction needs [I - A; H C] of full column rank: say so now (host fp64, no GPU involved), not at the first solve
            sv = np.linalg.svd(np.vstack((np.eye(self.Nx) - A, H @ C)), compute_uv=False)
            if sv.min() <= 1e-10 * sv.max():
                raise ValueError("TargetSelector(backend='hip'): [I - A; H C] is rank deficient (integrating modes invisible to H C); "
                                 "pass backend='host' for the full-space problem in the host solver")
        self._setup_fixed_matrices()
        self._cache = {}
        self._batched = None

    def _setup_fixed_matrices(self):
        Nx, Nu, Ny, Nz = self.Nx, self.Nu, self.Ny, self.Nz
        E = np.vstack((np.eye(Nu), -np.eye(Nu)))
        self.F = np.vstack((np.eye(Ny), -np.eye(Ny)))
        if self.ylb is not None and self.yub is not None:
            self.G = np.block([[self.F @ self.C, np.zeros((2 * Ny, Nu))], [np.zeros((2 * Nu, Nx)), E]])
            self.h = None
        else:
            self.G = np.hstack((np.zeros((2 * Nu, Nx)), E))
            self.h = np.vstack((self.uub, -self.ulb))
        self.tA = np.block([[np.eye(Nx) - self.A, -self.B], [self.H @ self.C, np.zeros((Nz, Nu))]])
        self.tb = np.block([[np.zeros((Nx, Ny)), self.Bd], [self.H, -(self.H @ self.Cd)]])
        self.P = scipy.linalg.block_diag(self.C.T @ (self.Qs @ self.C), self.Rs)

    def _setup_changing_matrices(self, ysp, dhats):
        q = np.vstack((-(self.C.T @ (self.Qs @ (ysp - self.Cd @ dhats))), -(self.Rs @ self.usp)))
        if self.h is None:
            h1 = np.vstack((self.yub, -self.ylb)) - self.F @ (self.Cd @ dhats)
            h = np.vstack((h1, self.uub, -self.ulb))
        else:
            h = self.h
        b = self.tb @ np.vstack((ysp, dhats))
        return (q, h, b)

    def _device(self):
        if self._batched is None:
            from .target import BatchedTargetSelector
            self._batched = BatchedTargetSelector(self.A, self.B, self.C, self.H, self.Bd, self.Cd, self.Qs, self.Rs,
                                                  self.usp, self.ulb, self.uub)
        return self._batched

    def solve_batch(self, Ysp, Dhat):
        """Rows (ysp, dhat) -> (Xs (M, Nx), Us (M, Nu)).  Identical rows (piecewise-constant PRBS signals: a few
        thousand distinct pairs in 357 600 CDU steps, cdu_parameters.py:135-143) are solved once."""
        Ysp = np.asarray(Ysp, float).reshape(-1, self.Ny)
        Dhat = np.asarray(Dhat, float).reshape(-1, self.Nd)
        if self.backend == "host":
            key = np.concatenate((Ysp, Dhat), axis=1)
            uniq, inv = np.unique(key, axis=0, return_inverse=True)
            sol = np.empty((uniq.shape[0], self.Nx + self.Nu))
            for i, row in enumerate(uniq):
                (q, h, b) = self._setup_changing_matrices(row[:self.Ny, None], row[self.Ny:, None])
                sol[i] = solve_small_qp(self.P, q, self.G, h, self.tA, b)
            sol = sol[np.ravel(inv)]
            return sol[:, :self.Nx], sol[:, self.Nx:]
        Xs, Us, st = self._device().solve_batch(Ysp, Dhat)
        if (st != 0).any():
            bad = int(np.flatnonzero(st != 0)[0])
            raise ArithmeticError(f"target selector: {int((st != 0).sum())} of {st.size} problems not solved (first: row {bad}, "
                                  f"status {int(st[bad])}): no steady state reaches that setpoint inside the input bounds "
                                  "(the equalities restricted to the free inputs lose rank)")
        return Xs, Us

    def solve(self, ysp, dhats):
        """(xs, us) for one (ysp, dhat), reference signature (:298-311).  Identical consecutive inputs
        (piecewise-constant signals) are answered from a one-entry cache."""
        key = (ysp.tobytes(), dhats.tobytes())
        if key not in self._cache:
            Xs, Us = self.solve_batch(ysp.reshape(1, -1), dhats.reshape(1, -1))
            self._cache = {key: np.concatenate((Xs[0], Us[0]))}
        (xs, us) = np.split(self._cache[key].reshape(-1, 1), [self.Nx])
        self.xs.append(xs); self.us.append(us); self.ysp.append(ysp); self.dhats.append(dhats)
        return (xs, us)


class DenseQPRegulator:
    """Condensed regulator QP (reference :321-517) solved on the GPU.

    min 1/2 sum_k (x'Qx + u'Ru + 2x'Mu) + 1/2 x_N' Pf x_N,  x+ = Ax + Bu,  ulb <= u <= uub.
    """

    def __init__(self, *, A, B, Q, R, M, N, ulb, uub, max_batch=1024, solver_options=None):
        self.A, self.B, self.Q, self.R, self.M, self.N = A, B, Q, R, M, N
        self.ulb, self.uub = ulb, uub
        self.Nx, self.Nu = A.shape[0], B.shape[1]
        (self.Krep, self.Pf) = dlqr(A, B, Q, R, M)
        self._reparameterize()
        (self.P, self.tq) = condense.condense(self.A, self.B, self.Q, self.R, self.M, self.Pf, N)
        self._max_batch = max_batch
        self._opts = dict(solver_options or {})
        # this class solves at most max_batch problems per call (one per call from the reference's own loops): size the
        # active-set workspace for that, not for a quarter of the HBM (qp.BatchedBoxQP's default for bulk batches)
        self._opts.setdefault("seg_max", max(128, int(max_batch)))
        self._qp = None
        self.x0, self.useq = [], []
        self.last_info = None

    def _reparameterize(self):
        """u = Kx + v when A is not stable (reference :366-382)."""
        if np.any(np.abs(np.linalg.eigvals(self.A)) >= 1.0):
            K = self.Krep
            self.A = self.A + self.B @ K
            self.Q = self.Q + K.T @ (self.R @ K) + self.M @ K + K.T @ self.M.T
            self.M = K.T @ self.R + self.M
            self.reparameterize = True
        else:
            self.reparameterize = False

    # ---- dense views of the reference's fixed matrices (built on demand; small problems only)
    @property
    def tE(self):
        E = np.vstack((np.eye(self.Nu), -np.eye(self.Nu)))
        return scipy.linalg.block_diag(*([E] * self.N))

    @property
    def G(self):
        """Inequality matrix of the dense QP (reference :476-482)."""
        if self.reparameterize:
            Mg, _ = condense.constraint_map(self.A, self.B, self.Krep, self.N)
            return self.tE @ Mg
        return self.tE

    def _get_h(self, x0):
        """(reference :484-493)"""
        te = np.tile(np.vstack((self.uub, -self.ulb)), (self.N, 1))
        if self.reparameterize:
            _, KA = condense.constraint_map(self.A, self.B, self.Krep, self.N)
            return te - self.tE @ (KA @ x0)
        return te

    def _box_form(self):
        """(P_box, tq_box) of the box-constrained QP the GPU solves.

        Stable plant: the condensed (P, tq) themselves (G = tE, reference :481).
        Re-parameterised plant (u = Kx + v, :366-382, G = tE (I + tK tB) =: tE Mg, :479):
        the inequality rows are box rows of the *input* sequence w = Mg v + tK tA x0 (which is
        exactly what the reference returns after un-doing the re-parameterisation, :507-509).
        Mg is unit block-lower-triangular, so substituting v = Mg^-1 (w - tK tA x0) gives a box
        QP in w with the shared Hessian Mg^-T P Mg^-1 and a linear term that is again linear in
        x0 -- same optimum, same active rows, no dense-G contraction per sample.
        """
        if not self.reparameterize:
            return self.P, self.tq
        Mg, KA = condense.constraint_map(self.A, self.B, self.Krep, self.N)
        Y = scipy.linalg.solve_triangular(Mg, np.eye(Mg.shape[0]), lower=True, unit_diagonal=True)  # Mg^-1
        Pw = Y.T @ self.P @ Y
        Pw = 0.5 * (Pw + Pw.T)
        return Pw, Y.T @ self.tq - Pw @ KA

    def _solver(self):
        if self._qp is None:
            from .qp import BatchedBoxQP
            Pb, tqb = self._box_form()
            self._qp = BatchedBoxQP(Pb, tqb, self.Nu, max_batch=self._max_batch, **self._opts)
        return self._qp

    def solve_batch(self, X0, ulb=None, uub=None, first_move_only=False, guess=None):
        """B problems at once.  X0 (B, n_aug); ulb/uub (B, Nu) or (Nu,[1]) (default: self.ulb/uub).

        Returns (U (B, N*Nu) or (B, Nu), info) with info = dict(active (B, 2 N Nu) bool in the
        row order of G, status (B,), ipm_iters, factorizations).  ``guess`` (B, N*Nu) uint8
        (0 free / 1 upper / 2 lower) warm-starts the solver on an estimated active set.
        """
        lb = self.ulb if ulb is None else ulb
        ub = self.uub if uub is None else uub
        out = self._solver().solve_batch(np.asarray(X0, float).reshape(-1, self.Nx),
                                         np.asarray(lb, float).reshape(-1, self.Nu),
                                         np.asarray(ub, float).reshape(-1, self.Nu), guess=guess,
                                         first_move_only=first_move_only)
        U = out.pop("u")
        self.last_info = out
        return U, out

    def solve(self, x0):
        """One problem, reference signature: x0 (n_aug, 1) -> useq (N*Nu, 1)  (reference :495-512)."""
        U, info = self.solve_batch(x0.reshape(1, -1), self.ulb.reshape(1, -1), self.uub.reshape(1, -1))
        if info["status"][0] == 2:                 # cvxopt raises on a singular KKT matrix / invalid data
            raise ArithmeticError("regulator QP: numerical failure (NaN / Inf in the data, lb > ub, or a non-positive pivot)")
        if info["status"][0] != 0:                 # cvxopt would return status 'unknown' (the reference does not look at it)
            warnings.warn("regulator QP: iteration budget exhausted, the returned sequence is NOT certified optimal",
                          RuntimeWarning)
        useq = U.reshape(-1, 1)
        self._save_data(x0, useq)
        return useq

    def _save_data(self, x0, useq):
        self.x0.append(x0)
        self.useq.append(useq)


class LinearMPCController:
    """Kalman filter + target selector + regulator (reference :519-701)."""

    def __init__(self, *, A, B, C, H, Qwx, Qwd, Rv, xprior, dprior, Rs, Qs, Bd, Cd, usp, uprev,
                 Q, R, S, ulb, uub, N):
        self.A, self.B, self.C, self.H = A, B, C, H
        self.Qwx, self.Qwd, self.Rv, self.xprior, self.dprior = Qwx, Qwd, Rv, xprior, dprior
        self.Rs, self.Qs, self.Bd, self.Cd, self.usp = Rs, Qs, Bd, Cd, usp
        self.uprev = uprev
        self.useq = np.tile(uprev, (N, 1))
        self.Q, self.R, self.S, self.ulb, self.uub, self.N = Q, R, S, ulb, uub, N
        self.Nx, self.Nu, self.Ny, self.Nd = A.shape[0], B.shape[1], C.shape[0], Bd.shape[1]
        self.filter = LinearMPCController.setup_filter(A=A, B=B, C=C, Bd=Bd, Cd=Cd, Qwx=Qwx, Qwd=Qwd, Rv=Rv,
                                                       xprior=xprior, dprior=dprior)
        self.target_selector = LinearMPCController.setup_target_selector(A=A, B=B, C=C, H=H, Bd=Bd, Cd=Cd, usp=usp,
                                                                         Qs=Qs, Rs=Rs, ulb=ulb, uub=uub)
        self.regulator = LinearMPCController.setup_regulator(A=A, B=B, Q=Q, R=R, S=S, N=N, ulb=ulb, uub=uub)
        (_, _, self.Qaug, self.Raug, self.Maug) = LinearMPCController.get_augmented_matrices_for_regulator(A, B, Q, R, S)
        self.average_stage_costs = [np.zeros((1, 1))]
        self.computation_times = []

    @staticmethod
    def setup_filter(A, B, C, Bd, Cd, Qwx, Qwd, Rv, xprior, dprior):
        (Aaug, Baug, Caug, Qwaug) = LinearMPCController.get_augmented_matrices_for_filter(A, B, C, Bd, Cd, Qwx, Qwd)
        return KalmanFilter(A=Aaug, B=Baug, C=Caug, Qw=Qwaug, Rv=Rv, xprior=np.concatenate((xprior, dprior)))

    @staticmethod
    def setup_target_selector(A, B, C, H, Bd, Cd, usp, Qs, Rs, ulb, uub):
        return TargetSelector(A=A, B=B, C=C, H=H, Bd=Bd, Cd=Cd, usp=usp, Rs=Rs, Qs=Qs, ulb=ulb, uub=uub)

    @staticmethod
    def setup_regulator(A, B, Q, R, S, N, ulb, uub, **kw):
        (Aaug, Baug, Qaug, Raug, Maug) = LinearMPCController.get_augmented_matrices_for_regulator(A, B, Q, R, S)
        return DenseQPRegulator(A=Aaug, B=Baug, Q=Qaug, R=Raug, N=N, M=Maug, ulb=ulb, uub=uub, **kw)

    @staticmethod
    def get_augmented_matrices_for_filter(A, B, C, Bd, Cd, Qwx, Qwd):
        """Integrating-disturbance augmentation (reference :606-624)."""
        Nx, Nu, Nd = A.shape[0], B.shape[1], Bd.shape[1]
        Aaug = np.block([[A, Bd], [np.zeros((Nd, Nx)), np.eye(Nd)]])
        Baug = np.vstack((B, np.zeros((Nd, Nu))))
        Caug = np.hstack((C, Cd))
        Qwaug = scipy.linalg.block_diag(Qwx, Qwd)
        assert_detectable(Aaug, Caug)
        return (Aaug, Baug, Caug, Qwaug)

    get_augmented_matrices_for_regulator = staticmethod(augmented_matrices_for_regulator)

    def control_law(self, ysp, y):
        """Measurement -> input; only the regulator solve is timed (reference :646-669)."""
        (xhat, dhat) = LinearMPCController.get_state_estimates(self.filter, y, self.uprev, self.Nx)
        (xs, us) = LinearMPCController.get_target_pair(self.target_selector, ysp, dhat)
        tstart = time.time()
        self.useq = LinearMPCController.get_control_sequence(self.regulator, xhat, self.uprev, xs, us,
                                                             self.ulb, self.uub)
        tend = time.time()
        avg_ell = LinearMPCController.get_updated_average_stage_cost(
            xhat, self.uprev, xs, us, self.useq[0:self.Nu, :], self.Qaug, self.Raug, self.Maug,
            self.average_stage_costs[-1], len(self.average_stage_costs))
        self.average_stage_costs.append(avg_ell)
        self.uprev = self.useq[0:self.Nu, :]
        self.computation_times.append(tend - tstart)
        return self.uprev

    @staticmethod
    def get_state_estimates(filter, y, uprev, Nx):
        return np.split(filter.solve(y, uprev), [Nx])

    @staticmethod
    def get_target_pair(target_selector, ysp, dhat):
        return target_selector.solve(ysp, dhat)

    @staticmethod
    def get_control_sequence(regulator, x, uprev, xs, us, ulb, uub):
        """Deviation variables in, absolute input sequence out (reference :682-689)."""
        regulator.ulb = ulb - us
        regulator.uub = uub - us
        x0 = np.concatenate((x - xs, uprev - us))
        return regulator.solve(x0) + np.tile(us, (regulator.N, 1))

    @staticmethod
    def get_control_sequence_batch(regulator, X, Uprev, Xs, Us, ulb, uub, first_move_only=True, guess=None):
        """Batched counterpart of get_control_sequence: rows are samples (B, .)."""
        X0 = np.concatenate((X - Xs, Uprev - Us), axis=1)
        U, info = regulator.solve_batch(X0, ulb.reshape(1, -1) - Us, uub.reshape(1, -1) - Us,
                                        first_move_only=first_move_only, guess=guess)
        return U + (Us if first_move_only else np.tile(Us, (1, regulator.N))), info

    @staticmethod
    def get_updated_average_stage_cost(x, uprev, xs, us, u, Qaug, Raug, Maug, average_stage_cost, time_index):
        """Running mean of the stage cost (reference :691-701)."""
        x = np.concatenate((x - xs, uprev - us), axis=0)
        u = u - us
        stage_cost = x.T @ (Qaug @ x) + u.T @ (Raug @ u) + x.T @ (Maug @ u) + u.T @ (Maug.T @ x)
        return (average_stage_cost * (time_index - 1) + stage_cost) / time_index


def online_simulation(plant, controller, *, setpoints=None, disturbances=None, Nsim=None, stdout_filename=None):
    """Closed loop plant <-> controller (reference :703-718)."""
    if stdout_filename is not None:
        sys.stdout = open(stdout_filename, 'w')
    measurement = plant.y[0]
    setpoints = setpoints[..., np.newaxis]
    disturbances = disturbances[..., np.newaxis]
    for (setpoint, disturbance, i) in zip(setpoints, disturbances, range(Nsim)):
        print("Simulation Step:" + f"{i}")
        control_input = controller.control_law(setpoint, measurement)
        print("Computation time:" + str(controller.computation_times[-1]))
        measurement = plant.step(control_input, disturbance)
    return plant


def _save_training_data(dictionary, filename):
    """One dataset per key (reference lib/python_utils.py:41-57); .h5 when h5py exists, else .npz."""
    try:
        import h5py
        with h5py.File(filename, "w") as f:
            for k, v in dictionary.items():
                f.create_dataset(k, data=v)
        return filename
    except ImportError:
        np.savez(filename + ".npz", **dictionary)
        return filename + ".npz"


def _target_pairs(target_selectors, setpoints, disturbances):
    """(Xs (T, nc, Nx), Us (T, nc, Nu)) for every chain and step (reference :851, one QP per step and chain).

    Chains whose selectors are TargetSelector objects go through ONE deduplicated batched solve (all chains of a task
    share the plant, so the first selector's matrices serve them all); anything else with a ``solve(ysp, dhat)``
    method is asked step by step."""
    nc, T = len(setpoints), setpoints[0].shape[0]
    if all(isinstance(ts, TargetSelector) for ts in target_selectors):
        # chain-major rows: consecutive rows are consecutive steps of ONE chain, so the piecewise-constant signals give long runs
        # of equal rows (target.unique_rows_piecewise)
        Ysp = np.stack([np.asarray(setpoints[c], float) for c in range(nc)], axis=0)       # (nc, T, Ny)
        Dh = np.stack([np.asarray(disturbances[c], float) for c in range(nc)], axis=0)     # (nc, T, Nd)
        Xs, Us = target_selectors[0].solve_batch(Ysp.reshape(nc * T, -1), Dh.reshape(nc * T, -1))
        return (np.ascontiguousarray(np.swapaxes(Xs.reshape(nc, T, -1), 0, 1)),
                np.ascontiguousarray(np.swapaxes(Us.reshape(nc, T, -1), 0, 1)))
    first = target_selectors[0].solve(setpoints[0][0][:, None], disturbances[0][0][:, None])
    Xs, Us = np.empty((T, nc, first[0].size)), np.empty((T, nc, first[1].size))
    for c in range(nc):
        for t in range(T):
            (xs, us) = LinearMPCController.get_target_pair(target_selectors[c], setpoints[c][t][:, None],
                                                           disturbances[c][t][:, None])
            Xs[t, c], Us[t, c] = xs[:, 0], us[:, 0]
    return Xs, Us


def simulate_chains(x0, uprev0, A, B, Bd, regulator, ulb, uub, target_selectors, setpoints, disturbances,
                    warm_start=True, device_resident=True, allow_uncertified=False):
    """Lock-step closed-loop chains: chain c follows setpoints[c] (T, Ny), disturbances[c] (T, Nd).

    Target pairs of all chains and steps first (deduplicated, batched), then per step ONE batched regulator solve for
    all chains and the model step x+ = A x + B u + Bd d.  Same recurrences and outputs as the reference's
    simulate_offline (:845-872), which runs one chain per OS process.  ``device_resident`` (default): the whole loop runs
    in the library (nnmpc_chain_run) with the chain state and the records in HBM; otherwise the loop is driven from
    here, one host round trip per step (also reports ``factorizations``).  With ``warm_start`` the active set of step t,
    shifted by one stage, seeds the solve of step t+1; results are identical either way (every solve is KKT-certified).
    A solve that ends uncertified (status != 0) raises unless ``allow_uncertified``; ``status`` is part of the result.
    """
    nc = len(setpoints)
    T = setpoints[0].shape[0]
    Nx, Nu = B.shape
    Xs, Us = _target_pairs(target_selectors, setpoints, disturbances)
    D = np.stack([np.asarray(disturbances[c], float) for c in range(nc)], axis=1)             # (T, nc, Nd)
    if device_resident:
        from .chain import DeviceChains
        ch = DeviceChains(regulator._solver(), nc, A, B, Bd, ulb, uub, x0, uprev0)
        rec = ch.run(Xs, Us, D, warm_start=warm_start)
        ch.close()
        out = {k: np.ascontiguousarray(np.swapaxes(rec[k], 0, 1)) for k in ("x", "uprev", "u")}
        out["xs"], out["us"] = np.ascontiguousarray(np.swapaxes(Xs, 0, 1)), np.ascontiguousarray(np.swapaxes(Us, 0, 1))
        out["status"] = np.ascontiguousarray(rec["status"].T)
        out["factorizations"] = np.zeros((nc, T), np.int32)
    else:
        X = np.tile(x0.T, (nc, 1))
        Uprev = np.tile(uprev0.T, (nc, 1))
        out = dict(x=np.empty((nc, T, Nx)), uprev=np.empty((nc, T, Nu)), xs=np.empty((nc, T, Nx)),
                   us=np.empty((nc, T, Nu)), u=np.empty((nc, T, Nu)))
        status = np.zeros((nc, T), np.int32)
        nfac = np.zeros((nc, T), np.int32)
        guess = None
        for t in range(T):
            U, info = LinearMPCController.get_control_sequence_batch(regulator, X, Uprev, Xs[t], Us[t], ulb, uub, guess=guess)
            status[:, t] = info["status"]
            nfac[:, t] = info["factorizations"]
            if warm_start:
                st = regulator._solver().active_to_state(info["active"])      # (nc, N*Nu)
                guess = np.concatenate((st[:, Nu:], st[:, -Nu:]), axis=1)      # shift one stage, repeat the last
            out["x"][:, t], out["uprev"][:, t], out["xs"][:, t], out["us"][:, t], out["u"][:, t] = X, Uprev, Xs[t], Us[t], U
            X = X @ A.T + U @ B.T + D[t] @ Bd.T
            Uprev = U
        out["status"] = status
        out["factorizations"] = nfac
    if not allow_uncertified and (out["status"] != 0).any():
        c, t = np.argwhere(out["status"] != 0)[0]
        raise RuntimeError(f"simulate_chains: {int((out['status'] != 0).sum())} regulator solve(s) not certified optimal "
                           f"(first: chain {c}, step {t}, status {int(out['status'][c, t])}); the trajectories after that "
                           "step are not the MPC law's -- pass allow_uncertified=True to get them anyway")
    return out


def simulate_offline(task_number, process_number, data_filename, x0, uprev0, A, B, Bd,
                     regulator, ulb, uub, target_selector, setpoints, disturbances):
    """One chain, reference signature (:827-880); runs through the batched driver with B = 1."""
    t0 = time.time()
    res = simulate_chains(x0, uprev0, A, B, Bd, regulator, ulb, uub, [target_selector], [setpoints], [disturbances])
    data = {k: res[k][0] for k in ("x", "uprev", "xs", "us", "u", "status")}
    data["data_gen_time"] = time.time() - t0
    return _save_training_data(data, str(task_number) + '-' + str(process_number) + '-' + data_filename)


class OfflineSimulator:
    """Offline data generation (reference :720-825): the long PRBS signal is cut into
    num_data_gen_task * num_process_per_task contiguous chains (_split_scenarios, :786-801);
    generate_data(task) advances all chains of the task in lock-step on the GPU."""

    def __init__(self, *, A, B, C, H, Rs, Qs, Bd, Cd, usp, uprev, Q, R, S, ulb, uub, N, xprior,
                 setpoints, disturbances, num_data_gen_task, num_process_per_task):
        self.A, self.B, self.C, self.H = A, B, C, H
        self.Rs, self.Qs, self.Bd, self.Cd, self.usp = Rs, Qs, Bd, Cd, usp
        self.Q, self.R, self.S, self.ulb, self.uub, self.N = Q, R, S, ulb, uub, N
        self.num_data_gen_task, self.num_process_per_task = num_data_gen_task, num_process_per_task
        self.Nx, self.Nu, self.Ny, self.Nd = A.shape[0], B.shape[1], C.shape[0], Bd.shape[1]
        self.x0, self.uprev0 = xprior, uprev
        # one regulator (the GPU handle is shared by all chains), one target selector per chain
        self.regulator = LinearMPCController.setup_regulator(A=A, B=B, Q=Q, R=R, S=S, N=N, ulb=ulb, uub=uub)
        self.regulators = [self.regulator] * num_process_per_task
        self.target_selectors = [LinearMPCController.setup_target_selector(
            A=A, B=B, C=C, H=H, Bd=Bd, Cd=Cd, usp=usp, Qs=Qs, Rs=Rs, ulb=ulb, uub=uub)
            for _ in range(num_process_per_task)]
        (self.setpoints, self.disturbances) = self._split_scenarios(setpoints=setpoints, disturbances=disturbances)

    def _split_scenarios(self, *, setpoints, disturbances):
        nt, npp = self.num_data_gen_task, self.num_process_per_task
        each = int(setpoints.shape[0] / (nt * npp))
        sp = [setpoints[k * each:(k + 1) * each, :] for k in range(nt * npp)]
        ds = [disturbances[k * each:(k + 1) * each, :] for k in range(nt * npp)]
        return ([sp[t * npp:(t + 1) * npp] for t in range(nt)], [ds[t * npp:(t + 1) * npp] for t in range(nt)])

    def generate_data(self, *, task_number, data_filename, stdout_filename=None):
        """Writes '<task>-<proc>-<data_filename>' per chain, like the reference (:877-880)."""
        if stdout_filename is not None:
            sys.stdout = open(stdout_filename, 'w')
        t0 = time.time()
        res = simulate_chains(self.x0, self.uprev0, self.A, self.B, self.Bd, self.regulator, self.ulb, self.uub,
                              self.target_selectors, self.setpoints[task_number], self.disturbances[task_number])
        dt = time.time() - t0
        files = []
        for proc in range(self.num_process_per_task):
            data = {k: res[k][proc] for k in ("x", "uprev", "xs", "us", "u", "status")}   # status: 0 = certified optimal
            data["data_gen_time"] = dt
            files.append(_save_training_data(data, str(task_number) + '-' + str(proc) + '-' + data_filename))
        return files

    _REC_KEYS = ("x", "uprev", "xs", "us", "u")

    def generate_dataset(self, *, data_filename, task_numbers=None, comm=None, rank=None, world=None, gather=None,
                         write_files=True, allow_uncertified=False):
        """The whole data set, sharded over the ranks of a job (one process per GPU).

        The reference runs every task as its own OS process / cluster job on a contiguous slice of the scenario signal and
        lets the per-task files meet on the file system (:786-825, controller_evaluation.py:273-295).  Here task -> rank by
        contiguous blocks (``distributed.shard_bounds``), every rank advances ALL chains of its tasks in one lock-step batch
        on its GPU (``simulate_chains``: state, targets and records in HBM), and ONE gather of the records -- ``comm``
        (``distributed.Comm``: nnmpc_comm_gather_rows, RCCL over xGMI) or, in the CPU tests, a ``gather(rows, counts)``
        callable -- brings them to rank 0, which writes the reference's per-chain files '<task>-<proc>-<data_filename>' and
        returns what ``_post_process_data`` would return for them.  Other ranks return None.  Without ``comm`` / ``rank`` /
        ``world`` this is the single-process case: every task in one batch.
        """
        from . import distributed as dd
        tasks = list(range(self.num_data_gen_task)) if task_numbers is None else [int(t) for t in task_numbers]
        if comm is not None:
            rank, world = comm.rank, comm.world
        rank, world = int(rank or 0), int(world or 1)
        npp = self.num_process_per_task
        cuts = [dd.shard_bounds(len(tasks), r, world) for r in range(world)]
        mine = tasks[cuts[rank][0]:cuts[rank][1]]
        T = self.setpoints[tasks[0]][0].shape[0] if tasks else 0
        Nx, Nu = self.Nx, self.Nu
        width = 2 * Nx + 3 * Nu + 2                                   # x, uprev, xs, us, u, status, data_gen_time
        t0 = time.time()
        if mine:
            sp = [self.setpoints[t][c] for t in mine for c in range(npp)]
            ds = [self.disturbances[t][c] for t in mine for c in range(npp)]
            sel = [self.target_selectors[c] for _ in mine for c in range(npp)]
            res = simulate_chains(self.x0, self.uprev0, self.A, self.B, self.Bd, self.regulator, self.ulb, self.uub,
                                  sel, sp, ds, allow_uncertified=True)
            dt = time.time() - t0
        else:
            res, dt = None, 0.0
        counts = [(hi - lo) * npp * T for lo, hi in cuts]
        if world == 1:
            # single process: the records are already chain-major (nc, T, width of the key) -- no packing, no copy
            views = {k: res[k] for k in self._REC_KEYS} if res is not None else {k: np.empty((0, T, 0)) for k in self._REC_KEYS}
            status = res["status"].astype(np.int32) if res is not None else np.empty((0, T), np.int32)
            dtimes = np.full(len(tasks) * npp, dt)
        else:
            if res is not None:
                rec = np.concatenate([res[k] for k in self._REC_KEYS] + [res["status"][..., None].astype(float),
                                                                        np.full(res["status"].shape + (1,), dt)], axis=2)
                rec = np.ascontiguousarray(rec.reshape(-1, width))
            else:
                rec = np.empty((0, width))
            if comm is not None:
                from . import _lib
                send = _lib.DeviceArray.from_host(rec) if rec.size else None
                recv = _lib.DeviceArray((sum(counts), width), np.float64) if rank == 0 else None
                comm.gather_rows(send, counts, width, recv, root=0)        # the single collective of the job
                full = recv.to_host() if rank == 0 else None
                for a in (send, recv):
                    if a is not None:
                        a.free()
            elif gather is not None:
                full = gather(rec, counts)
            else:
                raise ValueError("generate_dataset: world > 1 needs comm= (distributed.Comm) or gather=")
            if rank != 0:
                return None
            full = full.reshape(len(tasks) * npp, T, width)
            status = full[:, :, -2].astype(np.int32)
            dtimes = full[:, 0, -1]
            splits = np.concatenate(([0], np.cumsum([Nx, Nu, Nx, Nu, Nu])))
            views = {k: full[:, :, splits[i]:splits[i + 1]] for i, k in enumerate(self._REC_KEYS)}
        if not allow_uncertified and (status != 0).any():
            c, t = np.argwhere(status != 0)[0]
            raise RuntimeError(f"generate_dataset: {int((status != 0).sum())} regulator solve(s) not certified optimal (first: chain "
                               f"{c}, step {t}); pass allow_uncertified=True to keep the data anyway")
        if write_files:
            for i, task in enumerate(tasks):
                for proc in range(npp):
                    j = i * npp + proc
                    one = {k: np.ascontiguousarray(views[k][j]) for k in self._REC_KEYS}
                    one["status"] = status[j]
                    one["data_gen_time"] = float(dtimes[j])
                    _save_training_data(one, str(task) + '-' + str(proc) + '-' + data_filename)
        out = {k: np.ascontiguousarray(views[k]).reshape(len(tasks) * npp * T, -1) for k in self._REC_KEYS}
        out["status"] = status.reshape(-1)
        out["data_gen_time"] = float(np.mean(dtimes)) if len(dtimes) else 0.0
        return out


