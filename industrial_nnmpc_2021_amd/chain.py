"""Device-resident lock-step chains (host wrapper over nnmpc_chain_*).

The reference's simulate_offline (lib/linearMPC.py:827-880) runs ONE chain per OS process: per step a target pair, a
regulator QP, a model step.  ``DeviceChains`` advances all nc chains of a task together with the chain state, the
target pairs / disturbances of all T steps and the recorded trajectories in HBM (see include/nnmpc.h).
"""
import ctypes as C

import numpy as np

from . import _lib


class DeviceChains:
    """regulator_qp: the ``qp.BatchedBoxQP`` of the regulator (n_aug = Nx + Nu); it must outlive this object."""

    def __init__(self, regulator_qp, nc, A, B, Bd, ulb, uub, x0, uprev0):
        lib = _lib.load()
        f = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        A, B = f(A), f(B)
        self.nx, self.nu = B.shape
        Bd = f(Bd).reshape(self.nx, -1)
        self.nd, self.nc = Bd.shape[1], int(nc)
        self._qp = regulator_qp                       # keeps the regulator handle alive
        self._h = C.c_void_p()
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        ulb, uub, x0, uprev0 = f(ulb).ravel(), f(uub).ravel(), f(x0).ravel(), f(uprev0).ravel()
        if ulb.size != self.nu or uub.size != self.nu or x0.size != self.nx or uprev0.size != self.nu:
            raise ValueError("DeviceChains: ulb/uub/uprev0 need Nu entries, x0 needs Nx")
        _lib.check(lib.nnmpc_chain_create(C.byref(self._h), regulator_qp._h, self.nc, self.nx, self.nu, self.nd,
                                          p(A), p(B), p(Bd), p(ulb), p(uub), p(x0), p(uprev0)), "nnmpc_chain_create")
        self._lib = lib

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nnmpc_chain_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def reset(self):
        _lib.check(self._lib.nnmpc_chain_reset(self._h), "nnmpc_chain_reset")

    def run(self, Xs, Us, D, warm_start=True):
        """Xs (T, nc, Nx), Us (T, nc, Nu), D (T, nc, Nd) -> dict(x (T, nc, Nx), uprev, u (T, nc, Nu), status (T, nc)).

        x[t], uprev[t]: state and previous input before the move of step t; u[t]: that move (absolute)."""
        f = lambda a, w: np.ascontiguousarray(a, dtype=np.float64).reshape(-1, self.nc, w)
        Xs, Us = f(Xs, self.nx), f(Us, self.nu)
        T = Xs.shape[0]
        D = f(D, self.nd) if self.nd else np.zeros((T, self.nc, 0))
        if Us.shape[0] != T or D.shape[0] != T:
            raise ValueError("DeviceChains.run: Xs, Us, D need the same number of steps")
        out = dict(x=np.empty((T, self.nc, self.nx)), uprev=np.empty((T, self.nc, self.nu)),
                   u=np.empty((T, self.nc, self.nu)), status=np.empty((T, self.nc), np.int32))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.nnmpc_chain_run(self._h, T, p(Xs), p(Us), p(D) if self.nd else None, p(out["x"]),
                                             p(out["uprev"]), p(out["u"]), p(out["status"]), int(bool(warm_start)),
                                             _lib.HOST), "nnmpc_chain_run")
        self._after_run()
        return out

    def run_device(self, T, Xs, Us, D, x_rec, uprev_rec, u_rec, status, warm_start=True):
        """Same with HBM-resident buffers (objects with data_ptr())."""
        q = lambda a: None if a is None else C.c_void_p(a.data_ptr())
        _lib.check(self._lib.nnmpc_chain_run(self._h, int(T), q(Xs), q(Us), q(D), q(x_rec), q(uprev_rec), q(u_rec),
                                             q(status), int(bool(warm_start)), _lib.DEVICE), "nnmpc_chain_run")
        self._after_run()

    def _after_run(self):
        """More than 256 chains advance in lock-step rounds whose full-width pass wants far-field factors for its window: the
        library queues the windows it met without (nnmpc_qp_farfield_missing), the regulator's wrapper factors them."""
        fa = getattr(self._qp, "_farfield_auto", None)
        if fa is not None:
            fa()

    def last_ms(self):
        """(hipEvent time of the last run, host time spent inside its regulator solves), milliseconds."""
        a, b = C.c_double(), C.c_double()
        self._lib.nnmpc_chain_last_ms(self._h, C.byref(a), C.byref(b))
        return a.value, b.value
