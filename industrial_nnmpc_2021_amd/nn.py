"""Structured NN controller forward on the GPU (host wrapper over the C ABI).

    u = clip(us + NN(x/xscale, [uprev], xs/xscale, us) - NN(xs/xscale, [us], xs/xscale, us))

Reference: RegulatorLayerWithUprev / RegulatorLayerWithoutUprev.call
(lib/LinearMPCLayers.py:40-61, :91-112) and NeuralNetworkController
._get_control_input (lib/controller_evaluation.py:863-892).
"""
import ctypes as C
import numpy as np

from . import _lib


class StructuredNN:
    """``weights``: Keras get_weights() order [W1 (in x h), b1, ..., Wout (h x nu)].

    ``use_bf16``: False -- f32 MFMA GEMMs; True -- bf16 operands, f32 accumulation (~2e-2 relative error on the CDU
    architecture); "split" (or 2) -- activations and weights as bf16 pairs hi + lo, every layer ONE bf16 GEMM of three times
    the depth (hi hi' + hi lo' + lo hi'): f32-grade results (~1e-5) from the bf16 matrix pipes."""

    def __init__(self, weights, nx, nu, *, nnwithuprev=True, xscale=None, ulb=None, uub=None,
                 max_batch=65536, use_bf16=False):
        lib = _lib.load()
        Ws = [np.ascontiguousarray(w, np.float64) for w in weights[0:-1:2]] + \
             [np.ascontiguousarray(weights[-1], np.float64)]
        bs = [np.ascontiguousarray(b, np.float64).ravel() for b in weights[1::2]]
        L = len(Ws)
        if len(bs) != L - 1:
            raise ValueError("weights must be [W1, b1, ..., W_{L-1}, b_{L-1}, Wout]")
        dims = [Ws[0].shape[0]] + [w.shape[1] for w in Ws]
        self.nx, self.nu, self.nnwithuprev = nx, nu, bool(nnwithuprev)
        dims_c = (C.c_int32 * (L + 1))(*dims)
        Wp = (C.c_void_p * L)(*[w.ctypes.data for w in Ws])
        bp = (C.c_void_p * L)(*([b.ctypes.data for b in bs] + [None]))
        opt = lambda a: None if a is None else np.ascontiguousarray(np.ravel(a), np.float64)
        xs_, lb_, ub_ = opt(xscale), opt(ulb), opt(uub)
        self._keep = (Ws, bs, xs_, lb_, ub_)
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        self._h = C.c_void_p()
        _lib.check(lib.nnmpc_nn_create(C.byref(self._h), L, dims_c, Wp, bp, nx, nu, int(nnwithuprev),
                                       p(xs_), p(lb_), p(ub_), 2 if use_bf16 in ("split", 2) else int(bool(use_bf16)), max_batch),
                   "nnmpc_nn_create")
        self._lib = lib

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.nnmpc_nn_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def forward(self, x, uprev, xs, us):
        """numpy (B, nx), (B, nu), (B, nx), (B, nu) -> (B, nu)."""
        c = lambda a, w: np.ascontiguousarray(a, np.float64).reshape(-1, w)
        x, xs, us = c(x, self.nx), c(xs, self.nx), c(us, self.nu)
        up = c(uprev, self.nu) if self.nnwithuprev else None
        B = x.shape[0]
        u = np.empty((B, self.nu))
        p = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        _lib.check(self._lib.nnmpc_nn_forward(self._h, B, p(x), p(up), p(xs), p(us), p(u), _lib.HOST),
                   "nnmpc_nn_forward")
        return u

    def forward_device(self, B, x, uprev, xs, us, u):
        """HBM-resident f64 buffers (objects with data_ptr())."""
        q = lambda a: None if a is None else C.c_void_p(a.data_ptr())
        _lib.check(self._lib.nnmpc_nn_forward(self._h, B, q(x), q(uprev), q(xs), q(us), q(u), _lib.DEVICE),
                   "nnmpc_nn_forward")

    def last_ms(self):
        g, t = C.c_double(), C.c_double()
        self._lib.nnmpc_nn_last_ms(self._h, C.byref(g), C.byref(t))
        return g.value, t.value

    def last_hidden_ms(self):
        """(hipEvent ms of the hidden-layer GEMMs of the last forward, number of those launches)."""
        g, k = C.c_double(), C.c_int32()
        self._lib.nnmpc_nn_last_hidden_ms(self._h, C.byref(g), C.byref(k))
        return g.value, k.value
