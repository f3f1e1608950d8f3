"""ctypes binding of libnnmpc_hip.so (C ABI: include/nnmpc.h).  Fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnnmpc_hip.so")

HOST, DEVICE = 0, 1
ST_OPTIMAL, ST_MAXITER, ST_NUMERIC = 0, 1, 2
OUT_SEQUENCE, OUT_FIRST_MOVE = 0, 1
EINVAL, EHIP, ENOMEM, ENOTIMPL = -1, -2, -3, -4


class QpOpts(C.Structure):
    _fields_ = [("max_batch", C.c_int32), ("nb", C.c_int32), ("max_ipm_iters", C.c_int32),
                ("max_polish_rounds", C.c_int32), ("max_refine", C.c_int32),
                ("max_rounds", C.c_int32), ("sub_steps", C.c_int32), ("stale_max_changes", C.c_int32),
                ("stale_cg_limit", C.c_int32), ("method", C.c_int32), ("asm_max_active", C.c_int32),
                ("asm_max_rounds", C.c_int32), ("asm_f32_rounds", C.c_int32), ("seg_max", C.c_int32), ("asm_tail_batch", C.c_int32), ("asm_predict_iters", C.c_int32), ("ipm_tol", C.c_float), ("refine_tol", C.c_double),
                ("bound_tol", C.c_double)]


class QpStats(C.Structure):
    _fields_ = [("problems", C.c_int64), ("rounds", C.c_int64), ("factorizations", C.c_int64),
                ("ipm_iterations", C.c_int64), ("panel_launches", C.c_int64),
                ("panel_ms", C.c_double), ("diag_ms", C.c_double), ("trsv_ms", C.c_double),
                ("total_ms", C.c_double), ("panel_flops", C.c_double), ("trsv_solves", C.c_int64), ("asm_solved", C.c_int64),
                ("asm_rounds", C.c_int64), ("asm_gemm_launches", C.c_int64), ("asm_gemm_ms", C.c_double),
                ("asm_gemm_flops", C.c_double), ("asm_lambda_ms", C.c_double), ("asm_update_ms", C.c_double), ("asm_lambda_flops", C.c_double),
                ("asm_lambda_bytes", C.c_double), ("asm_e1max", C.c_double),
                ("asm_e2max", C.c_double), ("asm_full_checks", C.c_int64), ("asm_lambda32_ms", C.c_double),
                ("asm_lambda64_ms", C.c_double), ("asm_lambda32_flops", C.c_double),
                ("asm_lambda32_launches", C.c_int64), ("asm_lambda64_launches", C.c_int64), ("asm_far_passes", C.c_int64), ("asm_side_ms", C.c_double),
                ("asm_small_passes", C.c_int64), ("asm_predict_launches", C.c_int64), ("asm_predict_ms", C.c_double),
                ("asm_predict_flops", C.c_double)]


EXPORTS = ["nnmpc_last_error", "nnmpc_qp_create", "nnmpc_qp_destroy", "nnmpc_qp_solve_batch",
           "nnmpc_qp_solve_batch_warm", "nnmpc_qp_solve_batch_ex", "nnmpc_qp_set_inverse", "nnmpc_qp_dims",
           "nnmpc_qp_first_moves", "nnmpc_qp_set_farfield", "nnmpc_qp_farfield_missing",
           "nnmpc_qp_set_profiling", "nnmpc_qp_get_stats", "nnmpc_qp_debug_factor_solve",
           "nnmpc_nn_create", "nnmpc_nn_destroy", "nnmpc_nn_forward", "nnmpc_nn_last_ms", "nnmpc_nn_last_hidden_ms",
           "nnmpc_chain_create", "nnmpc_chain_destroy", "nnmpc_chain_run", "nnmpc_chain_reset", "nnmpc_chain_last_ms",
           "nnmpc_ts_create", "nnmpc_ts_destroy", "nnmpc_ts_solve_batch",
           "nnmpc_device_count", "nnmpc_set_device", "nnmpc_device_synchronize", "nnmpc_dev_mem_info",
           "nnmpc_dev_malloc", "nnmpc_dev_free", "nnmpc_dev_memset", "nnmpc_memcpy_h2d", "nnmpc_memcpy_d2h",
           "nnmpc_memcpy_d2d", "nnmpc_host_alloc_pinned", "nnmpc_host_free_pinned",
           "nnmpc_comm_unique_id", "nnmpc_comm_init", "nnmpc_comm_destroy", "nnmpc_comm_rank", "nnmpc_comm_world",
           "nnmpc_comm_gather_rows", "nnmpc_comm_allreduce_max", "nnmpc_comm_barrier"]

_lib = None


class NnmpcError(RuntimeError):
    pass


def load():
    """Load the HIP library; raise (never fall back) if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NnmpcError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, dp = C.c_void_p, C.c_int32, C.c_void_p
    lib.nnmpc_last_error.restype = C.c_char_p
    lib.nnmpc_last_error.argtypes = []
    lib.nnmpc_qp_create.restype = i32
    lib.nnmpc_qp_create.argtypes = [C.POINTER(vp), i32, i32, i32, dp, dp, dp, C.POINTER(QpOpts)]
    lib.nnmpc_qp_destroy.restype = i32
    lib.nnmpc_qp_destroy.argtypes = [vp]
    lib.nnmpc_qp_solve_batch.restype = i32
    lib.nnmpc_qp_solve_batch.argtypes = [vp, i32, dp, dp, dp, dp, dp, dp, dp, i32]
    lib.nnmpc_qp_solve_batch_warm.restype = i32
    lib.nnmpc_qp_solve_batch_warm.argtypes = [vp, i32, dp, dp, dp, dp, dp, dp, dp, dp, i32]
    lib.nnmpc_qp_set_inverse.restype = i32
    lib.nnmpc_qp_set_inverse.argtypes = [vp, dp, dp]
    lib.nnmpc_qp_set_farfield.restype = i32
    lib.nnmpc_qp_set_farfield.argtypes = [vp, i32, i32, dp, dp, dp]
    lib.nnmpc_qp_farfield_missing.restype = i32
    lib.nnmpc_qp_farfield_missing.argtypes = [vp, C.POINTER(i32)]
    lib.nnmpc_qp_set_profiling.restype = i32
    lib.nnmpc_qp_set_profiling.argtypes = [vp, i32]
    lib.nnmpc_qp_get_stats.restype = i32
    lib.nnmpc_qp_get_stats.argtypes = [vp, C.POINTER(QpStats), i32]
    lib.nnmpc_qp_debug_factor_solve.restype = i32
    lib.nnmpc_qp_debug_factor_solve.argtypes = [vp, i32, dp, dp, dp, dp]
    lib.nnmpc_nn_create.restype = i32
    lib.nnmpc_nn_create.argtypes = [C.POINTER(vp), i32, C.POINTER(i32), C.POINTER(dp), C.POINTER(dp),
                                    i32, i32, i32, dp, dp, dp, i32, i32]
    lib.nnmpc_nn_destroy.restype = i32
    lib.nnmpc_nn_destroy.argtypes = [vp]
    lib.nnmpc_nn_forward.restype = i32
    lib.nnmpc_nn_forward.argtypes = [vp, i32, dp, dp, dp, dp, dp, i32]
    lib.nnmpc_nn_last_ms.restype = i32
    lib.nnmpc_nn_last_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.nnmpc_nn_last_hidden_ms.restype = i32
    lib.nnmpc_nn_last_hidden_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    u64 = C.c_uint64
    lib.nnmpc_qp_solve_batch_ex.restype = i32
    lib.nnmpc_qp_solve_batch_ex.argtypes = [vp, i32, dp, dp, dp, dp, dp, dp, dp, dp, i32, i32]
    lib.nnmpc_qp_first_moves.restype = i32
    lib.nnmpc_qp_first_moves.argtypes = [dp, C.c_int64, dp, i32, i32, dp]
    lib.nnmpc_qp_dims.restype = i32
    lib.nnmpc_qp_dims.argtypes = [vp, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    lib.nnmpc_chain_create.restype = i32
    lib.nnmpc_chain_create.argtypes = [C.POINTER(vp), vp, i32, i32, i32, i32, dp, dp, dp, dp, dp, dp, dp]
    lib.nnmpc_chain_destroy.restype = i32
    lib.nnmpc_chain_destroy.argtypes = [vp]
    lib.nnmpc_chain_run.restype = i32
    lib.nnmpc_chain_run.argtypes = [vp, i32, dp, dp, dp, dp, dp, dp, dp, i32, i32]
    lib.nnmpc_chain_reset.restype = i32
    lib.nnmpc_chain_reset.argtypes = [vp]
    lib.nnmpc_chain_last_ms.restype = i32
    lib.nnmpc_chain_last_ms.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.nnmpc_ts_create.restype = i32
    lib.nnmpc_ts_create.argtypes = [C.POINTER(vp), i32, i32, dp, dp, dp, dp]
    lib.nnmpc_ts_destroy.restype = i32
    lib.nnmpc_ts_destroy.argtypes = [vp]
    lib.nnmpc_ts_solve_batch.restype = i32
    lib.nnmpc_ts_solve_batch.argtypes = [vp, i32, dp, dp, dp, dp, dp, dp, i32]
    lib.nnmpc_device_count.restype = i32
    lib.nnmpc_device_count.argtypes = []
    lib.nnmpc_set_device.restype = i32
    lib.nnmpc_set_device.argtypes = [i32]
    lib.nnmpc_device_synchronize.restype = i32
    lib.nnmpc_device_synchronize.argtypes = []
    lib.nnmpc_dev_mem_info.restype = i32
    lib.nnmpc_dev_mem_info.argtypes = [C.POINTER(u64), C.POINTER(u64)]
    lib.nnmpc_dev_malloc.restype = i32
    lib.nnmpc_dev_malloc.argtypes = [C.POINTER(vp), u64]
    lib.nnmpc_dev_free.restype = i32
    lib.nnmpc_dev_free.argtypes = [vp]
    lib.nnmpc_dev_memset.restype = i32
    lib.nnmpc_dev_memset.argtypes = [vp, i32, u64]
    for name in ("nnmpc_memcpy_h2d", "nnmpc_memcpy_d2h", "nnmpc_memcpy_d2d"):
        getattr(lib, name).restype = i32
        getattr(lib, name).argtypes = [vp, vp, u64]
    lib.nnmpc_host_alloc_pinned.restype = i32
    lib.nnmpc_host_alloc_pinned.argtypes = [C.POINTER(vp), u64]
    lib.nnmpc_host_free_pinned.restype = i32
    lib.nnmpc_host_free_pinned.argtypes = [vp]
    lib.nnmpc_comm_unique_id.restype = i32
    lib.nnmpc_comm_unique_id.argtypes = [vp]
    lib.nnmpc_comm_init.restype = i32
    lib.nnmpc_comm_init.argtypes = [C.POINTER(vp), vp, i32, i32]
    lib.nnmpc_comm_destroy.restype = i32
    lib.nnmpc_comm_destroy.argtypes = [vp]
    lib.nnmpc_comm_rank.restype = i32
    lib.nnmpc_comm_rank.argtypes = [vp]
    lib.nnmpc_comm_world.restype = i32
    lib.nnmpc_comm_world.argtypes = [vp]
    lib.nnmpc_comm_gather_rows.restype = i32
    lib.nnmpc_comm_gather_rows.argtypes = [vp, dp, C.POINTER(C.c_int64), i32, dp, i32]
    lib.nnmpc_comm_allreduce_max.restype = i32
    lib.nnmpc_comm_allreduce_max.argtypes = [vp, C.POINTER(C.c_double)]
    lib.nnmpc_comm_barrier.restype = i32
    lib.nnmpc_comm_barrier.argtypes = [vp]
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().nnmpc_last_error().decode(errors="replace")
        raise NnmpcError(f"{what} failed (code {rc}): {msg}")


class DeviceArray:
    """HBM-resident array owned through the library (nnmpc_dev_malloc): shape, numpy dtype, data_ptr().

    What a torch CUDA tensor is used for elsewhere -- device memory that outlives a call -- without binding
    anything but libnnmpc_hip.so.  ``DeviceArray.from_host(a)`` uploads, ``.to_host()`` downloads.
    """

    def __init__(self, shape, dtype):
        import numpy as np
        self.shape = tuple(int(v) for v in (shape if hasattr(shape, "__len__") else (shape,)))
        self.dtype = np.dtype(dtype)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        self._p = C.c_void_p()
        check(load().nnmpc_dev_malloc(C.byref(self._p), self.nbytes), "nnmpc_dev_malloc")

    @classmethod
    def from_host(cls, a):
        import numpy as np
        a = np.ascontiguousarray(a)
        d = cls(a.shape, a.dtype)
        d.upload(a)
        return d

    def data_ptr(self):
        return self._p.value or 0

    def upload(self, a):
        import numpy as np
        a = np.ascontiguousarray(a, dtype=self.dtype)
        if a.nbytes != self.nbytes:
            raise ValueError("upload: size mismatch")
        check(load().nnmpc_memcpy_h2d(self._p, a.ctypes.data_as(C.c_void_p), self.nbytes), "nnmpc_memcpy_h2d")

    def to_host(self, rows=None):
        """The whole array, or its first ``rows`` rows."""
        import numpy as np
        shape = self.shape if rows is None else (int(rows),) + self.shape[1:]
        out = np.empty(shape, self.dtype)
        check(load().nnmpc_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self._p, out.nbytes), "nnmpc_memcpy_d2h")
        return out

    def row_ptr(self, row):
        """Device address of row ``row`` (leading dimension)."""
        import numpy as np
        stride = int(np.prod(self.shape[1:], dtype=np.int64)) * self.dtype.itemsize
        return C.c_void_p((self._p.value or 0) + int(row) * stride)

    def free(self):
        if getattr(self, "_p", None) is not None and self._p.value:
            load().nnmpc_dev_free(self._p)
            self._p = C.c_void_p()

    __del__ = free


def set_device(dev):
    check(load().nnmpc_set_device(int(dev)), "nnmpc_set_device")


def synchronize():
    check(load().nnmpc_device_synchronize(), "nnmpc_device_synchronize")


def device_count():
    return int(load().nnmpc_device_count())
