"""ctypes binding of libnnmpc_hip.so (C ABI: include/nnmpc.h).  Fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libnnmpc_hip.so")

HOST, DEVICE = 0, 1
ST_OPTIMAL, ST_MAXITER, ST_NUMERIC = 0, 1, 2


class QpOpts(C.Structure):
    _fields_ = [("max_batch", C.c_int32), ("nb", C.c_int32), ("max_ipm_iters", C.c_int32),
                ("max_polish_rounds", C.c_int32), ("max_refine", C.c_int32),
                ("max_rounds", C.c_int32), ("sub_steps", C.c_int32), ("stale_max_changes", C.c_int32),
                ("stale_cg_limit", C.c_int32), ("method", C.c_int32), ("asm_max_active", C.c_int32),
                ("asm_max_rounds", C.c_int32), ("asm_f32_rounds", C.c_int32), ("seg_max", C.c_int32), ("ipm_tol", C.c_float), ("refine_tol", C.c_double),
                ("bound_tol", C.c_double)]


class QpStats(C.Structure):
    _fields_ = [("problems", C.c_int64), ("rounds", C.c_int64), ("factorizations", C.c_int64),
                ("ipm_iterations", C.c_int64), ("panel_launches", C.c_int64),
                ("panel_ms", C.c_double), ("diag_ms", C.c_double), ("trsv_ms", C.c_double),
                ("total_ms", C.c_double), ("panel_flops", C.c_double), ("trsv_solves", C.c_int64), ("asm_solved", C.c_int64),
                ("asm_rounds", C.c_int64), ("asm_gemm_launches", C.c_int64), ("asm_gemm_ms", C.c_double),
                ("asm_gemm_flops", C.c_double), ("asm_lambda_ms", C.c_double), ("asm_update_ms", C.c_double), ("asm_lambda_flops", C.c_double),
                ("asm_lambda_bytes", C.c_double), ("asm_e1max", C.c_double),
                ("asm_e2max", C.c_double), ("asm_full_checks", C.c_int64)]


EXPORTS = ["nnmpc_last_error", "nnmpc_qp_create", "nnmpc_qp_destroy", "nnmpc_qp_solve_batch",
           "nnmpc_qp_solve_batch_warm", "nnmpc_qp_set_inverse",
           "nnmpc_qp_set_profiling", "nnmpc_qp_get_stats", "nnmpc_qp_debug_factor_solve",
           "nnmpc_nn_create", "nnmpc_nn_destroy", "nnmpc_nn_forward", "nnmpc_nn_last_ms", "nnmpc_nn_last_hidden_ms"]

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
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().nnmpc_last_error().decode(errors="replace")
        raise NnmpcError(f"{what} failed (code {rc}): {msg}")
