"""
ctypes binding of libgpnative.so (the C ABI declared in include/gpnative.h).

There is deliberately NO fallback: if the shared library is missing or a call
returns a non-zero status the caller gets an exception.  The product path never
routes through a CPU implementation.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GPN_LIB", os.path.join(_HERE, "lib", "libgpnative.so"))   # GPN_LIB: A/B builds (tools/)

c_void_p, c_int, c_int64, c_double = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_double

# name -> (restype, argtypes); must list every symbol of include/gpnative.h
SIGNATURES = {
    "gpn_version": (c_int, []),
    "gpn_arch": (ctypes.c_char_p, []),
    "gpn_last_hip_error": (ctypes.c_char_p, []),
    "gpn_factor_ld": (c_int64, [c_int64, c_int64]),
    "gpn_factor_rows": (c_int64, [c_int64, c_int64]),
    "gpn_winv_bytes": (c_int64, [c_int64]),
    "gpn_kernel_matrix": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                  c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64]),
    "gpn_pack_rhs": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_int64]),
    "gpn_potrf_lower": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "gpn_potrf_lower_panel": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "gpn_potrf_lower_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_int]),
    "gpn_lml_reduce_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int]),
    "gpn_lml_forward_batched": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int,
                                        c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "gpn_potrf_panel_width": (c_int64, [c_int64]),
    "gpn_potrf_panel_levels": (c_int, [c_int64, c_void_p]),
    "gpn_potrf_lower_persistent": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p]),
    "gpn_potrf_persistent_supported": (c_int, [c_int64, c_int64]),
    "gpn_potrf_persistent_plan": (c_int, [c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64]),
    "gpn_release_stream": (c_int, [c_void_p]),
    "gpn_trtri_diag": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "gpn_trsm_right_lt": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64]),
    "gpn_lml_reduce": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "gpn_gemm_nt": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64,
                            c_void_p, c_int64, c_double, c_void_p, c_int64, c_int, c_int]),
    "gpn_gemm_nt_stair": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64,
                                  c_void_p, c_int64, c_double, c_void_p, c_int64, c_int64, c_int]),
    "gpn_gemm_nt_batched": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_double, c_void_p, c_int64, c_int64,
                                    c_void_p, c_int64, c_int64, c_double, c_void_p, c_int64, c_int64, c_int, c_int, c_int]),
    "gpn_gemm_nt_batched_scaled": (c_int, [c_void_p, c_int64, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64,
                                           c_void_p, c_int64, c_int64, c_double, c_void_p, c_int64, c_int64, c_int, c_int, c_int]),
    "gpn_kernel_matrix_batched": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int,
                                          c_void_p, c_void_p, c_int, c_void_p, c_int, c_void_p, c_int64, c_int64]),
    "gpn_trsm_right_lt_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
                                          c_void_p, c_int64, c_int64, c_int64, c_int]),
    "gpn_trtri_upper_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64,
                                        c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int]),
    "gpn_kernel_grad_batched": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int,
                                        c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "gpn_kernel_grad_x2_batched": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int,
                                           c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int64, c_double, c_int, c_void_p, c_void_p]),
    "gpn_dot2d_batched": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int64, c_int64, c_void_p, c_int]),
    "gpn_lml_forward_ragged": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p, c_int64, c_int,
                                       c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_void_p, c_void_p]),
    "gpn_lml_backward_ragged": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                        c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "gpn_trtri_upper": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64]),
    "gpn_trtri_upper_ws": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64]),
    "gpn_grad_work_bytes": (c_int64, [c_int64, c_int64, c_int, c_int]),
    "gpn_lml_grad": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int,
                             c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "gpn_kernel_grad": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "gpn_grad_x2_work_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "gpn_kernel_grad_x2": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                   c_int, c_void_p, c_int64, c_double, c_int, c_void_p, c_void_p]),
    "gpn_lml_forward": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "gpn_lml_forward_saving": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                       c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    "gpn_lml_refine_work_bytes": (c_int64, [c_int64, c_int]),
    "gpn_lml_refine": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                               c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "gpn_lml_backward_work_bytes": (c_int64, [c_int64, c_int, c_int]),
    "gpn_lml_backward": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int,
                                 c_void_p, c_int64, c_void_p, c_int, c_void_p, c_void_p, c_void_p]),
    "gpn_lml_backward_batched_work_bytes": (c_int64, [c_int64, c_int, c_int, c_int]),
    "gpn_lml_backward_batched": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int,
                                         c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_void_p]),
    "gpn_lml_kinv_layout": (c_int, [c_int64, c_int, ctypes.POINTER(c_int64)]),
    "gpn_lml_kinv_batched_work_bytes": (c_int64, [c_int64, c_int, c_int]),
    "gpn_lml_kinv_batched": (c_int, [c_void_p, c_int, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p]),
    "gpn_lml_grad_batched": (c_int, [c_void_p, c_int, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int,
                                     c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p]),
    "gpn_predict_work_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "gpn_predict": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                            c_void_p, c_int64, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gpn_block_inverse_bytes": (c_int64, [c_int64]),
    "gpn_block_inverse": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p]),
    "gpn_trsm_right_lt_blocked": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_int64, c_int64, c_void_p, c_int64]),
    "gpn_predict_blocked": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int,
                                    c_void_p, c_int64, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "gpn_transpose": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64]),
    "gpn_copy_matrix": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int]),
    "gpn_row_sumsq": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p]),
    "gpn_fill_zero": (c_int, [c_void_p, c_void_p, c_int64]),
    "gpn_gemv_t_work_bytes": (c_int64, [c_int64, c_int64, c_int]),
    "gpn_gemv_t_acc": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p]),
    "gpn_refine_tile_count": (c_int64, [c_int64]),
    "gpn_refine_resid_part_work_bytes": (c_int64, [c_int, c_int64]),
    "gpn_refine_resid_part": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p,
                                      c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "gpn_refine_finish": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
}
class ExprTerm(ctypes.Structure):
    """include/gpnative.h gpn_expr_term: one leaf of a sum-of-products covariance expression."""
    _fields_ = [("type", c_int), ("kind", c_int), ("var_off", c_int), ("ls_off", c_int), ("nls", c_int), ("nvar", c_int)]


TERM_STATIONARY, TERM_LINEAR, TERM_CONSTANT, TERM_WHITE = 0, 1, 2, 3
EXPR_MAX_TERMS, EXPR_MAX_GROUPS = 16, 8
SIGNATURES.update({
    "gpn_lml_refine_dense": (c_int, [c_void_p, c_void_p, c_int64, c_double, c_int64, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "gpn_lml_refine_expr": (c_int, [c_void_p, ctypes.POINTER(ExprTerm), c_int, ctypes.POINTER(c_int), c_int, c_void_p, c_void_p, c_int64, c_int,
                                    c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "gpn_refine_resid_part_expr": (c_int, [c_void_p, ctypes.POINTER(ExprTerm), c_int, ctypes.POINTER(c_int), c_int, c_void_p, c_void_p, c_int64, c_int,
                                           c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]),
    "gpn_kernel_matrix_expr": (c_int, [c_void_p, ctypes.POINTER(ExprTerm), c_int, ctypes.POINTER(c_int), c_int, c_void_p, c_void_p,
                                       c_int64, c_void_p, c_int64, c_int, c_void_p, c_int, c_void_p, c_int64]),
    "gpn_kernel_expr_grad_work_bytes": (c_int64, [c_int64, c_int64, c_int, c_int]),
    "gpn_kernel_matrix_expr_batched": (c_int, [c_void_p, ctypes.POINTER(ExprTerm), c_int, ctypes.POINTER(c_int), c_int, c_int, c_void_p,
                                               c_int64, c_void_p, c_int64, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int64]),
    "gpn_kernel_expr_grad_batched": (c_int, [c_void_p, ctypes.POINTER(ExprTerm), c_int, ctypes.POINTER(c_int), c_int, c_int, c_void_p,
                                             c_int64, c_int, c_void_p, c_int64, c_int64, c_int, c_void_p, c_int64, c_int64, c_void_p,
                                             c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "gpn_kernel_expr_grad": (c_int, [c_void_p, ctypes.POINTER(ExprTerm), c_int, ctypes.POINTER(c_int), c_int, c_void_p, c_int,
                                     c_void_p, c_int64, c_void_p, c_int64, c_int, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int,
                                     c_void_p, c_void_p]),
})

# callback table of the distributed driver (include/gpnative.h gpn_dist_comm)
BCAST_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_int, c_void_p, c_int64, c_int, c_void_p)
ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, c_void_p, c_void_p, c_int64, c_void_p)


class DistComm(ctypes.Structure):
    _fields_ = [("ctx", c_void_p), ("bcast", BCAST_FN), ("allreduce", ALLREDUCE_FN), ("flags", c_int)]


SIGNATURES.update({
    "gpn_dist_work_bytes": (c_int64, [c_int, c_int, c_int, c_int64, c_int, c_int, c_int64]),
    "gpn_dist_lml_refine_work_bytes": (c_int64, [c_int, c_int, c_int, c_int64, c_int, c_int, c_int64]),
    "gpn_dist_grad_work_bytes": (c_int64, [c_int, c_int, c_int, c_int64, c_int, c_int, c_int64]),
    "gpn_dist_lml_grad": (c_int, [c_void_p, ctypes.POINTER(DistComm), c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int,
                                  c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p,
                                  c_void_p, c_void_p]),
    "gpn_dist_layout": (c_int, [c_int, c_int, c_int, c_int, c_int64, c_int, c_int, c_int64, ctypes.POINTER(c_int64), c_int]),
    "gpn_dist_predict_work_bytes": (c_int64, [c_int, c_int, c_int, c_int64, c_int, c_int, c_int64, c_int64, c_int]),
    "gpn_dist_predict": (c_int, [c_void_p, ctypes.POINTER(DistComm), c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int,
                                 c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_int,
                                 c_void_p, c_int64, c_void_p, c_void_p, c_void_p]),
    "gpn_dist_lml_forward": (c_int, [c_void_p, ctypes.POINTER(DistComm), c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int,
                                     c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    "gpn_dist_lml_refine": (c_int, [c_void_p, ctypes.POINTER(DistComm), c_int, c_int, c_int, c_int, c_void_p, c_int64, c_int,
                                    c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
})
# libgpnative_rccl.so: the RCCL adapter of that table (declared in the same header)
RCCL_LIB_PATH = os.environ.get("GPN_RCCL_LIB", os.path.join(os.path.dirname(LIB_PATH), "libgpnative_rccl.so"))   # (GPN_RCCL_LIB: the sanitizer leg)
RCCL_SIGNATURES = {
    "gpn_rccl_comm_create": (ctypes.POINTER(DistComm), [c_void_p, c_void_p, c_void_p]),
    "gpn_rccl_comm_destroy": (None, [ctypes.POINTER(DistComm)]),
    "gpn_mesh_plan": (c_int64, [c_int, c_int, c_int, c_int64, c_int, c_int64, ctypes.POINTER(c_int64), c_int64]),
}
DIST_FORCE_COLLECTIVES, DIST_MESH_EXCHANGE = 1, 2      # gpn_dist_comm.flags
_rccl_lib = None


def rccl_lib():
    """the RCCL adapter library (loads librccl); raises if it was not built."""
    global _rccl_lib
    if _rccl_lib is None:
        if not os.path.exists(RCCL_LIB_PATH):
            raise NativeError("libgpnative_rccl.so not found at %s -- run gptorch_amd/csrc/build.sh" % RCCL_LIB_PATH)
        handle = ctypes.CDLL(RCCL_LIB_PATH)
        for name, (res, args) in RCCL_SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype, fn.argtypes = res, args
        _rccl_lib = handle
    return _rccl_lib


# The launch profiler (include/gpnative.h, last section) is part of the public header since round 5.
PROFILE_SIGNATURES = {
    "gpn_profile_enable": (c_int, [c_int]),
    "gpn_profile_collect": (c_int, [ctypes.POINTER(c_double)]),
    "gpn_profile_collect_classes": (c_int, [ctypes.POINTER(c_double), c_int]),
}
SIGNATURES.update(PROFILE_SIGNATURES)
# not part of the public header: the A/B switches (kernel / driver variants per calling thread, CU-masked streams, the
# instrumented leaf) exist only in the tools' build libgpnative_dbg.so (same sources, -DGPN_DEBUG_SWITCHES): the shipped
# library carries no mutable debug state.
DEBUG_SIGNATURES = {
    "gpn_debug_set_gemm_variant": (c_int, [c_int]),
    "gpn_debug_set_potrf_variant": (c_int, [c_int]),
    "gpn_debug_set_outer_width": (c_int, [c_int, c_int]),
    "gpn_debug_set_extra_rows": (c_int, [c_int]),
    "gpn_debug_set_thin_tiles": (c_int, [c_int]),
    "gpn_debug_masked_stream": (c_int, [ctypes.POINTER(ctypes.c_uint32), c_int, ctypes.POINTER(c_void_p)]),
    "gpn_debug_set_backsub_persistent": (c_int, [c_int]),
    "gpn_debug_set_persistent": (c_int, [c_int, c_int]),
    "gpn_debug_persistent_trace": (c_int, [c_void_p]),
}
DEBUG_LIB_PATH = os.path.join(_HERE, "lib", "libgpnative_dbg.so")

_lib = None
_product_lib = None
_debug_lib = None


class NativeError(RuntimeError):
    pass


def _load(path, extra):
    handle = ctypes.CDLL(path)
    for name, (res, args) in list(SIGNATURES.items()) + list(extra.items()):
        fn = getattr(handle, name)
        fn.restype, fn.argtypes = res, args
    return handle


def lib():
    """Load (once) and return the ctypes handle; raises if the library is absent."""
    global _lib, _product_lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "libgpnative.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or gptorch_amd/csrc/build.sh). gptorch_amd has no CPU fallback." % LIB_PATH)
        extra = DEBUG_SIGNATURES if os.path.basename(LIB_PATH).startswith("libgpnative_dbg") else {}
        _product_lib = _lib = _load(LIB_PATH, extra)
    return _lib


class debug_library:
    """`with _native.debug_library() as lib:` -- inside the block every native call of this process goes through the tools'
    build libgpnative_dbg.so, whose gpn_debug_set_* switches select kernel / driver variants for the calling thread (A/B
    tools, forced-variant tests).  On exit the product library is back (and the switches of the debug build are reset)."""

    def __enter__(self):
        global _lib, _debug_lib
        lib()
        if _debug_lib is None:
            if not os.path.exists(DEBUG_LIB_PATH):
                raise NativeError("libgpnative_dbg.so not found at %s -- run gptorch_amd/csrc/build.sh" % DEBUG_LIB_PATH)
            _debug_lib = _load(DEBUG_LIB_PATH, DEBUG_SIGNATURES)
        _lib = _debug_lib
        return _debug_lib

    def __exit__(self, *exc):
        global _lib
        _debug_lib.gpn_debug_set_gemm_variant(0)
        _debug_lib.gpn_debug_set_potrf_variant(0)
        _lib = _product_lib
        return False


_debug_ctx = None


def debug_begin():
    """switch this process to the tools' build (see debug_library) and return its handle; debug_end() switches back."""
    global _debug_ctx
    if _debug_ctx is None:
        _debug_ctx = debug_library()
        return _debug_ctx.__enter__()
    return _debug_lib


def debug_end():
    global _debug_ctx
    if _debug_ctx is not None:
        _debug_ctx.__exit__(None, None, None)
        _debug_ctx = None


def check(status, what):
    if status != 0:
        detail = lib().gpn_last_hip_error().decode() if status == -100 else ""
        raise NativeError("%s failed with status %d %s" % (what, status, detail))
