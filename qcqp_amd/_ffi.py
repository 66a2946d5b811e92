"""Thin binding of the C ABI declared in include/qcqp_mi.h.

The north star asks for cffi; cffi is not installed in the image, so the binding uses the
standard library's ctypes in the same ABI-mode style (dlopen + declared prototypes).  There is
no Python/NumPy fallback: if the shared library is missing this module raises.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIBPATH = os.path.join(HERE, 'libqcqp_mi.so')

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int64)
c_bp = C.POINTER(C.c_uint8)

# (name, restype, argtypes) -- every symbol of include/qcqp_mi.h
PROTOTYPES = [
    ('qcqpmi_abi_version', C.c_int, []),
    ('qcqpmi_device_count', C.c_int, []),
    ('qcqpmi_last_error', C.c_char_p, [C.c_void_p]),
    ('qcqpmi_ctx_create', C.c_int, [C.POINTER(C.c_void_p), C.c_int64, C.c_int64, C.c_int]),
    ('qcqpmi_ctx_destroy', None, [C.c_void_p]),
    ('qcqpmi_set_quad', C.c_int, [C.c_void_p, C.c_int64, C.c_int, c_dp, c_ip, c_ip, C.c_int64, c_dp,
                                  C.c_double, C.c_int]),
    ('qcqpmi_set_quad_generated', C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int]),
    ('qcqpmi_weighted_matrix', C.c_int, [C.c_void_p, c_dp, c_dp]),
    ('qcqpmi_get_linear', C.c_int, [C.c_void_p, C.c_int64, c_dp, c_dp, C.POINTER(C.c_int)]),
    ('qcqpmi_pop_eval_parts', C.c_int, [C.c_void_p, c_dp, c_dp]),
    ('qcqpmi_pop_weighted_product', C.c_int, [C.c_void_p, c_dp, c_dp]),
    ('qcqpmi_sdr_solve_unitdiag', C.c_int, [C.c_void_p, c_dp, C.c_int64, c_dp, C.c_int, C.c_double, c_dp, C.POINTER(C.c_int)]),
    ('qcqpmi_finalize', C.c_int, [C.c_void_p]),
    ('qcqpmi_is_separable', C.c_int, [C.c_void_p]),
    ('qcqpmi_pop_upload', C.c_int, [C.c_void_p, c_dp, C.c_int64]),
    ('qcqpmi_pop_download', C.c_int, [C.c_void_p, c_dp, C.c_int64]),
    ('qcqpmi_pop_size', C.c_int64, [C.c_void_p]),
    ('qcqpmi_pop_randn', C.c_int, [C.c_void_p, C.c_int64, C.c_uint64, C.c_uint64]),
    ('qcqpmi_pop_sdr_sample', C.c_int, [C.c_void_p, c_dp, c_dp, C.c_int64, C.c_uint64, C.c_uint64, c_dp]),
    ('qcqpmi_pop_eval', C.c_int, [C.c_void_p, c_dp, c_dp, c_dp]),
    ('qcqpmi_sdr_sample_eval', C.c_int, [C.c_void_p, c_dp, c_dp, C.c_int64, C.c_uint64, C.c_uint64, c_dp, c_dp, c_dp]),
    ('qcqpmi_eval_batch', C.c_int, [C.c_void_p, c_dp, C.c_int64, c_dp, c_dp, c_dp]),
    ('qcqpmi_cd_run', C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_uint64,
                                C.c_uint64, c_ip, c_ip, c_ip, c_ip, c_bp, c_dp, c_dp]),
    ('qcqpmi_cd_run_stage', C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_uint64,
                                      C.c_uint64, c_ip, c_ip, c_ip, c_ip, c_bp, c_dp, c_dp]),
    ('qcqpmi_cd_status', C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ('qcqpmi_feasible_intervals_batch', C.c_int, [C.c_int, C.c_int64, c_dp, C.POINTER(C.c_int), c_dp]),
    ('qcqpmi_onevar_coeffs', C.c_int, [C.c_void_p, c_ip, c_dp]),
    ('qcqpmi_onevar_qcqp_batch', C.c_int, [C.c_int, C.c_int64, c_dp, c_dp, C.POINTER(C.c_int), c_dp, C.c_uint64, c_dp,
                                           C.POINTER(C.c_int), c_dp, C.POINTER(C.c_int)]),
    ('qcqpmi_admm_set_eig', C.c_int, [C.c_void_p, c_dp, c_dp]),
    ('qcqpmi_admm_setup', C.c_int, [C.c_void_p]),
    ('qcqpmi_admm_set_basis', C.c_int, [C.c_void_p, C.c_int64, c_dp, c_dp, c_dp]),
    ('qcqpmi_admm_apply_constraints', C.c_int, [C.c_void_p, C.c_int, c_dp, C.c_int, c_dp]),
    ('qcqpmi_admm_onecons', C.c_int, [C.c_void_p, C.c_int64, c_dp]),
    ('qcqpmi_admm_set_bracket', C.c_int, [C.c_void_p, c_dp, c_dp]),
    ('qcqpmi_admm_zsolver_device', C.c_int, [C.c_void_p, C.c_double, C.c_int64, c_dp, c_ip]),
    ('qcqpmi_p0_lambda_min', C.c_int, [C.c_void_p, C.c_int64, C.c_double, c_dp, c_ip]),
    ('qcqpmi_admm_run', C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_double, c_dp,
                                  c_ip, c_ip, c_dp, c_dp]),
    ('qcqpmi_admm_fused', C.c_int, [C.c_void_p, C.c_int]),
    ('qcqpmi_admm_unit_bases', C.c_int, [C.c_void_p, C.c_int]),
    ('qcqpmi_last_admm_kernel', C.c_char_p, [C.c_void_p, C.POINTER(C.c_int)]),
    ('qcqpmi_select_best', C.c_int, [C.c_void_p, C.c_double, c_ip, c_dp, c_dp, c_dp]),
    ('qcqpmi_last_kernel_ms', C.c_int, [C.c_void_p, C.c_int, c_dp]),
    ('qcqpmi_last_cd_kernel', C.c_char_p, [C.c_void_p]),
    ('qcqpmi_cd_reference_order', C.c_int, [C.c_void_p, C.c_int]),
    ('qcqpmi_cd_dense_block_step', C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_double, C.c_uint64, C.c_uint64, c_dp]),
    ('qcqpmi_cd_stream_reserve', C.c_int, [C.c_void_p, C.c_int64, C.c_int64]),
    ('qcqpmi_cd_stream_run', C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int64, C.c_double, C.c_double, C.c_uint64, C.c_uint64,
                                       C.c_uint64, C.c_uint64, C.c_double, c_ip, c_ip, c_ip, c_ip, c_bp, c_dp, c_dp, c_ip, c_dp, c_dp, c_dp]),
    ('qcqpmi_cd_queue', C.c_int, [C.c_void_p, C.c_int]),
    ('qcqpmi_cd_life_version', C.c_int, [C.c_void_p, C.c_int]),
    ('qcqpmi_cd_set_objective_factor', C.c_int, [C.c_void_p, c_dp, C.c_int64]),
    ('qcqpmi_sync', C.c_int, [C.c_void_p]),
    ('qcqpmi_debug_profile', C.c_int, [C.c_void_p, C.c_int, c_ip]),
    ('qcqpmi_debug_admm_profile', C.c_int, [C.c_void_p, c_ip]),
    ('qcqpmi_debug_dense_profile', C.c_int, [C.c_void_p, c_ip]),
    ('qcqpmi_debug_life_profile', C.c_int, [C.c_void_p, c_ip]),
    ('qcqpmi_dense_chain_mode', C.c_int, [C.c_void_p, C.c_int]),
    ('qcqpmi_dense_chain_geometry', C.c_int, [C.c_int64, C.POINTER(C.c_int)]),
    ('qcqpmi_debug_trace', C.c_int, [C.c_void_p, c_ip, C.c_int]),
    ('qcqpmi_comm_unique_id', C.c_int, [c_bp]),
    ('qcqpmi_comm_init', C.c_int, [C.c_void_p, C.c_int, C.c_int, c_bp]),
    ('qcqpmi_comm_select_best', C.c_int, [C.c_void_p, C.c_double, C.c_int64, c_ip, c_dp, c_dp, c_dp]),
    ('qcqpmi_comm_barrier', C.c_int, [C.c_void_p]),
    ('qcqpmi_comm_allreduce', C.c_int, [C.c_void_p, c_dp, C.c_int64, C.c_int]),
    ('qcqpmi_comm_allgather', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
]

_lib = None


def lib():
    """Loads libqcqp_mi.so (built in-tree by qcqp_amd._build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBPATH):
            raise ImportError(
                'libqcqp_mi.so is not built (%s missing). Run `python -m qcqp_amd._build` or '
                '__graft_entry__.build(); the engine has no CPU fallback.' % LIBPATH)
        L = C.CDLL(os.environ.get('QCQPMI_LIB') or LIBPATH)     # QCQPMI_LIB: an alternative build (e.g. with stage timers)
        for name, res, args in PROTOTYPES:
            fn = getattr(L, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
