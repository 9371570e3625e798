"""Loader of the HIP shared library (C ABI of include/remixt_amd.h).

There is no fallback: if the library is missing or cannot be loaded the import
of the product path fails with a clear error.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (RMX_LIB_PATH: another build of the same library -- A/B measurements of two builds on one box, tools/ab_library.sh; never a fallback)
LIB_PATH = os.environ.get("RMX_LIB_PATH") or os.path.join(HERE, "libremixt_hip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int64)
_i32p = C.POINTER(C.c_int32)


class RmxProblem(C.Structure):
    """struct rmx_problem (include/remixt_amd.h)."""
    _fields_ = [
        ("num_clones", C.c_int32), ("num_segments", C.c_int32), ("num_breakpoints", C.c_int32),
        ("num_cn_states", C.c_int32), ("num_brk_states", C.c_int32), ("num_classes", C.c_int32),
        ("normal_contamination", C.c_int32), ("reserved", C.c_int32),
        ("cn_classes", _ip), ("seg_class", _i32p), ("brk_states", _ip),
        ("l", _dp), ("x", _dp), ("y", _dp),
        ("is_telomere", _ip), ("breakpoint_idx", _ip), ("breakpoint_orient", _ip),
        ("transition_penalty", C.c_double),
    ]


# name -> (restype, argtypes); every symbol include/remixt_amd.h declares
SYMBOLS = {
    "rmx_batch_create": (C.c_int, [C.POINTER(RmxProblem), C.c_int32, _dp, _dp, C.c_int32, C.POINTER(C.c_void_p)]),
    "rmx_batch_destroy": (C.c_int, [C.c_void_p]),
    "rmx_compress_cn_states": (C.c_int, [_ip, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _i32p, _ip, _i32p]),
    "rmx_last_error": (C.c_char_p, []),
    "rmx_last_error_restarts": (C.c_int, [_i32p, C.c_int32]),
    "rmx_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "rmx_synchronize": (C.c_int, [C.c_void_p]),
    "rmx_info": (C.c_int, [C.c_void_p, C.c_int32, _ip]),
    "rmx_set_default_option": (C.c_int, [C.c_int32, C.c_int32]),
    "rmx_set_option": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rmx_get_option": (C.c_int, [C.c_void_p, C.c_int32, _i32p]),
    "rmx_set_param": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_double]),
    "rmx_get_param": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp]),
    "rmx_set_transition_model": (C.c_int, [C.c_void_p, C.c_int32]),
    "rmx_set_array": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "rmx_get_array": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "rmx_fetch_indicators": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(_dp), C.POINTER(_dp)]),
    "rmx_calculate_log_transmat": (C.c_int, [C.c_void_p, C.c_int32, _dp]),
    "rmx_weighted_search": (C.c_int, [_dp, C.c_int64, _dp, C.c_int32, _ip, _ip]),
    "rmx_weighted_sample_round": (C.c_int, [_dp, C.c_int64, C.c_int64, C.c_double, _dp, C.c_int32, _ip, _i32p, C.c_int32, _ip]),
    "rmx_get_state_table": (C.c_int, [C.c_void_p, C.c_int32, _ip]),
    "rmx_update_framelogprob": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rmx_update_p_cn": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rmx_update_p_breakpoint": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rmx_update_p_outlier_total": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rmx_update_p_outlier_allele": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rmx_update_p_allele_swap": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rmx_variational_update": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32]),
    "rmx_calculate_elbo": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp]),
    "rmx_calculate_elbo_begin": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "rmx_calculate_elbo_end": (C.c_int, [C.c_void_p, _dp]),
    "rmx_calculate_variational_energy": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp]),
    "rmx_calculate_variational_entropy": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp]),
    "rmx_expected_log_likelihood": (C.c_int, [C.c_void_p, C.c_int32, _ip, _dp, _dp]),
    "rmx_set_sample": (C.c_int, [C.c_void_p, C.c_int32, _ip]),
    "rmx_expected_ll_param_grid": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp, C.c_int32, _dp]),
    "rmx_expected_ll_batch": (C.c_int, [C.c_void_p, C.c_int32, _i32p, C.c_int32, _dp, _dp]),
    "rmx_param_search": (C.c_int, [C.c_void_p, C.c_int32, _i32p, C.c_int32, C.c_double, C.c_double, _dp, C.c_int32, _dp]),
    "rmx_set_sample_slot": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _ip]),
    "rmx_set_sample_lists": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _i32p, _i32p, _i32p]),
    "rmx_param_search_multi": (C.c_int, [C.c_void_p, C.c_int32, _i32p, C.c_int32, _i32p, _dp, _dp, _dp, C.c_int32, _dp, _dp]),
    "rmx_expected_ll_h_batch": (C.c_int, [C.c_void_p, C.c_int32, _i32p, _dp, _dp]),
    "rmx_expected_ll_full": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp]),
    "rmx_expected_ll_full_trial": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp]),
    "rmx_expected_ll_components": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "rmx_trial_rollback": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _dp]),
    "rmx_log_likelihood_total": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "rmx_log_likelihood_allele": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "rmx_cell_quantity": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _dp]),
    "rmx_infer_cn": (C.c_int, [C.c_void_p, C.c_int32, _ip, _dp]),
    "rmx_infer_cn_batch": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, _ip, _dp]),
    "rmx_sum_product": (C.c_int, [_dp, _dp, _dp, _dp, C.c_int32, C.c_int32, C.c_int32]),
    "rmx_max_product": (C.c_int, [_dp, _dp, _ip, _dp, C.c_int32, C.c_int32, C.c_int32]),
    "rmx_timer_start": (C.c_int, [C.c_void_p]),
    "rmx_timer_stop": (C.c_int, [C.c_void_p, _dp]),
    "rmx_profile_enable": (C.c_int, [C.c_void_p, C.c_int32]),
    "rmx_profile_get": (C.c_int, [C.c_void_p, C.c_int32, _dp, _ip]),
    "rmx_profile_reset": (C.c_int, [C.c_void_p]),
    "rmx_kernel_name": (C.c_char_p, [C.c_int32]),
    "rmx_num_kernels": (C.c_int, []),
}

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "remixt_amd: %s not found. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    # One HIP runtime per process: PyTorch wheels bundle their own libamdhip64 with the same SONAME as
    # /opt/rocm's.  If this library pulled in the system copy first and torch were imported later
    # (bench.py, torch.distributed), the process would mix two runtimes and the second one sees no
    # device.  Importing torch first makes the dynamic linker resolve our dependency to the copy
    # torch already loaded.  (torch is used for nothing else here.)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib
