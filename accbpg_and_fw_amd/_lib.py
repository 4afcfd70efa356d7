"""ctypes binding of libaccbpg_hip.so (C-ABI: include/accbpg_hip.h).

There is no CPU fallback: if the shared library is missing or a device tensor is
required and absent, the call raises.  Error codes are mapped to the exception
types the reference raises in the same situations (accbpg/functions.py:44-50,
233-234, 251-252, 269-270, 340).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (ACCBPG_HIP_LIB: development builds of the same library, e.g. the experiment of DESIGN.md section 4)
LIB_PATH = os.environ.get("ACCBPG_HIP_LIB") or os.path.join(_HERE, "lib", "libaccbpg_hip.so")

OK, ERR_ASSERT, ERR_NOT_PD, ERR_HIP, ERR_ARG = 0, 1, 2, 3, 4

_lib = None


class FwProbe(C.Structure):
    _fields_ = [("i", C.c_int64), ("j", C.c_int64), ("w_i", C.c_double), ("w_j", C.c_double),
                ("x_j", C.c_double), ("logdet_H", C.c_double), ("q_prev", C.c_double)]


_P = C.c_void_p
_SIGS = {
    "accbpg_abi_version": (C.c_int, []),
    "accbpg_last_error": (C.c_char_p, []),
    "accbpg_dopt_create": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _P, C.POINTER(_P), C.c_int]),
    "accbpg_dopt_destroy": (C.c_int, [_P]),
    "accbpg_dopt_set_stream": (C.c_int, [_P, _P]),
    "accbpg_dopt_func_grad": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_double), _P]),
    "accbpg_dopt_func_grad_begin": (C.c_int, [_P, _P, C.c_int, _P]),
    "accbpg_dopt_func_grad_end": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "accbpg_dopt_eval_gap_ms": (C.c_int, [_P, _P, C.POINTER(C.c_double)]),
    "accbpg_dopt_gram": (C.c_int, [_P, _P, _P]),
    "accbpg_dopt_factor": (C.c_int, [_P, _P, C.POINTER(C.c_double)]),
    "accbpg_dopt_grad": (C.c_int, [_P, _P]),
    "accbpg_tri_pack": (C.c_int, [_P, C.c_int64, _P, _P]),
    "accbpg_tri_unpack": (C.c_int, [_P, C.c_int64, _P, _P]),
    "accbpg_vec_count_bad": (C.c_int, [_P, C.c_int64, _P, _P]),
    "accbpg_dopt_shard_bounds": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "accbpg_shard_unique_id": (C.c_int, [_P]),
    "accbpg_dopt_shard_create": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, _P, _P, C.POINTER(_P)]),
    "accbpg_dopt_shard_func_grad": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_double), _P]),
    "accbpg_dopt_shard_destroy": (C.c_int, [_P]),
    "accbpg_debug_shard_pad": (C.c_int, [_P, C.c_int64]),
    "accbpg_dopt_gram_lincomb": (C.c_int, [_P, C.c_double, _P, C.c_double, _P, _P]),
    "accbpg_dopt_eval_gram": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_double), _P]),
    "accbpg_dopt_batch_create": (C.c_int, [C.POINTER(_P), C.c_int, C.c_int64, C.c_int64, C.c_int64, _P, C.POINTER(_P)]),
    "accbpg_dopt_batch_destroy": (C.c_int, [_P]),
    "accbpg_dopt_batch_set_stream": (C.c_int, [_P, _P]),
    "accbpg_dopt_batch_size": (C.c_int, [_P]),
    "accbpg_dopt_batch_is_fused": (C.c_int, [_P]),
    "accbpg_dopt_batch_chunk": (C.c_int, [_P]),
    "accbpg_dopt_batch_instance": (_P, [_P, C.c_int]),
    "accbpg_dopt_batch_func_grad": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_double), _P,
                                              C.c_int64, C.POINTER(C.c_int)]),
    "accbpg_dopt_batch_func_grad_begin": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_int), C.c_int, _P, C.c_int64]),
    "accbpg_dopt_batch_func_grad_end": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "accbpg_dopt_batch_burg_simplex_div_prox": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_double), C.c_double, _P,
                                                          C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "accbpg_dopt_batch_ls_terms": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.POINTER(C.c_int), C.POINTER(C.c_double),
                                             C.POINTER(C.c_int)]),
    "accbpg_dopt_batch_axpby": (C.c_int, [_P, C.POINTER(C.c_double), _P, C.POINTER(C.c_double), _P, C.c_int64,
                                          C.POINTER(C.c_int), _P]),
    "accbpg_vec_workspace_doubles": (C.c_int64, [C.c_int64]),
    "accbpg_burg_simplex_div_prox": (C.c_int, [_P, _P, C.c_double, C.c_double, C.c_int64, _P, _P,
                                               C.POINTER(C.c_int), _P]),
    "accbpg_burg_divergence": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_double), _P, _P]),
    "accbpg_ls_terms": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.POINTER(C.c_double), _P, _P]),
    "accbpg_vec_axpby": (C.c_int, [C.c_double, _P, C.c_double, _P, C.c_int64, _P, _P]),
    "accbpg_vec_dot_diff": (C.c_int, [_P, _P, _P, C.c_int64, C.POINTER(C.c_double), _P, _P]),
    "accbpg_vec_min_sum": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_double), _P, _P]),
    "accbpg_vec_div_scalar": (C.c_int, [_P, C.c_double, C.c_int64, _P, _P]),
    "accbpg_vec_vertex": (C.c_int, [C.c_int64, C.c_double, C.c_double, C.c_int64, _P, _P]),
    "accbpg_vec_argminmax": (C.c_int, [_P, C.c_int64, C.POINTER(C.c_int64), C.POINTER(C.c_double), _P, _P]),
    "accbpg_dopt_vt_times": (C.c_int, [_P, _P, _P]),
    "accbpg_dopt_get_column": (C.c_int, [_P, C.c_int64, _P]),
    "accbpg_fw_init": (C.c_int, [_P, _P, C.POINTER(C.c_double)]),
    "accbpg_fw_probe_step": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(FwProbe)]),
    "accbpg_fw_logdet_flush": (C.c_int, [_P, C.POINTER(C.c_double)]),
    "accbpg_fw_logdet_ring": (C.c_int, [_P, C.c_int, C.c_int]),
    "accbpg_fw_logdet_pending": (C.c_int, [_P]),
    "accbpg_fw_update": (C.c_int, [_P, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double]),
    "accbpg_fw_get_state": (C.c_int, [_P, _P, _P, _P]),
    "accbpg_poisson_create": (C.c_int, [_P, C.c_int64, C.c_int64, C.c_int64, _P, _P, C.POINTER(_P)]),
    "accbpg_poisson_destroy": (C.c_int, [_P]),
    "accbpg_poisson_set_stream": (C.c_int, [_P, _P]),
    "accbpg_poisson_func_grad": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_double), _P]),
    "accbpg_poisson_get_ax": (C.c_int, [_P, _P]),
    "accbpg_burg_reg_div_prox": (C.c_int, [C.c_int, _P, _P, C.c_double, C.c_double, C.c_int64, _P, _P]),
    "accbpg_vec_dot": (C.c_int, [_P, _P, C.c_int64, C.POINTER(C.c_double), _P, _P]),
    "accbpg_dopt_profile_enable": (C.c_int, [_P, C.c_int]),
    "accbpg_dopt_profile_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "accbpg_dopt_profile_reset": (C.c_int, [_P]),
    "accbpg_debug_pipe_probe": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_double), _P]),
    "accbpg_mfma_f64_peak": (C.c_int, [C.c_int, C.POINTER(C.c_double), _P]),
    "accbpg_dopt_factor_in_small_launches": (C.c_int, [_P, C.c_int]),
    "accbpg_debug_chol_variant": (C.c_int, [_P, C.c_int]),
    "accbpg_debug_plan_flags": (C.c_int, [C.c_int]),
    "accbpg_debug_chol_trace": (C.c_int, [_P, _P, C.c_int, C.POINTER(C.c_int64)]),
    "accbpg_debug_gram_variant": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "accbpg_test_gemm": (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P, C.c_int64, C.c_int64, C.c_int64,
                                   C.c_int64, C.c_int, C.c_double, C.c_double, C.c_int, _P]),
}

EXPORTS = tuple(sorted(_SIGS))


def load():
    """Load the shared library once; raise loudly if it was not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "accbpg_and_fw_amd: %s is missing -- build it with "
            "`make -C accbpg_and_fw_amd/csrc` or `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error():
    msg = load().accbpg_last_error()
    return msg.decode() if msg else ""


def check(rc, what, assert_msg=None):
    """Map a C-ABI status to the reference's exception types."""
    if rc == OK:
        return
    if rc == ERR_ASSERT:
        raise AssertionError(assert_msg or last_error() or what)
    if rc == ERR_NOT_PD:
        raise ValueError("HXHT is singular or not positive definite")   # functions.py:50
    if rc == ERR_ARG:
        raise ValueError("%s: bad argument (%s)" % (what, last_error()))
    raise RuntimeError("%s: HIP error: %s" % (what, last_error()))
