"""ctypes binding of the C-ABI HIP library (include/lsqr_hip.h -> liblsqr_hip.so).

There is no fallback: if the shared library is missing or no HIP device is usable, every
operation raises.  Nothing here imports or calls the CPU oracle under oracle/."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblsqr_hip.so")

OK, EMPTY, ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_STATE = range(6)
PLANE, SPHERE, LINE, DENSE, US_SINGLE, US_POINTER, ABSOR, PIVOT, RAY, LINE2D, PHANTOM = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11
LS_ALGEBRAIC, LS_GEOMETRIC = 0, 1
LS_ANALYTIC, LS_ITERATIVE = 0, 1
KERNEL_IDS = {"sample": 0, "estimate": 1, "scan": 2, "mask": 3, "moments": 4, "solve": 5,
              "index": 6, "absmax": 7}


class LsqrError(RuntimeError):
    def __init__(self, status, msg):
        RuntimeError.__init__(self, "lsqr_hip status %d: %s" % (status, msg))
        self.status = status


class ModelCfg(C.Structure):
    _fields_ = [("model", C.c_int32), ("dim", C.c_int32), ("delta", C.c_double),
                ("ls_type", C.c_int32), ("reserved", C.c_int32), ("aux", C.c_double)]


class FitInfo(C.Structure):
    _fields_ = [("n_params", C.c_int32), ("lm_info", C.c_int32), ("lm_nfev", C.c_int32),
                ("reserved", C.c_int32), ("n_used", C.c_uint64), ("cost", C.c_double)]


class RansacInfo(C.Structure):
    _fields_ = [("fraction", C.c_double), ("iterations", C.c_uint64), ("evaluated", C.c_uint64),
                ("best_index", C.c_uint64), ("best_votes", C.c_uint32), ("n_params", C.c_int32),
                ("fit", FitInfo)]


_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_ctx = C.c_void_p

# name -> (restype, argtypes); must list every LSQR_API symbol of include/lsqr_hip.h
SIGNATURES = {
    "lsqr_version": (C.c_char_p, []),
    "lsqr_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "lsqr_status_string": (C.c_char_p, [C.c_int]),
    "lsqr_ctx_create": (C.c_int, [C.c_int, C.POINTER(_ctx)]),
    "lsqr_ctx_destroy": (None, [_ctx]),
    "lsqr_last_error": (C.c_char_p, [_ctx]),
    "lsqr_synchronize": (C.c_int, [_ctx]),
    "lsqr_min_subset": (C.c_int, [C.POINTER(ModelCfg)]),
    "lsqr_num_params": (C.c_int, [C.POINTER(ModelCfg)]),
    "lsqr_record_doubles": (C.c_int, [C.POINTER(ModelCfg)]),
    "lsqr_set_model": (C.c_int, [_ctx, C.POINTER(ModelCfg)]),
    "lsqr_upload": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_size_t]),
    "lsqr_attach": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_size_t]),
    "lsqr_count": (C.c_size_t, [_ctx]),
    "lsqr_hypotheses_from_subsets": (C.c_int, [_ctx, C.c_void_p, C.c_size_t]),
    "lsqr_hypotheses_sample": (C.c_int, [_ctx, C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p]),
    "lsqr_sample_subsets": (C.c_int, [C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint64, C.c_int, C.c_void_p]),
    "lsqr_agree_host": (C.c_int, [C.POINTER(ModelCfg), C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]),
    "lsqr_estimate_host": (C.c_int, [C.POINTER(ModelCfg), C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p,
                                     C.POINTER(C.c_int)]),
    "lsqr_scan": (C.c_int, [_ctx]),
    "lsqr_get_hypotheses": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lsqr_num_hypotheses": (C.c_size_t, [_ctx]),
    "lsqr_get_hypothesis": (C.c_int, [_ctx, C.c_size_t, C.c_void_p, C.c_void_p]),
    "lsqr_best": (C.c_int, [_ctx, _u64p]),
    "lsqr_mask": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p, _u64p]),
    "lsqr_mask_from_hypothesis": (C.c_int, [_ctx, C.c_size_t, C.c_void_p, _u64p]),
    "lsqr_set_mask": (C.c_int, [_ctx, C.c_void_p]),
    "lsqr_ls_fit": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.POINTER(FitInfo)]),
    "lsqr_moments_len": (C.c_int, [C.POINTER(ModelCfg), C.c_int]),
    "lsqr_moments": (C.c_int, [_ctx, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
                               C.c_void_p]),
    "lsqr_moments_dev": (C.c_int, [_ctx, C.c_int, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p,
                               C.c_void_p]),
    "lsqr_solve_moments": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.POINTER(FitInfo)]),
    "lsqr_winner_moments": (C.c_int, [_ctx, C.c_uint64, C.c_uint64, C.c_size_t, C.c_size_t,
                                      C.c_void_p, C.c_void_p, C.c_void_p, _u64p]),
    "lsqr_lm_begin": (C.c_int, [_ctx, C.c_void_p, C.c_void_p]),
    "lsqr_lm_step": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p,
                               C.POINTER(FitInfo)]),
    "lsqr_stats": (C.c_int, [_ctx, C.c_void_p, C.c_int, C.c_void_p]),
    "lsqr_residuals": (C.c_int, [_ctx, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]),
    "lsqr_set_stream": (C.c_int, [_ctx, C.c_void_p, C.c_int]),
    "lsqr_step_scan": (C.c_int, [_ctx, C.c_uint64, C.c_uint64, C.c_size_t, C.c_uint32, C.c_void_p]),
    "lsqr_step_winner": (C.c_int, [_ctx, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_size_t,
                                   C.c_void_p]),
    "lsqr_step_finish": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.POINTER(RansacInfo)]),
    "lsqr_step_finish_enqueue": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.c_int]),
    "lsqr_step_finish_wait": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(RansacInfo)]),
    "lsqr_ransac": (C.c_int, [_ctx, C.c_double, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p,
                              C.c_void_p, C.POINTER(RansacInfo)]),
    "lsqr_batch_fit": (C.c_int, [_ctx, C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p, C.c_void_p,
                                 C.POINTER(RansacInfo)]),
    "lsqr_batch_fit_enqueue": (C.c_int, [_ctx, C.c_uint64, C.c_uint64, C.c_size_t, C.c_int]),
    "lsqr_batch_fit_wait": (C.c_int, [_ctx, C.c_int, C.c_void_p, C.POINTER(RansacInfo)]),
    "lsqr_multi_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "lsqr_multi_destroy": (None, [C.c_void_p]),
    "lsqr_multi_size": (C.c_int, [C.c_void_p]),
    "lsqr_multi_transport": (C.c_char_p, [C.c_void_p, C.POINTER(C.c_double)]),
    "lsqr_multi_ctx": (C.c_void_p, [C.c_void_p, C.c_int]),
    "lsqr_multi_last_error": (C.c_char_p, [C.c_void_p]),
    "lsqr_multi_set_model": (C.c_int, [C.c_void_p, C.POINTER(ModelCfg)]),
    "lsqr_multi_upload": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]),
    "lsqr_multi_batch_fit": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_size_t, C.c_void_p, C.c_void_p,
                                       C.POINTER(RansacInfo)]),
    "lsqr_multi_ransac": (C.c_int, [C.c_void_p, C.c_double, C.c_uint64, C.c_void_p, C.c_void_p,
                                    C.POINTER(RansacInfo)]),
    "lsqr_ransac_exhaustive": (C.c_int, [_ctx, C.c_void_p, C.c_void_p, C.POINTER(RansacInfo)]),
    "lsqr_replay_init": (C.c_int, [C.c_size_t, C.c_int, C.c_double, _u64p]),
    "lsqr_replay": (C.c_size_t, [C.c_size_t, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                 C.c_void_p, C.c_size_t, C.c_uint64, C.c_void_p, _u64p]),
    "lsqr_dedup_create": (C.c_void_p, [C.c_int]),
    "lsqr_dedup_destroy": (None, [C.c_void_p]),
    "lsqr_set_option": (C.c_int, [_ctx, C.c_char_p, C.c_int]),
    "lsqr_index_info": (C.c_int, [_ctx, _u64p]),
    "lsqr_scan_workload": (C.c_int, [_ctx, C.c_void_p, _u64p]),
    "lsqr_scan_work": (C.c_int, [_ctx, _u64p]),
    "lsqr_lm_persist_info": (C.c_int, [_ctx, _u64p, _u64p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "lsqr_profile_enable": (C.c_int, [_ctx, C.c_int]),
    "lsqr_profile_get": (C.c_int, [_ctx, C.c_int, _u64p, _dp]),
    "lsqr_profile_reset": (C.c_int, [_ctx]),
}

_lib = None


def load():
    """Load liblsqr_hip.so; raises if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LsqrError(ERR_NO_DEVICE, "%s not built: run `python -c 'import __graft_entry__ as"
                            " g; g.build()'` or make -C lsqrrecipes_amd/csrc" % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def device_count():
    n = C.c_int(0)
    load().lsqr_device_count(C.byref(n))
    return n.value


def ptr(a):
    return None if a is None else a.ctypes.data
