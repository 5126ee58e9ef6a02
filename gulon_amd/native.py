"""ctypes binding of libgulon_hip.so (include/gulon_hip.h).

This is the ONLY way the Python host layer reaches the hot path: there is no CPU
fallback.  If the shared library is missing, importing this module's `lib()`
raises -- loudly -- instead of degrading to anything slower.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GULON_HIP_LIB: another build of the same library (A/B measurements of kernel changes on one GPU box)
LIB_PATH = os.environ.get("GULON_HIP_LIB") or os.path.join(_HERE, "lib", "libgulon_hip.so")

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_ILLEGAL_STATE = -2
ERR_UNSUPPORTED = -3
ERR_DEVICE = -4
ERR_OOM = -5

FLAG_BOUNDARY_TIE = 1
FLAG_INTERIOR_TIE = 2
FLAG_EXACT_REPLAY = 4
FLAG_NONFINITE = 8
REPLAY_MAX_FLAGGED = 16
REPLAY_POOL = 2048
MAX_K = 63
MAX_K_PEELED = 8191


class GulonDeviceError(RuntimeError):
    """HIP runtime failure (maps to RuntimeException in the JNI glue)."""


class KMeansReport(C.Structure):
    """KMeans.ProgressReport (KMeans.scala:119-127)."""
    _fields_ = [("num_iterations", C.c_int32), ("converged", C.c_int32),
                ("step_count", C.c_int32), ("step_mean", C.c_float), ("step_s", C.c_float)]


class KMeansTraceTotals(C.Structure):
    """gulon_kmeans_trace_totals: stage times of the training loop (bench.py's C3 record)."""
    _fields_ = [("iterations", C.c_int32), ("update_ms", C.c_double), ("assign_ms", C.c_double),
                ("recheck_ms", C.c_double), ("converge_ms", C.c_double), ("mfma_flops", C.c_double),
                ("update_bytes", C.c_double), ("rows_rechecked", C.c_double), ("rows_total", C.c_double)]


_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
_vp = C.c_void_p
_i32 = C.c_int32

# name -> (restype, argtypes).  Must list every symbol include/gulon_hip.h declares
# (tests/test_abi.py checks the header against this table and the .so).
SIGNATURES = {
    "gulon_last_error": (C.c_char_p, []),
    "gulon_abi_version": (_i32, []),
    "gulon_device_count": (_i32, [C.POINTER(_i32)]),
    "gulon_set_device": (_i32, [_i32]),
    "gulon_device_synchronize": (_i32, []),
    "gulon_dev_malloc": (_i32, [C.POINTER(_vp), C.c_size_t]),
    "gulon_dev_free": (_i32, [_vp]),
    "gulon_memcpy_h2d": (_i32, [_vp, _vp, C.c_size_t]),
    "gulon_memcpy_d2h": (_i32, [_vp, _vp, C.c_size_t]),
    "gulon_subvectors": (_i32, [_i32, _i32, _i32p, _i32p]),
    "gulon_coder_width": (_i32, [_i32, C.POINTER(_i32)]),
    "gulon_coder_bytes": (_i32, [_i32, _i32, C.POINTER(_i32)]),
    "gulon_coder_build": (_i32, [_i32, _i32p, _i32, _u8p]),
    "gulon_coder_unpack": (_i32, [_i32, _u8p, _i32, _i32p]),
    "gulon_dataset_create": (_i32, [_f32p, _i32, _i32, C.POINTER(_vp)]),
    "gulon_dataset_create_synth": (_i32, [_i32, _i32, _i32, C.c_uint64, _i32, C.POINTER(_vp)]),
    "gulon_dataset_destroy": (_i32, [_vp]),
    "gulon_dataset_shape": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "gulon_dataset_device_ptr": (_i32, [_vp, C.POINTER(_vp)]),
    "gulon_dataset_get_rows": (_i32, [_vp, _i32p, _i32, _f32p]),
    "gulon_kmeans_init": (_i32, [_vp, _i32, _i32, _i32, _i32, _f32p, _vp]),
    "gulon_kmeans_assign": (_i32, [_vp, _i32, _i32, _f32p, _i32, _i32, _i32p]),
    "gulon_kmeans_update": (_i32, [_vp, _i32, _i32, _i32, _i32p, _f32p]),
    "gulon_kmeans_iterate": (_i32, [_vp, _i32, _i32, _f32p, _i32, _i32, _f32p]),
    "gulon_kmeans_train": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _f32p, C.POINTER(KMeansReport), _i32,
                                  C.POINTER(_i32)]),
    "gulon_kmeans_trace": (_i32, [_i32]),
    "gulon_kmeans_trace_read": (_i32, [C.POINTER(KMeansTraceTotals)]),
    "gulon_pq_train": (_i32, [_vp, _i32, _i32, _i32, _f32p, C.POINTER(KMeansReport), _i32, _vp]),
    "gulon_pq_encode": (_i32, [_vp, _i32, _i32, _f32p, _u8p]),
    "gulon_pq_train_range": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _f32p, C.POINTER(KMeansReport), _i32, _vp]),
    "gulon_pq_encode_range": (_i32, [_vp, _i32, _i32, _f32p, _i32, _i32, _u8p]),
    "gulon_prepare_query": (_i32, [_f32p, _i32, _i32, _i32, _f32p, _i32, _f32p]),
    "gulon_index_create": (_i32, [_u8p, _i32, _i32, _i32, _i32, _f32p, _i32, C.POINTER(_vp)]),
    "gulon_index_destroy": (_i32, [_vp]),
    "gulon_index_context_create": (_i32, [_vp, C.POINTER(_vp)]),
    "gulon_sharded_index_create": (_i32, [_u8p, _i32, _i32, _i32, _i32, _f32p, _i32p, _i32, C.POINTER(_vp)]),
    "gulon_sharded_index_destroy": (_i32, [_vp]),
    "gulon_sharded_index_batch_query": (_i32, [_vp, _f32p, _i32, _i32, _i32p, _f32p, _i32p, _i32p]),
    "gulon_sharded_index_info": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32),
                                        C.POINTER(_i32)]),
    "gulon_index_batch_query": (_i32, [_vp, _f32p, _i32, _i32, _i32, _i32, _i32p, _f32p, _i32p, _i32p]),
    "gulon_index_batch_query_dev": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gulon_index_scan_partial_dev": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp]),
    "gulon_index_scan_bounds_dev": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp]),
    "gulon_index_scan_partial_bounded_dev": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _vp, _vp, _vp]),
    "gulon_topk_merge_dev": (_i32, [_vp, _vp, _i32, C.c_int64, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gulon_index_profile": (_i32, [_vp, _i32]),
    "gulon_index_profile_read": (_i32, [_vp, C.POINTER(C.c_double), C.POINTER(_i32)]),
    "gulon_dataset_group_residuals": (_i32, [_vp, _i32p, _i32p, _f32p, _i32, C.POINTER(_vp)]),
    "gulon_grouped_index_create": (_i32, [_u8p, _i32, _i32, _i32, _i32, _f32p, _f32p, _i32p, _i32, C.POINTER(_vp)]),
    "gulon_grouped_index_destroy": (_i32, [_vp]),
    "gulon_grouped_index_batch_query": (_i32, [_vp, _f32p, _i32, _i32, _i32, _i32, _i32p, _f32p, _i32p]),
    "gulon_grouped_index_batch_query_dev": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "gulon_replay_pack_words": (C.c_int64, [_i32, _i32]),
    "gulon_index_replay_collect_dev": (_i32, [_vp, _vp, _i32, _i32, _i32, _i32, _vp, _i32, _i32, _i32, _vp, _vp]),
    "gulon_replay_apply_dev": (_i32, [_vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gulon_index_profile_read_ex": (_i32, [_vp, C.POINTER(C.c_double), C.POINTER(_i32), C.POINTER(C.c_int64)]),
    "gulon_index_filter_stats": (_i32, [_vp, C.POINTER(_i32), C.POINTER(_i32)]),
    "gulon_index_tuning": (_i32, [_vp, C.c_char_p, _i32]),
    "gulon_nan_queries_fix_dev": (_i32, [_vp, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "gulon_topk_merge": (_i32, [_f32p, _i32p, _i32, _i32, _i32, _i32p, _f32p, _i32p, _i32p]),
    "gulon_exact_knn": (_i32, [_vp, _i32, _i32, _f32p, _i32, _i32, _i32p, _f32p, _i32p, _i32p]),
    "gulon_distance_sq_rows": (_i32, [_vp, _f32p, _i32, _i32p, _i32, _f32p]),
}

_lib = None


# libgulon_hip_testhooks.so (tests only): the product library's objects plus the kernels' self-tests and the
# measured-and-dropped fused k-means update (GULON_UPDATE_FUSED=1) -- include/gulon_hip.h under GULON_TEST_HOOKS
HOOKS_LIB_PATH = os.path.join(_HERE, "lib", "libgulon_hip_testhooks.so")
TEST_HOOK_SIGNATURES = {
    "gulon_selftest_mean_division": (_i32, [_i32, _i32, C.c_uint64, C.POINTER(C.c_int64)]),
    "gulon_selftest_stream_update": (_i32, [C.c_void_p, _i32, _i32, _i32, _i32, _i32, C.c_void_p, C.c_void_p]),
    "gulon_selftest_conflict_order": (_i32, [C.c_void_p, C.c_int64, _i32, C.c_void_p, C.c_void_p]),
    "gulon_selftest_assign_band": (_i32, [_i32, C.c_uint64, C.c_float, C.POINTER(C.c_double)]),
}
_hooks = None


def hooks_lib():
    """The test-hook build of the library (tests only; nothing in gulon_amd calls it)."""
    global _hooks
    if _hooks is None:
        L = C.CDLL(HOOKS_LIB_PATH)
        for name, (res, args) in TEST_HOOK_SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        L.gulon_last_error.restype = C.c_char_p
        _hooks = L
    return _hooks


def lib():
    """Load libgulon_hip.so; raises if it is not built (no fallback exists)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `make` (or __graft_entry__.build()). "
                "gulon_amd has no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)   # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.gulon_abi_version() != 3:
            raise ImportError("libgulon_hip.so ABI version mismatch")
        _lib = L
    return _lib


def last_error():
    return lib().gulon_last_error().decode("utf-8", "replace")


def check(rc):
    """Map a C-ABI status to the exception class the reference would throw."""
    if rc == OK:
        return
    msg = last_error()
    if rc == ERR_INVALID_ARGUMENT:
        raise ValueError("requirement failed: " + msg)       # IllegalArgumentException
    if rc == ERR_ILLEGAL_STATE:
        raise RuntimeError(msg)                              # IllegalStateException
    if rc == ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    if rc == ERR_OOM:
        raise MemoryError(msg)
    raise GulonDeviceError(msg)


def device_count():
    n = _i32(0)
    check(lib().gulon_device_count(C.byref(n)))
    return n.value


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)
