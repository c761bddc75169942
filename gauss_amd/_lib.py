"""ctypes binding of libgauss_hip.so (include/gauss_hip.h).

There is deliberately no fallback: if the HIP library is missing or fails to load, importing the
numeric entry points raises.  The CPU oracle under ``oracle/`` is test infrastructure and is never
imported from here.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libgauss_hip.so")

MODE_POOLED = 0
MODE_WEIGHTED = 1
ST_CLAMPED = 1
ST_NONFINITE = 2
WIN_IMPUTE = 0
WIN_QCAT = 1
WIN_LD = 2
GENO_U8, GENO_2BIT = 0, 1
CODE_ADDITIVE, CODE_DOMINANT, CODE_RECESSIVE = 1, 2, 4
GRAM_F32 = 0
GRAM_I8 = 1

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_u8p = C.c_void_p


class WindowDesc(C.Structure):
    _fields_ = [
        ("mode", C.c_int), ("n_pop", C.c_int), ("pop_off", _ip), ("pop_wgt", _dp),
        ("n_measured", C.c_int), ("n_unmeasured", C.c_int),
        ("geno_m", _u8p), ("geno_u", _u8p), ("ld", C.c_int64), ("z1", _dp),
        ("lambda_", C.c_double), ("min_abs_eig", C.c_double),
        ("out_z", _dp), ("out_info", _dp), ("out_status", _ip), ("out_b11", _dp), ("out_b21", _dp),
        ("kind", C.c_int), ("n_head_measured", C.c_int), ("n_pred_measured", C.c_int), ("eig_cutoff", C.c_double),
        ("out_r", _dp), ("out_num_eig", _ip), ("u_codings", C.c_int),
        ("geno_format", C.c_int), ("rows_m", _ip), ("rows_u", _ip), ("pop_src_off", _ip),
    ]


class GaussHipError(RuntimeError):
    pass


_lib = None

# every symbol include/gauss_hip.h declares
SYMBOLS = [
    "gauss_hip_init", "gauss_hip_device_count", "gauss_hip_device_of", "gauss_hip_destroy", "gauss_last_error", "gauss_hip_version", "gauss_hip_set_gram_dtype", "gauss_pinned_alloc", "gauss_pinned_free", "gauss_store_upload", "gauss_store_free", "gauss_pack2bit_device", "gauss_ld", "gauss_ld_per_pop", "gauss_ld_per_pop_pairs",
    "gauss_impute_window", "gauss_gene_ld_batch", "gauss_gram_counts", "gauss_job_create",
    "gauss_ld_rows", "gauss_gene_ld_batch_rows", "gauss_job_run", "gauss_job_fetch", "gauss_job_destroy", "gauss_job_span_ms", "gauss_job_profile",
    "gauss_job_profile_get", "gauss_job_work", "gauss_job_stats", "gauss_synth_device",
    "gauss_store_upload_async", "gauss_store_upload_fd_async", "gauss_store_wait", "gauss_store_alloc", "gauss_store_fill", "gauss_store_fill_fd", "gauss_store_upload_fd", "gauss_hip_context_id", "gauss_hip_add_destroy_hook", "gauss_hip_trim_cache", "gauss_hip_source_hash", "gauss_hip_counters", "gauss_job_counters", "gauss_hip_queues",
]


def load():
    """Load libgauss_hip.so; raise loudly if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GaussHipError(
            f"{LIB_PATH} not found: build it with `python -m gauss_amd.build` "
            "(gauss_amd has no CPU fallback by design)")
    lib = C.CDLL(LIB_PATH)
    for name in SYMBOLS:
        if not hasattr(lib, name):
            raise GaussHipError(f"libgauss_hip.so does not export {name}")
    lib.gauss_last_error.restype = C.c_char_p
    lib.gauss_hip_version.restype = C.c_char_p
    lib.gauss_hip_source_hash.restype = C.c_char_p
    lib.gauss_hip_init.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.gauss_hip_device_count.argtypes = [C.POINTER(C.c_int)]
    lib.gauss_hip_device_of.argtypes = [C.c_void_p]
    lib.gauss_hip_destroy.argtypes = [C.c_void_p]
    lib.gauss_hip_destroy.restype = None
    lib.gauss_hip_set_gram_dtype.argtypes = [C.c_void_p, C.c_int]
    lib.gauss_pinned_alloc.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    lib.gauss_pinned_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.gauss_store_upload.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    lib.gauss_store_free.argtypes = [C.c_void_p, C.c_void_p]
    lib.gauss_store_upload_async.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    lib.gauss_store_wait.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
    lib.gauss_store_alloc.argtypes = [C.c_void_p, C.c_int64, C.POINTER(C.c_void_p)]
    lib.gauss_store_fill.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]
    lib.gauss_store_fill_fd.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64]
    lib.gauss_store_upload_fd.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.POINTER(C.c_void_p)]
    lib.gauss_ld.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_int, C.c_int64, _ip, _dp, C.c_int,
                             C.c_double, _dp]
    lib.gauss_pack2bit_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, _ip, C.c_int]
    lib.gauss_ld_per_pop.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int64, _ip, C.c_int, _dp]
    lib.gauss_ld_per_pop_pairs.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int64, _ip, C.c_int, _ip, C.c_int, _ip, _ip, C.c_int64, _dp]
    lib.gauss_impute_window.argtypes = [C.c_void_p, C.POINTER(WindowDesc)]
    lib.gauss_gene_ld_batch.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_int, C.c_int64, _ip, _dp,
                                        C.c_int, _ip, C.c_int, C.c_double, _dp]
    lib.gauss_ld_rows.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_int64, C.c_int, _ip, C.c_int, _ip, _ip, _dp, C.c_int, C.c_double,
                                  C.c_int, _dp]
    lib.gauss_gene_ld_batch_rows.argtypes = [C.c_void_p, C.c_int, _u8p, C.c_int64, C.c_int, _ip, C.c_int, _ip, _ip, _dp, C.c_int,
                                             _ip, C.c_int, C.c_double, C.c_int, _dp]
    lib.gauss_gram_counts.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, C.c_int64,
                                      C.POINTER(C.c_int64)]
    lib.gauss_job_create.argtypes = [C.c_void_p, C.POINTER(WindowDesc), C.c_int, C.c_int,
                                     C.POINTER(C.c_void_p)]
    lib.gauss_job_run.argtypes = [C.c_void_p]
    lib.gauss_job_fetch.argtypes = [C.c_void_p]
    lib.gauss_job_destroy.argtypes = [C.c_void_p]
    lib.gauss_job_destroy.restype = None
    lib.gauss_job_span_ms.argtypes = [C.c_void_p, C.c_void_p, _dp]
    lib.gauss_job_profile.argtypes = [C.c_void_p, C.c_int]
    lib.gauss_job_profile_get.argtypes = [C.c_void_p, C.c_int, _dp, C.POINTER(C.c_int64)]
    lib.gauss_job_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.gauss_job_work.argtypes = [C.c_void_p, _dp, _dp, _dp, C.POINTER(C.c_int64)]
    lib.gauss_job_stats.argtypes = [C.c_void_p, _dp]
    lib.gauss_hip_context_id.argtypes = [C.c_void_p]
    lib.gauss_hip_context_id.restype = C.c_uint64
    lib.gauss_hip_trim_cache.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.gauss_hip_counters.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
    lib.gauss_hip_queues.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    lib.gauss_synth_device.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, _ip, C.c_int,
                                       C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint64]
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise GaussHipError(f"gauss_hip error {rc}: {load().gauss_last_error().decode()}")


def as_u8(g):
    g = np.ascontiguousarray(g)
    if g.dtype != np.uint8:
        g = g.astype(np.uint8)
    return g


def ptr(a, t):
    return a.ctypes.data_as(t) if a is not None else None
