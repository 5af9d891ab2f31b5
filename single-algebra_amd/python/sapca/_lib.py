"""ctypes binding of libsapca.so (include/sapca.h).  No fallbacks: if the HIP library is
missing or there is no GPU, calls fail loudly."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# SAPCA_LIB_PATH points at another build of the same library (kernel experiments, tools/abl_build.sh)
LIB_PATH = os.environ.get("SAPCA_LIB_PATH") or os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libsapca.so"))
# the variant that reads the SAPCA_* experiment / route switches (csrc/switches.h; the release library reads none of them)
DEBUG_LIB_PATH = os.path.normpath(os.path.join(_HERE, "..", "..", "lib", "libsapca_dbg.so"))

OK, ERR_ARG, ERR_MASK_LEN, ERR_NOT_FITTED, ERR_SVD, ERR_HIP, ERR_COMM, ERR_NOMEM = range(8)
LANCZOS, RANDOM = 0, 1
NORM_QR, NORM_LU, NORM_NONE = 0, 1, 2
TRANSFORM_REFERENCE, TRANSFORM_CENTERED = 0, 1


class Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("random_seed", C.c_uint32), ("n_components", C.c_uint64),
        ("alpha", C.c_double), ("tolerance", C.c_double),
        ("center", C.c_uint8), ("verbose", C.c_uint8), ("collect_timings", C.c_uint8), ("reserved0", C.c_uint8),
        ("method", C.c_int32), ("n_oversamples", C.c_uint64), ("n_power_iterations", C.c_uint64),
        ("normalizer", C.c_int32), ("transform_semantics", C.c_int32), ("device_id", C.c_int32),
        ("spmm_variant", C.c_int32), ("stream", C.c_void_p),
    ]


class Timings(C.Structure):
    _fields_ = [
        ("upload_ms", C.c_double), ("prepare_ms", C.c_double), ("stats_ms", C.c_double),
        ("spmm_ms", C.c_double), ("spmmt_ms", C.c_double), ("ortho_ms", C.c_double),
        ("small_svd_ms", C.c_double), ("lanczos_ms", C.c_double), ("transform_ms", C.c_double),
        ("comm_ms", C.c_double), ("fit_total_ms", C.c_double),
        ("n_spmm", C.c_uint32), ("n_spmmt", C.c_uint32),
        ("spmm_sweep_ms", C.c_double * 32), ("spmmt_sweep_ms", C.c_double * 32),
        ("bytes_per_sweep", C.c_double), ("lanczos_steps", C.c_uint64),
        ("sweep_kernel", C.c_uint32), ("at_sweep_pieces", C.c_uint32), ("sweep_slots_a", C.c_uint64), ("sweep_slots_at", C.c_uint64),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p)

# every symbol include/sapca.h declares (the CPU test checks the library exports each one)
_TYPED = [
    "sapca_set_omega", "sapca_fit_csr", "sapca_transform_csr", "sapca_fit_transform_csr",
    "sapca_fit_csr_device", "sapca_transform_csr_device", "sapca_fit_transform_csr_device",
    "sapca_transform_csr_device_to_host", "sapca_fit_transform_csr_device_to_host",
    "sapca_get_components", "sapca_get_singular_values", "sapca_get_explained_variance", "sapca_get_mean",
    "sapca_get_explained_variance_ratio", "sapca_get_cumulative_explained_variance_ratio",
    "sapca_get_feature_importances", "sapca_colstats_csr", "sapca_spmm_csr", "sapca_spmmt_csr",
    "sapca_normalize_panel", "sapca_generate_omega",
    "sapca_upload_csr", "sapca_normalize_csr_device", "sapca_log1p_csr_device", "sapca_stats_csr_device",
    "sapca_multi_fit_csr", "sapca_multi_transform_csr", "sapca_multi_fit_transform_csr",
    "sapca_multi_upload_csr", "sapca_multi_transform_resident", "sapca_multi_fit_transform_resident",
]
_PLAIN = [
    "sapca_options_default", "sapca_abi_version", "sapca_create", "sapca_destroy", "sapca_last_error",
    "sapca_set_mask", "sapca_get_dims", "sapca_get_total_variance", "sapca_get_mask_index_maps",
    "sapca_get_timings", "sapca_partition_rows", "sapca_comm_unique_id", "sapca_comm_rccl_available", "sapca_comm_init_rank",
    "sapca_comm_set_callback", "sapca_comm_allreduce", "sapca_comm_abort", "sapca_comm_async_error", "sapca_comm_has_side_lane",
    "sapca_upload_values_changed", "sapca_measure_copy_gbs", "sapca_multi_fit_resident", "sapca_multi_resident_shard",
    "sapca_multi_create", "sapca_multi_destroy", "sapca_multi_last_error", "sapca_multi_n_devices", "sapca_multi_member",
    "sapca_multi_uses_rccl", "sapca_multi_set_mask",
]
EXPORTED_SYMBOLS = _PLAIN + [f"{n}_{s}" for n in _TYPED for s in ("f32", "f64")]

_lib = None


class SapcaError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(message)
        self.status = status


def load():
    """dlopen libsapca.so.  torch (if it is going to be used at all) must be imported first so the
    process shares ONE HIP runtime / RCCL: bench.py and the tests import torch at the top."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH) and not os.environ.get("SAPCA_NO_AUTOBUILD"):
        # building the product is not a fallback: same sources, same gfx950 target
        import shutil
        import subprocess
        if shutil.which("make") and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
            subprocess.call(["make", "-C", os.path.normpath(os.path.join(_HERE, "..", "..")), "-j", "8"])
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `make -C single-algebra_amd` "
            "(or python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
    _lib = _open(LIB_PATH)
    return _lib


_debug_lib = None


def load_debug():
    """The -DSAPCA_DEBUG_SWITCHES build of the same sources (make debug): for tests and tools that steer routes through
    SAPCA_* environment variables.  A second library instance beside the release one (its own handles: objects do not cross)."""
    global _debug_lib
    if _debug_lib is None:
        if not os.path.exists(DEBUG_LIB_PATH):
            raise ImportError(f"{DEBUG_LIB_PATH} is missing: build it with `make -C single-algebra_amd debug`")
        _debug_lib = _open(DEBUG_LIB_PATH)
    return _debug_lib


def _open(path):
    lib = C.CDLL(path)
    lib.sapca_last_error.restype = C.c_char_p
    lib.sapca_last_error.argtypes = [C.c_void_p]
    lib.sapca_create.argtypes = [C.POINTER(Options), C.POINTER(C.c_void_p)]
    lib.sapca_destroy.argtypes = [C.c_void_p]
    lib.sapca_destroy.restype = None
    lib.sapca_options_default.argtypes = [C.POINTER(Options)]
    lib.sapca_options_default.restype = None
    lib.sapca_multi_create.argtypes = [C.POINTER(Options), C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_void_p)]
    lib.sapca_multi_destroy.argtypes = [C.c_void_p]
    lib.sapca_multi_destroy.restype = None
    lib.sapca_multi_last_error.argtypes = [C.c_void_p]
    lib.sapca_multi_last_error.restype = C.c_char_p
    lib.sapca_multi_member.argtypes = [C.c_void_p, C.c_uint32]
    lib.sapca_multi_member.restype = C.c_void_p
    lib.sapca_multi_n_devices.argtypes = [C.c_void_p]
    lib.sapca_multi_n_devices.restype = C.c_uint32
    lib.sapca_multi_uses_rccl.argtypes = [C.c_void_p]
    if hasattr(lib, "sapca_comm_abort"):   # (ABI 4; an older build loaded through SAPCA_LIB_PATH for an A/B run lacks them)
        lib.sapca_multi_fit_resident.argtypes = [C.c_void_p]
        lib.sapca_comm_abort.argtypes = [C.c_void_p]
        lib.sapca_comm_async_error.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
        lib.sapca_comm_has_side_lane.argtypes = [C.c_void_p]
    if hasattr(lib, "sapca_measure_copy_gbs"):
        lib.sapca_measure_copy_gbs.argtypes = [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]
    return lib


def default_options() -> Options:
    o = Options()
    load().sapca_options_default(C.byref(o))
    return o


def check(handle, status):
    if status != OK:
        msg = load().sapca_last_error(handle)
        raise SapcaError(status, (msg or b"").decode() or f"sapca status {status}")


def as_u64(a):
    """A CSR offset / index array in the library's u64 (nalgebra_sparse `usize`) layout: int64 arrays -- what scipy holds for
    large matrices -- are reinterpreted in place (a negative entry reads as an index the library refuses), anything else
    is converted (a copy)."""
    import numpy as np
    a = np.asarray(a)
    if a.dtype == np.uint64 and a.flags.c_contiguous:
        return a
    if a.dtype == np.int64 and a.flags.c_contiguous:
        return a.view(np.uint64)
    return np.ascontiguousarray(a, dtype=np.uint64)
