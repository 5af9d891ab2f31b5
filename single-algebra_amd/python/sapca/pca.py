"""Host-side mirror of single-algebra's sparse-PCA API over the C ABI.

Same names, argument meaning and error behaviour as the reference so the parity tests read
like the reference's own (paths relative to /root/reference):

  SVDMethod, PowerIterationNormalizer      src/dimred/pca/mod.rs:41-68
  SparsePCABuilder / SparsePCA             src/dimred/pca/sparse/mod.rs:33-484
  MaskedSparsePCABuilder / MaskedSparsePCA src/dimred/pca/sparse_masked/mod.rs:37-620

Inputs are either a scipy.sparse.csr_matrix (host; indices are widened to the usize layout of
nalgebra_sparse::CsrMatrix and go through the sapca_*_csr_* entry points) or a DeviceCsr of
torch CUDA tensors (HBM-resident; sapca_*_csr_device_* entry points, outputs are torch tensors).
All compute happens in libsapca.so; this module only marshals.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib as L
from ._lib import as_u64


class PowerIterationNormalizer(enum.IntEnum):
    QR = L.NORM_QR
    LU = L.NORM_LU
    NONE = L.NORM_NONE


@dataclass(frozen=True)
class SVDMethod:
    """enum SVDMethod { Lanczos, Random { n_oversamples, n_power_iterations, normalizer } }"""
    kind: str = "Lanczos"                       # Default: Lanczos (pca/mod.rs:64-68)
    n_oversamples: int = 10
    n_power_iterations: int = 4
    normalizer: PowerIterationNormalizer = PowerIterationNormalizer.QR

    @staticmethod
    def Lanczos() -> "SVDMethod":
        return SVDMethod("Lanczos")

    @staticmethod
    def Random(n_oversamples: int, n_power_iterations: int,
               normalizer: PowerIterationNormalizer = PowerIterationNormalizer.QR) -> "SVDMethod":
        return SVDMethod("Random", int(n_oversamples), int(n_power_iterations), PowerIterationNormalizer(normalizer))


class DeviceCsr:
    """HBM-resident CSR: row_offsets int64 [m+1], col_indices int32 [nnz], values f32/f64 [nnz]
    (torch CUDA tensors), columns ascending and unique per row."""

    def __init__(self, row_offsets, col_indices, values, shape):
        import torch
        assert row_offsets.dtype == torch.int64 and col_indices.dtype == torch.int32
        assert values.dtype in (torch.float32, torch.float64)
        assert row_offsets.is_cuda and col_indices.is_cuda and values.is_cuda
        self.row_offsets = row_offsets.contiguous()
        self.col_indices = col_indices.contiguous()
        self.values = values.contiguous()
        self.shape = (int(shape[0]), int(shape[1]))
        assert self.row_offsets.numel() == self.shape[0] + 1

    @property
    def nnz(self):
        return int(self.values.numel())

    def nrows(self):
        return self.shape[0]

    def ncols(self):
        return self.shape[1]


_SUF = {np.dtype(np.float32): ("f32", C.c_float), np.dtype(np.float64): ("f64", C.c_double)}


def _np_ptr(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class _Estimator:
    """State and marshalling shared by SparsePCA and MaskedSparsePCA."""

    def __init__(self, n_components, alpha, tolerance, random_seed, center, verbose, svdmethod, mask=None,
                 device=None, transform_semantics=L.TRANSFORM_REFERENCE, spmm_variant=0, collect_timings=False):
        self.n_components = int(n_components)
        self.alpha = float(alpha)
        self.tolerance = float(tolerance)
        self.random_seed = int(random_seed)
        self.center = bool(center)
        self.verbose = bool(verbose)
        self.svdmethod = svdmethod
        self._mask = None if mask is None else np.ascontiguousarray(np.asarray(mask, dtype=bool))
        self._h = C.c_void_p()
        lib = L.load()
        o = L.default_options()
        o.n_components = self.n_components
        o.alpha, o.tolerance = self.alpha, self.tolerance
        o.random_seed = self.random_seed & 0xFFFFFFFF
        o.center, o.verbose = int(self.center), int(self.verbose)
        o.collect_timings = int(bool(collect_timings))
        o.method = L.RANDOM if svdmethod.kind == "Random" else L.LANCZOS
        o.n_oversamples = svdmethod.n_oversamples
        o.n_power_iterations = svdmethod.n_power_iterations
        o.normalizer = int(svdmethod.normalizer)
        o.transform_semantics = int(transform_semantics)
        o.spmm_variant = int(spmm_variant)
        o.device_id = -1 if device is None else int(device)
        try:
            import torch
            if torch.cuda.is_available():
                if device is not None:
                    torch.cuda.set_device(int(device))
                o.stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        except ImportError:
            pass
        self._opts = o
        self._borrowed = False     # (a member of a sapca_multi: the multi handle owns it)
        st = lib.sapca_create(C.byref(o), C.byref(self._h))
        if st != L.OK:
            raise L.SapcaError(st, (lib.sapca_last_error(None) or b"").decode())
        if self._mask is not None and self._mask.size:
            m8 = self._mask.astype(np.uint8)
            L.check(self._h, lib.sapca_set_mask(self._h, _np_ptr(m8, C.c_uint8), C.c_size_t(m8.size)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and not getattr(self, "_borrowed", False):
                L.load().sapca_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # -- test hook -----------------------------------------------------------------------
    def set_omega(self, omega):
        """Inject the Gaussian test matrix of the next randomized fit (parity tests)."""
        om = np.ascontiguousarray(omega, dtype=np.float64)
        L.check(self._h, L.load().sapca_set_omega_f64(self._h, _np_ptr(om, C.c_double),
                                                      C.c_size_t(om.shape[0]), C.c_size_t(om.shape[1])))
        return self

    # -- marshalling ---------------------------------------------------------------------
    def _mask_check(self, ncols):
        # MaskedSparsePCA raises on ANY length mismatch, including an empty mask (masked :258-262)
        if self._mask is not None and self._mask.size != ncols:
            raise L.SapcaError(L.ERR_MASK_LEN,
                               "The mask vector length and the number of features (columns) have to be the same!")

    def _call(self, op, x, want_out):
        lib = L.load()
        if isinstance(x, DeviceCsr):
            import torch
            suf = "f32" if x.values.dtype == torch.float32 else "f64"
            m, n = x.shape
            self._mask_check(n)
            args = [self._h, C.c_uint64(m), C.c_uint64(n), C.c_uint64(x.nnz),
                    C.c_void_p(x.row_offsets.data_ptr()), C.c_void_p(x.col_indices.data_ptr()),
                    C.c_void_p(x.values.data_ptr())]
            out = None
            if want_out:
                out = torch.empty((m, self.n_components), dtype=x.values.dtype, device=x.values.device)
                args.append(C.c_void_p(out.data_ptr()))
            torch.cuda.current_stream().synchronize()
            L.check(self._h, getattr(lib, f"sapca_{op}_csr_device_{suf}")(*args))
            return out
        import scipy.sparse as sp
        if not sp.isspmatrix_csr(x):
            raise TypeError("expected a scipy.sparse.csr_matrix or a sapca.DeviceCsr")
        if not x.has_sorted_indices:
            x = x.sorted_indices()
        dt = np.dtype(x.dtype)
        if dt not in _SUF:
            raise TypeError("values must be float32 or float64")
        suf, ct = _SUF[dt]
        m, n = x.shape
        self._mask_check(n)
        ro = as_u64(x.indptr)     # nalgebra_sparse usize layout
        ci = as_u64(x.indices)
        va = np.ascontiguousarray(x.data)
        args = [self._h, C.c_uint64(m), C.c_uint64(n), C.c_uint64(va.size), _np_ptr(ro, C.c_uint64),
                _np_ptr(ci, C.c_uint64), _np_ptr(va, ct)]
        out = None
        if want_out:
            out = np.empty((m, self.n_components), dtype=dt)
            args.append(_np_ptr(out, ct))
        L.check(self._h, getattr(lib, f"sapca_{op}_csr_{suf}")(*args))
        return out

    # -- reference API ---------------------------------------------------------------------
    def fit(self, x):
        self._call("fit", x, False)
        return self

    def transform(self, x):
        return self._call("transform", x, True)

    def fit_transform(self, x):
        return self._call("fit_transform", x, True)

    def _dims(self):
        k, nu, nc = C.c_uint64(), C.c_uint64(), C.c_uint64()
        L.check(self._h, L.load().sapca_get_dims(self._h, C.byref(k), C.byref(nu), C.byref(nc)))
        return k.value, nu.value, nc.value

    def _get(self, name, shape_fn, dtype=None):
        k, nu, nc = self._dims()
        dtype = np.dtype(dtype or self._fitted_dtype())
        suf, ct = _SUF[dtype]
        shape = shape_fn(k, nu, nc)
        out = np.empty(shape, dtype=dtype)
        L.check(self._h, getattr(L.load(), f"sapca_get_{name}_{suf}")(self._h, _np_ptr(out, ct), C.c_size_t(out.size)))
        return out

    def _fitted_dtype(self):
        return np.float64 if getattr(self, "_dtype64", False) else np.float32

    def feature_importances(self, dtype=None):
        return self._get("feature_importances", lambda k, nu, nc: (k, nu), dtype)

    def explained_variance_ratio(self, dtype=None):
        return self._get("explained_variance_ratio", lambda k, nu, nc: (k,), dtype)

    def cumulative_explained_variance_ratio(self, dtype=None):
        return self._get("cumulative_explained_variance_ratio", lambda k, nu, nc: (k,), dtype)

    # -- fitted fields (private in the reference; exposed for wrappers and tests) ----------------
    def components_(self, dtype=None):
        return self._get("components", lambda k, nu, nc: (k, nu), dtype)

    def explained_variance_(self, dtype=None):
        return self._get("explained_variance", lambda k, nu, nc: (k,), dtype)

    def singular_values_(self, dtype=None):
        return self._get("singular_values", lambda k, nu, nc: (k,), dtype)

    def mean_(self, dtype=None):
        return self._get("mean", lambda k, nu, nc: (nc,), dtype)

    def total_variance_(self):
        v = C.c_double()
        L.check(self._h, L.load().sapca_get_total_variance(self._h, C.byref(v)))
        return v.value

    def mask_index_maps(self):
        n = 0 if self._mask is None else self._mask.size
        cols = np.zeros(max(n, 1), dtype=np.uint64)
        o2m = np.zeros(max(n, 1), dtype=np.int64)
        L.check(self._h, L.load().sapca_get_mask_index_maps(
            self._h, _np_ptr(cols, C.c_uint64), C.c_size_t(cols.size), _np_ptr(o2m, C.c_int64), C.c_size_t(o2m.size)))
        n_used = int(self._mask.sum()) if self._mask is not None else 0
        return cols[:n_used], o2m[:n]

    def timings(self):
        t = L.Timings()
        L.check(self._h, L.load().sapca_get_timings(self._h, C.byref(t)))
        return t

    # -- multi-GPU ---------------------------------------------------------------------------
    def comm_init_rank(self, nranks, rank, unique_id: bytes):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        L.check(self._h, L.load().sapca_comm_init_rank(self._h, C.c_uint32(nranks), C.c_uint32(rank), buf))

    def comm_set_callback(self, nranks, rank, fn):
        self._cb = L.ALLREDUCE_FN(fn)     # keep alive
        L.check(self._h, L.load().sapca_comm_set_callback(self._h, C.c_uint32(nranks), C.c_uint32(rank), self._cb, None))

    def comm_abort(self):
        """End every collective of this handle (a peer rank failed): sapca_comm_abort; callable from another thread."""
        L.check(self._h, L.load().sapca_comm_abort(self._h))

    def comm_async_error(self):
        st = C.c_int32(0)
        L.check(self._h, L.load().sapca_comm_async_error(self._h, C.byref(st)))
        return int(st.value)

    def comm_has_side_lane(self):
        return bool(L.load().sapca_comm_has_side_lane(self._h))

    def measure_copy_gbs(self, nbytes=1 << 30, reps=5):
        """GB/s (read + write) of the library's 16-byte-per-lane streaming copy on this handle's device"""
        g = C.c_double(0.0)
        L.check(self._h, L.load().sapca_measure_copy_gbs(self._h, C.c_uint64(nbytes), C.c_uint32(reps), C.byref(g)))
        return float(g.value)


def _note_dtype(est, x):
    if isinstance(x, DeviceCsr):
        import torch
        est._dtype64 = x.values.dtype == torch.float64
    else:
        est._dtype64 = np.dtype(x.dtype) == np.float64


class SparsePCA(_Estimator):
    """SparsePCA<T> (sparse/mod.rs:33-359)."""

    def __init__(self, n_components, alpha, tollerance=None, random_seed=None, center=True, verbose=False,
                 svdmethod=SVDMethod(), **ext):
        super().__init__(n_components, alpha, 1e-6 if tollerance is None else tollerance,      # :75
                         42 if random_seed is None else random_seed, center, verbose, svdmethod, None, **ext)  # :76

    def fit(self, x):
        _note_dtype(self, x)
        return super().fit(x)

    def fit_transform(self, x):
        _note_dtype(self, x)
        return super().fit_transform(x)


class MaskedSparsePCA(_Estimator):
    """MaskedSparsePCA<T> (sparse_masked/mod.rs:179-620)."""

    def __init__(self, n_components, alpha, tollerance=None, random_seed=None, mask=(), center=True, verbose=False,
                 svd_method=SVDMethod(), **ext):
        super().__init__(n_components, alpha, 1e-6 if tollerance is None else tollerance,
                         42 if random_seed is None else random_seed, center, verbose, svd_method,
                         np.asarray(mask, dtype=bool), **ext)

    def fit(self, x):
        _note_dtype(self, x)
        return super().fit(x)

    def fit_transform(self, x):
        _note_dtype(self, x)
        return super().fit_transform(x)


class _BuilderBase:
    def __init__(self):
        # defaults: sparse/mod.rs:392-401, sparse_masked/mod.rs:55-66
        self._n_components = 50
        self._alpha = 1.0
        self._tolerance = 1e-6
        self._random_seed: Optional[int] = 42
        self._center = True
        self._verbose = False
        self._svdmethod = SVDMethod()
        self._ext = {}

    @classmethod
    def new(cls):
        return cls()

    @classmethod
    def default(cls):
        return cls()

    def n_components(self, n):
        self._n_components = int(n)
        return self

    def alpha(self, a):
        self._alpha = float(a)
        return self

    def tolerance(self, t):
        self._tolerance = float(t)
        return self

    def random_seed(self, seed):
        self._random_seed = int(seed)
        return self

    def center(self, c):
        self._center = bool(c)
        return self

    def verbose(self, v):
        self._verbose = bool(v)
        return self

    def svd_method(self, m: SVDMethod):
        self._svdmethod = m
        return self

    # extensions that have no reference counterpart (device placement, diagnostics)
    def device(self, ordinal):
        self._ext["device"] = int(ordinal)
        return self

    def transform_semantics(self, sem):
        self._ext["transform_semantics"] = int(sem)
        return self

    def spmm_variant(self, v):
        self._ext["spmm_variant"] = int(v)
        return self

    def collect_timings(self, on=True):
        self._ext["collect_timings"] = bool(on)
        return self


class SparsePCABuilder(_BuilderBase):
    """SparsePCABuilder<T> (sparse/mod.rs:375-484)."""

    def build(self) -> SparsePCA:
        return SparsePCA(self._n_components, self._alpha, self._tolerance,
                         42 if self._random_seed is None else self._random_seed,       # :475
                         self._center, self._verbose, self._svdmethod, **self._ext)


class MaskedSparsePCABuilder(_BuilderBase):
    """MaskedSparsePCABuilder<T> (sparse_masked/mod.rs:37-160)."""

    def __init__(self):
        super().__init__()
        self._mask = np.zeros(0, dtype=bool)                                            # :62

    def mask(self, mask):
        self._mask = np.asarray(mask, dtype=bool)
        return self

    def build(self) -> MaskedSparsePCA:
        return MaskedSparsePCA(self._n_components, self._alpha, self._tolerance,
                               42 if self._random_seed is None else self._random_seed, self._mask,
                               self._center, self._verbose, self._svdmethod, **self._ext)
