"""Stage-level operators of the C ABI (host buffers in and out) for the parity tests:
sum_col / sum_col_squared (src/sparse/csr.rs:259-312, 558-608), the two sweeps and the
normaliser inside randomized_svd, the Omega generator, the row partitioner."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from ._lib import as_u64

_SUF = {np.dtype(np.float32): ("f32", C.c_float), np.dtype(np.float64): ("f64", C.c_double)}


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


class Session:
    """A bare handle (default options) to run stage-level operators on."""

    def __init__(self, seed=42, spmm_variant=0):
        self._h = C.c_void_p()
        o = L.default_options()
        o.random_seed = seed
        o.spmm_variant = spmm_variant
        st = L.load().sapca_create(C.byref(o), C.byref(self._h))
        if st != L.OK:
            raise L.SapcaError(st, (L.load().sapca_last_error(None) or b"").decode())

    def __del__(self):
        try:
            if self._h:
                L.load().sapca_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def _csr_args(self, indptr, indices, data, m, n):
        ro = as_u64(indptr)
        ci = as_u64(indices)
        va = np.ascontiguousarray(data)
        suf, ct = _SUF[va.dtype]
        keep = (ro, ci, va)
        return suf, ct, keep, [self._h, C.c_uint64(m), C.c_uint64(n), C.c_uint64(va.size), _p(ro, C.c_uint64),
                               _p(ci, C.c_uint64), _p(va, ct)]

    def colstats(self, indptr, indices, data, m, n):
        suf, ct, keep, args = self._csr_args(indptr, indices, data, m, n)
        s = np.zeros(n, dtype=data.dtype)
        sq = np.zeros(n, dtype=data.dtype)
        cnt = np.zeros(n, dtype=np.uint64)
        L.check(self._h, getattr(L.load(), f"sapca_colstats_csr_{suf}")(*args, _p(s, ct), _p(sq, ct), _p(cnt, C.c_uint64)))
        return s, sq, cnt

    def spmm(self, indptr, indices, data, m, n, X, mu=None, transposed=False):
        suf, ct, keep, args = self._csr_args(indptr, indices, data, m, n)
        X = np.ascontiguousarray(X, dtype=data.dtype)
        l = X.shape[1]
        out = np.zeros((n if transposed else m, l), dtype=data.dtype)
        mu_p = _p(np.ascontiguousarray(mu, dtype=data.dtype), ct) if mu is not None else None
        fn = getattr(L.load(), f"sapca_spmm{'t' if transposed else ''}_csr_{suf}")
        L.check(self._h, fn(*args, mu_p, C.c_uint64(l), _p(X, ct), _p(out, ct)))
        return out

    def normalize_panel(self, P, normalizer):
        P = np.array(P, order="C", copy=True)
        suf, ct = _SUF[P.dtype]
        L.check(self._h, getattr(L.load(), f"sapca_normalize_panel_{suf}")(
            self._h, C.c_int32(int(normalizer)), C.c_uint64(P.shape[0]), C.c_uint64(P.shape[1]), _p(P, ct)))
        return P

    def generate_omega(self, rows, l, dtype=np.float64):
        out = np.zeros((rows, l), dtype=dtype)
        suf, ct = _SUF[np.dtype(dtype)]
        L.check(self._h, getattr(L.load(), f"sapca_generate_omega_{suf}")(self._h, C.c_uint64(rows), C.c_uint64(l), _p(out, ct)))
        return out


ROW, COLUMN = 0, 1   # Direction of the reference's Normalize / statistics traits (src/utils.rs)


class ResidentCsr:
    """A CSR matrix uploaded once into buffers owned by a Session (sapca_upload_csr_*): normalize -> log1p ->
    statistics -> PCA on it without crossing PCIe again.  `as_device_csr()` gives the
    sapca.DeviceCsr the estimators take (zero-copy torch views of the same memory)."""

    def __init__(self, session, shape, nnz, dtype, d_ptr, d_idx, d_val):
        self._s, self.shape, self.nnz, self.dtype = session, tuple(shape), int(nnz), np.dtype(dtype)
        self.d_ptr, self.d_idx, self.d_val = int(d_ptr), int(d_idx), int(d_val)

    def _args(self):
        m, n = self.shape
        return [self._s._h, C.c_uint64(m), C.c_uint64(n), C.c_uint64(self.nnz), C.c_void_p(self.d_ptr), C.c_void_p(self.d_idx),
                C.c_void_p(self.d_val)]

    def normalize(self, sums, target, direction):
        """Normalize<T>::normalize(&sums, target, &direction), csr.rs:1012-1066 (in place)"""
        suf, _ = _SUF[self.dtype]
        sums = np.ascontiguousarray(sums, dtype=np.float64)
        L.check(self._s._h, getattr(L.load(), f"sapca_normalize_csr_device_{suf}")(
            *self._args(), _p(sums, C.c_double), C.c_uint64(sums.size), C.c_double(float(target)), C.c_int32(int(direction))))
        return self

    def log1p(self):
        """Log1P::log1p_normalize, csr.rs:1069-1078 (in place)"""
        suf, _ = _SUF[self.dtype]
        L.check(self._s._h, getattr(L.load(), f"sapca_log1p_csr_device_{suf}")(self._s._h, C.c_uint64(self.nnz), C.c_void_p(self.d_val)))
        return self

    def values_changed(self):
        """the caller edited the device values itself: drop the statistics gathered at upload (sapca_upload_values_changed)"""
        L.check(self._s._h, L.load().sapca_upload_values_changed(self._s._h))
        return self

    def stats(self, direction):
        """(sum, sum_squared, nonzero, min, max) per row (ROW) or column (COLUMN): sum_row/col, sum_row/col_squared,
        nonzero_row/col, min_max_row/col of the reference (csr.rs:23-134, 259-392, 558-630, 917-1008)"""
        suf, ct = _SUF[self.dtype]
        m, n = self.shape
        ln = n if int(direction) == COLUMN else m
        sm, sq = np.zeros(ln), np.zeros(ln)
        nz = np.zeros(ln, dtype=np.uint64)
        lo, hi = np.zeros(ln, dtype=self.dtype), np.zeros(ln, dtype=self.dtype)
        L.check(self._s._h, getattr(L.load(), f"sapca_stats_csr_device_{suf}")(
            *self._args(), C.c_int32(int(direction)), _p(sm, C.c_double), _p(sq, C.c_double), _p(nz, C.c_uint64), _p(lo, ct), _p(hi, ct)))
        return sm, sq, nz, lo, hi

    def variance(self, direction):
        """var_row / var_col (csr.rs:632-726): host arithmetic on the sums"""
        sm, sq, _, _, _ = self.stats(direction)
        N = float(self.shape[0] if int(direction) == COLUMN else self.shape[1])
        if N <= 1:
            return np.zeros_like(sm)
        mean = sm / N
        return (sq / N - mean ** 2) * (N / (N - 1.0))

    def values(self):
        """the current (device) values, copied to the host"""
        return self.as_device_csr().values.cpu().numpy()

    def as_device_csr(self):
        import torch
        from .dist import _DevView  # noqa: F401  (same zero-copy view helper)
        from .pca import DeviceCsr

        class _View:
            def __init__(self, ptr, count, typestr):
                self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
        m, n = self.shape
        ptr = torch.as_tensor(_View(self.d_ptr, m + 1, "<i8"), device="cuda")
        idx = torch.as_tensor(_View(self.d_idx, max(self.nnz, 1), "<i4"), device="cuda")[: self.nnz]
        val = torch.as_tensor(_View(self.d_val, max(self.nnz, 1), "<f4" if self.dtype == np.float32 else "<f8"), device="cuda")[: self.nnz]
        return DeviceCsr(ptr, idx, val, (m, n))


def _upload(self, indptr, indices, data, m, n):
    """sapca_upload_csr_*: host CSR -> ResidentCsr in this session's buffers"""
    suf, ct, keep, args = self._csr_args(indptr, indices, data, m, n)
    dp, di, dv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    L.check(self._h, getattr(L.load(), f"sapca_upload_csr_{suf}")(*args, C.byref(dp), C.byref(di), C.byref(dv)))
    return ResidentCsr(self, (m, n), keep[2].size, keep[2].dtype, dp.value or 0, di.value or 0, dv.value or 0)


Session.upload = _upload


def partition_rows(indptr, nparts):
    """nnz-balanced contiguous row ranges (host-only code path of the library)."""
    ro = as_u64(indptr)
    bounds = np.zeros(nparts + 1, dtype=np.uint64)
    st = L.load().sapca_partition_rows(C.c_uint64(ro.size - 1), _p(ro, C.c_uint64), C.c_uint32(nparts), _p(bounds, C.c_uint64))
    if st != L.OK:
        raise L.SapcaError(st, "sapca_partition_rows failed")
    return bounds
