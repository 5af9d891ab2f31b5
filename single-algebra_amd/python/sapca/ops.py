"""Stage-level operators of the C ABI (host buffers in and out) for the parity tests:
sum_col / sum_col_squared (src/sparse/csr.rs:259-312, 558-608), the two sweeps and the
normaliser inside randomized_svd, the Omega generator, the row partitioner."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

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
        ro = np.ascontiguousarray(indptr, dtype=np.uint64)
        ci = np.ascontiguousarray(indices, dtype=np.uint64)
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


def partition_rows(indptr, nparts):
    """nnz-balanced contiguous row ranges (host-only code path of the library)."""
    ro = np.ascontiguousarray(indptr, dtype=np.uint64)
    bounds = np.zeros(nparts + 1, dtype=np.uint64)
    st = L.load().sapca_partition_rows(C.c_uint64(ro.size - 1), _p(ro, C.c_uint64), C.c_uint32(nparts), _p(bounds, C.c_uint64))
    if st != L.OK:
        raise L.SapcaError(st, "sapca_partition_rows failed")
    return bounds
