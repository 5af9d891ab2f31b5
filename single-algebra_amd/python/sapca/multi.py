"""One estimator, several GPUs, ONE calling thread (include/sapca.h, sapca_multi_*).

The reference call is a single `fit_transform(&CsrMatrix)` from one thread
(/root/reference/src/dimred/pca/sparse/mod.rs:355-358).  MultiDevice keeps that shape: the host
matrix is split into nnz-balanced row ranges inside the library, one member handle and one host
thread per device; the members' all-reduces run over RCCL between distinct devices and through
page-locked host memory when a device is listed more than once (a one-GPU box rehearsing the path).
The fitted state is replicated on all members, so every getter of the estimator classes works on
member 0 (`md.components_()`, `md.explained_variance_ratio()`, ...)."""
from __future__ import annotations

import copy
import ctypes as C

import numpy as np

from . import _lib as L
from .pca import _SUF, _np_ptr, as_u64


class MultiDevice:
    def __init__(self, estimator, devices):
        """`estimator`: an unfitted SparsePCA / MaskedSparsePCA from the builders (its configuration is copied, the object
        itself stays usable); `devices`: HIP device ordinals, one shard each."""
        lib = L.load()
        self._proto = estimator
        self.devices = [int(d) for d in devices]
        self.n_components = estimator.n_components
        ids = (C.c_int32 * len(self.devices))(*self.devices)
        self._mh = C.c_void_p()
        st = lib.sapca_multi_create(C.byref(estimator._opts), ids, C.c_uint32(len(self.devices)), C.byref(self._mh))
        if st != L.OK:
            raise L.SapcaError(st, (lib.sapca_multi_last_error(None) or b"").decode())
        mask = getattr(estimator, "_mask", None)
        self._mask = mask
        if mask is not None and mask.size:
            m8 = mask.astype(np.uint8)
            self._check(lib.sapca_multi_set_mask(self._mh, _np_ptr(m8, C.c_uint8), C.c_size_t(m8.size)))
        self._members = []
        for i in range(len(self.devices)):
            m = copy.copy(estimator)          # same class and configuration, bound to the member's handle, owning nothing
            m._h = C.c_void_p(lib.sapca_multi_member(self._mh, C.c_uint32(i)))
            m._borrowed = True
            self._members.append(m)

    def __del__(self):
        try:
            if getattr(self, "_mh", None):
                for m in self._members:
                    m._h = C.c_void_p()
                L.load().sapca_multi_destroy(self._mh)
                self._mh = C.c_void_p()
        except Exception:
            pass

    def _check(self, status):
        if status != L.OK:
            raise L.SapcaError(status, (L.load().sapca_multi_last_error(self._mh) or b"").decode())

    def uses_rccl(self):
        return bool(L.load().sapca_multi_uses_rccl(self._mh))

    def member(self, i):
        """the estimator view of member i (getters, timings)"""
        return self._members[i]

    def set_omega(self, omega):
        for m in self._members:
            m.set_omega(omega)
        return self

    def _call(self, op, x, want_out):
        import scipy.sparse as sp
        if not sp.isspmatrix_csr(x):
            raise TypeError("expected a scipy.sparse.csr_matrix (sapca_multi_* shard a HOST matrix)")
        if not x.has_sorted_indices:
            x = x.sorted_indices()
        dt = np.dtype(x.dtype)
        suf, ct = _SUF[dt]
        m, n = x.shape
        if self._mask is not None and self._mask.size != n:
            raise L.SapcaError(L.ERR_MASK_LEN, "The mask vector length and the number of features (columns) have to be the same!")
        ro, ci, va = as_u64(x.indptr), as_u64(x.indices), np.ascontiguousarray(x.data)
        args = [self._mh, C.c_uint64(m), C.c_uint64(n), C.c_uint64(va.size), _np_ptr(ro, C.c_uint64), _np_ptr(ci, C.c_uint64), _np_ptr(va, ct)]
        out = None
        if want_out:
            out = np.empty((m, self.n_components), dtype=dt)
            args.append(_np_ptr(out, ct))
        self._check(getattr(L.load(), f"sapca_multi_{op}_csr_{suf}")(*args))
        for mem in self._members:
            mem._dtype64 = dt == np.float64
        return out

    # ---- resident shards: upload once, fit / project repeatedly without PCIe (sapca_multi_upload_csr_*) ----
    def upload(self, x):
        import scipy.sparse as sp
        if not sp.isspmatrix_csr(x):
            raise TypeError("expected a scipy.sparse.csr_matrix")
        if not x.has_sorted_indices:
            x = x.sorted_indices()
        dt = np.dtype(x.dtype)
        suf, ct = _SUF[dt]
        m, n = x.shape
        if self._mask is not None and self._mask.size != n:
            raise L.SapcaError(L.ERR_MASK_LEN, "The mask vector length and the number of features (columns) have to be the same!")
        ro, ci, va = as_u64(x.indptr), as_u64(x.indices), np.ascontiguousarray(x.data)
        self._check(getattr(L.load(), f"sapca_multi_upload_csr_{suf}")(
            self._mh, C.c_uint64(m), C.c_uint64(n), C.c_uint64(va.size), _np_ptr(ro, C.c_uint64), _np_ptr(ci, C.c_uint64), _np_ptr(va, ct)))
        self._resident = (m, dt)
        return self

    def _resident_call(self, op):
        if getattr(self, "_resident", None) is None:
            raise L.SapcaError(L.ERR_ARG, "no resident matrix: call upload() first")
        m, dt = self._resident
        suf, ct = _SUF[dt]
        lib = L.load()
        out = None
        if op == "fit":
            self._check(lib.sapca_multi_fit_resident(self._mh))
        else:
            out = np.empty((m, self.n_components), dtype=dt)
            self._check(getattr(lib, f"sapca_multi_{op}_resident_{suf}")(self._mh, _np_ptr(out, ct)))
        for mem in self._members:
            mem._dtype64 = dt == np.float64
        return out

    def fit_resident(self):
        self._resident_call("fit")
        return self

    def transform_resident(self):
        return self._resident_call("transform")

    def fit_transform_resident(self):
        return self._resident_call("fit_transform")

    def resident_shard(self, i):
        """(first_row, rows, nnz) of member i's resident shard"""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        self._check(L.load().sapca_multi_resident_shard(self._mh, C.c_uint32(i), C.byref(a), C.byref(b), C.byref(c), None, None, None))
        return a.value, b.value, c.value

    def fit(self, x):
        self._resident = None
        self._call("fit", x, False)
        return self

    def transform(self, x):
        self._resident = None
        return self._call("transform", x, True)

    def fit_transform(self, x):
        self._resident = None
        return self._call("fit_transform", x, True)

    def __getattr__(self, name):
        # the fitted state is replicated: every getter of the estimator classes answers from member 0
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._members[0], name)
