"""ctypes access to oracle/build/liborc.so (the C/OpenMP restatement).

TEST INFRASTRUCTURE, NOT PRODUCT -- same rules as sapca_oracle.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "build", "liborc.so")


def build(force=False):
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "csrc", "sapca_oracle.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        # OpenMP's default is every CPU the machine shows; a GPU box hands a job a share of them (16 per GPU).  Hundreds of
        # threads on 16 cores turn the restatement's many short parallel regions (the Jacobi rotations, the Householder
        # columns) into minutes of barrier waits: the default here is the share this process may actually run on.
        try:
            share = len(os.sched_getaffinity(0))
        except AttributeError:
            share = os.cpu_count() or 1
        _lib.orc_set_num_threads(C.c_int(max(1, min(share, int(os.environ.get("ORC_MAX_THREADS", "16"))))))
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def _suf(dtype):
    return {"float32": ("f32", C.c_float), "float64": ("f64", C.c_double)}[np.dtype(dtype).name]


def _u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


def num_threads():
    return int(lib().orc_num_threads())


def set_num_threads(n):
    lib().orc_set_num_threads(C.c_int(int(n)))


def sum_col(indices, data, n, squared=False):
    suf, ct = _suf(data.dtype)
    out = np.zeros(n, dtype=data.dtype)
    idx = _u64(indices)
    getattr(lib(), f"orc_sum_col_{suf}")(C.c_uint64(len(data)), _p(idx, C.c_uint64), _p(data, ct),
                                         C.c_uint64(n), C.c_int(int(squared)), _p(out, ct))
    return out


def randomized_fit(indptr, indices, data, m, n, k, p, q, normalizer, center, omega):
    suf, ct = _suf(data.dtype)
    T = data.dtype
    ptr, idx = _u64(indptr), _u64(indices)
    omega = np.ascontiguousarray(omega, dtype=T)
    comps = np.zeros((k, n), dtype=T)
    sing = np.zeros(k, dtype=T)
    ev = np.zeros(k, dtype=T)
    mean = np.zeros(n, dtype=T)
    tv = np.zeros(1, dtype=T)
    norm = {"QR": 0, "LU": 1, "NONE": 2}[normalizer]
    rc = getattr(lib(), f"orc_randomized_fit_{suf}")(
        C.c_uint64(m), C.c_uint64(n), _p(ptr, C.c_uint64), _p(idx, C.c_uint64), _p(data, ct),
        C.c_uint64(k), C.c_uint64(p), C.c_uint64(q), C.c_int(norm), C.c_int(int(center)),
        _p(omega, ct), _p(comps, ct), _p(sing, ct), _p(ev, ct), _p(mean, ct), _p(tv, ct))
    return rc, comps, sing, ev, mean, tv[0]


def spmm(indptr, indices, data, m, X, c=None):
    suf, ct = _suf(data.dtype)
    ptr, idx = _u64(indptr), _u64(indices)
    X = np.ascontiguousarray(X, dtype=data.dtype)
    l = X.shape[1]
    Y = np.zeros((m, l), dtype=data.dtype)
    cp = _p(np.ascontiguousarray(c, dtype=data.dtype), ct) if c is not None else None
    getattr(lib(), f"orc_spmm_{suf}")(C.c_uint64(m), _p(ptr, C.c_uint64), _p(idx, C.c_uint64),
                                      _p(data, ct), _p(X, ct), C.c_uint64(l), cp, _p(Y, ct))
    return Y


def spmmt(indptr, indices, data, m, n, Y, mu=None):
    suf, ct = _suf(data.dtype)
    ptr, idx = _u64(indptr), _u64(indices)
    Y = np.ascontiguousarray(Y, dtype=data.dtype)
    l = Y.shape[1]
    Z = np.zeros((n, l), dtype=data.dtype)
    mp = _p(np.ascontiguousarray(mu, dtype=data.dtype), ct) if mu is not None else None
    getattr(lib(), f"orc_spmmt_{suf}")(C.c_uint64(m), C.c_uint64(n), _p(ptr, C.c_uint64),
                                       _p(idx, C.c_uint64), _p(data, ct), _p(Y, ct), C.c_uint64(l),
                                       mp, _p(Z, ct))
    return Z


def householder_q(P):
    suf, ct = _suf(P.dtype)
    P = np.array(P, order="C", copy=True)
    getattr(lib(), f"orc_householder_q_{suf}")(_p(P, ct), C.c_uint64(P.shape[0]), C.c_uint64(P.shape[1]))
    return P


def lu_pl(P):
    suf, ct = _suf(P.dtype)
    P = np.array(P, order="C", copy=True)
    getattr(lib(), f"orc_lu_pl_{suf}")(_p(P, ct), C.c_uint64(P.shape[0]), C.c_uint64(P.shape[1]))
    return P


def transform_masked(indptr, indices, data, m, comps, mean, center, o2m=None):
    suf, ct = _suf(data.dtype)
    ptr, idx = _u64(indptr), _u64(indices)
    comps = np.ascontiguousarray(comps, dtype=data.dtype)
    k, n_used = comps.shape
    out = np.zeros((m, k), dtype=data.dtype)
    mean = np.ascontiguousarray(mean, dtype=data.dtype)
    op = _p(np.ascontiguousarray(o2m, dtype=np.int64), C.c_int64) if o2m is not None else None
    getattr(lib(), f"orc_transform_masked_{suf}")(
        C.c_uint64(m), _p(ptr, C.c_uint64), _p(idx, C.c_uint64), _p(data, ct), op,
        C.c_uint64(n_used), C.c_uint64(k), _p(comps, ct), _p(mean, ct), C.c_int(int(center)), _p(out, ct))
    return out


def transform_sparse(indptr, indices, data, m, n, comps, mean, center):
    suf, ct = _suf(data.dtype)
    ptr, idx = _u64(indptr), _u64(indices)
    comps = np.ascontiguousarray(comps, dtype=data.dtype)
    k = comps.shape[0]
    out = np.zeros((m, k), dtype=data.dtype)
    mean = np.ascontiguousarray(mean, dtype=data.dtype)
    getattr(lib(), f"orc_transform_sparse_{suf}")(
        C.c_uint64(m), C.c_uint64(n), _p(ptr, C.c_uint64), _p(idx, C.c_uint64), _p(data, ct),
        C.c_uint64(k), _p(comps, ct), _p(mean, ct), C.c_int(int(center)), _p(out, ct))
    return out
