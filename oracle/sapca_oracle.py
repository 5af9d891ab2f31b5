"""CPU oracle for the sparse-PCA hot path -- TEST INFRASTRUCTURE, NOT PRODUCT.

Only tests/, __graft_entry__.smoke() and bench.py's ``cpu_baseline`` leg may
import this module (or the C restatement next to it).  The product path
(single-algebra_amd/) never does; it fails loudly when libsapca.so is missing.

PARITY UNPINNED.  The reference's arithmetic for this path lives in the
third-party crate ``single-svdlib = 1.0.9`` (Cargo.toml:37, Cargo.lock:1393-1409),
which is absent from /root/reference and from this image, there is no Rust
toolchain, and the reference's only PCA test asserts ``is_ok()``
(src/dimred/pca/sparse/mod.rs:539-562).  So this oracle is a restatement of
(a) the in-tree semantics, line by line, and (b) the published algorithm the
crate implements (Halko-Martinsson-Tropp randomized SVD in scikit-learn's
formulation, which the reference README:169 names as its inspiration; SVDLIBC
las2 for Lanczos).  It is cross-checked in tests/ against independent
implementations (numpy exact SVD, sklearn.utils.extmath.randomized_svd /
svd_flip, scipy.sparse.linalg.svds) and against the two data fixtures the
reference's own tests hold for column sums (src/sparse/csc.rs:1071-1094,1123-1129;
src/sparse/csr.rs:1385-1404).  The preprocessing / statistics functions at the end of this file ARE pinned by
reference-held test data (tests/golden/ref_pins_preproc.npz).

Every function cites the reference lines it restates (paths relative to
/root/reference).
"""
from __future__ import annotations

import numpy as np
import scipy.sparse as sp

# --------------------------------------------------------------------------
# R1 / R2  column sums                      src/sparse/csr.rs:259-312, 558-608
# --------------------------------------------------------------------------
PARALLEL_THRESHOLD = 200_000  # src/sparse/csr.rs:19


def sum_col(indptr, indices, data, n_cols, dtype=None):
    """s_j = sum_i a_ij, accumulated in T like the reference (csr.rs:273-284 serial
    branch; the Rayon branch :286-308 adds the same numbers in a scheduler-dependent
    order, so only agreement to rounding is defined)."""
    dtype = np.dtype(dtype or data.dtype)
    out = np.zeros(n_cols, dtype=dtype)
    if len(data) == 0 or n_cols == 0:       # csr.rs:268-270
        return out
    np.add.at(out, indices, data.astype(dtype))
    return out


def sum_col_squared(indptr, indices, data, n_cols, dtype=None):
    """sum_i a_ij^2 (csr.rs:558-608)."""
    dtype = np.dtype(dtype or data.dtype)
    out = np.zeros(n_cols, dtype=dtype)
    if len(data) == 0 or n_cols == 0:
        return out
    v = data.astype(dtype)
    np.add.at(out, indices, v * v)
    return out


def nonzero_col(indices, n_cols):
    """cnt_j = stored entries in column j (needed by the Q2 transform restatement)."""
    return np.bincount(indices, minlength=n_cols).astype(np.int64)


# --------------------------------------------------------------------------
# R3  mean / total variance                 sparse/mod.rs:106-131; masked :273-311
# --------------------------------------------------------------------------
def mean_and_total_var(indptr, indices, data, m, n, center, cols_to_use=None):
    T = data.dtype
    if center:
        s = sum_col(indptr, indices, data, n)                      # :107 / :279
        mean = (s / T.type(m)).astype(T)                           # :108-114
        s2 = sum_col(indptr, indices, data, n)                     # :121 / :299 (second pass)
        sq = sum_col_squared(indptr, indices, data, n)             # :122 / :300
        cols = range(n) if cols_to_use is None else cols_to_use    # :126 / :303
        tv = T.type(0)
        for j in cols:
            mj = s2[j] / T.type(m)
            tv += (sq[j] - mj * s2[j]) / T.type(m - 1)             # :127-129
        return mean, tv
    # sparse/mod.rs:116 allocates zeros(n_samples) (wrong length, never read);
    # sparse_masked/mod.rs:291 zeros(ncols).  Either way the mean is unused.
    return np.zeros(n, dtype=T), T.type(0)


# --------------------------------------------------------------------------
# R5 / R6  mask index maps                  sparse_masked/mod.rs:264-271, :313, :455-466
# --------------------------------------------------------------------------
def mask_index_maps(mask):
    """cols_to_use ascending (mod.rs:264-271) and orig->masked (-1 = dropped),
    the HashMap of :462-466 as a dense table.  Integer, bit-exact."""
    mask = np.asarray(mask, dtype=bool)
    cols_to_use = np.flatnonzero(mask).astype(np.uint64)
    orig_to_masked = np.full(mask.shape[0], -1, dtype=np.int64)
    orig_to_masked[cols_to_use.astype(np.int64)] = np.arange(cols_to_use.shape[0], dtype=np.int64)
    return cols_to_use, orig_to_masked


def masked_csr(indptr, indices, data, n, mask):
    """MaskedCSRMatrix::new (call site sparse_masked/mod.rs:313): the operator over
    the kept columns renumbered 0..n' in ascending order."""
    cols_to_use, o2m = mask_index_maps(mask)
    keep = o2m[indices] >= 0
    new_idx = o2m[indices][keep].astype(np.int64)
    new_val = data[keep]
    rows = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))[keep]
    cnt = np.bincount(rows, minlength=len(indptr) - 1)
    new_ptr = np.zeros(len(indptr), dtype=np.int64)
    new_ptr[1:] = np.cumsum(cnt)
    return new_ptr, new_idx, new_val, int(cols_to_use.shape[0])


# --------------------------------------------------------------------------
# R8 / R9  implicitly centred products (inside randomized_svd; call sites
#          sparse/mod.rs:170-180, sparse_masked/mod.rs:341-351)
# --------------------------------------------------------------------------
def _csr(indptr, indices, data, m, n):
    return sp.csr_matrix((data, indices, indptr), shape=(m, n))


def spmm_centered(A, X, mu=None):
    """Y = (A - 1 mu^T) X = A X - 1 (mu^T X)."""
    Y = A @ X
    if mu is not None:
        Y = Y - (mu @ X)[None, :]
    return Y


def spmmt_centered(A, Y, mu=None):
    """Z = (A - 1 mu^T)^T Y = A^T Y - mu (1^T Y)."""
    Z = A.T @ Y
    if mu is not None:
        Z = Z - np.outer(mu, Y.sum(axis=0))
    return Z


# --------------------------------------------------------------------------
# R10 normalizer                            pca/mod.rs:41 (re-export), README.md:64
# --------------------------------------------------------------------------
def normalize_panel(Y, normalizer):
    if normalizer == "QR":
        q, _ = np.linalg.qr(Y)
        return q
    if normalizer == "LU":
        import scipy.linalg as sl
        pl, _ = sl.lu(Y, permute_l=True)
        return pl
    if normalizer == "NONE":
        return Y
    raise ValueError(normalizer)


# --------------------------------------------------------------------------
# R7 / R11  randomized SVD                  call sites sparse/mod.rs:170-180
# --------------------------------------------------------------------------
def randomized_svd(A, k, n_oversamples, n_power_iterations, normalizer="QR",
                   mean=None, omega=None, seed=42):
    """Halko-Martinsson-Tropp Alg. 4.4 + 5.1 as written in scikit-learn 1.7.2
    extmath.py:287-353 / 374-590, with the centring folded into the products.

    ``omega`` (n x l) injects the Gaussian test matrix; when None a numpy
    Generator seeded with ``seed`` is used (the reference's rand-0.9 stream is not
    reproducible offline)."""
    m, n = A.shape
    l = k + n_oversamples
    T = A.dtype
    if omega is None:
        omega = np.random.default_rng(seed).standard_normal((n, l)).astype(T)
    Q = omega.astype(T)
    for _ in range(n_power_iterations):
        Q = normalize_panel(spmm_centered(A, Q, mean), normalizer)
        Q = normalize_panel(spmmt_centered(A, Q, mean), normalizer)
    Q, _ = np.linalg.qr(spmm_centered(A, Q, mean))
    B = spmmt_centered(A, Q, mean).T               # l x n  (= Q^T Ac)
    Uh, s, Vt = np.linalg.svd(B, full_matrices=False)
    U = Q @ Uh
    return U[:, :k], s[:k], Vt[:k, :]


# --------------------------------------------------------------------------
# R13 svd_flip(u, vt, u_based_decision=false)   sparse/mod.rs:201-206
# --------------------------------------------------------------------------
def svd_flip_v(u, vt):
    """Sign of the largest-|.| entry of each vt row (first index on ties) made positive."""
    idx = np.argmax(np.abs(vt), axis=1)
    signs = np.where(vt[np.arange(vt.shape[0]), idx] < 0, -1.0, 1.0).astype(vt.dtype)
    vt = vt * signs[:, None]
    if u is not None:
        u = u * signs[None, :]
    return u, vt


# --------------------------------------------------------------------------
# R12 Lanczos (las2 semantics: raw matrix, no centring -- quirk Q1)
#     call sites sparse/mod.rs:134-144, sparse_masked/mod.rs:316-331
# --------------------------------------------------------------------------
def lanczos_svd(A, k, kappa=1e-5, max_steps=None, seed=42):
    """Single-vector Lanczos on A^T A with full re-orthogonalisation, Ritz values of
    the tridiagonal, accepted when the residual bound <= kappa*|theta| (SVDLIBC las2
    acceptance test).  Any converged result is the acceptance oracle."""
    m, n = A.shape
    A = A.astype(np.float64)
    max_steps = min(n, max_steps or max(4 * k + 40, 100))
    rng = np.random.default_rng(seed)
    v = rng.standard_normal(n)
    v /= np.linalg.norm(v)
    V = np.zeros((n, max_steps + 1))
    V[:, 0] = v
    alpha, beta = [], []
    for j in range(max_steps):
        w = A.T @ (A @ V[:, j])
        a = float(w @ V[:, j])
        w -= a * V[:, j]
        if j > 0:
            w -= beta[-1] * V[:, j - 1]
        for _ in range(2):
            w -= V[:, : j + 1] @ (V[:, : j + 1].T @ w)
        b = float(np.linalg.norm(w))
        alpha.append(a)
        beta.append(b)
        steps = j + 1
        if b < 1e-300:
            break
        V[:, j + 1] = w / b
        if steps >= k and (steps % 5 == 0 or steps == max_steps):
            Tm = np.diag(alpha) + np.diag(beta[:-1], 1) + np.diag(beta[:-1], -1)
            th, S = np.linalg.eigh(Tm)
            bnd = np.abs(b * S[-1, :])
            top = np.argsort(th)[::-1][:k]
            if np.all(bnd[top] <= kappa * np.abs(th[top])):
                break
    Tm = np.diag(alpha) + np.diag(beta[:-1], 1) + np.diag(beta[:-1], -1)
    th, S = np.linalg.eigh(Tm)
    top = np.argsort(th)[::-1][:k]
    Vk = V[:, :steps] @ S[:, top]
    s = np.sqrt(np.maximum(th[top], 0.0))
    U = (A @ Vk) / np.where(s > 0, s, 1.0)[None, :]
    return U, s, Vk.T


# --------------------------------------------------------------------------
# R4 / R5 / R14  fit                        sparse/mod.rs:102-242; masked :255-419
# --------------------------------------------------------------------------
class FitResult:
    __slots__ = ("components", "explained_variance", "mean", "singular_values",
                 "total_var", "cols_to_use", "orig_to_masked", "n_features")


def fit(indptr, indices, data, m, n, *, n_components, method="RANDOM", n_oversamples=10,
        n_power_iterations=4, normalizer="QR", center=True, seed=42, mask=None,
        omega=None):
    T = data.dtype
    res = FitResult()
    if mask is not None:
        mask = np.asarray(mask, dtype=bool)
        if mask.shape[0] != n:                                      # masked :258-262
            raise ValueError("The mask vector length and the number of features (columns) have to be the same!")
        res.cols_to_use, res.orig_to_masked = mask_index_maps(mask)
        cols = res.cols_to_use.astype(np.int64)
    else:
        res.cols_to_use, res.orig_to_masked = None, None
        cols = None
    res.mean, res.total_var = mean_and_total_var(indptr, indices, data, m, n, center, cols)
    if mask is not None:
        ptr2, idx2, val2, n_used = masked_csr(indptr, indices, data, n, mask)   # :313
        A = _csr(ptr2, idx2, val2, m, n_used)
        mu = res.mean[cols] if center else None
    else:
        A = _csr(indptr, indices, data, m, n)
        n_used = n
        mu = res.mean if center else None
    res.n_features = n_used
    if method == "LANCZOS":
        # raw operator: no centring on this branch (Q1; sparse/mod.rs:134-143, masked :316-331)
        u, s, vt = lanczos_svd(A, n_components, kappa=10e-6, seed=seed)
        u, s, vt = u.astype(T), s.astype(T), vt.astype(T)
    else:
        u, s, vt = randomized_svd(A, n_components, n_oversamples, n_power_iterations,
                                  normalizer, mu, omega, seed)      # :170-180
    u, vt = svd_flip_v(u, vt)                                       # :201-206
    if s.shape[0] < n_components:                                   # s[i] would panic, :213-215
        raise RuntimeError("SVD computation failed: fewer singular values than n_components")
    res.components = vt                                             # :208
    res.singular_values = s
    res.explained_variance = (s[:n_components] ** 2 / T.type(m - 1)).astype(T)   # :210-216
    if not center:                                                  # :218-223
        res.total_var = T.type(np.sum(s ** 2 / T.type(m - 1)))
    return res


def explained_variance_ratio(ev):
    """ratio_i = ev_i / sum over the k computed components (Q4; sparse/mod.rs:312-322)."""
    return ev / ev.sum()


def cumulative_explained_variance_ratio(ev):
    """sparse/mod.rs:333-343: running sum in T."""
    r = explained_variance_ratio(ev)
    out = np.zeros_like(r)
    s = r.dtype.type(0)
    for i, x in enumerate(r):
        s += x
        out[i] = s
    return out


def feature_importances(components):
    """components^2 (sparse/mod.rs:295-302)."""
    return components * components


# --------------------------------------------------------------------------
# R15  SparsePCA::transform (quirk Q2)      sparse/mod.rs:255-285
# --------------------------------------------------------------------------
def transform_sparse_bruteforce(indptr, indices, data, m, n, components, mean, center):
    """Literal triple loop of sparse/mod.rs:268-282 (tiny inputs only): for each row
    and component, iterate over the WHOLE matrix's col_indices() and look the entry
    up in the row (stored value or zero)."""
    k = components.shape[0]
    out = np.zeros((m, k), dtype=data.dtype)
    dense = _csr(indptr, indices, data, m, n).toarray()
    for i in range(m):
        for kk in range(k):
            score = data.dtype.type(0)
            for c in indices:                      # x.col_indices(): every stored entry
                val = dense[i, c]
                eff = val - mean[c] if center else val
                score += eff * components[kk, c]
            out[i, kk] = score
    return out


def transform_sparse(indptr, indices, data, m, n, components, mean, center):
    """Closed form of the loop above: t_ik = sum_j cnt_j (x_ij - [center] mu_j) V_kj."""
    cnt = nonzero_col(indices, n).astype(data.dtype)
    W = (components * cnt[None, :]).T              # n x k
    A = _csr(indptr, indices, data, m, n)
    return spmm_centered(A, W, mean if center else None).astype(data.dtype)


# --------------------------------------------------------------------------
# R16  MaskedSparsePCA::transform (quirk Q3)   sparse_masked/mod.rs:438-546
# --------------------------------------------------------------------------
def transform_masked(indptr, indices, data, m, n, components, mean, center, mask):
    """t_ik = sum over STORED entries j of row i with mask[j]:
    (a_ij - [center] mu_j) * V[k, idx(j)]   (:488-529)."""
    mask = np.asarray(mask, dtype=bool)
    if mask.shape[0] != n:                                          # :440-444
        raise ValueError("The mask vector length and the number of features (columns) have to be the same!")
    _, o2m = mask_index_maps(mask)
    k = components.shape[0]
    out = np.zeros((m, k), dtype=data.dtype)
    for i in range(m):
        for e in range(indptr[i], indptr[i + 1]):
            c = indices[e]
            mi = o2m[c]
            if mi < 0:
                continue
            eff = data[e] - mean[c] if center else data[e]
            out[i, :] += eff * components[:, mi]
    return out


def transform_masked_fast(indptr, indices, data, m, n, components, mean, center, mask):
    """Vectorised form of transform_masked for larger inputs."""
    ptr2, idx2, val2, n_used = masked_csr(indptr, indices, data, n, mask)
    cols = np.flatnonzero(np.asarray(mask, dtype=bool))
    v = val2 - mean[cols][idx2] if center else val2
    A = _csr(ptr2, idx2, v.astype(data.dtype), m, n_used)
    return (A @ components.T).astype(data.dtype)


# --------------------------------------------------------------------------
# metrics used by the parity tests
# --------------------------------------------------------------------------
def subspace_angle(V1t, V2t):
    """Largest principal angle between the row spaces of V1t and V2t (k x n each)."""
    q1, _ = np.linalg.qr(np.asarray(V1t, dtype=np.float64).T)
    q2, _ = np.linalg.qr(np.asarray(V2t, dtype=np.float64).T)
    # sin-based formula: accurate for small angles (cos-based loses them below 1e-8)
    r = q2 - q1 @ (q1.T @ q2)
    s = np.linalg.svd(r, compute_uv=False)
    return float(np.arcsin(min(1.0, s.max())))

# --------------------------------------------------------------------------
# Preprocessing and the remaining CSR statistics.  PINNED by the reference's own test
# data: src/sparse/csr.rs:1516-1552 (normalize), :1385-1422 (nonzero_col/row), src/sparse/csc.rs:1071-1226
# (sum_row/col, min_max) -- tests/golden/ref_pins_preproc.npz.
# --------------------------------------------------------------------------
ROW, COLUMN = 0, 1


def normalize_csr(indptr, indices, data, sums, target, direction):
    """Normalize<T> for CsrMatrix, src/sparse/csr.rs:1012-1066 with U = f64: scale_i = target / sums_i where
    sums_i > 0 else 0 (:1019-1029); value = T(U(value) * scale) where scale > 0 (:1037-1041, :1052-1059)."""
    sums = np.asarray(sums, dtype=np.float64)
    scale = np.where(sums > 0, float(target) / np.where(sums > 0, sums, 1.0), 0.0)
    out = np.array(data, copy=True)
    if int(direction) == COLUMN:
        sc = scale[np.asarray(indices)]
    else:
        sc = np.repeat(scale, np.diff(np.asarray(indptr)))
    hit = sc > 0
    out[hit] = (out[hit].astype(np.float64) * sc[hit]).astype(out.dtype)
    return out


def log1p_csr(data):
    """Log1P for CsrMatrix, src/sparse/csr.rs:1069-1078: value = (1 + value).ln() in T"""
    data = np.asarray(data)
    return np.log((data.dtype.type(1) + data).astype(data.dtype)).astype(data.dtype)


def stats_csr(indptr, indices, data, m, n, direction):
    """(sum, sum_squared, nonzero, min, max) per row or per column: sum_row/col (csr.rs:259-392),
    sum_*_squared (:558-630), nonzero_row/col (:23-134: stored entries), min_max_row/col (:917-1008: over
    the stored entries, empty rows/columns keep Item::max_value()/min_value() = (T::MAX, -T::MAX))."""
    indptr, indices, data = np.asarray(indptr), np.asarray(indices), np.asarray(data)
    ln = n if int(direction) == COLUMN else m
    key = indices if int(direction) == COLUMN else np.repeat(np.arange(m), np.diff(indptr))
    d64 = data.astype(np.float64)
    sm = np.bincount(key, weights=d64, minlength=ln)
    sq = np.bincount(key, weights=d64 * d64, minlength=ln)
    nz = np.bincount(key, minlength=ln).astype(np.uint64)
    big = np.finfo(data.dtype).max
    lo = np.full(ln, big, dtype=data.dtype)
    hi = np.full(ln, -big, dtype=data.dtype)
    np.minimum.at(lo, key, data)
    np.maximum.at(hi, key, data)
    return sm, sq, nz, lo, hi


def variance_from_sums(sm, sq, N):
    """var_col / var_row, src/sparse/csr.rs:632-726: mean = sum/N; (sumsq/N - mean^2) * N/(N-1); 0 when N <= 1"""
    N = float(N)
    if N <= 1:
        return np.zeros_like(np.asarray(sm, dtype=np.float64))
    mean = np.asarray(sm, dtype=np.float64) / N
    return (np.asarray(sq, dtype=np.float64) / N - mean ** 2) * (N / (N - 1.0))
