#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- the committed golden vectors.

Run from the repo root in the build container:  python tests/golden/make_golden.py

What pins what:
  * ref_pins.npz   -- DATA the reference's own tests hold for column statistics
                      (the only numeric pins adjacent to this path):
                      src/sparse/csc.rs:1071-1094 matrix with sum_col == [5,3,7]
                      (:1128-1129); src/sparse/csr.rs:1385-1404 matrix with
                      nonzero_col == [2,2,2] (:1410-1412).
  * everything else is produced by INDEPENDENT implementations in this container
    (dense numpy float64 products, numpy.linalg.svd of the densified centred
    matrix, scikit-learn's randomized_svd / svd_flip, brute-force loops that
    transliterate the reference's transform code) -- never by the oracle under test,
    so the same files pin both oracle/ and the HIP path.
Nothing here reads /root/reference at run time and nothing imports the reference
(it is Rust; there is no toolchain -- SURVEY.md §8c).
"""
import os
import sys

import numpy as np
import scipy.sparse as sp
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))
from sapca import synth  # noqa: E402  (deterministic input generator only)

OUT = os.path.dirname(os.path.abspath(__file__))


def to_np(t):
    p, i, v = t
    return p.numpy().astype(np.int64), i.numpy().astype(np.int64), v.numpy()


def save(name, **kw):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **kw)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


def small_csr(m, n, density, seed, dtype, empty_rows=(), empty_cols=()):
    rng = np.random.default_rng(seed)
    D = (rng.random((m, n)) < density) * rng.uniform(-3, 3, (m, n))
    D[list(empty_rows), :] = 0
    D[:, list(empty_cols)] = 0
    A = sp.csr_matrix(D.astype(dtype))
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(dtype), D


def main():
    # ---- reference-held pins ------------------------------------------------
    save("ref_pins.npz",
         csc_dense=np.array([[1., 0, 2], [0, 3, 0], [4, 0, 5]]), csc_sum_col=np.array([5., 3, 7]),
         csc_sum_row=np.array([3., 3, 9]),
         csr_dense=np.array([[1., 0, 2], [0, 0, 0], [3, 4, 0], [0, 5, 6]]),
         csr_nonzero_col=np.array([2, 2, 2]), csr_nonzero_row=np.array([2, 0, 2, 2]))

    # ---- G1 column statistics, 64x48 with empty rows/cols --------------------
    ptr, idx, val, D = small_csr(64, 48, 0.2, 1, np.float64, empty_rows=(0, 17, 63), empty_cols=(3, 47))
    save("g1_colstats.npz", indptr=ptr, indices=idx, data=val, m=64, n=48,
         sum_col=D.sum(0), sum_col_sq=(D * D).sum(0), cnt=(D != 0).sum(0))

    # ---- G2 mask index maps (integers, exact) --------------------------------
    masks = {
        "all_true": np.ones(10, bool),
        "alternating": np.arange(11) % 2 == 0,
        "head_tail": np.array([1, 0, 0, 0, 0, 0, 0, 1], bool),
        "bernoulli60_seed7": synth.bernoulli_mask(257, 0.6, 7).numpy(),
    }
    g2 = {}
    for name, mk in masks.items():
        cols = [j for j in range(len(mk)) if mk[j]]                  # transliterates masked mod.rs:264-271
        o2m = [-1] * len(mk)
        for k_, c_ in enumerate(cols):                                # and the HashMap of :462-466
            o2m[c_] = k_
        g2[name + "_mask"] = mk
        g2[name + "_cols_to_use"] = np.array(cols, dtype=np.uint64)
        g2[name + "_orig_to_masked"] = np.array(o2m, dtype=np.int64)
    save("g2_masks.npz", **g2)

    # ---- G3 centred / uncentred SpMM and SpMM^T vs dense numpy ---------------
    ptr, idx, val, D = small_csr(512, 384, 0.05, 3, np.float64, empty_rows=(5, 100), empty_cols=(7,))
    mu = D.mean(0)
    g3 = dict(indptr=ptr, indices=idx, data=val, m=512, n=384, mu=mu)
    for l in (8, 30, 64):
        X = synth.gaussian_panel(384, l, 100 + l).numpy()
        Yp = synth.gaussian_panel(512, l, 200 + l).numpy()
        g3[f"X{l}"] = X
        g3[f"Yin{l}"] = Yp
        g3[f"AX{l}"] = D @ X
        g3[f"AcX{l}"] = (D - mu[None, :]) @ X
        g3[f"AtY{l}"] = D.T @ Yp
        g3[f"ActY{l}"] = (D - mu[None, :]).T @ Yp
    save("g3_spmm.npz", **g3)

    # ---- G4 randomized fit with injected Omega (independent: sklearn on the dense centred matrix)
    from sklearn.utils.extmath import svd_flip
    m, n, k, p, q = 1500, 400, 8, 6, 2
    ptr, idx, val = to_np(synth.gapped_csr(m, n, 0.08, k, seed=5, dtype=torch.float64))
    Dm = sp.csr_matrix((val, idx, ptr), shape=(m, n)).toarray()
    mu = Dm.mean(0)
    Ac = Dm - mu[None, :]
    omega = synth.gaussian_panel(n, k + p, 77).numpy()
    # scikit-learn's range finder + projection, written out with the injected Omega
    # (extmath.py:287-353 with power_iteration_normalizer="QR", then :560-590).
    Q = omega.copy()
    for _ in range(q):
        Q, _ = np.linalg.qr(Ac @ Q)
        Q, _ = np.linalg.qr(Ac.T @ Q)
    Q, _ = np.linalg.qr(Ac @ Q)
    Uh, s, Vt = np.linalg.svd(Q.T @ Ac, full_matrices=False)
    U, Vt = svd_flip(Q @ Uh, Vt, u_based_decision=False)
    s, Vt = s[:k], Vt[:k]
    ev = s ** 2 / (m - 1)
    # uncentred variant of the same
    Q2 = omega.copy()
    for _ in range(q):
        Q2, _ = np.linalg.qr(Dm @ Q2)
        Q2, _ = np.linalg.qr(Dm.T @ Q2)
    Q2, _ = np.linalg.qr(Dm @ Q2)
    Uh2, s2, Vt2 = np.linalg.svd(Q2.T @ Dm, full_matrices=False)
    _, Vt2 = svd_flip(Q2 @ Uh2, Vt2, u_based_decision=False)
    exact = np.linalg.svd(Ac, full_matrices=False)
    save("g4_randomized_fit.npz", indptr=ptr, indices=idx, data=val, m=m, n=n, k=k, p=p, q=q,
         omega=omega, mean=mu, s=s, vt=Vt, ev=ev, ratio=ev / ev.sum(), cum=np.cumsum(ev / ev.sum()),
         s_uncentred=s2[:k], vt_uncentred=Vt2[:k], exact_s=exact[1][:k + 2], exact_vt=exact[2][:k])

    # ---- G5 gapped generator at C1 scale: exact SVD of the densified centred matrix
    m, n, k = 10000, 2000, 20
    ptr, idx, val = to_np(synth.gapped_csr(m, n, 0.05, k, seed=42, dtype=torch.float64))
    Dm = sp.csr_matrix((val, idx, ptr), shape=(m, n)).toarray()
    Ac = Dm - Dm.mean(0)[None, :]
    _, s, Vt = np.linalg.svd(Ac, full_matrices=False)
    _, Vtf = svd_flip(np.zeros((1, k)), Vt[:k].copy(), u_based_decision=False)
    ev = s[:k] ** 2 / (m - 1)
    save("g5_gapped_c1.npz", m=m, n=n, k=k, density=0.05, seed=42, nnz=len(val),
         data_checksum=np.array([val.sum(), float(idx.sum()), float(ptr.sum())]),
         exact_s=s[:k + 3], exact_vt=Vtf.astype(np.float32), ratio=ev / ev.sum())

    # ---- G6 Lanczos (uncentred, quirk Q1): exact SVD of the raw matrix --------
    m, n, k = 3000, 800, 10
    ptr, idx, val = to_np(synth.gapped_csr(m, n, 0.06, k, seed=11, centred=False, dtype=torch.float64))
    Dm = sp.csr_matrix((val, idx, ptr), shape=(m, n)).toarray()
    _, s, Vt = np.linalg.svd(Dm, full_matrices=False)
    _, Vtf = svd_flip(np.zeros((1, k)), Vt[:k].copy(), u_based_decision=False)
    mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
    _, sm, Vtm = np.linalg.svd(Dm[:, mask], full_matrices=False)
    _, Vtmf = svd_flip(np.zeros((1, k)), Vtm[:k].copy(), u_based_decision=False)
    save("g6_lanczos.npz", m=m, n=n, k=k, density=0.06, seed=11, nnz=len(val),
         exact_s=s[:k + 2], exact_vt=Vtf, mask=mask, masked_s=sm[:k + 2], masked_vt=Vtmf)

    # ---- G7 transform semantics Q2 / Q3 by brute force on 200x60 -------------
    m, n, k = 200, 60, 5
    ptr, idx, val, D = small_csr(m, n, 0.15, 9, np.float64, empty_rows=(3,), empty_cols=(11,))
    rng = np.random.default_rng(10)
    comps = rng.standard_normal((k, n))
    mean = D.mean(0)
    mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
    cols = np.flatnonzero(mask)
    comps_m = rng.standard_normal((k, len(cols)))
    g7 = dict(indptr=ptr, indices=idx, data=val, m=m, n=n, k=k, comps=comps, mean=mean,
              mask=mask, comps_masked=comps_m)
    for center in (True, False):
        # Q2: literal loop of sparse/mod.rs:268-282 -- every stored column index of the
        # whole matrix is visited for every (row, component).
        t2 = np.zeros((m, k))
        for i in range(m):
            eff = D[i, idx] - (mean[idx] if center else 0.0)       # one term per stored entry
            t2[i, :] = comps[:, idx] @ eff
        g7[f"q2_center{int(center)}"] = t2
        # Q3: literal loop of sparse_masked/mod.rs:488-529
        lut = {c: j for j, c in enumerate(cols)}
        t3 = np.zeros((m, k))
        for i in range(m):
            for e in range(ptr[i], ptr[i + 1]):
                c = idx[e]
                if c in lut:
                    eff = val[e] - mean[c] if center else val[e]
                    t3[i, :] += eff * comps_m[:, lut[c]]
        g7[f"q3_center{int(center)}"] = t3
    save("g7_transform.npz", **g7)


if __name__ == "__main__":
    main()
