"""CPU tests (-m "not gpu"): the oracle against the committed golden vectors and against
independent implementations (numpy exact SVD, scikit-learn, scipy svds), and the C
restatement against the numpy one.  Tolerances are written next to each check."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import orc
import sapca_oracle as O
from sapca import synth


def csr_np(t):
    p, i, v = t
    return p.numpy().astype(np.int64), i.numpy().astype(np.int64), v.numpy()


# ---------------------------------------------------------------- reference-held pins
def test_ref_pins_sum_col(golden):
    g = golden("ref_pins.npz")
    A = sp.csr_matrix(g["csc_dense"])
    s = O.sum_col(A.indptr, A.indices, A.data, 3)
    assert s.tolist() == g["csc_sum_col"].tolist() == [5.0, 3.0, 7.0]      # csc.rs:1128-1129
    assert orc.sum_col(A.indices, A.data, 3).tolist() == [5.0, 3.0, 7.0]
    B = sp.csr_matrix(g["csr_dense"])
    assert O.nonzero_col(B.indices, 3).tolist() == g["csr_nonzero_col"].tolist() == [2, 2, 2]  # csr.rs:1410-1412


# ---------------------------------------------------------------- G1
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-13), (np.float32, 2e-6)])
def test_g1_colstats(golden, dtype, tol):
    g = golden("g1_colstats.npz")
    ptr, idx, val, n = g["indptr"], g["indices"], g["data"].astype(dtype), int(g["n"])
    for impl in (lambda sq: (O.sum_col_squared if sq else O.sum_col)(ptr, idx, val, n),
                 lambda sq: orc.sum_col(idx, val, n, squared=sq)):
        np.testing.assert_allclose(impl(False), g["sum_col"], rtol=tol, atol=tol)
        np.testing.assert_allclose(impl(True), g["sum_col_sq"], rtol=tol, atol=tol)
    assert O.nonzero_col(idx, n).tolist() == g["cnt"].tolist()


def test_sum_col_parallel_branch_matches_serial():
    # > PARALLEL_THRESHOLD entries exercises the chunk-of-8192 branch (csr.rs:286-308)
    ptr, idx, val = csr_np(synth.flat_csr(3000, 1000, 0.1, seed=3, dtype=torch.float64))
    assert len(val) > O.PARALLEL_THRESHOLD
    np.testing.assert_allclose(orc.sum_col(idx, val, 1000), O.sum_col(ptr, idx, val, 1000), rtol=1e-12)
    np.testing.assert_allclose(orc.sum_col(idx, val, 1000, True), O.sum_col_squared(ptr, idx, val, 1000), rtol=1e-12)


def test_sum_col_empty():
    z = np.zeros(0)
    assert O.sum_col(np.zeros(1, np.int64), np.zeros(0, np.int64), z, 0).shape == (0,)
    assert O.sum_col(np.zeros(4, np.int64), np.zeros(0, np.int64), z, 5).tolist() == [0] * 5
    assert orc.sum_col(np.zeros(0, np.int64), z, 5).tolist() == [0] * 5


# ---------------------------------------------------------------- G2 (bit-exact)
def test_g2_mask_maps(golden):
    g = golden("g2_masks.npz")
    for name in ("all_true", "alternating", "head_tail", "bernoulli60_seed7"):
        cols, o2m = O.mask_index_maps(g[name + "_mask"])
        assert cols.dtype == np.uint64 and o2m.dtype == np.int64
        assert np.array_equal(cols, g[name + "_cols_to_use"])
        assert np.array_equal(o2m, g[name + "_orig_to_masked"])


def test_mask_length_mismatch_is_an_error():
    ptr, idx, val = csr_np(synth.flat_csr(20, 10, 0.3, dtype=torch.float64))
    with pytest.raises(ValueError, match="mask vector length"):
        O.fit(ptr, idx, val, 20, 10, n_components=2, mask=np.ones(9, bool))


# ---------------------------------------------------------------- G3
@pytest.mark.parametrize("l", [8, 30, 64])
def test_g3_spmm(golden, l):
    g = golden("g3_spmm.npz")
    ptr, idx, val, m, n, mu = g["indptr"], g["indices"], g["data"], int(g["m"]), int(g["n"]), g["mu"]
    A = sp.csr_matrix((val, idx, ptr), shape=(m, n))
    X, Yin = g[f"X{l}"], g[f"Yin{l}"]
    np.testing.assert_allclose(O.spmm_centered(A, X), g[f"AX{l}"], atol=1e-11)
    np.testing.assert_allclose(O.spmm_centered(A, X, mu), g[f"AcX{l}"], atol=1e-11)
    np.testing.assert_allclose(O.spmmt_centered(A, Yin), g[f"AtY{l}"], atol=1e-11)
    np.testing.assert_allclose(O.spmmt_centered(A, Yin, mu), g[f"ActY{l}"], atol=1e-11)
    np.testing.assert_allclose(orc.spmm(ptr, idx, val, m, X, mu @ X), g[f"AcX{l}"], atol=1e-11)
    np.testing.assert_allclose(orc.spmmt(ptr, idx, val, m, n, Yin, mu), g[f"ActY{l}"], atol=1e-10)


# ---------------------------------------------------------------- G4
def test_g4_randomized_fit_injected_omega(golden):
    g = golden("g4_randomized_fit.npz")
    ptr, idx, val = g["indptr"], g["indices"], g["data"]
    m, n, k, p, q = (int(g[x]) for x in "mnkpq")
    r = O.fit(ptr, idx, val, m, n, n_components=k, n_oversamples=p, n_power_iterations=q,
              omega=g["omega"])
    np.testing.assert_allclose(r.mean, g["mean"], atol=1e-13)
    np.testing.assert_allclose(r.singular_values, g["s"], rtol=1e-10)
    np.testing.assert_allclose(r.components, g["vt"], atol=1e-9)           # signs included (svd_flip)
    np.testing.assert_allclose(r.explained_variance, g["ev"], rtol=1e-10)
    np.testing.assert_allclose(O.explained_variance_ratio(r.explained_variance), g["ratio"], rtol=1e-10)
    np.testing.assert_allclose(O.cumulative_explained_variance_ratio(r.explained_variance), g["cum"], rtol=1e-10)
    assert abs(O.explained_variance_ratio(r.explained_variance).sum() - 1) < 1e-12   # Q4
    # uncentred
    r2 = O.fit(ptr, idx, val, m, n, n_components=k, n_oversamples=p, n_power_iterations=q,
               omega=g["omega"], center=False)
    np.testing.assert_allclose(r2.singular_values, g["s_uncentred"], rtol=1e-10)
    np.testing.assert_allclose(r2.components, g["vt_uncentred"], atol=1e-9)
    # C restatement: same Omega, Householder + Jacobi instead of LAPACK
    rc, comps, sing, ev, mean, tv = orc.randomized_fit(ptr, idx, val, m, n, k, p, q, "QR", True, g["omega"])
    assert rc == 0
    np.testing.assert_allclose(sing, g["s"], rtol=1e-9)
    assert O.subspace_angle(comps, g["vt"]) < 1e-7
    np.testing.assert_allclose(comps, g["vt"], atol=1e-7)
    np.testing.assert_allclose(tv, r.total_var, rtol=1e-10)
    # f32 path, tolerance 1e-4 relative
    r32 = O.fit(ptr, idx, val.astype(np.float32), m, n, n_components=k, n_oversamples=p,
                n_power_iterations=q, omega=g["omega"].astype(np.float32))
    np.testing.assert_allclose(r32.singular_values, g["s"], rtol=1e-4)
    assert O.subspace_angle(r32.components, g["vt"]) < 1e-4
    rc, comps32, sing32, *_ = orc.randomized_fit(ptr, idx, val.astype(np.float32), m, n, k, p, q, "QR", True,
                                                 g["omega"].astype(np.float32))
    np.testing.assert_allclose(sing32, g["s"], rtol=1e-4)
    assert O.subspace_angle(comps32, g["vt"]) < 1e-4


@pytest.mark.parametrize("normalizer", ["LU", "NONE"])
def test_normalizers_span_the_same_space(golden, normalizer):
    g = golden("g4_randomized_fit.npz")
    ptr, idx, val = g["indptr"], g["indices"], g["data"]
    m, n, k, p, q = (int(g[x]) for x in "mnkpq")
    r = O.fit(ptr, idx, val, m, n, n_components=k, n_oversamples=p, n_power_iterations=q,
              omega=g["omega"], normalizer=normalizer)
    np.testing.assert_allclose(r.singular_values, g["s"], rtol=1e-7)
    assert O.subspace_angle(r.components, g["vt"]) < 1e-6
    rc, comps, sing, *_ = orc.randomized_fit(ptr, idx, val, m, n, k, p, q, normalizer, True, g["omega"])
    np.testing.assert_allclose(sing, g["s"], rtol=1e-7)
    assert O.subspace_angle(comps, g["vt"]) < 1e-6


def test_oracle_vs_sklearn_randomized_svd(golden):
    """Same Omega distribution is not reproducible across RNGs, so compare to sklearn on the
    quantity that is RNG-independent on a gapped input: the converged top-k subspace."""
    from sklearn.utils.extmath import randomized_svd
    g = golden("g4_randomized_fit.npz")
    ptr, idx, val = g["indptr"], g["indices"], g["data"]
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    Ac = sp.csr_matrix((val, idx, ptr), shape=(m, n)).toarray()
    Ac -= Ac.mean(0)
    _, s_sk, vt_sk = randomized_svd(Ac, k, n_oversamples=10, n_iter=12, power_iteration_normalizer="QR",
                                    flip_sign=True, random_state=0)
    r = O.fit(ptr, idx, val, m, n, n_components=k, n_oversamples=10, n_power_iterations=12, seed=1)
    np.testing.assert_allclose(r.singular_values, s_sk, rtol=1e-6)
    assert O.subspace_angle(r.components, vt_sk) < 1e-4
    assert O.subspace_angle(r.components, g["exact_vt"]) < 1e-4


# ---------------------------------------------------------------- G5
def test_g5_gapped_c1_vs_exact(golden):
    g = golden("g5_gapped_c1.npz")
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, float(g["density"]), k, seed=int(g["seed"]), dtype=torch.float64))
    assert len(val) == int(g["nnz"])                                      # generator is deterministic
    np.testing.assert_allclose([val.sum(), float(idx.sum()), float(ptr.sum())], g["data_checksum"], rtol=1e-12)
    om = synth.gaussian_panel(n, k + 10, 42).numpy()
    r = O.fit(ptr, idx, val, m, n, n_components=k, n_oversamples=10, n_power_iterations=4, omega=om)
    assert g["exact_s"][k - 1] / g["exact_s"][k] > 2.5                    # the planted gap
    assert O.subspace_angle(r.components, g["exact_vt"]) < 1e-4           # north-star tolerance
    np.testing.assert_allclose(O.explained_variance_ratio(r.explained_variance), g["ratio"], atol=1e-6)
    r32 = O.fit(ptr, idx, val.astype(np.float32), m, n, n_components=k, n_oversamples=10,
                n_power_iterations=4, omega=om.astype(np.float32))
    assert O.subspace_angle(r32.components, g["exact_vt"]) < 1e-4
    np.testing.assert_allclose(O.explained_variance_ratio(r32.explained_variance), g["ratio"], atol=1e-5)


# ---------------------------------------------------------------- G6
def test_g6_lanczos_uncentred(golden):
    g = golden("g6_lanczos.npz")
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, float(g["density"]), k, seed=int(g["seed"]),
                                            centred=False, dtype=torch.float64))
    assert len(val) == int(g["nnz"])
    r = O.fit(ptr, idx, val, m, n, n_components=k, method="LANCZOS", center=True)   # center ignored by SVD (Q1)
    np.testing.assert_allclose(r.singular_values, g["exact_s"][:k], rtol=1e-5)      # kappa
    assert O.subspace_angle(r.components, g["exact_vt"]) < 1e-4
    assert np.all(r.components[np.arange(k), np.argmax(np.abs(r.components), 1)] > 0)  # svd_flip
    # scipy's ARPACK as a second independent check
    from scipy.sparse.linalg import svds
    A = sp.csr_matrix((val, idx, ptr), shape=(m, n))
    s2 = np.sort(svds(A, k=k, return_singular_vectors=False))[::-1]
    np.testing.assert_allclose(r.singular_values, s2, rtol=1e-5)
    # masked
    rm = O.fit(ptr, idx, val, m, n, n_components=k, method="LANCZOS", mask=g["mask"])
    np.testing.assert_allclose(rm.singular_values, g["masked_s"][:k], rtol=1e-5)
    assert O.subspace_angle(rm.components, g["masked_vt"]) < 1e-4
    assert rm.components.shape == (k, int(g["mask"].sum()))
    assert rm.mean.shape == (n,)                                          # mean keeps FULL width (masked :275-291)


# ---------------------------------------------------------------- G7
@pytest.mark.parametrize("center", [True, False])
def test_g7_transform_semantics(golden, center):
    g = golden("g7_transform.npz")
    ptr, idx, val = g["indptr"], g["indices"], g["data"]
    m, n = int(g["m"]), int(g["n"])
    want2, want3 = g[f"q2_center{int(center)}"], g[f"q3_center{int(center)}"]
    t2 = O.transform_sparse(ptr, idx, val, m, n, g["comps"], g["mean"], center)
    np.testing.assert_allclose(t2, want2, atol=1e-10)
    np.testing.assert_allclose(orc.transform_sparse(ptr, idx, val, m, n, g["comps"], g["mean"], center), want2, atol=1e-10)
    t3 = O.transform_masked(ptr, idx, val, m, n, g["comps_masked"], g["mean"], center, g["mask"])
    np.testing.assert_allclose(t3, want3, atol=1e-11)
    np.testing.assert_allclose(O.transform_masked_fast(ptr, idx, val, m, n, g["comps_masked"], g["mean"], center, g["mask"]),
                               want3, atol=1e-11)
    _, o2m = O.mask_index_maps(g["mask"])
    np.testing.assert_allclose(orc.transform_masked(ptr, idx, val, m, g["comps_masked"], g["mean"], center, o2m),
                               want3, atol=1e-11)


def test_q2_bruteforce_equals_closed_form():
    """The literal O(m*k*nnz) loop of sparse/mod.rs:268-282 on a tiny input."""
    ptr, idx, val = csr_np(synth.flat_csr(12, 9, 0.3, seed=2, dtype=torch.float64))
    rng = np.random.default_rng(0)
    comps, mean = rng.standard_normal((3, 9)), rng.standard_normal(9)
    for center in (True, False):
        a = O.transform_sparse_bruteforce(ptr, idx, val, 12, 9, comps, mean, center)
        b = O.transform_sparse(ptr, idx, val, 12, 9, comps, mean, center)
        np.testing.assert_allclose(a, b, atol=1e-12)


# ---------------------------------------------------------------- C normalizers
def test_c_householder_and_lu():
    P = synth.gaussian_panel(300, 12, 5).numpy()
    Q = orc.householder_q(P)
    np.testing.assert_allclose(Q.T @ Q, np.eye(12), atol=1e-13)
    q_np, _ = np.linalg.qr(P)
    assert O.subspace_angle(Q.T, q_np.T) < 1e-10
    import scipy.linalg as sl
    pl, _ = sl.lu(P, permute_l=True)
    np.testing.assert_allclose(orc.lu_pl(P), pl, atol=1e-12)


def test_generator_is_offset_invariant():
    a = csr_np(synth.gapped_csr(64, 50, 0.2, 4, seed=9, dtype=torch.float64))
    b = csr_np(synth.gapped_csr(32, 50, 0.2, 4, seed=9, row_start=32, dtype=torch.float64))
    lo = a[0][32]
    assert np.array_equal(a[1][lo:], b[1]) and np.array_equal(a[2][lo:], b[2])
    assert np.array_equal(a[0][32:] - lo, b[0])
