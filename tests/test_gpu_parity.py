"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the committed
golden vectors and the CPU oracle on the same seeded inputs.

Bars: bit-exact for the mask index maps; f64 within 1e-10-ish of the golden
vectors; f32 within 1e-4 relative / 1e-4 rad subspace angle (the north-star tolerance)."""
import os
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import sapca
import sapca_oracle as O
from sapca import _lib as L
from sapca import ops, synth
from sapca import PowerIterationNormalizer as PIN
from sapca import SVDMethod

pytestmark = pytest.mark.gpu


def csr_np(t):
    p, i, v = t
    return p.cpu().numpy().astype(np.int64), i.cpu().numpy().astype(np.int64), v.cpu().numpy()


def mat(ptr, idx, val, m, n):
    return sp.csr_matrix((val, idx, ptr), shape=(m, n))


@pytest.fixture(scope="module")
def session():
    return ops.Session()


# ------------------------------------------------------------------ reference-held pins + G1
def test_ref_pins_sum_col(golden, session):
    g = golden("ref_pins.npz")
    A = sp.csr_matrix(g["csc_dense"])
    s, sq, cnt = session.colstats(A.indptr, A.indices, A.data, 3, 3)
    assert s.tolist() == [5.0, 3.0, 7.0]                      # csc.rs:1128-1129
    B = sp.csr_matrix(g["csr_dense"])
    s, sq, cnt = session.colstats(B.indptr, B.indices, B.data, 4, 3)
    assert cnt.tolist() == [2, 2, 2]                          # csr.rs:1410-1412


@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-13), (np.float32, 2e-6)])
def test_g1_colstats(golden, session, dtype, tol):
    g = golden("g1_colstats.npz")
    s, sq, cnt = session.colstats(g["indptr"], g["indices"], g["data"].astype(dtype), int(g["m"]), int(g["n"]))
    np.testing.assert_allclose(s, g["sum_col"], rtol=tol, atol=tol)
    np.testing.assert_allclose(sq, g["sum_col_sq"], rtol=tol, atol=tol)
    assert cnt.tolist() == g["cnt"].tolist()


def _exact_column_sums(idx, val, n):
    """correctly rounded exact column sums and sums of squares (math.fsum; squares as exact fractions)"""
    from fractions import Fraction
    import math
    cols = [[] for _ in range(n)]
    for j, v in zip(idx.tolist(), val.tolist()):
        cols[j].append(v)
    s = np.array([math.fsum(c) for c in cols])
    sq = np.array([float(sum((Fraction(v) * Fraction(v) for v in c), Fraction(0))) for c in cols])
    return s, sq


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_statistics_gathered_behind_the_upload_are_the_exact_sums(session, dtype):
    """the column statistics of a host matrix are accumulated chunk by chunk behind the DMA into long integer
    accumulators -- order-independent, rounded once.  Values over 80 binades with heavy cancellation: the result equals the
    correctly rounded exact sum bit for bit; the row sums of A^T (the route of device-resident input) agree to rounding."""
    rng = np.random.default_rng(11)
    m, n = 4000, 300
    A = sp.random(m, n, density=0.05, format="csr", random_state=5, dtype=np.float64)
    A.sort_indices()
    v = rng.standard_normal(A.nnz) * np.exp2(rng.integers(-40, 41, A.nnz))
    kk = (A.nnz - 1) // 7
    v[0:7 * kk:7] = -v[1:7 * kk + 1:7]                    # cancellation between entries that may share a column
    v[5] = 0.0                                            # a stored zero counts as an entry
    if dtype == np.float32:
        v[11] = np.float32(1e-42)                         # a subnormal
    val = v.astype(dtype)
    ptr, idx = A.indptr.astype(np.int64), A.indices.astype(np.int64)
    s, sq, cnt = session.colstats(ptr, idx, val, m, n)
    want_s, want_sq = _exact_column_sums(idx, val.astype(np.float64), n)
    assert np.array_equal(s, want_s.astype(dtype))        # (f32: the exact sum rounded to f64, then to f32 on the way out)
    assert np.array_equal(sq, want_sq.astype(dtype))
    assert np.array_equal(cnt.astype(np.int64), np.bincount(idx, minlength=n))
    # the other route, reached through a matrix the accumulators refuse: one inf switches them off for that upload
    val2 = val.copy()
    val2[3] = np.inf
    s2, sq2, cnt2 = session.colstats(ptr, idx, val2, m, n)
    keep = np.ones(n, bool)
    keep[idx[3]] = False
    np.testing.assert_allclose(s2[keep], want_s[keep], rtol=1e-12 if dtype == np.float64 else 1e-6, atol=1e-30)
    assert np.isinf(s2[idx[3]]) and np.array_equal(cnt2, cnt)


def test_large_projection_comes_back_through_the_pinned_ring():
    """a projection of more than 16 MB returns to the caller's (pageable) array in 8 MB pieces through the page-locked ring,
    copied out by a few threads while the next piece's DMA runs: the same numbers as the device-resident call"""
    m, n, k, p, q = 150000, 300, 40, 8, 1
    dev = synth.gapped_csr(m, n, 0.05, 10, seed=2, dtype=torch.float32, device="cuda")
    ptr, idx, val = csr_np(dev)
    om = synth.gaussian_panel(n, k + p, 4).numpy()
    host = _builder(k, p, q).build().set_omega(om)
    t_host = host.fit_transform(mat(ptr, idx, val, m, n))
    assert t_host.nbytes > 2 * (8 << 20) and t_host.shape == (m, k)
    res = _builder(k, p, q).build().set_omega(om)
    t_dev = res.fit_transform(sapca.DeviceCsr(*dev, (m, n))).cpu().numpy()
    np.testing.assert_allclose(t_host, t_dev, atol=2e-4 * np.abs(t_dev).max())
    assert np.isfinite(t_host).all() and np.abs(t_host[-1]).max() > 0          # the last piece arrived


def test_host_fits_are_bitwise_reproducible_and_match_resident_fits(monkeypatch):
    """the statistics gathered behind the upload do not depend on the order the chunks' atomics land in: two host fits are
    bit-identical; the same matrix fitted from device-resident arrays (row sums of A^T) gives the same model to rounding"""
    m, n, k, p, q = 20000, 1500, 10, 6, 2
    dev = synth.gapped_csr(m, n, 0.04, k, seed=9, dtype=torch.float32, device="cuda")
    ptr, idx, val = csr_np(dev)
    om = synth.gaussian_panel(n, k + p, 3).numpy()
    runs = []
    for _ in range(2):
        pca = _builder(k, p, q).build().set_omega(om)
        t = pca.fit_transform(mat(ptr, idx, val, m, n))
        runs.append((pca.mean_(np.float64), pca.explained_variance_ratio(np.float64), pca.components_(np.float64), t))
    for a, b in zip(*runs):
        np.testing.assert_array_equal(a, b)
    res = _builder(k, p, q).build().set_omega(om)
    t_res = res.fit_transform(sapca.DeviceCsr(*dev, (m, n))).cpu().numpy()
    np.testing.assert_allclose(runs[0][0], res.mean_(np.float64), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(runs[0][1], res.explained_variance_ratio(np.float64), rtol=1e-5)
    assert O.subspace_angle(runs[0][2], res.components_(np.float64)) < 1e-4
    np.testing.assert_allclose(runs[0][3], t_res, atol=2e-4 * np.abs(t_res).max())


def test_colstats_large_matches_oracle(session):
    ptr, idx, val = csr_np(synth.flat_csr(3000, 1000, 0.1, seed=3, dtype=torch.float64))
    s, sq, cnt = session.colstats(ptr, idx, val, 3000, 1000)
    np.testing.assert_allclose(s, O.sum_col(ptr, idx, val, 1000), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(sq, O.sum_col_squared(ptr, idx, val, 1000), rtol=1e-12)
    assert np.array_equal(cnt.astype(np.int64), O.nonzero_col(idx, 1000))


# ------------------------------------------------------------------ G2: bit-exact integers
def test_g2_mask_maps_bit_exact(golden):
    g = golden("g2_masks.npz")
    for name in ("all_true", "alternating", "head_tail", "bernoulli60_seed7"):
        est = sapca.MaskedSparsePCABuilder.new().mask(g[name + "_mask"]).build()
        cols, o2m = est.mask_index_maps()
        assert cols.dtype == np.uint64 and o2m.dtype == np.int64
        assert np.array_equal(cols, g[name + "_cols_to_use"])
        assert np.array_equal(o2m, g[name + "_orig_to_masked"])


# ------------------------------------------------------------------ G3: the two sweeps
@pytest.mark.parametrize("l", [8, 30, 64])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-11), (np.float32, 2e-4)])
def test_g3_spmm(golden, session, l, dtype, tol):
    g = golden("g3_spmm.npz")
    ptr, idx, val = g["indptr"], g["indices"], g["data"].astype(dtype)
    m, n, mu = int(g["m"]), int(g["n"]), g["mu"].astype(dtype)
    X, Yin = g[f"X{l}"].astype(dtype), g[f"Yin{l}"].astype(dtype)
    np.testing.assert_allclose(session.spmm(ptr, idx, val, m, n, X), g[f"AX{l}"], atol=tol * 10)
    np.testing.assert_allclose(session.spmm(ptr, idx, val, m, n, X, mu), g[f"AcX{l}"], atol=tol * 10)
    np.testing.assert_allclose(session.spmm(ptr, idx, val, m, n, Yin, None, transposed=True), g[f"AtY{l}"], atol=tol * 10)
    np.testing.assert_allclose(session.spmm(ptr, idx, val, m, n, Yin, mu, transposed=True), g[f"ActY{l}"], atol=tol * 10)


@pytest.mark.parametrize("l", [110, 128])
def test_spmm_wide_panels(session, l):
    ptr, idx, val = csr_np(synth.flat_csr(700, 300, 0.05, seed=4, dtype=torch.float32))
    X = synth.gaussian_panel(300, l, 1).numpy().astype(np.float32)
    want = mat(ptr, idx, val, 700, 300).astype(np.float64) @ X.astype(np.float64)
    np.testing.assert_allclose(session.spmm(ptr, idx, val, 700, 300, X), want, atol=2e-3)


def test_spmm_edge_rows(session):
    """empty matrix rows, an all-empty matrix, a single dense row"""
    ptr = np.array([0, 0, 0, 5, 5], dtype=np.int64)
    idx = np.array([0, 1, 2, 3, 4], dtype=np.int64)
    val = np.arange(1, 6, dtype=np.float64)
    X = np.arange(5 * 3, dtype=np.float64).reshape(5, 3)
    got = session.spmm(ptr, idx, val, 4, 5, X)
    want = mat(ptr, idx, val, 4, 5) @ X
    np.testing.assert_allclose(got, want, atol=1e-12)
    z = session.spmm(np.zeros(4, np.int64), np.zeros(0, np.int64), np.zeros(0), 3, 5, X)
    assert np.all(z == 0) and z.shape == (3, 3)


def _ragged_rows_csr(n, seed, dtype):
    """rows whose lengths straddle the 16-, 32- and 64-entry chunks of the DPP-fed row kernel"""
    lens = [0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63, 64, 65, 96, 100, 127, 128, 129, 200, 257]
    rng = np.random.default_rng(seed)
    rows = [np.sort(rng.choice(n, ln, replace=False)) for _ in range(12) for ln in lens]
    ptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    idx = np.concatenate(rows).astype(np.int64)
    val = rng.standard_normal(len(idx)).astype(dtype)
    return ptr, idx, val, len(rows)


@pytest.mark.parametrize("dtype,l,tol", [(np.float32, 60, 2e-5), (np.float32, 110, 2e-5), (np.float64, 30, 1e-12), (np.float64, 60, 1e-12)])
def test_row_kernel_chunk_boundaries(session, dtype, l, tol):
    """the row kernel's two ways through a row (whole chunks fed by DPP broadcasts, the remainder entry by entry) on rows of
    0 .. 257 entries, with panels of 16 and 32 lanes per row, with and without the centring vector"""
    n = 700
    ptr, idx, val, m = _ragged_rows_csr(n, 5, dtype)
    X = synth.gaussian_panel(n, l, 3).numpy().astype(dtype)
    mu = np.random.default_rng(2).standard_normal(n).astype(dtype)
    D = mat(ptr, idx, val, m, n).toarray().astype(np.float64)
    want = D @ X.astype(np.float64)
    scale = max(1.0, float(np.abs(want).max()))
    np.testing.assert_allclose(session.spmm(ptr, idx, val, m, n, X), want, atol=tol * scale)
    want_c = (D - mu.astype(np.float64)[None, :]) @ X.astype(np.float64)
    np.testing.assert_allclose(session.spmm(ptr, idx, val, m, n, X, mu), want_c, atol=tol * 20 * scale)


@pytest.mark.parametrize("dtype,k,tol", [(np.float64, 30, 1e-10), (np.float64, 40, 1e-10), (np.float32, 40, 5e-5)])
def test_masked_projection_row_kernel_with_long_rows(debug_switches, monkeypatch, dtype, k, tol):
    """quirk Q3 through the row kernel with the mean folded in (value - mu[column] per stored, kept entry), rows of every
    length around the chunk sizes, 16 and 32 lanes per panel row; against the oracle's entry loop"""
    monkeypatch.setenv("SAPCA_Q3_ROWKERNEL", "1")
    n = 700
    ptr, idx, val, m = _ragged_rows_csr(n, 9, np.float64)
    mask = synth.bernoulli_mask(n, 0.8, 3).numpy()
    est = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask)
           .svd_method(SVDMethod.Random(8, 2, PIN.QR)).build())
    got = est.fit_transform(mat(ptr, idx, val.astype(dtype), m, n))
    comps, mean = est.components_(np.float64), est.mean_(np.float64)
    want = O.transform_masked_fast(ptr, idx, val.astype(dtype).astype(np.float64), m, n, comps, mean, True, mask)
    np.testing.assert_allclose(got, want, atol=tol * max(1.0, float(np.abs(want).max())))


# ------------------------------------------------------------------ normaliser (R10)
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-12), (np.float32, 5e-6)])
def test_normalizer_qr_orthonormal_same_span(session, dtype, tol):
    P = synth.gaussian_panel(5000, 30, 3).numpy()
    P[:, 5] = P[:, 4] * (1 + 1e-3) + 1e-3 * P[:, 5]           # ill-conditioned on purpose (cond ~ 1e3)
    P = P.astype(dtype)
    Q = session.normalize_panel(P, PIN.QR)
    np.testing.assert_allclose(Q.T.astype(np.float64) @ Q.astype(np.float64), np.eye(30), atol=tol * 20)
    q_ref, _ = np.linalg.qr(P.astype(np.float64))
    assert O.subspace_angle(Q.T, q_ref.T) < (1e-9 if dtype == np.float64 else 2e-3)
    Lp = session.normalize_panel(P, PIN.LU)
    assert O.subspace_angle(Lp.T, q_ref.T) < (1e-9 if dtype == np.float64 else 2e-3)
    assert np.array_equal(session.normalize_panel(P, PIN.NONE), P)


@pytest.mark.parametrize("l", [1, 7, 16, 17, 33, 48, 60, 64, 90])
def test_normalizer_every_block_count_of_the_cholesky(session, l):
    """l <= 64 runs the blocked one-workgroup Cholesky (1-4 blocks of 16 columns, padded), wider panels the general one"""
    P = synth.gaussian_panel(4000, l, 11).numpy().astype(np.float64)
    P *= np.logspace(0, 3, l)[None, :]                          # graded column scales
    Q = session.normalize_panel(P, PIN.QR)
    np.testing.assert_allclose(Q.T @ Q, np.eye(l), atol=2e-11)
    q_ref, _ = np.linalg.qr(P)
    assert O.subspace_angle(Q.T, q_ref.T) < 1e-9
    # R = Q^T P is upper triangular with a positive diagonal (Cholesky factor of the Gram matrix)
    Rm = Q.T @ P
    assert np.allclose(np.tril(Rm, -1), 0, atol=1e-8 * np.abs(Rm).max()) and np.all(np.diag(Rm) > 0)


def test_omega_generator_matches_host_function(session):
    om = session.generate_omega(300, 30)
    np.testing.assert_allclose(om, synth.gaussian_panel(300, 30, 42).numpy(), atol=1e-12)
    assert abs(om.mean()) < 0.05 and abs(om.std() - 1) < 0.05


# ------------------------------------------------------------------ G4: full randomized fit
# Projection against the ORACLE's components at production size.  The two sides project with their own components: what
# separates them is (i) the f32 rounding of the sweeps (a row holds ~600 entries; Q2 multiplies every feature by its
# stored-entry count, up to 6e3 -- the same factor on both sides) and (ii) the rotation of neighbouring components inside the
# subspace both sides agree on to 1e-4 rad (adjacent singular values of the gapped generator differ by 1.1 %: an error of
# 1e-6 in the small SVD turns a pair by 1e-4).  Both are of order 1e-4 of the largest score: the bound is 2e-4, the one the
# small random configurations hold (measured, round 4: all four production-size cases pass at 1e-4 -- gpurun_out/r4_oracle_tests_*).
PROJ_ATOL = float(os.environ.get("SAPCA_TEST_PROJ_ATOL", "2e-4"))


def _builder(k, p, q, norm=PIN.QR, **kw):
    b = sapca.SparsePCABuilder.new().n_components(k).random_seed(42).svd_method(SVDMethod.Random(p, q, norm))
    for key, v in kw.items():
        getattr(b, key)(v)
    return b


@pytest.mark.parametrize("dtype,srel,ang,vabs", [(np.float64, 1e-10, 1e-9, 1e-8), (np.float32, 1e-4, 1e-4, 5e-4)])
def test_g4_randomized_fit_injected_omega(golden, dtype, srel, ang, vabs):
    g = golden("g4_randomized_fit.npz")
    m, n, k, p, q = (int(g[x]) for x in "mnkpq")
    A = mat(g["indptr"], g["indices"], g["data"].astype(dtype), m, n)
    pca = _builder(k, p, q).build().set_omega(g["omega"])
    pca.fit(A)
    np.testing.assert_allclose(pca.mean_(np.float64), g["mean"], atol=1e-13 if dtype == np.float64 else 1e-7)
    np.testing.assert_allclose(pca.singular_values_(np.float64), g["s"], rtol=srel)
    assert O.subspace_angle(pca.components_(np.float64), g["vt"]) < ang
    np.testing.assert_allclose(pca.components_(np.float64), g["vt"], atol=vabs)          # signs: svd_flip
    np.testing.assert_allclose(pca.explained_variance_(np.float64), g["ev"], rtol=2 * srel)
    np.testing.assert_allclose(pca.explained_variance_ratio(np.float64), g["ratio"], rtol=2 * srel, atol=1e-7)
    np.testing.assert_allclose(pca.cumulative_explained_variance_ratio(np.float64), g["cum"], rtol=2 * srel, atol=1e-6)
    np.testing.assert_allclose(pca.feature_importances(np.float64), g["vt"] ** 2, atol=vabs)
    assert pca.components_().dtype == dtype and pca.components_().shape == (k, n)
    # uncentred
    pca2 = _builder(k, p, q, center=False).build().set_omega(g["omega"])
    pca2.fit(A)
    np.testing.assert_allclose(pca2.singular_values_(np.float64), g["s_uncentred"], rtol=srel)
    assert O.subspace_angle(pca2.components_(np.float64), g["vt_uncentred"]) < ang
    assert np.all(pca2.mean_() == 0)
    np.testing.assert_allclose(pca2.total_variance_(), pca2.explained_variance_(np.float64).sum(), rtol=1e-6)


@pytest.mark.parametrize("norm", [PIN.LU, PIN.NONE])
def test_g4_other_normalizers(golden, norm):
    g = golden("g4_randomized_fit.npz")
    m, n, k, p, q = (int(g[x]) for x in "mnkpq")
    A = mat(g["indptr"], g["indices"], g["data"], m, n)
    pca = _builder(k, p, q, norm).build().set_omega(g["omega"])
    pca.fit(A)
    np.testing.assert_allclose(pca.singular_values_(), g["s"], rtol=1e-7)
    assert O.subspace_angle(pca.components_(), g["vt"]) < 1e-6


def test_fit_with_builtin_omega_converges_to_exact(golden):
    g = golden("g5_gapped_c1.npz")
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    ptr, idx, val = synth.gapped_csr(m, n, float(g["density"]), k, seed=int(g["seed"]), dtype=torch.float32, device="cuda")
    assert val.numel() == int(g["nnz"])
    x = sapca.DeviceCsr(ptr, idx, val, (m, n))
    pca = _builder(k, 10, 4).build()
    t = pca.fit_transform(x)
    assert t.shape == (m, k) and t.is_cuda
    assert O.subspace_angle(pca.components_(np.float64), g["exact_vt"]) < 1e-4           # north-star tolerance
    np.testing.assert_allclose(pca.explained_variance_ratio(np.float64), g["ratio"], atol=1e-5)
    np.testing.assert_allclose(pca.singular_values_(np.float64), g["exact_s"][:k], rtol=1e-4)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_rank_deficient_panel_keeps_the_leading_components(dtype):
    """rank(A - mean) = 4 < l = 13: the panel's Gram matrix is singular, the Cholesky floors the dead pivots
    (reported through chol_regularised) and the leading components still match the exact SVD"""
    rng = np.random.default_rng(3)
    m, n, r, k = 1500, 200, 5, 3
    U = rng.standard_normal((m, r)) * np.array([9.0, 7.0, 5.0, 3.0, 2.0])
    W = rng.standard_normal((r, n))
    D = U @ W
    c = sp.csr_matrix(D.astype(dtype))                    # dense low-rank matrix stored as CSR
    A = mat(c.indptr, c.indices, c.data, m, n)
    pca = _builder(k, 10, 3).build()
    pca.fit(A)
    Dc = D.astype(dtype).astype(np.float64)
    Dc = Dc - Dc.mean(axis=0)
    _, s_ex, vt_ex = np.linalg.svd(Dc, full_matrices=False)
    assert np.all(np.isfinite(pca.components_(np.float64)))
    np.testing.assert_allclose(pca.singular_values_(np.float64), s_ex[:k], rtol=1e-9 if dtype == np.float64 else 2e-4)
    assert O.subspace_angle(pca.components_(np.float64), vt_ex[:k]) < (1e-7 if dtype == np.float64 else 1e-3)


# ------------------------------------------------------------------ G5 vs oracle, device entry points
@pytest.mark.parametrize("dtype,ang", [(torch.float64, 1e-8), (torch.float32, 1e-4)])
def test_g5_c1_device_path_vs_oracle(golden, dtype, ang):
    g = golden("g5_gapped_c1.npz")
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    dev = synth.gapped_csr(m, n, float(g["density"]), k, seed=int(g["seed"]), dtype=dtype, device="cuda")
    ptr, idx, val = csr_np(dev)
    om = synth.gaussian_panel(n, k + 10, 42).numpy()
    want = O.fit(ptr, idx, val.astype(np.float64), m, n, n_components=k, n_oversamples=10, n_power_iterations=4, omega=om)
    pca = _builder(k, 10, 4).build().set_omega(om)
    pca.fit(sapca.DeviceCsr(*dev, (m, n)))
    assert O.subspace_angle(pca.components_(np.float64), want.components) < ang
    np.testing.assert_allclose(pca.explained_variance_ratio(np.float64), O.explained_variance_ratio(want.explained_variance),
                               atol=1e-6)
    assert O.subspace_angle(pca.components_(np.float64), g["exact_vt"]) < 1e-4


# ------------------------------------------------------------------ G7: transform semantics
@pytest.mark.parametrize("center", [True, False])
@pytest.mark.parametrize("dtype,tol", [(np.float64, 1e-9), (np.float32, 2e-3)])
def test_g7_transform_reference_semantics(golden, center, dtype, tol):
    """Fit on the fixture matrix, then check transform against the oracle's restatement of Q2/Q3
    evaluated with the FITTED components/mean (the fixture's brute-force loops pin the oracle)."""
    g = golden("g7_transform.npz")
    m, n = int(g["m"]), int(g["n"])
    ptr, idx, val = g["indptr"], g["indices"], g["data"].astype(dtype)
    A = mat(ptr, idx, val, m, n)
    k = 5
    pca = _builder(k, 5, 2, center=center).build()
    t = pca.fit_transform(A)
    comps, mean = pca.components_(np.float64), pca.mean_(np.float64)
    want = O.transform_sparse(ptr, idx, val.astype(np.float64), m, n, comps, mean, center)      # Q2
    np.testing.assert_allclose(t, want, atol=tol * max(1.0, np.abs(want).max()))
    t_again = pca.transform(A)                                                                  # separate call, re-upload
    np.testing.assert_allclose(t_again, t, atol=tol * max(1.0, np.abs(want).max()))
    mpca = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(g["mask"]).center(center)
            .svd_method(SVDMethod.Random(5, 2, PIN.QR)).build())
    tm = mpca.fit_transform(A)
    want_m = O.transform_masked(ptr, idx, val.astype(np.float64), m, n, mpca.components_(np.float64),
                                mpca.mean_(np.float64), center, g["mask"])                       # Q3
    np.testing.assert_allclose(tm, want_m, atol=tol * max(1.0, np.abs(want_m).max()))
    assert mpca.components_().shape == (k, int(g["mask"].sum())) and mpca.mean_().shape == (n,)
    np.testing.assert_allclose(mpca.transform(A), tm, atol=tol * max(1.0, np.abs(want_m).max()))


def test_transform_centered_semantics_opt_in(golden):
    g = golden("g7_transform.npz")
    m, n = int(g["m"]), int(g["n"])
    A = mat(g["indptr"], g["indices"], g["data"], m, n)
    pca = _builder(5, 5, 2).transform_semantics(L.TRANSFORM_CENTERED).build()
    t = pca.fit_transform(A)
    want = (A.toarray() - pca.mean_()[None, :]) @ pca.components_().T
    np.testing.assert_allclose(t, want, atol=1e-9)


# ------------------------------------------------------------------ masked fit
@pytest.mark.parametrize("dtype,ang", [(np.float64, 1e-8), (np.float32, 1e-4)])
def test_masked_randomized_fit_vs_oracle(dtype, ang):
    m, n, k = 4000, 900, 8
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.06, k, seed=21, dtype=torch.float64))
    mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
    n_used = int(mask.sum())
    om = synth.gaussian_panel(n_used, k + 8, 5).numpy()
    want = O.fit(ptr, idx, val, m, n, n_components=k, n_oversamples=8, n_power_iterations=3, omega=om, mask=mask)
    est = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask)
           .svd_method(SVDMethod.Random(8, 3, PIN.QR)).build().set_omega(om))
    est.fit(mat(ptr, idx, val.astype(dtype), m, n))
    assert est.components_().shape == (k, n_used)
    assert O.subspace_angle(est.components_(np.float64), want.components) < ang
    np.testing.assert_allclose(est.singular_values_(np.float64), want.singular_values, rtol=1e-9 if dtype == np.float64 else 1e-4)
    np.testing.assert_allclose(est.mean_(np.float64), want.mean, atol=1e-12 if dtype == np.float64 else 1e-6)   # FULL width
    np.testing.assert_allclose(est.total_variance_(), want.total_var, rtol=1e-9 if dtype == np.float64 else 1e-4)
    cols, o2m = est.mask_index_maps()
    assert np.array_equal(cols, want.cols_to_use) and np.array_equal(o2m, want.orig_to_masked)


def test_masked_projection_through_the_sweep_with_stored_zeros(debug_switches, monkeypatch):
    """quirk Q3 through the fitted matrix's tile-major format: A'W - P diag(mu) W as two sweeps (the second reads every stored
    non-zero value as 1) plus the stored zeros by hand -- here a few hundred stored +0.0 / -0.0 entries, which the format
    cannot tell from its padding.  Against the oracle's entry loop, and against the row kernel on shifted values."""
    m, n, k, p, q = 9000, 1400, 12, 6, 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.05, k, seed=33, dtype=torch.float32))
    rng = np.random.default_rng(4)
    z = rng.choice(len(val), 400, replace=False)
    val = val.copy()
    val[z[:200]] = 0.0
    val[z[200:]] = -0.0
    mask = synth.bernoulli_mask(n, 0.7, 3).numpy()
    n_used = int(mask.sum())
    om = synth.gaussian_panel(n_used, k + p, 5).numpy()
    A = sp.csr_matrix((val, idx, ptr), shape=(m, n))          # (scipy keeps the stored zeros)
    assert A.nnz == len(val)
    out = {}
    for route in ("sweep", "rowkernel"):
        if route == "rowkernel":
            monkeypatch.setenv("SAPCA_Q3_ROWKERNEL", "1")
        est = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).spmm_variant(2).collect_timings(True)
               .svd_method(SVDMethod.Random(p, q, PIN.QR)).build().set_omega(om))
        out[route] = est.fit_transform(A)
        comps, mean = est.components_(np.float64), est.mean_(np.float64)
    want = O.transform_masked_fast(ptr, idx, val.astype(np.float64), m, n, comps, mean, True, mask)
    scale = np.abs(want).max()
    np.testing.assert_allclose(out["sweep"], want, atol=2e-5 * scale)
    np.testing.assert_allclose(out["rowkernel"], want, atol=2e-5 * scale)
    # the stored zeros matter at this tolerance: dropping them moves their rows by |mu_j V_kj|
    dropped = O.transform_masked_fast(*(lambda B: (B.indptr.astype(np.int64), B.indices.astype(np.int64), B.data.astype(np.float64)))(
        (lambda B: (B.eliminate_zeros(), B)[1])(A.copy())), m, n, comps, mean, True, mask)
    assert np.abs(dropped - want).max() > 2e-4 * scale          # ten times the tolerance above


def test_masked_projection_keeps_its_digits_in_well_filled_columns_of_small_spread():
    """Q3 where the two-sweep form A'W - P diag(mu) W would cancel: a third of the kept columns are 90 % filled with values
    1000 +- 1, so a stored value sits within 10 % of its column mean and the two f32 sums agree in their leading digits.
    The reference subtracts entry by entry (sparse_masked/mod.rs:488-494); the library notices such columns in its
    statistics and projects through the row kernel, which does the same: the entry loop of the oracle to 1e-5 of the
    projection's scale."""
    m, n, k, p, q = 6000, 900, 8, 6, 2
    rng = np.random.default_rng(17)
    base = sp.random(m, n, density=0.04, format="csr", random_state=8, dtype=np.float64)
    base.data = rng.uniform(0.5, 2.0, base.nnz)
    D = base.toarray()
    heavy = rng.choice(n, n // 3, replace=False)
    fill = rng.random((m, len(heavy))) < 0.9
    D[:, heavy] = np.where(fill, 1000.0 + rng.standard_normal((m, len(heavy))), 0.0)
    A = sp.csr_matrix(D.astype(np.float32))
    A.sort_indices()
    mask = synth.bernoulli_mask(n, 0.7, 3).numpy()
    assert mask[heavy].sum() > 50
    om = synth.gaussian_panel(int(mask.sum()), k + p, 5).numpy()
    est = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).spmm_variant(2)
           .svd_method(SVDMethod.Random(p, q, PIN.QR)).build().set_omega(om))
    t = est.fit_transform(A)
    comps, mean = est.components_(np.float64), est.mean_(np.float64)
    ptr, idx, val = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64)
    want = O.transform_masked_fast(ptr, idx, val, m, n, comps, mean.astype(np.float32).astype(np.float64), True, mask)
    np.testing.assert_allclose(t, want, atol=1e-5 * np.abs(want).max())


def _random_case(seed):
    rng = np.random.default_rng(1000 + seed)
    k = int(rng.integers(1, 13))
    # at least 60 rows and 40 columns per planted cluster: smaller blocks drown in the background entries
    m = int(rng.integers(60 * (k + 1), 60 * (k + 1) + 1500))   # (sizes bounded by the dense SVD that certifies the gap)
    n = int(rng.integers(40 * (k + 1), 40 * (k + 1) + 700))
    return dict(
        m=m, n=n, k=k, dens=float(rng.uniform(0.04, 0.3)), p=int(rng.integers(1, 9)), q=int(rng.integers(0, 4)),
        norm=[PIN.QR, PIN.LU, PIN.NONE][int(rng.integers(0, 3))], center=bool(rng.integers(0, 2)), masked=bool(rng.integers(0, 2)),
        dtype=[np.float32, np.float64][int(rng.integers(0, 2))], variant=[0, 1, 2][int(rng.integers(0, 3))])


def _gapped_case(c, seed, mask):
    """the case's matrix with a GUARANTEED gap: sigma_k / sigma_{k+1} >= 2 of the operator the fit sees (mask applied,
    centred if the case centres; exact dense SVD).  The generator plants k+1 clusters for centred fits and k for uncentred
    ones; where the draw still leaves the gap below 2 (sparse, few rows per cluster) the density is raised, at most three
    times.  None if nothing helped (the caller skips: the tolerance is never loosened)."""
    m, n, k = c["m"], c["n"], c["k"]
    dens = c["dens"]
    for _ in range(4):
        ptr, idx, val = csr_np(synth.gapped_csr(m, n, dens, k, seed=seed, centred=c["center"], dtype=torch.float64))
        D = mat(ptr, idx, val, m, n).toarray()
        if mask is not None:
            D = D[:, mask]
        if c["center"]:
            D = D - D.mean(axis=0)
        sv = np.linalg.svd(D, compute_uv=False)
        if k < len(sv) and sv[k - 1] >= 2.0 * sv[k]:
            return ptr, idx, val, float(sv[k - 1] / sv[k])
        dens = min(0.5, dens * 1.6)
    return None


@pytest.mark.parametrize("seed", range(48))
def test_random_configurations_against_the_oracle(seed):
    """random shapes, densities, ranks, oversampling, power iterations, normalisers, centring on / off, with and without a
    mask, f32 / f64, each sweep kernel: fit and transform against the oracle run on the same matrix with the same Omega.
    Every case has sigma_k / sigma_{k+1} >= 2 (checked on the dense operator), so f32 is held to the north-star figures:
    1e-4 relative on the singular values, 1e-4 rad subspace angle."""
    c = _random_case(seed)
    m, n, k, p, q = c["m"], c["n"], c["k"], c["p"], c["q"]
    if c["norm"] == PIN.NONE:
        q = min(q, 2)                      # un-normalised power iterations square the conditioning each round
    mask = synth.bernoulli_mask(n, 0.7, seed).numpy() if c["masked"] else None
    n_used = int(mask.sum()) if c["masked"] else n
    l = min(k + p, m, n_used)
    if k > l:
        pytest.skip("mask left fewer columns than components")
    made = _gapped_case(c, seed, mask)
    if made is None:
        pytest.skip("no spectral gap of 2 at this shape (mask / density): skipped, not loosened")
    ptr, idx, val, gap = made
    om = synth.gaussian_panel(n_used, k + p, seed + 7).numpy()
    norm_name = {PIN.QR: "QR", PIN.LU: "LU", PIN.NONE: "NONE"}[c["norm"]]
    want = O.fit(ptr, idx, val, m, n, n_components=k, n_oversamples=p, n_power_iterations=q, normalizer=norm_name,
                 center=c["center"], omega=om, mask=mask)
    if c["masked"]:
        b = sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask)
    else:
        b = sapca.SparsePCABuilder.new().n_components(k)
    est = b.center(c["center"]).spmm_variant(c["variant"]).svd_method(SVDMethod.Random(p, q, c["norm"])).build().set_omega(om)
    A = mat(ptr, idx, val.astype(c["dtype"]), m, n)
    t = est.fit_transform(A)
    f32 = c["dtype"] == np.float32
    note = f"{c} gap {gap:.2f}"
    s_got, s_want = est.singular_values_(np.float64), want.singular_values
    np.testing.assert_allclose(s_got, s_want, rtol=1e-4 if f32 else 1e-7, err_msg=note)
    np.testing.assert_allclose(est.mean_(np.float64), want.mean, atol=1e-5 if f32 else 1e-12, err_msg=note)
    assert O.subspace_angle(est.components_(np.float64), want.components) < (1e-4 if f32 else 1e-5), note
    comps, mean = est.components_(np.float64), est.mean_(np.float64)
    if c["masked"]:
        tw = O.transform_masked_fast(ptr, idx, val, m, n, comps, mean, c["center"], mask)
    else:
        tw = O.transform_sparse(ptr, idx, val, m, n, comps, mean, c["center"])
    np.testing.assert_allclose(t, tw, atol=(2e-4 if f32 else 1e-9) * max(1.0, float(np.abs(tw).max())), err_msg=note)


# ------------------------------------------------------------------ errors (reference messages)
def test_error_behaviour():
    ptr, idx, val = csr_np(synth.flat_csr(50, 20, 0.3, dtype=torch.float64))
    A = mat(ptr, idx, val, 50, 20)
    pca = _builder(3, 3, 1).build()
    with pytest.raises(L.SapcaError, match="Must be fitted before transform!") as e:
        pca.transform(A)
    assert e.value.status == L.ERR_NOT_FITTED
    with pytest.raises(L.SapcaError, match="Model must be fitted first!"):
        pca.explained_variance_ratio()
    with pytest.raises(L.SapcaError, match="Model must be fitted first!"):
        pca.feature_importances()
    for bad in (np.ones(19, bool), np.zeros(0, bool)):
        mp = sapca.MaskedSparsePCABuilder.new().n_components(3).mask(bad).svd_method(SVDMethod.Random(3, 1)).build()
        with pytest.raises(L.SapcaError, match="mask vector length") as e:
            mp.fit(A)
        assert e.value.status == L.ERR_MASK_LEN
    big = _builder(30, 3, 1).build()                       # k > min(m, n): the reference would panic on s[i]
    with pytest.raises(L.SapcaError, match="Randomized SVD computation failed") as e:
        big.fit(A)
    assert e.value.status == L.ERR_SVD
    none = sapca.MaskedSparsePCABuilder.new().n_components(2).mask(np.zeros(20, bool)).svd_method(SVDMethod.Random(3, 1)).build()
    with pytest.raises(L.SapcaError) as e:
        none.fit(A)
    assert e.value.status == L.ERR_SVD


def test_malformed_host_csr_is_refused(session):
    """arrays a CsrMatrix could never hold: offsets that go backwards, a column past n -- an error, not a stray access"""
    X = np.ones((4, 2))
    val = np.ones(5)
    idx = np.array([0, 1, 2, 3, 0], dtype=np.int64)
    with pytest.raises(L.SapcaError, match="non-decreasing") as e:
        session.spmm(np.array([0, 5, 3, 5], dtype=np.int64), idx, val, 3, 4, X)
    assert e.value.status == L.ERR_ARG
    with pytest.raises(L.SapcaError, match="do not span"):
        session.spmm(np.array([0, 2, 3, 4], dtype=np.int64), idx, val, 3, 4, X)
    with pytest.raises(L.SapcaError, match="column index out of range"):
        session.spmm(np.array([0, 2, 3, 5], dtype=np.int64), np.array([0, 1, 2, 3, 4], dtype=np.int64), val, 3, 4, X)


def test_rows_whose_columns_do_not_ascend_give_numbers_not_stray_accesses():
    """include/sapca.h: a device CSR with unsorted rows gives wrong numbers, not stray accesses.  The gather fill of A^T's
    format reads runs of A's rows whose ends the histogram pass found assuming ascending columns: that pass now notices a
    row whose columns go back to an earlier block, and the builder takes the bucket route (every column through a table)
    instead.  The handle then fits a sane matrix as if nothing had happened."""
    m, n, k = 40_000, 2600, 6
    dev = synth.gapped_csr(m, n, 0.03, k, seed=11, dtype=torch.float32, device="cuda")
    ptr, idx, val = dev
    bad_idx, bad_val = idx.clone(), val.clone()
    p = ptr.cpu().numpy()
    for r in (17, 4242, m - 1):   # rows reversed end to end: their columns descend through every block
        lo, hi = int(p[r]), int(p[r + 1])
        assert hi - lo > 8
        bad_idx[lo:hi] = torch.flip(idx[lo:hi], dims=[0])
        bad_val[lo:hi] = torch.flip(val[lo:hi], dims=[0])
    pca = _builder(k, 6, 2).spmm_variant(2).build()
    try:
        t_bad = pca.fit_transform(sapca.DeviceCsr(ptr, bad_idx, bad_val, (m, n)))
        assert t_bad.shape == (m, k)
        torch.cuda.synchronize()
    except L.SapcaError as e:   # (a refusal would be fine too; a fault is not)
        assert e.status in (L.ERR_ARG, L.ERR_SVD)
    good = _builder(k, 6, 2).spmm_variant(2).build()
    t_ref = good.fit_transform(sapca.DeviceCsr(ptr, idx, val, (m, n)))
    t_again = pca.fit_transform(sapca.DeviceCsr(ptr, idx, val, (m, n)))
    assert torch.equal(t_again, t_ref)


def test_explained_variance_ratio_sums_to_one_q4():
    ptr, idx, val = csr_np(synth.gapped_csr(2000, 500, 0.08, 6, seed=3, dtype=torch.float32))
    pca = _builder(6, 6, 2).build()
    pca.fit(mat(ptr, idx, val, 2000, 500))
    r = pca.explained_variance_ratio()
    assert abs(float(r.sum()) - 1.0) < 1e-5 and np.all(np.diff(r) <= 1e-7)
    np.testing.assert_allclose(pca.cumulative_explained_variance_ratio()[-1], 1.0, atol=1e-5)


# ------------------------------------------------------------------ size-independent properties at a larger size
def test_properties_at_scale():
    """200k x 4k f32 (2.4e7 stored entries): linearity of the sweeps in the panel, orthonormality of the
    components, fit_transform == fit + transform, determinism."""
    m, n, k = 200_000, 4_000, 16
    dev = synth.gapped_csr(m, n, 0.03, k, seed=42, dtype=torch.float32, device="cuda")
    x = sapca.DeviceCsr(*dev, (m, n))
    pca = _builder(k, 8, 2).build()
    t1 = pca.fit_transform(x)
    c1 = pca.components_(np.float64)
    np.testing.assert_allclose(c1 @ c1.T, np.eye(k), atol=5e-5)
    t2 = pca.transform(x)
    # a separate transform on caller-owned device arrays does not reuse the fit's preparation (the arrays may have been
    # edited in between), so it may run another sweep kernel: the same projection to f32 rounding, not bit for bit
    assert float((t1 - t2).abs().max()) <= 2e-5 * float(t1.abs().max())
    pca_b = _builder(k, 8, 2).build()
    t3 = pca_b.fit_transform(x)
    assert torch.equal(t1, t3) and np.array_equal(pca_b.components_(), pca.components_())      # bitwise reproducible
    ev = pca.explained_variance_(np.float64)
    assert np.all(np.diff(ev) <= 0) and abs(pca.explained_variance_ratio(np.float64).sum() - 1) < 1e-5
    # the CENTERED projection of the data onto orthonormal components has variance == explained variance
    pc = _builder(k, 8, 2).transform_semantics(L.TRANSFORM_CENTERED).build()
    tc = pc.fit_transform(x).double()
    np.testing.assert_allclose((tc ** 2).sum(0).cpu().numpy() / (m - 1), pc.explained_variance_(np.float64), rtol=2e-3)

def _properties_of_a_randomized_fit(m, n, density, k, p, q, *, want_kernel=2):
    """What can be checked without the oracle at BASELINE's full sizes: orthonormal components, fit_transform == fit +
    transform (to rounding: the separate call may take another sweep kernel), bitwise reproducibility, ratios summing to
    one (Q4), the variance identity of the CENTERED projection, and the residual check
    ||Ac^T u_i - sigma_i v_i|| <= 2e-3 sigma_i  with u_i = Ac v_i / sigma_i  through independent torch index arithmetic on
    the same resident arrays.  `want_kernel`: which sweep the production dispatch must have picked (sapca_timings)."""
    dev = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device="cuda")
    x = sapca.DeviceCsr(*dev, (m, n))
    pca = _builder(k, p, q).collect_timings(True).build()
    t1 = pca.fit_transform(x)
    assert int(pca.timings().sweep_kernel) == want_kernel
    c = pca.components_(np.float64)
    np.testing.assert_allclose(c @ c.T, np.eye(k), atol=5e-5)
    assert float((t1 - pca.transform(x)).abs().max()) <= 2e-5 * float(t1.abs().max())   # another sweep kernel: see test_properties_at_scale
    pca_b = _builder(k, p, q).build()
    t3 = pca_b.fit_transform(x)
    assert torch.equal(t1, t3) and np.array_equal(pca_b.components_(), pca.components_())
    del t1, t3, pca_b
    r = pca.explained_variance_ratio(np.float64)
    assert abs(r.sum() - 1) < 1e-5 and np.all(np.diff(pca.explained_variance_(np.float64)) <= 0)
    pc = _builder(k, p, q).transform_semantics(L.TRANSFORM_CENTERED).build()
    tc = pc.fit_transform(x).double()          # = U S: column i has norm sigma_i and the columns are orthogonal
    sv = pc.singular_values_(np.float64)
    np.testing.assert_allclose((tc ** 2).sum(0).sqrt().cpu().numpy(), sv, rtol=1e-3)
    g = (tc.T @ tc).cpu().numpy() / np.outer(sv, sv)
    np.testing.assert_allclose(g, np.eye(k), atol=2e-3)
    # singular pair residual through A^T: Ac^T (Ac v_i) = sigma_i^2 v_i, Ac^T y = A^T y - mean * sum(y)
    ptr, idx, val = dev
    rows = torch.repeat_interleave(torch.arange(m, device="cuda"), ptr[1:] - ptr[:-1])
    mean = torch.as_tensor(pc.mean_(np.float64), device="cuda")
    V = torch.as_tensor(pc.components_(np.float64), device="cuda")
    idx64, val64 = idx.long(), val.double()
    for i in (0, k // 2, k - 1):
        y = tc[:, i]
        z = torch.zeros(n, dtype=torch.float64, device="cuda").index_add_(0, idx64, val64 * y[rows]) - mean * y.sum()
        resid = (z - sv[i] ** 2 * V[i]).norm().item() / sv[i] ** 2
        assert resid < 2e-3, (i, resid)


def test_properties_at_the_c2_size():
    """BASELINE's C2 (200k x 20k f32, 1.2e8 stored entries, k=50, p=10, q=4, QR) -- too big for the oracle."""
    _properties_of_a_randomized_fit(200_000, 20_000, 0.03, 50, 10, 4)


def test_properties_at_the_c4_size():
    """BASELINE's C4 on one GPU (1M x 30k f32, 9e8 stored entries, k=50, p=10, q=4, QR): the configuration the north star's
    roofline target is quoted on; 7.2 GB of CSR, 1024-row blocks on both operators."""
    _properties_of_a_randomized_fit(1_000_000, 30_000, 0.03, 50, 10, 4)


def test_properties_at_the_c5_size():
    """BASELINE's C5 on one GPU (2M x 50k f32, 1e9 stored entries, 1 % dense, k=100, p=10: l = 110 goes through the
    64-column tile geometry in two column passes)."""
    _properties_of_a_randomized_fit(2_000_000, 50_000, 0.01, 100, 10, 4)


def test_production_dispatch_against_the_oracle():
    """The sweep the AUTO dispatch picks for a large operator (spmm_variant = 0, 1.8e7 stored entries: above the
    staged-sweep floor) against the C restatement of the reference algorithm with the same injected Omega:
    sigma to 1e-4, subspace angle < 1e-4 (the north-star tolerance), explained-variance ratios to 1e-6."""
    import orc
    m, n, density, k, p, q = 30_000, 20_000, 0.03, 50, 10, 4
    dev = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device="cuda")
    ptr, idx, val = csr_np(dev)
    om = synth.gaussian_panel(n, k + p, 42).numpy().astype(np.float32)
    pca = _builder(k, p, q).collect_timings(True).build().set_omega(om)
    t = pca.fit_transform(sapca.DeviceCsr(*dev, (m, n))).cpu().numpy()
    assert int(pca.timings().sweep_kernel) == 2            # the DPP-fed quad sweep ran, not the row kernel
    val = val.astype(np.float64)                           # the oracle in f64: the f32 fit is held to the north-star tolerances
    rc, comps, sing, ev, mean, tv = orc.randomized_fit(ptr, idx, val, m, n, k, p, q, "QR", True, om.astype(np.float64))
    assert rc == 0
    np.testing.assert_allclose(pca.singular_values_(np.float64), sing[:k], rtol=1e-4)
    assert O.subspace_angle(pca.components_(np.float64), comps[:k].astype(np.float64)) < 1e-4
    ratio = (sing[:k].astype(np.float64) ** 2) / (sing[:k].astype(np.float64) ** 2).sum()
    np.testing.assert_allclose(pca.explained_variance_ratio(np.float64), ratio, atol=1e-6)
    want = orc.transform_sparse(ptr, idx, val, m, n, comps, mean, True)
    np.testing.assert_allclose(t, want, atol=PROJ_ATOL * np.abs(want).max())


def test_c2_at_full_size_against_the_oracle():
    """BASELINE configs[1] at its full size (200k x 20k f32, 1.2e8 stored entries, k=50, p=10, q=4, QR) against the C
    restatement of the reference algorithm (f64, all host cores, about a minute) with the same injected Omega: the
    north-star tolerances, not a property."""
    import orc
    m, n, density, k, p, q = 200_000, 20_000, 0.03, 50, 10, 4
    dev = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device="cuda")
    om = synth.gaussian_panel(n, k + p, 42).numpy().astype(np.float32)
    pca = _builder(k, p, q).collect_timings(True).build().set_omega(om)
    t = pca.fit_transform(sapca.DeviceCsr(*dev, (m, n)))
    assert int(pca.timings().sweep_kernel) == 2
    ptr, idx, val = csr_np(dev)
    val = val.astype(np.float64)
    rc, comps, sing, ev, mean, tv = orc.randomized_fit(ptr, idx, val, m, n, k, p, q, "QR", True, om.astype(np.float64))
    assert rc == 0
    np.testing.assert_allclose(pca.singular_values_(np.float64), sing[:k], rtol=1e-4)
    assert O.subspace_angle(pca.components_(np.float64), comps[:k].astype(np.float64)) < 1e-4
    ratio = (sing[:k].astype(np.float64) ** 2) / (sing[:k].astype(np.float64) ** 2).sum()
    np.testing.assert_allclose(pca.explained_variance_ratio(np.float64), ratio, atol=1e-6)
    np.testing.assert_allclose(pca.mean_(np.float64), mean, rtol=1e-5, atol=1e-7)
    rows = np.arange(0, m, 997)                                  # the projection on a sample of rows (the closed form of Q2)
    sub_ptr = np.zeros(len(rows) + 1, np.int64)
    sub_ptr[1:] = np.cumsum(ptr[rows + 1] - ptr[rows])
    take = np.concatenate([np.arange(ptr[r], ptr[r + 1]) for r in rows])
    cnt = np.bincount(idx, minlength=n).astype(np.float64)      # Q2 weights every feature by its stored-entry count over the WHOLE matrix
    X = sp.csr_matrix((val[take], idx[take], sub_ptr), shape=(len(rows), n))
    W = (comps[:k].astype(np.float64) * cnt[None, :]).T
    want = X @ W - (mean.astype(np.float64) @ W)[None, :]
    np.testing.assert_allclose(t[torch.as_tensor(rows, device=t.device)].cpu().numpy(), want, atol=PROJ_ATOL * np.abs(want).max())


def test_two_column_passes_at_production_size_against_the_oracle():
    """l = 110 (k = 100, p = 10: wider than the 64-column tile, so every sweep runs two column passes over the same
    format) through the AUTO dispatch at 1.8e7 stored entries, against the C restatement with the same injected Omega:
    the north-star tolerances of the 64-column case."""
    import orc
    m, n, density, k, p, q = 30_000, 20_000, 0.03, 100, 10, 4
    dev = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device="cuda")
    ptr, idx, val = csr_np(dev)
    om = synth.gaussian_panel(n, k + p, 42).numpy().astype(np.float32)
    pca = _builder(k, p, q).collect_timings(True).build().set_omega(om)
    t = pca.fit_transform(sapca.DeviceCsr(*dev, (m, n))).cpu().numpy()
    assert int(pca.timings().sweep_kernel) == 2            # the DPP-fed quad sweep ran, not the row kernel
    val = val.astype(np.float64)
    rc, comps, sing, ev, mean, tv = orc.randomized_fit(ptr, idx, val, m, n, k, p, q, "QR", True, om.astype(np.float64))
    assert rc == 0
    np.testing.assert_allclose(pca.singular_values_(np.float64), sing[:k], rtol=1e-4)
    assert O.subspace_angle(pca.components_(np.float64), comps[:k].astype(np.float64)) < 1e-4
    ratio = (sing[:k].astype(np.float64) ** 2) / (sing[:k].astype(np.float64) ** 2).sum()
    np.testing.assert_allclose(pca.explained_variance_ratio(np.float64), ratio, atol=1e-6)
    want = orc.transform_sparse(ptr, idx, val, m, n, comps, mean, True)
    np.testing.assert_allclose(t, want, atol=PROJ_ATOL * np.abs(want).max())


def test_masked_randomized_on_the_bucket_route_at_production_size_against_the_oracle():
    """MaskedSparsePCA with the randomized method through the AUTO dispatch at 2.7e7 stored entries (1.6e7 kept by the 60 %
    mask): compaction first, A'^T's format straight from the compacted matrix (the bucket route), the masked-out columns'
    sums from the dropped pairs.  Against the C restatement run on the compacted operator (MaskedCSRMatrix::new,
    sparse_masked/mod.rs:313) with the same Omega; mean_ is full width (:279-286), the projection is Q3 (:488-529)."""
    import orc
    m, n, density, k, p, q = 30_000, 30_000, 0.03, 30, 10, 4
    dev = synth.gapped_csr(m, n, density, k, seed=7, dtype=torch.float32, device="cuda")
    ptr, idx, val = csr_np(dev)
    mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
    n_used = int(mask.sum())
    om = synth.gaussian_panel(n_used, k + p, 11).numpy().astype(np.float32)
    est = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).collect_timings(True)
           .svd_method(SVDMethod.Random(p, q, PIN.QR)).build().set_omega(om))
    t = est.fit_transform(sapca.DeviceCsr(*dev, (m, n))).cpu().numpy()
    assert int(est.timings().sweep_kernel) == 2
    cols, o2m = est.mask_index_maps()
    assert np.array_equal(cols, np.flatnonzero(mask))                        # bit-exact index maps
    val = val.astype(np.float64)
    ptr2, idx2, val2, nu = O.masked_csr(ptr, idx, val, n, mask)
    assert nu == n_used and len(val2) > 1e7
    rc, comps, sing, ev, mean_used, tv = orc.randomized_fit(ptr2, idx2, val2, m, n_used, k, p, q, "QR", True, om.astype(np.float64))
    assert rc == 0
    np.testing.assert_allclose(est.singular_values_(np.float64), sing[:k], rtol=1e-4)
    assert O.subspace_angle(est.components_(np.float64), comps[:k].astype(np.float64)) < 1e-4
    mean_full = np.bincount(idx, weights=val, minlength=n) / m               # every column, masked-out ones included
    np.testing.assert_allclose(est.mean_(np.float64), mean_full, rtol=2e-6, atol=1e-8)
    want = orc.transform_masked(ptr, idx, val, m, comps[:k], mean_full, True, o2m)
    np.testing.assert_allclose(t, want, atol=PROJ_ATOL * np.abs(want).max())


def test_properties_at_the_c3_size():
    """BASELINE's C3 (200k x 30k f64, 60 % feature mask, Lanczos k=30): exact integer mask maps, orthonormal
    components with svd_flip signs, descending singular values, and the eigen-residual of the UNCENTRED masked
    operator (Q1)  ||A'^T A' v_i - sigma_i^2 v_i|| <= 1e-4 sigma_i^2  (las2's kappa is 1e-5 on the Ritz value),
    the masked transform (Q3) against torch index arithmetic on the same resident arrays."""
    m, n, k = 200_000, 30_000, 30
    ptr, idx, val = synth.gapped_csr(m, n, 0.03, k, seed=42, centred=False, dtype=torch.float64, device="cuda")
    mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
    x = sapca.DeviceCsr(ptr, idx, val, (m, n))
    est = sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).svd_method(SVDMethod.Lanczos()).build()
    t = est.fit_transform(x)
    cols, o2m = est.mask_index_maps()
    want_cols = np.flatnonzero(mask)
    assert np.array_equal(cols, want_cols)                                   # ascending, bit-exact
    want_o2m = np.full(n, -1, np.int64)
    want_o2m[want_cols] = np.arange(want_cols.size)
    assert np.array_equal(o2m, want_o2m)
    c = est.components_(np.float64)
    n_used = want_cols.size
    assert c.shape == (k, n_used)
    np.testing.assert_allclose(c @ c.T, np.eye(k), atol=1e-9)
    assert np.all(c[np.arange(k), np.argmax(np.abs(c), 1)] > 0)
    sv = est.singular_values_(np.float64)
    assert np.all(np.diff(sv) < 0)
    # residual on the GPU with torch: entries of kept columns only, columns renumbered
    o2m_d = torch.as_tensor(want_o2m, device="cuda")
    rows = torch.repeat_interleave(torch.arange(m, device="cuda"), ptr[1:] - ptr[:-1])
    cj = o2m_d[idx.long()]
    keep = cj >= 0
    rows_k, cj_k, val_k = rows[keep], cj[keep], val[keep]
    V = torch.as_tensor(c, device="cuda")
    for i in (0, k // 2, k - 1):
        y = torch.zeros(m, dtype=torch.float64, device="cuda").index_add_(0, rows_k, val_k * V[i][cj_k])
        z = torch.zeros(n_used, dtype=torch.float64, device="cuda").index_add_(0, cj_k, val_k * y[rows_k])
        resid = (z - sv[i] ** 2 * V[i]).norm().item() / sv[i] ** 2
        assert resid < 1e-4, (i, resid)
    # Q3: t_ik = sum over stored, kept entries of (a_ij - mean_j) V[k, idx(j)]
    mean = torch.as_tensor(est.mean_(np.float64), device="cuda")
    assert mean.numel() == n                                               # unmasked indexing (sparse_masked/mod.rs:291)
    i = 3
    want = torch.zeros(m, dtype=torch.float64, device="cuda").index_add_(0, rows_k, (val_k - mean[idx.long()[keep]]) * V[i][cj_k])
    np.testing.assert_allclose(t[:, i].cpu().numpy(), want.cpu().numpy(), atol=1e-9 * float(want.abs().max()))


# ------------------------------------------------------------------ G6: Lanczos (uncentred: quirk Q1)
@pytest.mark.parametrize("dtype,srel,ang", [(torch.float64, 1e-5, 1e-4), (torch.float32, 1e-4, 1e-4)])
def test_g6_lanczos_uncentred(golden, dtype, srel, ang):
    g = golden("g6_lanczos.npz")
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    dev = synth.gapped_csr(m, n, float(g["density"]), k, seed=int(g["seed"]), centred=False, dtype=dtype, device="cuda")
    assert dev[2].numel() == int(g["nnz"])
    x = sapca.DeviceCsr(*dev, (m, n))
    pca = sapca.SparsePCABuilder.new().n_components(k).build()          # default method: Lanczos (pca/mod.rs:64-68)
    pca.fit(x)
    np.testing.assert_allclose(pca.singular_values_(np.float64), g["exact_s"][:k], rtol=srel)    # kappa = 1e-5
    assert O.subspace_angle(pca.components_(np.float64), g["exact_vt"]) < ang
    c = pca.components_(np.float64)
    assert np.all(c[np.arange(k), np.argmax(np.abs(c), 1)] > 0)                                   # svd_flip
    np.testing.assert_allclose(np.abs(c @ g["exact_vt"].T), np.eye(k), atol=2e-3)                 # vector by vector
    # centring is computed (mean_, total variance) but the SVD ignores it
    ptr, idx, val = csr_np(dev)
    np.testing.assert_allclose(pca.mean_(np.float64), np.asarray(mat(ptr, idx, val, m, n).mean(0)).ravel(), rtol=1e-5, atol=1e-7)
    ev = pca.explained_variance_(np.float64)
    np.testing.assert_allclose(ev, g["exact_s"][:k] ** 2 / (m - 1), rtol=2 * srel)
    assert pca.timings().lanczos_steps >= k
    # transform still uses the mean (Q2 restatement evaluated with the fitted state)
    t = pca.transform(x).cpu().numpy()
    want = O.transform_sparse(ptr, idx, val.astype(np.float64), m, n, c, pca.mean_(np.float64), True)
    np.testing.assert_allclose(t, want, atol=(1e-8 if dtype == torch.float64 else 2e-3) * np.abs(want).max())


def test_g6_lanczos_masked(golden):
    g = golden("g6_lanczos.npz")
    m, n, k = int(g["m"]), int(g["n"]), int(g["k"])
    dev = synth.gapped_csr(m, n, float(g["density"]), k, seed=int(g["seed"]), centred=False, dtype=torch.float64, device="cuda")
    est = sapca.MaskedSparsePCABuilder.new().n_components(k).mask(g["mask"]).svd_method(SVDMethod.Lanczos()).build()
    est.fit(sapca.DeviceCsr(*dev, (m, n)))
    np.testing.assert_allclose(est.singular_values_(), g["masked_s"][:k], rtol=1e-5)
    assert O.subspace_angle(est.components_(), g["masked_vt"]) < 1e-4
    assert est.components_().shape == (k, int(g["mask"].sum())) and est.mean_().shape == (n,)


@pytest.mark.parametrize("dtype,masked", [(np.float64, True), (np.float64, False), (np.float32, True)])
def test_lanczos_without_a_transposed_operator(debug_switches, monkeypatch, dtype, masked):
    """Lanczos fits whose transposed side fits LDS build no A^T: the second product of a step scatters A's rows into
    per-workgroup FIXED-POINT copies of the output (scatter.hip), the column statistics come from the same kind of pass.
    (1) bit-for-bit reproducible (integer sums: no order dependence); (2) the same fit as the transposed route
    (SAPCA_LANCZOS_TRANSPOSE=1: radix sort + gather along A^T's rows) to rounding; (3) mean_ / total variance against the
    oracle; (4) the exact SVD of the dense operator."""
    m, n, k = 7000, 1100, 6
    t = torch.float64 if dtype == np.float64 else torch.float32
    dev = synth.gapped_csr(m, n, 0.05, k, seed=17, centred=False, dtype=t, device="cuda")
    ptr, idx, val = csr_np(dev)
    mask = synth.bernoulli_mask(n, 0.6, 5).numpy() if masked else None
    def build():
        b = sapca.MaskedSparsePCABuilder.new().mask(mask) if masked else sapca.SparsePCABuilder.new()
        return b.n_components(k).svd_method(SVDMethod.Lanczos()).build()
    x = sapca.DeviceCsr(*dev, (m, n))
    a, b = build(), build()
    ta, tb = a.fit_transform(x), b.fit_transform(x)
    assert np.array_equal(a.components_(np.float64), b.components_(np.float64)) and torch.equal(ta, tb)      # (1)
    assert np.array_equal(a.mean_(np.float64), b.mean_(np.float64))
    monkeypatch.setenv("SAPCA_LANCZOS_TRANSPOSE", "1")
    c = build()
    tc = c.fit_transform(x)
    f32 = dtype == np.float32
    np.testing.assert_allclose(a.singular_values_(np.float64), c.singular_values_(np.float64), rtol=1e-6 if f32 else 1e-10)   # (2)
    assert O.subspace_angle(a.components_(np.float64), c.components_(np.float64)) < (1e-5 if f32 else 1e-8)
    np.testing.assert_allclose(a.mean_(np.float64), c.mean_(np.float64), rtol=1e-6 if f32 else 1e-13, atol=1e-12)
    np.testing.assert_allclose(ta.cpu().numpy(), tc.cpu().numpy(), atol=(1e-3 if f32 else 1e-8) * float(tc.abs().max()))
    v64 = val.astype(np.float64)                                                                             # (3)
    want_mean = np.bincount(idx, weights=v64, minlength=n) / m
    np.testing.assert_allclose(a.mean_(np.float64), want_mean, rtol=2e-6 if f32 else 1e-13, atol=1e-12)
    D = mat(ptr, idx, v64, m, n).toarray()                                                                   # (4)
    if masked:
        D = D[:, mask]
    _, sv, vt = np.linalg.svd(D, full_matrices=False)
    np.testing.assert_allclose(a.singular_values_(np.float64), sv[:k], rtol=1e-5)
    assert O.subspace_angle(a.components_(np.float64), vt[:k]) < 1e-4


def test_lanczos_wide_matrix_uses_the_small_side():
    """m < n: las2 iterates on A A^T and recovers the right vectors."""
    m, n, k = 300, 2000, 5
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.08, k, seed=13, centred=False, dtype=torch.float64))
    A = mat(ptr, idx, val, m, n)
    pca = sapca.SparsePCABuilder.new().n_components(k).build()
    pca.fit(A)
    _, s, vt = np.linalg.svd(A.toarray(), full_matrices=False)
    np.testing.assert_allclose(pca.singular_values_(), s[:k], rtol=1e-5)
    assert O.subspace_angle(pca.components_(), vt[:k]) < 1e-4


@pytest.mark.parametrize("seed", range(16))
def test_random_lanczos_configurations_against_the_exact_svd(seed):
    """svd_las2 on random shapes (tall and wide), with and without a mask, f32 / f64: singular values and the top-k right
    subspace of the raw operator (quirk Q1: Lanczos fits are uncentred) against numpy's SVD of the dense matrix"""
    rng = np.random.default_rng(500 + seed)
    k = int(rng.integers(1, 9))
    m, n = int(rng.integers(60, 1800)), int(rng.integers(40, 1500))
    dens = float(rng.uniform(0.03, 0.25))
    masked, f32 = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, dens, k, seed=seed, centred=False, dtype=torch.float64))
    A = mat(ptr, idx, val.astype(np.float32 if f32 else np.float64), m, n)
    mask = synth.bernoulli_mask(n, 0.7, seed).numpy() if masked else np.ones(n, bool)
    if masked:
        est = sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).svd_method(SVDMethod.Lanczos()).build()
    else:
        est = sapca.SparsePCABuilder.new().n_components(k).build()
    D = A.toarray().astype(np.float64)[:, mask]
    if min(D.shape) < k + 2:
        pytest.skip("mask left too few columns")
    est.fit(A)
    _, s, vt = np.linalg.svd(D, full_matrices=False)
    np.testing.assert_allclose(est.singular_values_(np.float64), s[:k], rtol=2e-4 if f32 else 1e-5)
    assert O.subspace_angle(est.components_(np.float64), vt[:k]) < (2e-3 if f32 else 1e-4)
    t = est.transform(A)
    comps, mean = est.components_(np.float64), est.mean_(np.float64)
    if masked:
        tw = O.transform_masked_fast(ptr, idx, val, m, n, comps, mean, True, mask)
    else:
        tw = O.transform_sparse(ptr, idx, val, m, n, comps, mean, True)
    np.testing.assert_allclose(t, tw, atol=(5e-4 if f32 else 1e-9) * max(1.0, float(np.abs(tw).max())))


def test_lanczos_vs_oracle_and_rank_error():
    ptr, idx, val = csr_np(synth.flat_csr(400, 60, 0.2, seed=5, dtype=torch.float64))
    A = mat(ptr, idx, val, 400, 60)
    want = O.fit(ptr, idx, val, 400, 60, n_components=6, method="LANCZOS")
    pca = sapca.SparsePCABuilder.new().n_components(6).build()
    pca.fit(A)
    np.testing.assert_allclose(pca.singular_values_(), want.singular_values, rtol=1e-5)
    too_many = sapca.SparsePCABuilder.new().n_components(61).build()
    with pytest.raises(L.SapcaError, match="SVD computation failed") as e:
        too_many.fit(A)
    assert e.value.status == L.ERR_SVD


# ------------------------------------------------------------------ the LDS-staged sweep on its own (variant 2)
@pytest.fixture(scope="module")
def session_tiled():
    return ops.Session(spmm_variant=2)


@pytest.mark.parametrize("l", [8, 30, 64])
def test_g3_spmm_tiled(golden, session_tiled, l):
    g = golden("g3_spmm.npz")
    ptr, idx, val = g["indptr"], g["indices"], g["data"].astype(np.float32)
    m, n, mu = int(g["m"]), int(g["n"]), g["mu"].astype(np.float32)
    X, Yin = g[f"X{l}"].astype(np.float32), g[f"Yin{l}"].astype(np.float32)
    s = session_tiled
    np.testing.assert_allclose(s.spmm(ptr, idx, val, m, n, X), g[f"AX{l}"], atol=2e-3)
    np.testing.assert_allclose(s.spmm(ptr, idx, val, m, n, X, mu), g[f"AcX{l}"], atol=2e-3)
    np.testing.assert_allclose(s.spmm(ptr, idx, val, m, n, Yin, None, transposed=True), g[f"AtY{l}"], atol=2e-3)
    np.testing.assert_allclose(s.spmm(ptr, idx, val, m, n, Yin, mu, transposed=True), g[f"ActY{l}"], atol=2e-3)


@pytest.mark.parametrize("shape,dens,l", [((5000, 3000), 0.02, 60), ((700, 300), 0.05, 110), ((3000, 40000), 0.004, 60),
                                          ((70000, 900), 0.01, 20), ((1000, 1000), 0.0, 16), ((20000, 640), 0.9, 60),
                                          ((3000, 700), 0.5, 64), ((3, 1), 1.0, 4), ((64, 2), 0.7, 8), ((1025, 321), 0.2, 60),
                                          ((2, 700), 0.5, 30), ((4097, 319), 0.05, 64)])
def test_spmm_tiled_matches_row_kernel(session, session_tiled, shape, dens, l):
    """ragged shapes: several row blocks, split tile ranges, wide panels, an empty matrix, nearly dense operators (a
    wave's stream in one tile far longer than the 32 chunks one descriptor register covers, with 1024- and 512-row
    blocks), panels of one to a few rows (every LDS-DMA piece clamps its source rows) and of one row short of / past a
    tile; the two sweep kernels must agree to f32 rounding on both A and A^T"""
    m, n = shape
    ptr, idx, val = csr_np(synth.flat_csr(m, n, dens, seed=8, dtype=torch.float32))
    X = synth.gaussian_panel(n, l, 3).numpy().astype(np.float32)
    Yin = synth.gaussian_panel(m, l, 4).numpy().astype(np.float32)
    a1, a2 = session.spmm(ptr, idx, val, m, n, X), session_tiled.spmm(ptr, idx, val, m, n, X)
    scale = max(1.0, float(np.abs(a1).max()))
    np.testing.assert_allclose(a2, a1, atol=2e-5 * scale)
    b1 = session.spmm(ptr, idx, val, m, n, Yin, None, transposed=True)
    b2 = session_tiled.spmm(ptr, idx, val, m, n, Yin, None, transposed=True)
    np.testing.assert_allclose(b2, b1, atol=2e-5 * max(1.0, float(np.abs(b1).max())))
    if dens > 0:
        want = mat(ptr, idx, val, m, n).astype(np.float64) @ X.astype(np.float64)
        np.testing.assert_allclose(a2, want, atol=1e-4 * scale)


@pytest.mark.parametrize("l", [8, 30, 64])
def test_g3_spmm_tiled_f64(golden, session_tiled, l):
    """the staged sweep with f64 values and panels (512-byte panel rows), golden sweeps to f64 rounding"""
    g = golden("g3_spmm.npz")
    ptr, idx, val = g["indptr"], g["indices"], g["data"].astype(np.float64)
    m, n, mu = int(g["m"]), int(g["n"]), g["mu"].astype(np.float64)
    X, Yin = g[f"X{l}"].astype(np.float64), g[f"Yin{l}"].astype(np.float64)
    s = session_tiled
    np.testing.assert_allclose(s.spmm(ptr, idx, val, m, n, X), g[f"AX{l}"], atol=1e-10)
    np.testing.assert_allclose(s.spmm(ptr, idx, val, m, n, X, mu), g[f"AcX{l}"], atol=1e-10)
    np.testing.assert_allclose(s.spmm(ptr, idx, val, m, n, Yin, None, transposed=True), g[f"AtY{l}"], atol=1e-10)
    np.testing.assert_allclose(s.spmm(ptr, idx, val, m, n, Yin, mu, transposed=True), g[f"ActY{l}"], atol=1e-10)


@pytest.mark.parametrize("shape,dens,l", [((5000, 3000), 0.02, 60), ((3000, 40000), 0.004, 33), ((70000, 900), 0.01, 20),
                                          ((1027, 333), 0.1, 7), ((1000, 1000), 0.0, 16), ((700, 300), 0.05, 110),
                                          ((40000, 900), 0.02, 100)])
def test_spmm_tiled_f64_matches_row_kernel(session, session_tiled, shape, dens, l):
    m, n = shape
    ptr, idx, val = csr_np(synth.flat_csr(m, n, dens, seed=8, dtype=torch.float64))
    X = synth.gaussian_panel(n, l, 3).numpy()
    Yin = synth.gaussian_panel(m, l, 4).numpy()
    a1, a2 = session.spmm(ptr, idx, val, m, n, X), session_tiled.spmm(ptr, idx, val, m, n, X)
    np.testing.assert_allclose(a2, a1, atol=1e-11 * max(1.0, float(np.abs(a1).max())))
    b1 = session.spmm(ptr, idx, val, m, n, Yin, None, transposed=True)
    b2 = session_tiled.spmm(ptr, idx, val, m, n, Yin, None, transposed=True)
    np.testing.assert_allclose(b2, b1, atol=1e-11 * max(1.0, float(np.abs(b1).max())))
    if dens > 0:
        np.testing.assert_allclose(a2, mat(ptr, idx, val, m, n) @ X, atol=1e-10 * max(1.0, float(np.abs(a1).max())))


def test_f64_fit_through_the_staged_sweep(golden):
    """G4 in f64 with the staged sweep forced on: the oracle's numbers to 1e-10, like the row kernel"""
    g = golden("g4_randomized_fit.npz")
    m, n, k, p, q = (int(g[x]) for x in "mnkpq")
    A = mat(g["indptr"], g["indices"], g["data"].astype(np.float64), m, n)
    res = []
    for variant in (1, 2):
        pca = _builder(k, p, q).spmm_variant(variant).build().set_omega(g["omega"])
        t = pca.fit_transform(A)
        np.testing.assert_allclose(pca.singular_values_(np.float64), g["s"], rtol=1e-10)
        assert O.subspace_angle(pca.components_(np.float64), g["vt"]) < 1e-9
        res.append(t)
    np.testing.assert_allclose(res[1], res[0], atol=1e-9 * np.abs(res[0]).max())


@pytest.mark.parametrize("masked", [False, True])
def test_f64_staged_sweep_on_many_tiles_and_blocks(masked):
    """f64 fits whose A^T format comes from the tile-major transposition and the run-wise fill (125 tiles, a dozen row blocks
    of A^T, rows of A^T long enough to leave the LDS-staged fill): the row kernel's fit to f64 rounding, with and without a
    column mask"""
    m, n, k, p, q = 20000, 3000, 10, 6, 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.04, k, seed=14, dtype=torch.float64))
    mask = synth.bernoulli_mask(n, 0.7, 5).numpy() if masked else None
    n_used = int(mask.sum()) if masked else n
    om = synth.gaussian_panel(n_used, k + p, 5).numpy()
    A = mat(ptr, idx, val, m, n)
    res = []
    for variant in (1, 2):
        b = (sapca.MaskedSparsePCABuilder.new().mask(mask) if masked else sapca.SparsePCABuilder.new())
        est = (b.n_components(k).spmm_variant(variant).collect_timings(True).svd_method(SVDMethod.Random(p, q, PIN.QR)).build().set_omega(om))
        t = est.fit_transform(A)
        res.append((t, est.singular_values_(np.float64), est.components_(np.float64), est.mean_(np.float64)))
    np.testing.assert_allclose(res[1][1], res[0][1], rtol=1e-10)
    assert O.subspace_angle(res[1][2], res[0][2]) < 1e-9
    np.testing.assert_allclose(res[1][3], res[0][3], atol=1e-13)
    np.testing.assert_allclose(res[1][0], res[0][0], atol=1e-9 * np.abs(res[0][0]).max())


def test_fit_is_the_same_with_either_sweep_kernel(golden):
    g = golden("g4_randomized_fit.npz")
    m, n, k, p, q = (int(g[x]) for x in "mnkpq")
    A = mat(g["indptr"], g["indices"], g["data"].astype(np.float32), m, n)
    res = []
    for variant in (1, 2):
        pca = _builder(k, p, q).spmm_variant(variant).build().set_omega(g["omega"])
        t = pca.fit_transform(A)
        res.append((pca.singular_values_(np.float64), pca.components_(np.float64), t))
        np.testing.assert_allclose(res[-1][0], g["s"], rtol=1e-4)
        assert O.subspace_angle(res[-1][1], g["vt"]) < 1e-4
    np.testing.assert_allclose(res[0][2], res[1][2], atol=2e-3 * np.abs(res[0][2]).max())


# ------------------------------------------------------------------ wide panels, new-matrix transform, limits
def test_fit_wide_panel_k100():
    """l = 110 (the C5 shape of the panel): the 128-wide sweep path through a whole fit"""
    m, n, k, p, q = 3000, 400, 100, 10, 2
    ptr, idx, val = csr_np(synth.flat_csr(m, n, 0.08, seed=31, dtype=torch.float32))
    om = synth.gaussian_panel(n, k + p, 9).numpy()
    want = O.fit(ptr, idx, val.astype(np.float64), m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    for variant in (1, 2):
        pca = _builder(k, p, q).spmm_variant(variant).build().set_omega(om)
        pca.fit(mat(ptr, idx, val, m, n))
        np.testing.assert_allclose(pca.singular_values_(np.float64), want.singular_values, rtol=2e-4)
        # flat spectrum: compare the projector on the leading, well separated part only
        np.testing.assert_allclose(pca.explained_variance_ratio(np.float64), O.explained_variance_ratio(want.explained_variance), atol=2e-5)


def test_wide_panel_fit_transform_two_column_passes():
    """l = 100 over the 64-wide tile geometry: two column passes per sweep, on an operator whose A^T side splits
    its tile range (900 rows -> 2 row blocks) and through the projection (k = 90 > 64); against the row kernel
    and the oracle"""
    m, n, k, p, q = 40000, 900, 90, 10, 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.03, 12, seed=17, dtype=torch.float32))
    om = synth.gaussian_panel(n, k + p, 5).numpy()
    want = O.fit(ptr, idx, val.astype(np.float64), m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    res = []
    for variant in (1, 2):
        pca = _builder(k, p, q).spmm_variant(variant).build().set_omega(om)
        t = pca.fit_transform(mat(ptr, idx, val, m, n))
        assert t.shape == (m, k) and np.isfinite(t).all()
        np.testing.assert_allclose(pca.singular_values_(np.float64)[:12], want.singular_values[:12], rtol=1e-4)
        res.append((pca.singular_values_(np.float64), t))
    np.testing.assert_allclose(res[1][0], res[0][0], rtol=2e-4)
    # the leading, well separated components project the same way with either kernel
    np.testing.assert_allclose(res[1][1][:, :12], res[0][1][:, :12], atol=2e-3 * np.abs(res[0][1][:, :12]).max())


@pytest.mark.parametrize("variant", [0, 2])
def test_fit_transform_is_fit_followed_by_transform(variant):
    """The reference's fit_transform IS fit() then transform() (sparse/mod.rs:355-358).  Here fit_transform of an unmasked f32
    randomized fit projects with the un-rotated panel while the host solves the l x l problem (engine.cpp, finish_small_svd)
    and a separate transform projects with the fitted components: one model, two routes, the same scores to f32 rounding
    (the 2e-4 bound of PROJ_ATOL), and the stage timings count the held-back small SVD once."""
    m, n, k, p, q = 30000, 2500, 12, 8, 3
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.04, k, seed=29, dtype=torch.float32))
    om = synth.gaussian_panel(n, k + p, 3).numpy()
    x = mat(ptr, idx, val, m, n)
    a = _builder(k, p, q).spmm_variant(variant).collect_timings(True).build().set_omega(om)
    t_a = a.fit_transform(x)
    tm = a.timings()
    assert tm.transform_ms > 0 and tm.small_svd_ms > 0
    b = _builder(k, p, q).spmm_variant(variant).build().set_omega(om)
    b.fit(x)
    t_b = b.transform(x)
    np.testing.assert_allclose(a.singular_values_(np.float64), b.singular_values_(np.float64), rtol=1e-6)
    np.testing.assert_allclose(a.components_(np.float64), b.components_(np.float64), atol=1e-6)
    np.testing.assert_allclose(t_a, t_b, atol=2e-4 * np.abs(t_b).max())
    t_c = a.transform(x)   # and the same handle projects the same way afterwards
    np.testing.assert_allclose(t_c, t_b, atol=2e-4 * np.abs(t_b).max())


def test_dpp_fed_sweep_on_random_shapes_against_the_row_kernel():
    """Both products, centred and not, of 30 random operators -- ragged rows, a run of empty rows, one row that holds every
    column, 1 to 128 panel columns, blocks of a few quads -- through the DPP-fed sweep (forced) against the row kernel: the
    generated main loop's corner cases (a quad of one step, a stream that ends inside a position, a wave without quads,
    a tile range split over workgroups) with numbers behind them.  tools/fuzz_sweep.py runs more of the same."""
    rng = np.random.default_rng(11)
    dq, row = ops.Session(spmm_variant=2), ops.Session(spmm_variant=1)
    for case in range(30):
        m, n = int(rng.integers(1, 5000)), int(rng.integers(1, 4000))
        l = int(rng.choice([1, 3, 16, 30, 60, 64, 65, 100, 128]))
        A = sp.random(m, n, density=float(rng.choice([0.002, 0.01, 0.05, 0.2])), format="csr", dtype=np.float32,
                      random_state=int(rng.integers(1 << 30)))
        if case % 2 and m > 8:
            r = rng.choice(m, size=3, replace=False)
            A = (A + sp.csr_matrix((np.full(n, 0.5, np.float32), (np.full(n, r[0]), np.arange(n))), shape=(m, n))).tolil()
            A[r[1], :] = 0
            A[r[2], :] = 0
            A = A.tocsr()
            A.eliminate_zeros()
        A.sort_indices()
        for transposed in (False, True):
            X = rng.standard_normal(((m if transposed else n), l)).astype(np.float32)
            mu = rng.standard_normal(n).astype(np.float32) if case % 3 else None
            a = dq.spmm(A.indptr, A.indices, A.data, m, n, X, mu, transposed)
            b = row.spmm(A.indptr, A.indices, A.data, m, n, X, mu, transposed)
            assert np.isfinite(a).all()
            np.testing.assert_allclose(a, b, atol=2e-5 * max(1e-30, float(np.abs(b).max())), err_msg=f"case {case} m {m} n {n} l {l} T {transposed}")


def test_f64_wide_panel_fit_through_the_staged_sweep():
    """l = 100 in f64: two 64-column passes over 512-byte-row tiles, fit and projection (k = 90), against the oracle"""
    m, n, k, p, q = 3000, 400, 90, 10, 2
    ptr, idx, val = csr_np(synth.flat_csr(m, n, 0.08, seed=31, dtype=torch.float64))
    om = synth.gaussian_panel(n, k + p, 9).numpy()
    want = O.fit(ptr, idx, val, m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    res = []
    for variant in (1, 2):
        pca = _builder(k, p, q).spmm_variant(variant).build().set_omega(om)
        t = pca.fit_transform(mat(ptr, idx, val, m, n))
        np.testing.assert_allclose(pca.singular_values_(np.float64), want.singular_values, rtol=1e-9)
        res.append(t)
    np.testing.assert_allclose(res[1], res[0], atol=1e-8 * np.abs(res[0]).max())


def test_panel_width_limit_is_an_error():
    """panels of up to 1024 columns are taken (the reference has no limit: above 128 the dense steps run block-wise);
    beyond that the fit is refused with an argument error, not a crash"""
    ptr, idx, val = csr_np(synth.flat_csr(1500, 1400, 0.05, seed=2, dtype=torch.float32))
    pca = _builder(1020, 10, 1).build()
    with pytest.raises(L.SapcaError, match="above 1024") as e:
        pca.fit(mat(ptr, idx, val, 1500, 1400))
    assert e.value.status == L.ERR_ARG


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_transform_of_a_matrix_that_was_not_fitted(dtype):
    """transform(B) with B != the fitted matrix: column counts (Q2) come from B, the mean from the fit"""
    m, n, k = 1500, 300, 6
    a = csr_np(synth.gapped_csr(m, n, 0.08, k, seed=3, dtype=torch.float64))
    b = csr_np(synth.gapped_csr(700, n, 0.05, k, seed=4, dtype=torch.float64))
    A, B = mat(a[0], a[1], a[2].astype(dtype), m, n), mat(b[0], b[1], b[2].astype(dtype), 700, n)
    tol = 1e-9 if dtype == np.float64 else 2e-3
    pca = _builder(k, 6, 2).build()
    pca.fit(A)
    t = pca.transform(B)
    want = O.transform_sparse(b[0], b[1], b[2], 700, n, pca.components_(np.float64), pca.mean_(np.float64), True)
    np.testing.assert_allclose(t, want, atol=tol * np.abs(want).max())
    mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
    mp_ = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).svd_method(SVDMethod.Random(6, 2)).build())
    mp_.fit(A)
    tm = mp_.transform(B)
    want_m = O.transform_masked_fast(b[0], b[1], b[2], 700, n, mp_.components_(np.float64), mp_.mean_(np.float64), True, mask)
    np.testing.assert_allclose(tm, want_m, atol=tol * np.abs(want_m).max())
    with pytest.raises(L.SapcaError):
        pca.transform(mat(a[0][:11], a[1][:a[0][10]], a[2][:a[0][10]].astype(dtype), 10, n + 1) if False else
                      sp.csr_matrix((10, n + 1), dtype=dtype))


def test_transform_of_a_large_matrix_that_was_not_fitted_builds_its_own_format():
    """transform(B) with B != the fitted matrix and B above the staged sweep's break-even (here forced: spmm_variant 2): the
    projection builds B's tile-major format for its one sweep instead of falling back to the row kernel (5 ms against 1.5 at
    C2's size).  Same numbers as the row kernel's, Q2 counts from B, mean from the fit; masked: Q3 through the same format."""
    m, n, k = 9000, 2600, 8
    a = csr_np(synth.gapped_csr(m, n, 0.04, k, seed=3, dtype=torch.float32))
    b = csr_np(synth.gapped_csr(5000, n, 0.05, k, seed=4, dtype=torch.float32))
    A, B = mat(a[0], a[1], a[2], m, n), mat(b[0], b[1], b[2], 5000, n)
    for variant in (2, 1):   # the tile-major format built for B; the row kernel
        pca = _builder(k, 6, 2).spmm_variant(variant).build()
        pca.fit(A)
        t = pca.transform(B)
        want = O.transform_sparse(b[0], b[1], b[2].astype(np.float64), 5000, n, pca.components_(np.float64), pca.mean_(np.float64), True)
        np.testing.assert_allclose(t, want, atol=2e-4 * np.abs(want).max())
        # the handle still fits and projects its own matrix afterwards (B's format does not pass for A's)
        t2 = pca.fit_transform(A)
        want2 = O.transform_sparse(a[0], a[1], a[2].astype(np.float64), m, n, pca.components_(np.float64), pca.mean_(np.float64), True)
        np.testing.assert_allclose(t2, want2, atol=2e-4 * np.abs(want2).max())
    mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
    mp_ = sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).spmm_variant(2).svd_method(SVDMethod.Random(6, 2)).build()
    mp_.fit(A)
    tm = mp_.transform(B)
    want_m = O.transform_masked_fast(b[0], b[1], b[2].astype(np.float64), 5000, n, mp_.components_(np.float64), mp_.mean_(np.float64), True, mask)
    np.testing.assert_allclose(tm, want_m, atol=2e-4 * np.abs(want_m).max())


def test_transform_after_the_values_were_edited_in_place():
    """fit(X) on caller-owned device arrays, X.values edited in place through torch, transform(X): the projection must
    see the new values (the preparation kept from the fit -- a tile-major copy of the values -- is not reused across
    separate calls on caller-owned pointers), with the column counts (Q2) of the unchanged pattern."""
    m, n, k = 3000, 600, 6
    dev = synth.gapped_csr(m, n, 0.05, k, seed=12, dtype=torch.float32, device="cuda")
    x = sapca.DeviceCsr(*dev, (m, n))
    pca = _builder(k, 6, 2).spmm_variant(2).build()
    t_fit = pca.fit_transform(x).cpu().numpy()
    dev[2].mul_(3.0).add_(0.25)
    t = pca.transform(x).cpu().numpy()
    ptr, idx, val = csr_np(dev)
    want = O.transform_sparse(ptr, idx, val.astype(np.float64), m, n, pca.components_(np.float64), pca.mean_(np.float64), True)
    np.testing.assert_allclose(t, want, atol=2e-3 * np.abs(want).max())
    assert np.abs(t - t_fit).max() > 0.1 * np.abs(t_fit).max()   # and it is not the projection of the fitted values


def test_fit_after_the_uploaded_values_were_edited_in_place():
    """sapca_upload_csr_* on the ESTIMATOR's handle hands back a writable d_values; the statistics gathered behind that
    upload serve a fit of the arrays as uploaded.  A caller that edits the values with its own kernel calls
    sapca_upload_values_changed first: mean_ and the components then follow the edited values (the C restatement on them)."""
    import ctypes as C
    from sapca import _lib as L
    m, n, k, p, q = 3000, 500, 6, 6, 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.06, k, seed=31, dtype=torch.float32))
    om = synth.gaussian_panel(n, k + p, 9).numpy()
    pca = _builder(k, p, q).build().set_omega(om)
    lib = L.load()
    ro, ci = np.ascontiguousarray(ptr, dtype=np.uint64), np.ascontiguousarray(idx, dtype=np.uint64)
    dp, di, dv = C.c_void_p(), C.c_void_p(), C.c_void_p()
    L.check(pca._h, lib.sapca_upload_csr_f32(pca._h, C.c_uint64(m), C.c_uint64(n), C.c_uint64(val.size), ro.ctypes.data_as(C.c_void_p),
                                             ci.ctypes.data_as(C.c_void_p), val.ctypes.data_as(C.c_void_p), C.byref(dp), C.byref(di), C.byref(dv)))

    class _View:
        def __init__(self, ptr, count, typestr):
            self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}
    d_val = torch.as_tensor(_View(dv.value, val.size, "<f4"), device="cuda")
    d_val.mul_(2.0).add_(0.5)                       # the caller's own kernel
    torch.cuda.synchronize()
    L.check(pca._h, lib.sapca_upload_values_changed(pca._h))
    L.check(pca._h, lib.sapca_fit_csr_device_f32(pca._h, C.c_uint64(m), C.c_uint64(n), C.c_uint64(val.size), dp, di, dv))
    pca._dtype64 = False
    val2 = (val * np.float32(2.0) + np.float32(0.5)).astype(np.float64)
    want = O.fit(ptr, idx, val2, m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    np.testing.assert_allclose(pca.mean_(np.float64), want.mean, rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(pca.singular_values_(np.float64), want.singular_values, rtol=1e-4)
    assert O.subspace_angle(pca.components_(np.float64), want.components) < 1e-4


def test_every_route_to_the_transposed_format_gives_the_same_fit(debug_switches, monkeypatch):
    """A^T's tile-major format can come from the tile-major transposed rows still packed by the sort (default),
    from the same rows unpacked into a CSR, from a naturally ordered transposed CSR, or straight from A: the
    bytes are the same, hence bit-identical fits"""
    m, n, k, p, q = 6000, 1500, 12, 6, 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.05, k, seed=5, dtype=torch.float32))
    om = synth.gaussian_panel(n, k + p, 3).numpy()
    out = []
    routes = ("SAPCA_AT_SORT", "SAPCA_AT_UNPACK", "SAPCA_AT_NATURAL", "SAPCA_TILED_FROM_A", None)
    monkeypatch.setenv("SAPCA_TILE_DEFAULT", "1")   # same LDS split on every route (a natural-order A^T could take the bigger tile)
    monkeypatch.setenv("SAPCA_NO_ROWSORT", "1")     # and the same row order: the builder straight from A does not sort rows by length
    for route in routes:
        for r in routes[:-1]:
            monkeypatch.delenv(r, raising=False)
        if route:
            monkeypatch.setenv(route, "1")
            if route != "SAPCA_AT_SORT":
                monkeypatch.setenv("SAPCA_AT_SORT", "1")   # the other switches select among the transposition's routes
        pca = _builder(k, p, q).spmm_variant(2).build().set_omega(om)
        dev = sapca.DeviceCsr(*(torch.as_tensor(x, device="cuda") for x in (ptr.astype(np.int64), idx.astype(np.int32), val)), (m, n))
        t = pca.fit_transform(dev).cpu().numpy()     # (device arrays: every route computes its own column statistics)
        out.append((pca.singular_values_(np.float64), pca.components_(np.float64), t, pca.mean_(np.float64)))
    for o in out[1:-1]:
        for a, b in zip(out[0], o):
            np.testing.assert_array_equal(a, b)
    # the default (bucket) route writes the same format bytes, but adds the column statistics per tile and then over the
    # tiles instead of along the transposed rows: means equal to f64 rounding, hence the same model to f32 rounding
    np.testing.assert_allclose(out[-1][3], out[0][3], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(out[-1][0], out[0][0], rtol=1e-5)
    np.testing.assert_allclose(out[-1][2], out[0][2], atol=1e-4 * np.abs(out[0][2]).max())
    want = O.fit(ptr, idx, val.astype(np.float64), m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    np.testing.assert_allclose(out[0][0], want.singular_values, rtol=1e-4)


@pytest.mark.parametrize("shape", [(30000, 2500), (700, 5000), (2500, 70), (100000, 300), (3000, 70000)])
def test_bucket_route_to_the_transposed_format_on_skewed_columns(debug_switches, monkeypatch, shape):
    """A^T's format straight from A (per-chunk buckets) against the transposition route on matrices whose columns differ
    in density by two orders of magnitude (A^T's rows get sorted by length: blocks cut by entry count, rows permuted),
    tall, wide and narrow, and with more columns than the bucket route takes (65536: the transposition steps in while the
    helper thread is already building A's format): same model to rounding, and the oracle's"""
    m, n = shape
    k, p, q = 6, 6, 2
    rng = np.random.default_rng(m + n)
    dens_col = np.clip(rng.lognormal(np.log(0.03), 1.2, n), 2e-4, 0.6)
    cols = [np.flatnonzero(rng.random(m) < d) for d in dens_col]
    rows = np.concatenate(cols)
    cidx = np.concatenate([np.full(len(c), j) for j, c in enumerate(cols)])
    vals = rng.standard_normal(len(rows)).astype(np.float32) + 1.5
    A = sp.csr_matrix((vals, (rows, cidx)), shape=(m, n))
    A.sort_indices()
    om = synth.gaussian_panel(n, k + p, 9).numpy()
    dev = sapca.DeviceCsr(torch.as_tensor(A.indptr.astype(np.int64), device="cuda"), torch.as_tensor(A.indices.astype(np.int32), device="cuda"),
                          torch.as_tensor(A.data, device="cuda"), (m, n))
    out = []
    for sort_route in (False, True):
        if sort_route:
            monkeypatch.setenv("SAPCA_AT_SORT", "1")
        else:
            monkeypatch.delenv("SAPCA_AT_SORT", raising=False)
        pca = _builder(k, p, q).spmm_variant(2).build().set_omega(om)
        t = pca.fit_transform(dev).cpu().numpy()
        out.append((pca.singular_values_(np.float64), pca.mean_(np.float64), pca.explained_variance_ratio(np.float64), t))
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=2e-5)
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(out[0][3], out[1][3], atol=2e-4 * np.abs(out[1][3]).max())
    A64 = A.astype(np.float64)
    want = O.fit(A64.indptr.astype(np.int64), A64.indices.astype(np.int64), A64.data, m, n, n_components=k, n_oversamples=p,
                 n_power_iterations=q, omega=om)
    np.testing.assert_allclose(out[0][0], want.singular_values, rtol=2e-4)
    np.testing.assert_allclose(out[0][1], want.mean, atol=1e-6)


@pytest.mark.parametrize("case", [(6000, 1500, 0.05), (40000, 3000, 0.02), (3000, 20000, 0.01), (4000, 2000, 0.5), (2500, 70, 0.2),
                                  (1000, 40000, 0.004)])
def test_gather_fill_and_bucket_route_write_the_same_format(debug_switches, monkeypatch, case):
    """A^T's format in natural row order: every chunk gathers its runs of A's rows itself (default; the run ends come from
    the histogram pass) or reads the bucket a scatter pass filled (SAPCA_AT_BUCKETS=1) -- the same bytes and the same
    per-tile column sums, hence bit-identical fits.  Blocks of 512 and of 1024 rows, chunks above the ten entries a thread
    holds in registers, rows without entries, a last tile with few rows; and the oracle's model."""
    m, n, d = case
    k, p, q = 6, 6, 2
    ptr, idx, val = csr_np(synth.flat_csr(m, n, d, seed=m + n, dtype=torch.float32))
    A = mat(ptr, idx, val, m, n).tolil()
    for r in (0, 17, m - 1):
        A[r, :] = 0                                  # rows without entries (first and last of the matrix among them)
    A = A.tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    om = synth.gaussian_panel(n, k + p, 4).numpy()
    dev = sapca.DeviceCsr(torch.as_tensor(A.indptr.astype(np.int64), device="cuda"), torch.as_tensor(A.indices.astype(np.int32), device="cuda"),
                          torch.as_tensor(A.data.astype(np.float32), device="cuda"), (m, n))
    out = []
    for buckets in (False, True):
        if buckets:
            monkeypatch.setenv("SAPCA_AT_BUCKETS", "1")
        else:
            monkeypatch.delenv("SAPCA_AT_BUCKETS", raising=False)
        pca = _builder(k, p, q).spmm_variant(2).build().set_omega(om)
        t = pca.fit_transform(dev).cpu().numpy()
        out.append((pca.singular_values_(np.float64), pca.mean_(np.float64), pca.components_(np.float64), t))
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)
    A64 = A.astype(np.float64)
    want = O.fit(A64.indptr.astype(np.int64), A64.indices.astype(np.int64), A64.data, m, n, n_components=k, n_oversamples=p,
                 n_power_iterations=q, omega=om)
    np.testing.assert_allclose(out[0][0], want.singular_values, rtol=2e-4)
    np.testing.assert_allclose(out[0][1], want.mean, atol=1e-6)


def test_bucket_route_on_a_half_dense_matrix(debug_switches, monkeypatch):
    """chunks of 80 000 entries (512 columns x 320 rows at density 0.5): the fill keeps ten entries per thread in registers
    and takes the rest of its bucket from memory; (row, tile) segments of up to 320 entries.  Against the transposition
    route and the oracle."""
    m, n, k, p, q = 4000, 2000, 8, 6, 2
    ptr, idx, val = csr_np(synth.flat_csr(m, n, 0.5, seed=17, dtype=torch.float32))
    om = synth.gaussian_panel(n, k + p, 2).numpy()
    dev = sapca.DeviceCsr(torch.as_tensor(ptr.astype(np.int64), device="cuda"), torch.as_tensor(idx.astype(np.int32), device="cuda"),
                          torch.as_tensor(val, device="cuda"), (m, n))
    out = []
    for sort_route in (False, True):
        if sort_route:
            monkeypatch.setenv("SAPCA_AT_SORT", "1")
        else:
            monkeypatch.delenv("SAPCA_AT_SORT", raising=False)
        pca = _builder(k, p, q).spmm_variant(2).build().set_omega(om)
        t = pca.fit_transform(dev).cpu().numpy()
        out.append((pca.singular_values_(np.float64), pca.mean_(np.float64), t))
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=2e-5)
    np.testing.assert_allclose(out[0][2], out[1][2], atol=2e-4 * np.abs(out[1][2]).max())
    want = O.fit(ptr, idx, val.astype(np.float64), m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    np.testing.assert_allclose(out[0][0], want.singular_values, rtol=2e-4)


def test_tile_major_builder_with_and_without_its_lds_table(debug_switches, monkeypatch):
    """A^T's format: the row-segment bounds staged in LDS (few tiles) or read from global memory one tile ahead
    (many tiles, C4/C5) -- the same bytes, hence bit-identical fits"""
    m, n, k, p, q = 9000, 700, 8, 6, 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.06, k, seed=12, dtype=torch.float32))
    om = synth.gaussian_panel(n, k + p, 3).numpy()
    out = []
    for lim in ("100000", "0"):
        monkeypatch.setenv("SAPCA_RUNS_SEG_LDS_MAX", lim)
        pca = _builder(k, p, q).spmm_variant(2).build().set_omega(om)
        t = pca.fit_transform(mat(ptr, idx, val, m, n))
        out.append((pca.singular_values_(np.float64), pca.components_(np.float64), t))
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)
    want = O.fit(ptr, idx, val.astype(np.float64), m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    np.testing.assert_allclose(out[0][0], want.singular_values, rtol=1e-4)


def test_staged_and_direct_format_fill_agree(debug_switches, monkeypatch):
    """the LDS-staged builder of A's tile-major format and the direct one write the same bytes: bit-identical
    fits; a matrix with a few very long rows sends some quads down the direct route inside the staged kernel"""
    m, n, k, p, q = 5000, 3000, 10, 6, 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.04, k, seed=8, dtype=torch.float32))
    A = mat(ptr, idx, val, m, n).tolil()
    rng = np.random.default_rng(0)
    for r in (7, 1234, 4999):            # dense rows: 3000 entries each, a quad of > 6144 padded entries
        A[r, :] = rng.uniform(0.5, 1.5, n).astype(np.float32)
    A = A.tocsr()
    A.sort_indices()
    om = synth.gaussian_panel(n, k + p, 3).numpy()
    out = []
    for direct in (False, True):
        if direct:
            monkeypatch.setenv("SAPCA_FILL_DIRECT", "1")
        else:
            monkeypatch.delenv("SAPCA_FILL_DIRECT", raising=False)
        pca = _builder(k, p, q).spmm_variant(2).build().set_omega(om)
        t = pca.fit_transform(A)
        out.append((pca.singular_values_(np.float64), t))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1], out[1][1])
    A64 = A.astype(np.float64)
    want = O.fit(A64.indptr.astype(np.int64), A64.indices.astype(np.int64), A64.data, m, n, n_components=k, n_oversamples=p,
                 n_power_iterations=q, omega=om)
    np.testing.assert_allclose(out[0][0], want.singular_values, rtol=1e-4)


@pytest.mark.parametrize("m,n,dens,k,p", [(1027, 333, 0.1, 7, 5), (37, 1200, 0.2, 3, 4), (5, 70, 0.5, 2, 2), (2051, 65, 0.3, 9, 3)])
def test_staged_sweep_on_ragged_shapes(m, n, dens, k, p):
    """odd row counts (incomplete quads and blocks), more columns than rows, empty rows and columns, one-tile and
    many-tile operators: the staged path forced on, against the oracle with the same Omega"""
    rng = np.random.default_rng(m * 7 + n)
    ptr, idx, val = csr_np(synth.flat_csr(m, n, dens, seed=m + n, dtype=torch.float32))
    A = mat(ptr, idx, val, m, n).tolil()
    A[m // 2, :] = 0            # an empty row
    A[:, n // 3] = 0            # an empty column
    A = A.tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    om = synth.gaussian_panel(n, k + p, 11).numpy()
    A64 = A.astype(np.float64)
    want = O.fit(A64.indptr.astype(np.int64), A64.indices.astype(np.int64), A64.data, m, n, n_components=k, n_oversamples=p,
                 n_power_iterations=2, omega=om)
    pca = _builder(k, p, 2).spmm_variant(2).build().set_omega(om)
    t = pca.fit_transform(A)
    np.testing.assert_allclose(pca.singular_values_(np.float64), want.singular_values, rtol=2e-4)
    np.testing.assert_allclose(pca.explained_variance_ratio(np.float64), O.explained_variance_ratio(want.explained_variance), atol=5e-5)
    assert t.shape == (m, k) and np.isfinite(t).all()


# ------------------------------------------------------------------ device-resident preprocessing and statistics
def _resident(A, sess=None):
    sess = sess or ops.Session()
    A = A.tocsr()
    A.sort_indices()
    return sess, sess.upload(A.indptr, A.indices, A.data, A.shape[0], A.shape[1])


def test_normalize_and_statistics_on_the_reference_test_vectors(golden):
    """the data the reference's own tests hold (csr.rs:1385-1422, 1516-1552; csc.rs:1071-1226), through the C ABI"""
    import scipy.sparse as sp
    g = golden("ref_pins_preproc.npz")
    A = sp.coo_matrix((g["norm_vals"], (g["norm_rows"], g["norm_cols"])), shape=(3, 3)).tocsr()
    for direction, sums, want in ((ops.COLUMN, g["norm_col_sums"], g["norm_expected_col"]), (ops.ROW, g["norm_row_sums"], g["norm_expected_row"])):
        for dt in (np.float64, np.float32):
            sess, R = _resident(A.astype(dt))
            got = R.normalize(sums, float(g["norm_target"]), direction).values()
            assert np.abs(got - want).max() < (float(g["norm_tol"]) if dt == np.float64 else 1e-6)
    sess, R = _resident(sp.csr_matrix(g["nz_dense"]))
    np.testing.assert_array_equal(R.stats(ops.COLUMN)[2], g["nz_col"])
    np.testing.assert_array_equal(R.stats(ops.ROW)[2], g["nz_row"])
    sess, R = _resident(sp.csr_matrix(g["sum_dense"]))
    sc, _, _, loc, hic = R.stats(ops.COLUMN)
    sr, _, _, lor, hir = R.stats(ops.ROW)
    np.testing.assert_array_equal(sc, g["sum_col"])
    np.testing.assert_array_equal(sr, g["sum_row"])
    assert loc[0] == g["min_col0"] and hic[0] == g["max_col0"] and lor[2] == g["min_row2"] and hir[2] == g["max_row2"]


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_preprocessing_and_statistics_against_the_oracle(dt):
    m, n = 3000, 700
    ptr, idx, val = csr_np(synth.flat_csr(m, n, 0.06, seed=12, dtype=torch.float32 if dt == np.float32 else torch.float64))
    val = np.abs(val).astype(dt)                       # count-like: log1p of negative values is not the use case
    A = mat(ptr, idx, val, m, n).tolil()
    A[5, :] = 0
    A[:, 9] = 0
    A = A.tocsr()
    A.eliminate_zeros()
    A.sort_indices()
    ptr, idx, val = A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data
    sess, R = _resident(A)
    for direction in (ops.ROW, ops.COLUMN):
        want = O.stats_csr(ptr, idx, val, m, n, direction)
        got = R.stats(direction)
        np.testing.assert_allclose(got[0], want[0], rtol=1e-12 if dt == np.float64 else 1e-6)
        np.testing.assert_allclose(got[1], want[1], rtol=1e-12 if dt == np.float64 else 1e-6)
        np.testing.assert_array_equal(got[2], want[2])
        np.testing.assert_array_equal(got[3], want[3])     # min / max: exact, including the (MAX, -MAX) of empty rows and columns
        np.testing.assert_array_equal(got[4], want[4])
        N = m if direction == ops.COLUMN else n
        np.testing.assert_allclose(R.variance(direction), O.variance_from_sums(want[0], want[1], N), rtol=1e-9, atol=1e-12)
    row_sums = R.stats(ops.ROW)[0]
    got = R.normalize(row_sums, 1e4, ops.ROW).values()
    want = O.normalize_csr(ptr, idx, val, row_sums, 1e4, ops.ROW)
    np.testing.assert_array_equal(got, want)               # IEEE multiply in f64, one rounding: bit-exact
    got2 = R.log1p().values()
    np.testing.assert_allclose(got2, O.log1p_csr(want), rtol=3e-7 if dt == np.float32 else 1e-15)   # libm vs device log: <= 2 ulp
    col_sums = R.stats(ops.COLUMN)[0]
    got3 = R.normalize(col_sums, 1.0, ops.COLUMN).values()
    np.testing.assert_array_equal(got3, O.normalize_csr(ptr, idx, got2, col_sums, 1.0, ops.COLUMN))
    with pytest.raises(L.SapcaError, match="Length of sums") as e:
        R.normalize(col_sums[:-1], 1.0, ops.COLUMN)
    assert e.value.status == L.ERR_ARG


def test_resident_workflow_normalize_log1p_pca():
    """upload once -> normalize -> log1p -> fit_transform on the resident copy (src/lib.rs:28-33), against the
    oracle run on the CPU-preprocessed matrix"""
    m, n, k, p, q = 4000, 900, 8, 6, 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.05, k, seed=21, dtype=torch.float32))
    sess, R = _resident(mat(ptr, idx, val, m, n))
    row_sums = R.stats(ops.ROW)[0]
    R.normalize(row_sums, 1e3, ops.ROW).log1p()
    v2 = O.log1p_csr(O.normalize_csr(ptr, idx, val, row_sums, 1e3, ops.ROW))
    np.testing.assert_allclose(R.values(), v2, rtol=3e-7)
    om = synth.gaussian_panel(n, k + p, 2).numpy()
    pca = _builder(k, p, q).build().set_omega(om)
    t = pca.fit_transform(R.as_device_csr()).cpu().numpy()
    want = O.fit(ptr, idx, R.values().astype(np.float64), m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    np.testing.assert_allclose(pca.singular_values_(np.float64), want.singular_values, rtol=1e-4)
    assert O.subspace_angle(pca.components_(np.float64), want.components) < 1e-4
    assert t.shape == (m, k)


def test_rows_sorted_by_length_on_a_skewed_matrix(debug_switches, monkeypatch):
    """log-normal row lengths and power-law column popularity (cell depth, gene detection rate): the staged formats
    sort rows by length and cut blocks by entry count; forced on and off, against the oracle"""
    m, n, k, p, q = 5000, 800, 8, 6, 2
    rng = np.random.default_rng(3)
    depth = np.exp(0.9 * rng.standard_normal(m))
    pop = (np.arange(1, n + 1) ** -0.8)[rng.permutation(n)]
    P = np.minimum(1.0, depth[:, None] * pop[None, :] * (0.06 * m * n / (depth.sum() * pop.sum())))
    D = (rng.random((m, n)) < P) * rng.uniform(1.0, 5.0, (m, n))
    import scipy.sparse as sp
    A = sp.csr_matrix(D.astype(np.float32))
    A.sort_indices()
    lens = np.diff(A.indptr)
    assert lens.max() > 8 * np.median(lens)          # really skewed
    om = synth.gaussian_panel(n, k + p, 5).numpy()
    A64 = A.astype(np.float64)
    want = O.fit(A64.indptr.astype(np.int64), A64.indices.astype(np.int64), A64.data, m, n, n_components=k, n_oversamples=p,
                 n_power_iterations=q, omega=om)
    outs = []
    for env in ("SAPCA_ROWSORT_ALWAYS", "SAPCA_NO_ROWSORT", None):
        monkeypatch.delenv("SAPCA_ROWSORT_ALWAYS", raising=False)
        monkeypatch.delenv("SAPCA_NO_ROWSORT", raising=False)
        if env:
            monkeypatch.setenv(env, "1")
        pca = _builder(k, p, q).spmm_variant(2).build().set_omega(om)
        t = pca.fit_transform(A)
        np.testing.assert_allclose(pca.singular_values_(np.float64), want.singular_values, rtol=1e-4)
        assert O.subspace_angle(pca.components_(np.float64), want.components) < 1e-4
        outs.append((pca.singular_values_(np.float64), t))
    for o in outs[1:]:
        np.testing.assert_allclose(o[0], outs[0][0], rtol=1e-6)
        np.testing.assert_allclose(o[1], outs[0][1], atol=1e-4 * np.abs(outs[0][1]).max())


# ------------------------------------------------------------------ panels wider than 128 columns
# (the reference puts no limit on n_components + n_oversamples: sweeps in column passes of 64, the dense steps in blocks of
#  64 / 128 columns, the Cholesky factor on the host -- correct, not tuned)
@pytest.mark.parametrize("l", [129, 150, 192, 200, 270])
def test_normalizer_on_panels_wider_than_128_columns(session, l):
    P = synth.gaussian_panel(3000, l, 5).numpy().astype(np.float64)
    P *= np.logspace(0, 2, l)[None, :]
    Q = session.normalize_panel(P, PIN.QR)
    np.testing.assert_allclose(Q.T @ Q, np.eye(l), atol=5e-11)
    q_ref, _ = np.linalg.qr(P)
    assert O.subspace_angle(Q.T, q_ref.T) < 1e-9
    Rm = Q.T @ P
    assert np.allclose(np.tril(Rm, -1), 0, atol=1e-8 * np.abs(Rm).max()) and np.all(np.diag(Rm) > 0)
    Qf = session.normalize_panel(P.astype(np.float32), PIN.QR).astype(np.float64)
    np.testing.assert_allclose(Qf.T @ Qf, np.eye(l), atol=2e-4)


@pytest.mark.parametrize("dtype,l", [(np.float32, 150), (np.float32, 260), (np.float64, 150)])
def test_sweeps_on_panels_wider_than_128_columns(session, session_tiled, dtype, l):
    """both sweep kernels, A and A^T, with and without the centring vector"""
    m, n = 5000, 1300
    ptr, idx, val = csr_np(synth.flat_csr(m, n, 0.05, seed=6, dtype=torch.float32 if dtype == np.float32 else torch.float64))
    X = synth.gaussian_panel(n, l, 3).numpy().astype(dtype)
    Yin = synth.gaussian_panel(m, l, 4).numpy().astype(dtype)
    mu = (np.arange(n) % 7 / 7.0).astype(dtype)
    A64 = mat(ptr, idx, val, m, n).astype(np.float64)
    tol = 2e-5 if dtype == np.float32 else 1e-11
    for sess in (session, session_tiled):
        a = sess.spmm(ptr, idx, val, m, n, X)
        want = A64 @ X.astype(np.float64)
        np.testing.assert_allclose(a, want, atol=tol * np.abs(want).max())
        ac = sess.spmm(ptr, idx, val, m, n, X, mu)
        want_c = want - np.outer(np.ones(m), mu.astype(np.float64) @ X.astype(np.float64))
        np.testing.assert_allclose(ac, want_c, atol=20 * tol * np.abs(want).max())
        b = sess.spmm(ptr, idx, val, m, n, Yin, None, transposed=True)
        want_t = A64.T @ Yin.astype(np.float64)
        np.testing.assert_allclose(b, want_t, atol=tol * np.abs(want_t).max())


@pytest.mark.parametrize("dtype,variant,k,p,shape", [(np.float32, 2, 140, 12, (8000, 2800, 0.20, 14.0, 11.5, 2.0)),
                                                     (np.float32, 0, 140, 12, (8000, 2800, 0.20, 14.0, 11.5, 2.0)),
                                                     (np.float64, 2, 136, 10, (8000, 2800, 0.20, 14.0, 11.5, 2.0)),
                                                     (np.float32, 2, 250, 20, (10000, 3600, 0.15, 30.0, 26.0, 1.7))])
def test_randomized_fit_with_more_than_128_panel_columns(dtype, variant, k, p, shape):
    """n_components + n_oversamples = 146 .. 270 against the oracle on the same Omega: singular values, subspace, mean,
    the projection (k > 128 columns of it), f32 on both sweep kernels and f64.  The matrices have a certified gap at k
    (sigma_k / sigma_{k+1} of the centred operator, dense SVD)."""
    m, n, dens, w_hi, w_lo, min_gap = shape
    q = 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, dens, k, seed=77, centred=True, dtype=torch.float64, w_hi=w_hi, w_lo=w_lo))
    D = mat(ptr, idx, val, m, n).toarray()
    sv = np.linalg.svd(D - D.mean(axis=0), compute_uv=False)
    assert sv[k - 1] >= min_gap * sv[k], "the generator left no gap at k"
    om = synth.gaussian_panel(n, k + p, 5).numpy()
    pca = _builder(k, p, q).spmm_variant(variant).build().set_omega(om)
    t = pca.fit_transform(mat(ptr, idx, val.astype(dtype), m, n))
    want = O.fit(ptr, idx, val.astype(dtype).astype(np.float64), m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    srel, ang = (1e-4, 1e-4) if dtype == np.float32 else (1e-9, 1e-8)
    np.testing.assert_allclose(pca.singular_values_(np.float64), want.singular_values, rtol=srel)
    assert O.subspace_angle(pca.components_(np.float64), want.components) < ang
    np.testing.assert_allclose(pca.mean_(np.float64), want.mean, rtol=1e-5, atol=1e-7)
    assert t.shape == (m, k)
    wt = O.transform_sparse(ptr, idx, val.astype(dtype).astype(np.float64), m, n, pca.components_(np.float64), pca.mean_(np.float64), True)
    np.testing.assert_allclose(t, wt, atol=(5e-4 if dtype == np.float32 else 1e-8) * max(1.0, float(np.abs(wt).max())))


def test_masked_fit_and_lanczos_with_more_than_128_components():
    """a masked randomized fit (l = 150 over the kept columns) against the oracle, and a Lanczos fit with 140 components
    against the exact SVD of the dense operator"""
    m, n, k, p, q = 8000, 3600, 138, 12, 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.2, k, seed=78, centred=True, dtype=torch.float64, w_hi=14.0, w_lo=11.5))
    mask = synth.bernoulli_mask(n, 0.8, 3).numpy()
    D = mat(ptr, idx, val, m, n).toarray()[:, mask]
    sv = np.linalg.svd(D - D.mean(axis=0), compute_uv=False)
    if sv[k - 1] < 1.8 * sv[k]:
        pytest.skip("no spectral gap at k under this mask: skipped, not loosened")
    n_used = int(mask.sum())
    om = synth.gaussian_panel(n_used, k + p, 6).numpy()
    est = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).random_seed(42)
           .svd_method(SVDMethod.Random(p, q, PIN.QR)).spmm_variant(2).build().set_omega(om))
    t = est.fit_transform(mat(ptr, idx, val.astype(np.float32), m, n))
    want = O.fit(ptr, idx, val.astype(np.float32).astype(np.float64), m, n, n_components=k, n_oversamples=p, n_power_iterations=q,
                 omega=om, mask=mask)
    np.testing.assert_allclose(est.singular_values_(np.float64), want.singular_values, rtol=1e-4)
    assert O.subspace_angle(est.components_(np.float64), want.components) < 1e-4
    assert t.shape == (m, k)
    # Lanczos, uncentred (Q1), k = 140 on a smaller operator
    m2, n2, k2 = 5000, 2800, 140
    ptr2, idx2, val2 = csr_np(synth.gapped_csr(m2, n2, 0.2, k2, seed=79, centred=False, dtype=torch.float64, w_hi=14.0, w_lo=11.5))
    lz = sapca.SparsePCABuilder.new().n_components(k2).svd_method(SVDMethod.Lanczos()).build()
    t2 = lz.fit_transform(mat(ptr2, idx2, val2, m2, n2))
    _, sv2, vt2 = np.linalg.svd(mat(ptr2, idx2, val2, m2, n2).toarray(), full_matrices=False)
    np.testing.assert_allclose(lz.singular_values_(np.float64), sv2[:k2], rtol=1e-8)
    if sv2[k2 - 1] >= 1.5 * sv2[k2]:
        assert O.subspace_angle(lz.components_(np.float64), vt2[:k2]) < 1e-6
    assert t2.shape == (m2, k2)


@pytest.mark.parametrize("semantics", ["reference", "centred"])
def test_projection_through_an_operator_whose_tile_range_is_split(semantics):
    """a matrix with few row blocks (a shard of a strong-scaled fit: 125k rows of C4 make 123 blocks) splits the tile range of
    A's operator over workgroups; fit_transform's projection runs through that operator too (it used to fall back to the row
    kernel: 2.0 instead of 0.55 ms on such a shard) -- against the oracle's projection, unmasked and masked"""
    m, n, k, p, q = 30000, 2500, 8, 6, 2
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.06, k, seed=21, dtype=torch.float32))
    om = synth.gaussian_panel(n, k + p, 3).numpy()
    b = _builder(k, p, q).spmm_variant(2)
    if semantics == "centred":
        b = b.transform_semantics(L.TRANSFORM_CENTERED)
    pca = b.build().set_omega(om)
    t = pca.fit_transform(mat(ptr, idx, val, m, n))
    comps, mean = pca.components_(np.float64), pca.mean_(np.float64)
    if semantics == "reference":
        want = O.transform_sparse(ptr, idx, val.astype(np.float64), m, n, comps, mean, True)      # Q2
    else:
        want = (mat(ptr, idx, val.astype(np.float64), m, n).toarray() - mean[None, :]) @ comps.T
    np.testing.assert_allclose(t, want, atol=5e-4 * max(1.0, float(np.abs(want).max())))
    t2 = pca.transform(mat(ptr, idx, val, m, n))
    np.testing.assert_allclose(t2, t, atol=5e-4 * max(1.0, float(np.abs(t).max())))
    mask = synth.bernoulli_mask(n, 0.8, 3).numpy()
    est = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).random_seed(42)
           .svd_method(SVDMethod.Random(p, q, PIN.QR)).spmm_variant(2).build())
    tm = est.fit_transform(mat(ptr, idx, val, m, n))
    wm = O.transform_masked_fast(ptr, idx, val.astype(np.float64), m, n, est.components_(np.float64), est.mean_(np.float64), True, mask)
    np.testing.assert_allclose(tm, wm, atol=5e-4 * max(1.0, float(np.abs(wm).max())))


@pytest.mark.parametrize("l", [1, 2, 3, 7, 16, 33, 60, 61, 90, 111, 112, 113])
def test_device_eigensolver_against_the_host_one(debug_switches, monkeypatch, l):
    """the l x l Gram of the small SVD: parallel Jacobi in one workgroup on the device (l <= 112, SAPCA_EIG_DEVICE=1: an
    experiment, four times slower than the default) against the host's Householder + QL (l = 113 takes the host on both runs) -- the same singular values to 1e-6 relative
    (f32 panels upstream: both solvers see the same f64 Gram, and agree to ~1e-12 of sigma_0), the same subspace, the same
    projection; odd l (a dummy index in the tournament), l = 1, and the oracle's numbers"""
    m, n = 4000, 900
    k = max(1, l - 4) if l > 8 else l
    p, q = l - k, 1
    ptr, idx, val = csr_np(synth.gapped_csr(m, n, 0.08, min(k, 40), seed=l, dtype=torch.float32))
    om = synth.gaussian_panel(n, l, 4).numpy()
    out = []
    for host in (False, True):
        if host:
            monkeypatch.delenv("SAPCA_EIG_DEVICE", raising=False)
        else:
            monkeypatch.setenv("SAPCA_EIG_DEVICE", "1")
        pca = _builder(k, p, q).build().set_omega(om)
        t = pca.fit_transform(mat(ptr, idx, val, m, n))
        out.append((pca.singular_values_(np.float64), pca.components_(np.float64), t, pca.explained_variance_ratio(np.float64)))
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=1e-6, atol=1e-9 * out[1][0][0])
    np.testing.assert_allclose(out[0][3], out[1][3], rtol=1e-5, atol=1e-9)
    want = O.fit(ptr, idx, val.astype(np.float64), m, n, n_components=k, n_oversamples=p, n_power_iterations=q, omega=om)
    np.testing.assert_allclose(out[0][0], want.singular_values, rtol=2e-4)
    # the leading components (well separated in this generator) agree between the two solvers; so does the projection on them
    j = min(k, 5)
    assert O.subspace_angle(out[0][1][:j], out[1][1][:j]) < 1e-4
    np.testing.assert_allclose(np.abs(out[0][2][:, :j]), np.abs(out[1][2][:, :j]), atol=2e-3 * np.abs(out[1][2]).max())


def test_device_eigensolver_on_a_rank_deficient_gram(debug_switches, monkeypatch):
    """more panel columns than the matrix has rank: the Gram has zero eigenvalues; the solver must converge (rotations against
    rounding noise are skipped) and the surplus directions come out as zero components with sigma ~ 0"""
    rng = np.random.default_rng(5)
    m, n, r, k, p = 3000, 500, 6, 10, 6
    U = rng.standard_normal((m, r)) * (rng.random((m, r)) < 0.2)
    W = rng.standard_normal((r, n)) * (rng.random((r, n)) < 0.3)
    D = (U @ W).astype(np.float32)
    A = sp.csr_matrix(D)
    A.sort_indices()
    monkeypatch.setenv("SAPCA_EIG_DEVICE", "1")
    pca = _builder(k, p, 1).center(False).build()
    t = pca.fit_transform(A)
    sv = pca.singular_values_(np.float64)
    s_exact = np.linalg.svd(D.astype(np.float64), compute_uv=False)
    np.testing.assert_allclose(sv[:r], s_exact[:r], rtol=1e-4)
    assert np.all(sv[r:] <= 1e-3 * sv[0])
    assert np.all(np.isfinite(t))


@pytest.mark.parametrize("seed", range(int(os.environ.get("SAPCA_FUZZ_SEEDS", "20"))))   # (SAPCA_FUZZ_SEEDS=300: a longer soak)
def test_gather_fill_on_random_shapes_is_the_bucket_route_bit_for_bit(debug_switches, monkeypatch, seed):
    """random shapes (1 to 500 tiles of A rows, 70 to 60 000 columns: block counts from a few to 64, blocks of 512 and of 1024
    rows), densities from 0.2 % to 30 %, a few dense rows and empty rows thrown in: A^T's format through the gather fill and
    through the bucket route must give bit-identical fits (same bytes, same per-tile column sums)"""
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.choice([300, 321, 2000, 5000, 20000, 60000, 160000]))
    n = int(rng.choice([70, 600, 1023, 1025, 4000, 17000, 60000]))
    dens = float(rng.choice([0.002, 0.01, 0.05, 0.3]))
    if m * n > 4e8:                       # (the generator hashes every cell: keep the host side of the test short)
        m = int(4e8 / n)
    if m * n * dens > 1e7:
        dens = 1e7 / (m * n)
    if m * n * dens < 2000:
        dens = min(0.5, 2000 / (m * n))
    k, p, q = 5, 5, 1
    ptr, idx, val = csr_np(synth.flat_csr(m, n, dens, seed=seed, dtype=torch.float32, device="cuda"))
    A = mat(ptr, idx, val, m, n)
    for r in rng.choice(m, 3, replace=False):                     # rows without entries
        A.data[A.indptr[int(r)]:A.indptr[int(r) + 1]] = 0
    A.eliminate_zeros()
    if n <= 4000:                                                 # one dense row
        r = int(rng.integers(m))
        dense = sp.csr_matrix(rng.uniform(0.5, 1.5, (1, n)).astype(np.float32))
        A = sp.vstack([A[:r], dense, A[r + 1:]]).tocsr()
    A.sort_indices()
    om = synth.gaussian_panel(n, k + p, 4).numpy()
    dev = sapca.DeviceCsr(torch.as_tensor(A.indptr.astype(np.int64), device="cuda"), torch.as_tensor(A.indices.astype(np.int32), device="cuda"),
                          torch.as_tensor(A.data.astype(np.float32), device="cuda"), (m, n))
    out = []
    for buckets in (False, True):
        if buckets:
            monkeypatch.setenv("SAPCA_AT_BUCKETS", "1")
        else:
            monkeypatch.delenv("SAPCA_AT_BUCKETS", raising=False)
        pca = _builder(k, p, q).spmm_variant(2).build().set_omega(om)
        t = pca.fit_transform(dev).cpu().numpy()
        out.append((pca.singular_values_(np.float64), pca.mean_(np.float64), pca.components_(np.float64), t))
    for a, b in zip(out[0], out[1]):
        np.testing.assert_array_equal(a, b)
    want_mean = np.asarray(A.astype(np.float64).sum(axis=0)).ravel() / m
    np.testing.assert_allclose(out[0][1], want_mean, rtol=1e-6, atol=1e-9)
