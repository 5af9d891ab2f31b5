"""Oracle of the preprocessing / statistics traits against the data the reference's own tests hold
(tests/golden/ref_pins_preproc.npz; csr.rs:1385-1422, 1516-1552; csc.rs:1071-1226).  CPU only."""
import numpy as np
import scipy.sparse as sp

import sapca_oracle as O


def _csr(dense):
    A = sp.csr_matrix(dense)
    A.sort_indices()
    return A.indptr.astype(np.int64), A.indices.astype(np.int64), A.data.astype(np.float64), A.shape


def test_normalize_matches_the_reference_test_vectors(golden):
    g = golden("ref_pins_preproc.npz")
    A = sp.coo_matrix((g["norm_vals"], (g["norm_rows"], g["norm_cols"])), shape=(3, 3)).tocsr()
    A.sort_indices()
    ptr, idx, val = A.indptr, A.indices, A.data
    got_c = O.normalize_csr(ptr, idx, val, g["norm_col_sums"], float(g["norm_target"]), O.COLUMN)
    got_r = O.normalize_csr(ptr, idx, val, g["norm_row_sums"], float(g["norm_target"]), O.ROW)
    assert np.abs(got_c - g["norm_expected_col"]).max() < float(g["norm_tol"])     # csr.rs:1535-1538
    assert np.abs(got_r - g["norm_expected_row"]).max() < float(g["norm_tol"])     # csr.rs:1546-1549


def test_normalize_leaves_rows_with_non_positive_sums_alone():
    ptr, idx = np.array([0, 2, 3, 3]), np.array([0, 1, 1])
    val = np.array([2.0, -2.0, 5.0])
    out = O.normalize_csr(ptr, idx, val, [0.0, 5.0, 1.0], 10.0, O.ROW)      # csr.rs:1021-1027, 1053
    np.testing.assert_array_equal(out, [2.0, -2.0, 10.0])


def test_statistics_match_the_reference_test_vectors(golden):
    g = golden("ref_pins_preproc.npz")
    ptr, idx, val, (m, n) = _csr(g["nz_dense"])
    _, _, nzc, _, _ = O.stats_csr(ptr, idx, val, m, n, O.COLUMN)
    _, _, nzr, _, _ = O.stats_csr(ptr, idx, val, m, n, O.ROW)
    np.testing.assert_array_equal(nzc, g["nz_col"])                        # csr.rs:1410-1412
    np.testing.assert_array_equal(nzr, g["nz_row"])                        # csr.rs:1419-1421
    ptr, idx, val, (m, n) = _csr(g["sum_dense"])
    sc, _, _, loc, hic = O.stats_csr(ptr, idx, val, m, n, O.COLUMN)
    sr, _, _, lor, hir = O.stats_csr(ptr, idx, val, m, n, O.ROW)
    np.testing.assert_array_equal(sc, g["sum_col"])                        # csc.rs:1128-1129
    np.testing.assert_array_equal(sr, g["sum_row"])                        # csc.rs:1132-1133
    assert loc[0] == g["min_col0"] and hic[0] == g["max_col0"]             # csc.rs:1209-1210
    assert lor[2] == g["min_row2"] and hir[2] == g["max_row2"]             # csc.rs:1224-1225


def test_min_max_of_an_empty_row_keeps_the_initial_values():
    ptr, idx, val = np.array([0, 1, 1]), np.array([0]), np.array([3.0], dtype=np.float32)
    _, _, nz, lo, hi = O.stats_csr(ptr, idx, val, 2, 2, O.ROW)
    assert nz.tolist() == [1, 0] and lo[1] == np.finfo(np.float32).max and hi[1] == -np.finfo(np.float32).max   # csr.rs:932-933


def test_variance_formula():
    rng = np.random.default_rng(0)
    D = (rng.random((50, 7)) < 0.4) * rng.normal(size=(50, 7))
    ptr, idx, val, (m, n) = _csr(D)
    sm, sq, _, _, _ = O.stats_csr(ptr, idx, val, m, n, O.COLUMN)
    np.testing.assert_allclose(O.variance_from_sums(sm, sq, m), D.var(axis=0, ddof=1), rtol=1e-12, atol=1e-14)   # csr.rs:646-655


def test_log1p():
    v = np.array([0.0, 1.0, 9.0, 0.5], dtype=np.float32)
    np.testing.assert_allclose(O.log1p_csr(v), np.log(np.float32(1) + v), rtol=0)
