#!/usr/bin/env python3
"""tests/golden/ref_pins_preproc.npz -- DATA the reference's own tests hold for the preprocessing and statistics
traits, typed in from the test sources (inputs and expected outputs only):

  * src/sparse/csr.rs:1516-1552  test_csr_normalize: 3x3 triplets, column sums [2,7,3] / row sums [5,5,2], target 1
  * src/sparse/csr.rs:1385-1422  4x3 matrix with nonzero_col == [2,2,2], nonzero_row == [2,0,2,2]
  * src/sparse/csc.rs:1071-1226  3x3 matrix with sum_col == [5,3,7], sum_row == [3,3,9], col 0 min/max == 1/4,
                                 row 2 min/max == 4/5

Run from the repo root:  python tests/golden/make_golden_preproc.py
"""
import os

import numpy as np

OUT = os.path.dirname(os.path.abspath(__file__))
np.savez_compressed(
    os.path.join(OUT, "ref_pins_preproc.npz"),
    norm_rows=np.array([0, 0, 1, 1, 2]), norm_cols=np.array([0, 1, 1, 2, 2]), norm_vals=np.array([2.0, 3.0, 4.0, 1.0, 2.0]),
    norm_col_sums=np.array([2.0, 7.0, 3.0]), norm_row_sums=np.array([5.0, 5.0, 2.0]), norm_target=np.array(1.0),
    norm_expected_col=np.array([1.0, 3.0 / 7.0, 4.0 / 7.0, 1.0 / 3.0, 2.0 / 3.0]),
    norm_expected_row=np.array([0.4, 0.6, 0.8, 0.2, 1.0]), norm_tol=np.array(1e-10),
    nz_dense=np.array([[1.0, 0, 2], [0, 0, 0], [3, 4, 0], [0, 5, 6]]), nz_col=np.array([2, 2, 2]), nz_row=np.array([2, 0, 2, 2]),
    sum_dense=np.array([[1.0, 0, 2], [0, 3, 0], [4, 0, 5]]), sum_col=np.array([5.0, 3, 7]), sum_row=np.array([3.0, 3, 9]),
    min_col0=np.array(1.0), max_col0=np.array(4.0), min_row2=np.array(4.0), max_row2=np.array(5.0))
print("ref_pins_preproc.npz written")
