"""Random shapes through the DPP-fed sweep (spmm_variant 2: forced) against the row kernel (variant 1), both products, centred and not:
   python3 tools/fuzz_sweep.py [cases] [seed]     -- ragged rows, empty rows and columns, very long rows, 1..128 panel columns"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))
import numpy as np, scipy.sparse as sp
import torch  # noqa: F401
from sapca import ops
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
dq, row = ops.Session(spmm_variant=2), ops.Session(spmm_variant=1)
worst = 0.0
for c in range(cases):
    m = int(rng.integers(1, 6000)); n = int(rng.integers(1, 5000)); l = int(rng.choice([1, 3, 16, 30, 60, 64, 65, 100, 128]))
    dens = float(rng.choice([0.002, 0.01, 0.05, 0.2]))
    A = sp.random(m, n, density=dens, format="csr", dtype=np.float32, random_state=int(rng.integers(1 << 30)))
    if rng.random() < 0.5 and m > 8:      # a few very long rows and a run of empty ones
        rows = rng.choice(m, size=3, replace=False)
        dense = sp.csr_matrix((np.ones(n, np.float32) * 0.5, (np.full(n, rows[0]), np.arange(n))), shape=(m, n))
        A = (A + dense).tocsr(); A.sum_duplicates()
        lil = A.tolil(); lil[rows[1], :] = 0; lil[rows[2], :] = 0; A = lil.tocsr(); A.eliminate_zeros()
    A.sort_indices()
    for transposed in (False, True):
        X = rng.standard_normal(((m if transposed else n), l)).astype(np.float32)
        mu = rng.standard_normal(n).astype(np.float32) if rng.random() < 0.5 else None
        a = dq.spmm(A.indptr, A.indices, A.data, m, n, X, mu, transposed)
        b = row.spmm(A.indptr, A.indices, A.data, m, n, X, mu, transposed)
        err = float(np.abs(a - b).max() / max(1e-30, np.abs(b).max()))
        worst = max(worst, err)
        if not np.isfinite(a).all() or err > 2e-5:
            print(f"MISMATCH case {c}: m {m} n {n} l {l} density {dens} transposed {transposed} centred {mu is not None}: rel err {err:.3e}")
            raise SystemExit(1)
print(f"{cases} shapes x 2 products agree with the row kernel; worst relative difference {worst:.2e}")
