"""Timing of the device-resident preprocessing / statistics entry points on the C2 matrix.
Wall-clock per call (each call synchronises), best of 5, with the algorithmic bytes each one has to move."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "single-algebra_amd", "python"))
import numpy as np, torch
from sapca import ops, synth
import sapca

m, n, density, k = 200_000, 20_000, 0.03, 50
ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device="cuda")
nnz = int(val.numel())
sess = ops.Session()
# a ResidentCsr over torch-owned device memory (same entry points as after sapca_upload_csr_*)
R = ops.ResidentCsr(sess, (m, n), nnz, np.float32, ptr.data_ptr(), idx.data_ptr(), val.data_ptr())


def best(fn, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts)


rs = R.stats(ops.ROW)[0]
cs = R.stats(ops.COLUMN)[0]
rows = [
    ("stats ROW (sum, sumsq, nonzero, min, max)", lambda: R.stats(ops.ROW), nnz * 4 + (m + 1) * 8),
    ("stats COLUMN (transposition + the same)", lambda: R.stats(ops.COLUMN), nnz * 8 + (m + 1) * 8),
    ("normalize ROW", lambda: R.normalize(rs, 1e4, ops.ROW), nnz * 8 + (m + 1) * 8),
    ("normalize COLUMN", lambda: R.normalize(cs, 1.0, ops.COLUMN), nnz * 12),
    ("log1p", lambda: R.log1p(), nnz * 8),
]
print(f"C2 resident matrix: {m} x {n}, {nnz} stored entries, f32")
for name, fn, nbytes in rows:
    t = best(fn)
    print(f"{name:45s} {t * 1e3:8.3f} ms   {nbytes / t / 1e9:8.1f} GB/s algorithmic   ({nbytes / t / 8e12 * 100:5.1f} % of 8 TB/s)")
