"""Separate transform() of a C2-sized matrix on caller-owned device arrays (no preparation is kept across calls there)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))
import torch, sapca
from sapca import synth
m, n, density, k, p, q = 200_000, 20_000, 0.03, 50, 10, 4
x = sapca.DeviceCsr(*synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device="cuda"), (m, n))
y = sapca.DeviceCsr(*synth.gapped_csr(m, n, density, k, seed=43, dtype=torch.float32, device="cuda"), (m, n))
pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42)
       .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
pca.fit(x)
for name, mtx in (("the fitted matrix (caller-owned arrays)", x), ("another matrix", y)):
    for _ in range(2):
        pca.transform(mtx)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        pca.transform(mtx)
    torch.cuda.synchronize()
    print(f"transform of {name}: {(time.perf_counter() - t0) * 100:.3f} ms", flush=True)
