"""PCIe-inclusive timing of the host entry point (sapca_fit_transform_csr_f32 with nalgebra-style u64 indices) on C2."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "single-algebra_amd", "python"))
import numpy as np, scipy.sparse as sp, torch
import sapca
from sapca import synth
m, n, density, k, p, q = 200_000, 20_000, 0.03, 50, 10, 4
ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device="cuda")
A = sp.csr_matrix((val.cpu().numpy(), idx.cpu().numpy().astype(np.int64), ptr.cpu().numpy()), shape=(m, n))
del ptr, idx, val
torch.cuda.empty_cache()
pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).collect_timings(True)
       .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
for it in range(4):
    t0 = time.perf_counter()
    out = pca.fit_transform(A)
    dt = (time.perf_counter() - t0) * 1e3
    t = pca.timings()
    print(f"run {it}: wall {dt:.1f} ms  upload {t.upload_ms:.1f} ms  fit {t.fit_total_ms:.1f} ms  transform {t.transform_ms:.1f} ms", flush=True)
