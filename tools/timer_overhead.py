"""Wall time of the C2 fit_transform loop with and without the per-stage HIP events (collect_timings)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))
import torch
import sapca
from sapca import synth
m, n, density, k, p, q = 200_000, 20_000, 0.03, 50, 10, 4
ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device="cuda")
x = sapca.DeviceCsr(ptr, idx, val, (m, n))
for rep in range(2):
    for flag in (True, False):
        pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).device(0).collect_timings(flag)
               .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
        for _ in range(3):
            pca.fit_transform(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            pca.fit_transform(x)
        torch.cuda.synchronize()
        print(f"collect_timings={flag}: {(time.perf_counter() - t0) * 100:.3f} ms/step", flush=True)
