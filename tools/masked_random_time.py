"""fit_transform time of MaskedSparsePCA with SVDMethod::Random on the C3 matrix (200k x 30k, 60 % mask): python tools/masked_random_time.py [f32|f64]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))
import numpy as np, torch
import sapca
from sapca import synth
dtype = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
m, n, k = 200_000, 30_000, 30
ptr, idx, val = synth.gapped_csr(m, n, 0.03, k, seed=42, dtype=dtype, device="cuda")
mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
x = sapca.DeviceCsr(ptr, idx, val, (m, n))
est = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).collect_timings(True)
       .svd_method(sapca.SVDMethod.Random(10, 4, sapca.PowerIterationNormalizer.QR)).build())
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = est.fit_transform(x)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) * 1e3
    t = est.timings()
    print(f"fit {it}: {dt:.1f} ms  prepare {t.prepare_ms:.1f} stats {t.stats_ms:.1f} A sweeps {np.mean(t.spmm_sweep_ms[:t.n_spmm]):.2f} x{t.n_spmm} "
          f"At sweeps {np.mean(t.spmmt_sweep_ms[:t.n_spmmt]):.2f} x{t.n_spmmt} ortho {t.ortho_ms:.1f} small {t.small_svd_ms:.1f} transform {t.transform_ms:.1f}", flush=True)
