"""Per-sweep and preparation times of one randomized fit_transform on a gapped matrix of any shape:
   python tools/shape_time.py M N DENSITY K P Q [fits] [f32|f64]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))
import numpy as np, torch
import sapca
from sapca import synth
m, n, density, k, p, q = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
fits = int(sys.argv[7]) if len(sys.argv) > 7 else 2
dtype = torch.float64 if len(sys.argv) > 8 and sys.argv[8] == "f64" else torch.float32
dev = torch.device("cuda", 0)
ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, dtype=dtype, device=dev)
x = sapca.DeviceCsr(ptr, idx, val, (m, n))
pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).device(0).collect_timings(True)
       .spmm_variant(int(os.environ.get("SHAPE_VARIANT", "0")))
       .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
for it in range(fits):
    torch.cuda.synchronize()
    import time
    t0 = time.perf_counter()
    out = pca.fit_transform(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    t = pca.timings()
    print(f"fit {it}: {dt:.1f} ms  prepare {t.prepare_ms:.1f} stats {t.stats_ms:.1f}  A sweeps {np.mean(t.spmm_sweep_ms[:t.n_spmm]):.2f} ms x{t.n_spmm}  "
          f"At sweeps {np.mean(t.spmmt_sweep_ms[:t.n_spmmt]):.2f} ms x{t.n_spmmt}  ortho {t.ortho_ms:.1f} small {t.small_svd_ms:.1f} transform {t.transform_ms:.1f}  nnz {x.nnz}", flush=True)
