"""Times the C2 sweeps with whatever library SAPCA_LIB_PATH / SAPCA_TILED_MODE select (kernel experiments).
Results of ablated builds are numerically meaningless; only the per-sweep times are read."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))
import numpy as np, torch
import sapca
from sapca import synth
m, n, density, k, p, q = 200_000, 20_000, 0.03, 50, 10, 2
dev = torch.device("cuda", 0)
if os.environ.get("GEN") == "flat":
    ptr, idx, val = synth.flat_csr(m, n, density, seed=42, dtype=torch.float32, device=dev)
else:
    ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device=dev)
x = sapca.DeviceCsr(ptr, idx, val, (m, n))
pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).device(0).collect_timings(True)
       .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.NONE)).build())
a, at = [], []
for it in range(3):
    try:
        pca.fit(x)
    except Exception as e:
        pass
    t = pca.timings()
    if it:
        a += list(t.spmm_sweep_ms[: t.n_spmm]); at += list(t.spmmt_sweep_ms[: t.n_spmmt])
tag = os.path.basename(os.environ.get("SAPCA_LIB_PATH", "default")) + " mode " + os.environ.get("SAPCA_TILED_MODE", "0") + " fmt " + os.environ.get("SAPCA_TILED_FMT", "1") + " " + os.environ.get("GEN", "gapped")
print(f"{tag:52s} A sweep {np.mean(a):.3f} ms   At sweep {np.mean(at):.3f} ms   (n={len(a)},{len(at)})")
