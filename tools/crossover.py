import sys, time; sys.path.insert(0, "single-algebra_amd/python")
import torch, sapca
from sapca import synth
dt = torch.float64 if len(sys.argv) > 1 and sys.argv[1] == "f64" else torch.float32
for (m, n, d) in ((30000, 8000, 0.04), (50000, 8000, 0.04), (100000, 10000, 0.03), (60000, 20000, 0.03), (150000, 4000, 0.03), (400000, 4000, 0.02)):
    ptr, idx, val = synth.gapped_csr(m, n, d, 20, seed=1, dtype=dt, device="cuda")
    x = sapca.DeviceCsr(ptr, idx, val, (m, n))
    res = []
    for v in (1, 2):
        pca = (sapca.SparsePCABuilder.new().n_components(50).random_seed(1).spmm_variant(v)
               .svd_method(sapca.SVDMethod.Random(10, 4, sapca.PowerIterationNormalizer.QR)).build())
        for i in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter(); pca.fit_transform(x); torch.cuda.synchronize(); t = (time.perf_counter() - t0) * 1e3
        res.append(t)
    print(f"{m} x {n} d={d} nnz={val.numel():.2e}: row {res[0]:.2f} ms  staged {res[1]:.2f} ms", flush=True)
