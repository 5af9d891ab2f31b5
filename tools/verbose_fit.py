import os, sys
sys.path.insert(0, "" + os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "single-algebra_amd", "python") + "")
import torch, sapca
from sapca import synth
m, n, density, k, p, q = 200_000, 20_000, 0.03, 50, 10, 4
dev = torch.device("cuda", 0)
ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device=dev)
x = sapca.DeviceCsr(ptr, idx, val, (m, n))
pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).device(0).verbose(True)
       .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
for _ in range(3): pca.fit(x)
