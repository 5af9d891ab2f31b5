"""How the staged sweep copes with skewed rows and columns (log-normal cell depth, Zipf-like gene popularity):
prints the formats' padding (stored slots / entries) and the per-sweep times next to the homogeneous C2 matrix."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "single-algebra_amd", "python"))
import numpy as np, torch
import sapca
from sapca import synth

m, n, k, p, q = 200_000, 20_000, 50, 10, 2
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(1)
# per-row depth (log-normal, sigma 0.8) and per-column popularity (power law), Bernoulli entries with p_ij = min(1, d_i * w_j)
depth = torch.exp(0.8 * torch.randn(m, device=dev, generator=g))
pop = (torch.arange(1, n + 1, device=dev, dtype=torch.float64) ** -0.7)
pop = pop[torch.randperm(n, device=dev, generator=g)]
scale = 0.03 * m * n / float(depth.double().sum() * pop.sum())
rows, cols = [], []
for r0 in range(0, m, 2000):
    r1 = min(m, r0 + 2000)
    pr = (depth[r0:r1, None].double() * pop[None, :] * scale).clamp_(max=1.0)
    nz = torch.rand(pr.shape, device=dev, generator=g, dtype=torch.float64) < pr
    ri, ci = torch.nonzero(nz, as_tuple=True)
    rows.append(ri + r0); cols.append(ci)
rows = torch.cat(rows); cols = torch.cat(cols).to(torch.int32)
vals = (1.0 + torch.rand(rows.numel(), device=dev, generator=g) * 4).float()
counts = torch.bincount(rows, minlength=m)
ptr = torch.zeros(m + 1, dtype=torch.int64, device=dev); ptr[1:] = torch.cumsum(counts, 0)
print("skewed: nnz", rows.numel(), "row length min/median/max", int(counts.min()), int(counts.median()), int(counts.max()),
      "col count min/median/max", *[int(x) for x in (torch.bincount(cols.long(), minlength=n).min(), torch.bincount(cols.long(), minlength=n).median(), torch.bincount(cols.long(), minlength=n).max())], flush=True)
x = sapca.DeviceCsr(ptr, cols, vals, (m, n))
pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).device(0).collect_timings(True).verbose(True)
       .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
for it in range(2):
    pca.fit(x)
t = pca.timings()
print("skewed: A sweeps", [round(v, 3) for v in t.spmm_sweep_ms[: t.n_spmm]], "At sweeps", [round(v, 3) for v in t.spmmt_sweep_ms[: t.n_spmmt]], "prepare", round(t.prepare_ms, 2))
