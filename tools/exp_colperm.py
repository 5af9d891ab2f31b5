"""Would dealing A^T's quads to the waves round-robin pay?  Upper-bound experiment: the C2 matrix with its COLUMNS permuted so
that neighbouring columns come from different cluster blocks (what a wave of the A^T sweep walks is 64 consecutive columns of
A) against the generator's own column order; per-sweep times of A and A^T sweeps from the library's own events."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))
import numpy as np
import torch
import sapca
from sapca import synth

m, n, density, k, p, q = 200_000, 20_000, 0.03, 50, 10, 4
ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device="cuda")


def run(ptr, idx, val, tag):
    x = sapca.DeviceCsr(ptr, idx, val, (m, n))
    pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).collect_timings(True)
           .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
    for _ in range(3):
        pca.fit_transform(x)
    a, at, tot = [], [], []
    import time
    for _ in range(10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pca.fit_transform(x)
        torch.cuda.synchronize(); tot.append((time.perf_counter() - t0) * 1e3)
        t = pca.timings()
        a += list(t.spmm_sweep_ms[: t.n_spmm]); at += list(t.spmmt_sweep_ms[: t.n_spmmt])
    print(f"{tag:28s} step {np.median(tot):.3f} ms   A sweep {np.median(a):.4f}   A^T sweep {np.median(at):.4f}   slots A {t.sweep_slots_a} A^T {t.sweep_slots_at}", flush=True)


def permuted(kind):
    g = torch.Generator(device="cpu"); g.manual_seed(1)
    if kind == "random":
        perm = torch.randperm(n, generator=g)
    else:   # stride interleave: column c -> position (c % 51) * ceil(n / 51) + c // 51  (51 cluster blocks of ~392 columns)
        c = torch.arange(n)
        w = -(-n // 51)
        key = (c // w) + (c % w) * 51          # neighbours in the new order come from different blocks
        perm = torch.empty(n, dtype=torch.long); perm[torch.argsort(key)] = torch.arange(n)
    perm = perm.to(idx.device)
    rows = torch.repeat_interleave(torch.arange(m, device=idx.device), ptr[1:] - ptr[:-1])
    newc = perm[idx.long()]
    order = torch.argsort(rows * n + newc)
    return ptr, newc[order].to(torch.int32), val[order]


for rep in range(2):
    run(ptr, idx, val, "generator's column order")
    run(*permuted("stride"), "columns interleaved by block")
    run(*permuted("random"), "columns in random order")
