"""Kernel census of a row-sharded fit on ONE GPU (two members of a sapca_multi sharing the device, in-process transport):
   rocprofv3 --kernel-trace --stats -- python3 tools/multi_kernels.py [M N DENSITY K P Q]
Timings of shared-GPU members mean nothing; the list of kernels shows which routes a shard takes (row kernels, sorts)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))
import numpy as np, torch, scipy.sparse as sp
import sapca
from sapca import synth
a = sys.argv[1:]
m, n, d, k, p, q = (int(a[0]), int(a[1]), float(a[2]), int(a[3]), int(a[4]), int(a[5])) if len(a) >= 6 else (250000, 30000, 0.03, 50, 10, 4)
ptr, idx, val = synth.gapped_csr(m, n, d, k, seed=42, dtype=torch.float32)
A = sp.csr_matrix((val.numpy(), idx.numpy(), ptr.numpy()), shape=(m, n))
est = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).collect_timings(True)
       .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
md = sapca.MultiDevice(est, [0, 0])
for it in range(2):
    t = md.fit_transform(A)
tm = md.member(0).timings()
print("member 0: prepare %.1f spmm %.2f x%d spmmt %.2f x%d ortho %.1f small %.1f transform %.1f comm %.1f pieces %d" % (
    tm.prepare_ms, np.mean(tm.spmm_sweep_ms[:tm.n_spmm]), tm.n_spmm, np.mean(tm.spmmt_sweep_ms[:tm.n_spmmt]), tm.n_spmmt,
    tm.ortho_ms, tm.small_svd_ms, tm.transform_ms, tm.comm_ms, tm.at_sweep_pieces))
