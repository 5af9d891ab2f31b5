#!/usr/bin/env python3
"""bench.py -- SparsePCA fit_transform on synthetic CSR, MI355X, one process per GPU.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one fit_transform of the hot path over a device-resident CSR (inputs already in
HBM when the timed region starts).  At N=1 the workload is BASELINE.json configs[1] (C2:
200k x 20k f32, ~97 % sparse, randomized SVD k=50, p=10, q=4, QR normalizer).  For N>1 every
rank holds a 200k-row shard of the (N*200k) x 20k matrix (weak scaling), rows range-partitioned,
panels all-reduced over RCCL inside the library.

One JSON line on rank 0: metric/value = whole-job algorithmic GB/s of fit_transform (SURVEY.md
§8d formula / wall-clock), ms_per_step = fit_transform wall-clock, `roofline` for the dominant
kernel (the sparse x dense sweep, HIP-event timed on the library's stream), `cpu_baseline` = the
C/OpenMP restatement of the reference algorithm timed on a bounded row sample of the same
workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

WORKLOADS = {
    # name: (rows per GPU, cols, density, k, p, q)
    "c2": (200_000, 20_000, 0.03, 50, 10, 4),
    "c4": (1_000_000, 30_000, 0.03, 50, 10, 4),
    "c5": (2_000_000, 50_000, 0.01, 100, 10, 4),
    "small": (20_000, 4_000, 0.03, 20, 10, 4),
    # BASELINE.json configs[2]: MaskedSparsePCA, f64, 60 % feature mask, SVDMethod::Lanczos k=30 (p, q unused)
    "c3": (200_000, 30_000, 0.03, 30, 0, 0),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def alg_bytes(m, n, nnz, l, k, q, tsize=4):
    sweep = nnz * (tsize + 4) + (m + 1) * 8 + n * l * tsize + m * l * tsize
    stats = nnz * (tsize + 4) + (m + 1) * 8 + 2 * n * 8
    transform = nnz * (tsize + 4) + (m + 1) * 8 + n * k * tsize + m * k * tsize
    return sweep, (2 * q + 2) * sweep + stats + transform


def cpu_baseline(name, n, density, k, p, q, seed, gen_device="cpu"):
    """The oracle's C restatement ("port") on a bounded row sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from sapca import synth
    ms = 30000 if name != "small" else 5000
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    orc.set_num_threads(min(share, 16))      # a 1-GPU box gives this job 16 host cores
    ptr, idx, val = synth.gapped_csr(ms, n, density, k, seed=seed, dtype=torch.float32, device=gen_device, chunk_elems=1 << 24)
    ptr, idx, val = ptr.cpu().numpy(), idx.cpu().numpy().astype(np.int64), val.cpu().numpy()
    om = synth.gaussian_panel(n, k + p, 42).numpy().astype(np.float32)
    t0 = time.perf_counter()
    rc, comps, sing, ev, mean, tv = orc.randomized_fit(ptr, idx, val, ms, n, k, p, q, "QR", True, om)
    orc.transform_sparse(ptr, idx, val, ms, n, comps, mean, True)   # closed form of the Q2 loop (the literal loop is O(m*k*nnz))
    dt = time.perf_counter() - t0
    _, total = alg_bytes(ms, n, len(val), k + p, k, q)
    return {"value": total / dt / 1e9, "unit": "GB/s", "cores": orc.num_threads(), "kind": "port",
            "sample": f"{ms} x {n} row sample of the workload ({len(val)} stored entries), fit + closed-form transform, "
                      f"{dt:.2f} s; restatement of the reference algorithm, not the reference binary"}


def bench_lanczos(args, rank, local_rank, world, dev):
    """configs[2]: MaskedSparsePCA fit_transform, f64, Lanczos on the mask-compacted (uncentred) operator."""
    import sapca
    from sapca import synth
    assert world == 1, "the c3 workload is a 1-GPU configuration"
    m, n, density, k, _, _ = WORKLOADS["c3"]
    ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, centred=False, dtype=torch.float64, device=dev)
    mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
    x = sapca.DeviceCsr(ptr, idx, val, (m, n))
    est = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).device(local_rank).collect_timings(True)
           .svd_method(sapca.SVDMethod.Lanczos()).build())
    for _ in range(args.warmup):
        out = est.fit_transform(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lz_ms, steps = 0.0, 0
    for _ in range(args.steps):
        out = est.fit_transform(x)
        t = est.timings()
        lz_ms += t.lanczos_ms
        steps += int(t.lanczos_steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_used = int(mask.sum())
    nnz_used = int(mask[idx.cpu().numpy()].sum())
    # per Lanczos step 2*[nnz'*(8+4) + (m+1)*8] + (m + 2n')*8 bytes
    step_bytes = 2 * (nnz_used * 12 + (m + 1) * 8) + (m + 2 * n_used) * 8
    achieved = step_bytes * steps / (lz_ms * 1e-3) / 1e9
    total_bytes = step_bytes * steps / args.steps + 2 * (val.numel() * 12 + (m + 1) * 8)   # + stats and transform passes
    print(json.dumps({
        "metric": "masked_sparse_pca_lanczos_fit_transform_algorithmic_throughput", "value": total_bytes / (dt / args.steps) / 1e9,
        "unit": "GB/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"c3: MaskedSparsePCA {m} x {n} CSR f64 density {density}, 60 % Bernoulli mask seed 7 ({n_used} kept), "
                               f"SVDMethod::Lanczos k={k} (uncentred operator), inputs resident in HBM",
                   "nnz": int(val.numel()), "nnz_masked": nnz_used, "lanczos_steps_per_fit": steps / args.steps},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": None, "kernel": "Lanczos step (SpMV pair + re-orthogonalisation), HIP events on the library stream",
                     "algorithmic_bytes_per_launch": step_bytes, "avg_launch_ms": lz_ms / max(steps, 1)}}))
    assert out.shape == (m, k)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--spmm-variant", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    import sapca
    from sapca import synth
    m, n, density, k, p, q = WORKLOADS[args.workload]
    seed = 42
    if args.workload == "c3":
        return bench_lanczos(args, rank, local_rank, world, dev)
    ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=seed, row_start=rank * m, dtype=torch.float32, device=dev)
    x = sapca.DeviceCsr(ptr, idx, val, (m, n))
    nnz = x.nnz
    pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).device(local_rank).collect_timings(True)
           .spmm_variant(args.spmm_variant)
           .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
    transport = "none"
    if world > 1:
        from sapca import dist as sdist
        transport = sdist.init_comm(pca)        # RCCL inside the library; torch.distributed callback as the fallback

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = pca.fit_transform(x)
    barrier()
    t0 = time.perf_counter()
    sweep_ms = []
    stage = {}
    for _ in range(args.steps):
        out = pca.fit_transform(x)
        t = pca.timings()
        sweep_ms += list(t.spmm_sweep_ms[: t.n_spmm]) + list(t.spmmt_sweep_ms[: t.n_spmmt])
        for f in ("prepare_ms", "stats_ms", "spmm_ms", "spmmt_ms", "ortho_ms", "small_svd_ms", "transform_ms", "comm_ms", "fit_total_ms"):
            stage[f] = stage.get(f, 0.0) + getattr(t, f) / args.steps
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nn = torch.tensor([float(nnz)], dtype=torch.float64, device=dev)
        dist.all_reduce(nn)
        nnz_total = float(nn.item())
    else:
        nnz_total = float(nnz)
    assert out.shape == (m, k) and bool(torch.isfinite(out).all())

    l = k + p
    sweep_bytes, _ = alg_bytes(m, n, nnz, l, k, q)                      # per rank (one launch)
    _, total_bytes = alg_bytes(m * world, n, nnz_total, l, k, q)        # whole job
    ms_per_step = dt / args.steps * 1e3
    avg_sweep_ms = float(np.mean(sweep_ms)) if sweep_ms else float("nan")
    achieved = sweep_bytes / (avg_sweep_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(args.workload, {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    if rank == 0:
        line = {
            "metric": "sparse_pca_fit_transform_algorithmic_throughput", "value": total_bytes / (dt / args.steps) / 1e9,
            "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {m * world} x {n} CSR f32, density {density}, gapped generator seed {seed}, "
                                   f"SparsePCA fit_transform, SVDMethod::Random k={k} p={p} q={q} QR, rows range-partitioned "
                                   f"over {world} GPU(s), inputs resident in HBM, collectives: {transport}",
                       "nnz": int(nnz_total), "rows_per_gpu": m, "sweeps_per_fit": 2 * q + 2, "stage_ms": stage},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "kernel": "spmm sweep (A*X and A^T*Y launches, HIP events on the library stream)",
                         "algorithmic_bytes_per_launch": sweep_bytes, "avg_launch_ms": avg_sweep_ms,
                         "launches_timed": len(sweep_ms)},
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, n, density, k, p, q, seed, dev)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
