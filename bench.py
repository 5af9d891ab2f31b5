#!/usr/bin/env python3
"""bench.py -- SparsePCA fit_transform on synthetic CSR, MI355X, one process per GPU.

    python bench.py --gpus N --steps K --warmup W            (N > 1: starts its own N ranks through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --gpus N --scaling strong [--workload c4|c5]   (one matrix split over the ranks: BASELINE configs[3], [4])

A "step" is one fit_transform of the hot path over a device-resident CSR (inputs already in
HBM when the timed region starts).  At N=1 the workload is BASELINE.json configs[1] (C2:
200k x 20k f32, ~97 % sparse, randomized SVD k=50, p=10, q=4, QR normalizer); the line also
carries every other BASELINE config on the one GPU as a sub-record: `c3` (configs[2]: masked f64
Lanczos, 3 steps, roofline per Lanczos step), `c4_1gpu` (configs[3]'s 1M x 30k matrix: the shape
north_star's roofline target is quoted on, 3 steps) and `c5_1gpu` (configs[4]: 2M x 50k, k=100,
2 steps, sweep roofline + executed slots per stored entry).  For N>1 the default is BASELINE configs[3]
STRONG-scaled: the 1M x 30k matrix split by rows over the ranks (`"scaling": "strong"`), rows
range-partitioned, panels all-reduced over RCCL inside the library; `--scaling weak` gives
every rank a 200k-row shard of the (N*200k) x 20k matrix instead.

One JSON line on rank 0: metric/value = whole-job algorithmic GB/s of fit_transform (SURVEY.md
§8d formula / wall-clock), ms_per_step = fit_transform wall-clock, `roofline` for the dominant
kernel (the sparse x dense sweep, HIP-event timed on the library's stream), `cpu_baseline` = the
C/OpenMP restatement of the reference algorithm timed on the WHOLE workload on the box's host
cores (rank 0, N=1 only), `cpu_baseline_reference` = the reference's own crate (baseline/rust_ref)
where a Rust toolchain with a populated registry exists, else {"available": false, "reason": ...}.
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
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy; the line carries this box's own: peak_measured)


def alg_bytes(m, n, nnz, l, k, q, tsize=4):
    sweep = nnz * (tsize + 4) + (m + 1) * 8 + n * l * tsize + m * l * tsize
    stats = nnz * (tsize + 4) + (m + 1) * 8 + 2 * n * 8
    transform = nnz * (tsize + 4) + (m + 1) * 8 + n * k * tsize + m * k * tsize
    return sweep, (2 * q + 2) * sweep + stats + transform


def cpu_baseline(name, m, n, density, k, p, q, seed, gen_device="cpu", rows=None):
    """The oracle's C restatement ("port") on the whole workload; `rows` bounds the sample (--cpu-sample-rows)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    from sapca import synth
    ms = m if not rows else min(m, rows)
    try:
        share = len(os.sched_getaffinity(0))
    except AttributeError:
        share = os.cpu_count() or 1
    orc.set_num_threads(min(share, 16))      # a 1-GPU box gives this job 16 host cores
    ptr, idx, val = synth.gapped_csr(ms, n, density, k, seed=seed, dtype=torch.float32, device=gen_device, chunk_elems=1 << 24)
    ptr, idx, val = ptr.cpu().numpy(), idx.cpu().numpy().astype(np.int64), val.cpu().numpy()
    torch.cuda.empty_cache()
    om = synth.gaussian_panel(n, k + p, 42).numpy().astype(np.float32)
    t0 = time.perf_counter()
    rc, comps, sing, ev, mean, tv = orc.randomized_fit(ptr, idx, val, ms, n, k, p, q, "QR", True, om)
    t1 = time.perf_counter()
    orc.transform_sparse(ptr, idx, val, ms, n, comps, mean, True)   # closed form of the Q2 loop (the literal loop is O(m*k*nnz))
    dt = time.perf_counter() - t0
    _, total = alg_bytes(ms, n, len(val), k + p, k, q)
    what = "the whole workload" if ms == m else f"{ms} x {n} row sample of the workload"
    return {"value": total / dt / 1e9, "unit": "GB/s", "cores": orc.num_threads(), "cpu_model": cpu_model(), "kind": "port",
            "ms_per_step": dt * 1e3, "fit_ms": (t1 - t0) * 1e3,
            "sample": f"{what} ({ms} x {n}, {len(val)} stored entries), one fit + closed-form transform, "
                      f"{dt:.2f} s; restatement of the reference algorithm, not the reference binary"}


def reference_binary_baseline(timeout_s=120):
    """the reference's own crate (baseline/rust_ref: single_algebra =0.9.2, SparsePCABuilder ... fit) timed
    on this box -- only where `cargo` AND an offline registry holding its dependencies exist.  Runs in a child process that
    never touches the GPU; otherwise says why not."""
    import shutil
    import subprocess
    cargo = shutil.which("cargo")
    if cargo is None:
        return {"available": False, "reason": "reference binary: unavailable on this box (no `cargo` on PATH; no Rust toolchain in the image)"}
    home = os.environ.get("CARGO_HOME", os.path.expanduser("~/.cargo"))
    crate = os.path.join(ROOT, "baseline", "rust_ref")
    if not (os.path.isdir(os.path.join(home, "registry")) or os.path.isdir(os.path.join(crate, "vendor"))):
        return {"available": False, "reason": f"reference binary: unavailable on this box (cargo at {cargo}, but no populated registry under "
                                              f"{home}/registry and no vendored crates; there is no network)"}
    try:
        env = dict(os.environ, CARGO_NET_OFFLINE="true")
        b = subprocess.run([cargo, "build", "--release", "--offline"], cwd=crate, env=env, capture_output=True, text=True, timeout=timeout_s)
        if b.returncode != 0:
            return {"available": False, "reason": "reference binary: `cargo build --release --offline` failed: " + b.stderr[-300:]}
        out = {}
        for cfg in ("c1", "c2"):
            r = subprocess.run([cargo, "run", "--release", "--offline", "--", cfg], cwd=crate, env=env, capture_output=True, text=True,
                               timeout=timeout_s)
            out[cfg] = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "failed: " + r.stderr[-200:]
        return {"available": True, "kind": "reference", "cores": os.cpu_count(), "runs": out}
    except Exception as e:   # never lose the line to the probe
        return {"available": False, "reason": "reference binary: " + repr(e)}


def cpu_model():
    """Model name of the host CPU the baseline ran on (SURVEY.md 8d asks for model + core count)."""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.lower().startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or "unknown"


def run_lanczos(steps, warmup, local_rank, dev):
    """configs[2]: MaskedSparsePCA fit_transform, f64, Lanczos on the mask-compacted (uncentred) operator: the record."""
    import sapca
    from sapca import synth
    m, n, density, k, _, _ = WORKLOADS["c3"]
    ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, centred=False, dtype=torch.float64, device=dev)
    mask = synth.bernoulli_mask(n, 0.6, 7).numpy()
    x = sapca.DeviceCsr(ptr, idx, val, (m, n))
    est = (sapca.MaskedSparsePCABuilder.new().n_components(k).mask(mask).device(local_rank).collect_timings(True)
           .svd_method(sapca.SVDMethod.Lanczos()).build())
    for _ in range(warmup):
        out = est.fit_transform(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    lz_ms, lsteps = 0.0, 0
    stage = {}
    for _ in range(steps):
        out = est.fit_transform(x)
        t = est.timings()
        lz_ms += t.lanczos_ms
        lsteps += int(t.lanczos_steps)
        for f in ("prepare_ms", "stats_ms", "transform_ms", "fit_total_ms"):
            stage[f] = stage.get(f, 0.0) + getattr(t, f) / steps
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert out.shape == (m, k) and bool(torch.isfinite(out).all())
    n_used = int(mask.sum())
    nnz_used = int(torch.as_tensor(mask, device=dev)[idx.long()].sum().item())
    # per Lanczos step 2*[nnz'*(8+4) + (m+1)*8] + (m + 2n')*8 bytes
    step_bytes = 2 * (nnz_used * 12 + (m + 1) * 8) + (m + 2 * n_used) * 8
    achieved = step_bytes * lsteps / (lz_ms * 1e-3) / 1e9
    total_bytes = step_bytes * lsteps / steps + 2 * (val.numel() * 12 + (m + 1) * 8)   # + stats and transform passes
    rec = {"metric": "masked_sparse_pca_lanczos_fit_transform_algorithmic_throughput", "value": total_bytes / (dt / steps) / 1e9,
           "unit": "GB/s", "steps": steps, "warmup": warmup, "ms_per_step": dt / steps * 1e3, "dtype": "f64",
           "config": {"workload": f"c3: MaskedSparsePCA {m} x {n} CSR f64 density {density}, 60 % Bernoulli mask seed 7 ({n_used} kept), "
                                  f"SVDMethod::Lanczos k={k} (uncentred operator), inputs resident in HBM",
                      "nnz": int(val.numel()), "nnz_masked": nnz_used, "lanczos_steps_per_fit": lsteps / steps, "stage_ms": stage},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                        "traffic": None, "kernel": "Lanczos step (SpMV pair + re-orthogonalisation), HIP events on the library stream",
                        "algorithmic_bytes_per_launch": step_bytes, "avg_launch_ms": lz_ms / max(lsteps, 1), "launches_timed": lsteps}}
    del out, x, ptr, idx, val, est
    torch.cuda.empty_cache()
    return rec


def bench_lanczos(args, rank, local_rank, world, dev):
    assert world == 1, "the c3 workload is a 1-GPU configuration"
    rec = run_lanczos(args.steps, args.warmup, local_rank, dev)
    rec.update({"n_gpus": 1, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "data": "synthetic"})
    print(json.dumps(rec))


def measure_traffic(workload, n_sweeps, timeout_s=150):
    """HBM-side bytes of one sweep, measured in THIS run: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a
    pass) over one step of the same workload, each a child process (`rocprofv3 ... -- python3 bench.py --steps 1 --no-extras`);
    the sweep kernels' counters are summed over the step and divided by its sweeps.  Units and gfx950 corrections as
    MI355X_MICROARCH.md prescribes (KB -> bytes; FETCH_SIZE doubled: an upper estimate for this kernel's 8-byte-per-lane
    entry loads, see profiles/traffic.json).  Returns (bytes per sweep, how) or (None, why not)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rp = shutil.which("rocprofv3")
    if rp is None:
        return None, "rocprofv3 is not on PATH"
    kb = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="sapca_pmc_", dir="/tmp")
        cmd = [rp, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__),
               "--workload", workload, "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-extras"]
        try:
            subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=timeout_s)
            files = glob.glob(os.path.join(d, "*", "*counter_collection.csv"))
            if not files:
                return None, f"rocprofv3 --pmc {counter} left no counter file"
            total = 0.0
            for r in csv.DictReader(open(max(files, key=os.path.getmtime))):
                if r["Counter_Name"] == counter and any(kname in r["Kernel_Name"] for kname in ("spmm_dq", "spmm_quad", "spmm_rowgather", "spmm_tiled")):
                    total += float(r["Counter_Value"])
            kb[counter] = total / n_sweeps
        except Exception as e:
            return None, f"rocprofv3 --pmc {counter}: {e!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    if kb["FETCH_SIZE"] <= 0:
        return None, "no sweep kernel in the counter file"
    return int(kb["FETCH_SIZE"] * 1024 * 2 + kb["WRITE_SIZE"] * 1024), (
        "measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of one step of this workload as child processes; "
        f"per sweep (the step's sweep-kernel counters / {n_sweeps} sweeps); KB x 1024, FETCH_SIZE doubled (gfx950: an upper estimate here)")


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """`python3 bench.py --gpus N` started by hand: one child per GPU through torch.distributed.run, started BEFORE this
    process touches a GPU; the parent relays rank 0's JSON line and the children's exit status."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    sys.stdout.write(proc.stdout)
    sys.stdout.flush()
    raise SystemExit(proc.returncode)


def measured_copy_gbs(local_rank, nbytes=1 << 30, reps=5):
    """Attainable HBM rate of this device: the library's own 16-byte-per-lane streaming copy of `nbytes` (read + write
    counted, best of `reps`; sapca_measure_copy_gbs) -- not a framework memcpy."""
    import sapca
    est = sapca.SparsePCABuilder.new().device(local_rank).build()
    return est.measure_copy_gbs(nbytes, reps)


def c1_comparison(dev, local_rank):
    """BASELINE.json configs[0] (the reference's CPU-runnable case: 10k x 2k f64, 5 %, k=20, p=10, q=4): the restatement
    on the full problem next to the GPU's fit_transform of the same matrix."""
    import sapca
    from sapca import synth
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    m, n, density, k, p, q = 10_000, 2_000, 0.05, 20, 10, 4
    ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float64, device=dev)
    x = sapca.DeviceCsr(ptr, idx, val, (m, n))
    pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).device(local_rank)
           .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
    for _ in range(3):
        pca.fit_transform(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        pca.fit_transform(x)
    torch.cuda.synchronize()
    gpu_ms = (time.perf_counter() - t0) / 10 * 1e3
    hp, hi, hv = ptr.cpu().numpy(), idx.cpu().numpy().astype(np.int64), val.cpu().numpy()
    om = synth.gaussian_panel(n, k + p, 42).numpy()
    t0 = time.perf_counter()
    rc, comps, sing, ev, mean, tv = orc.randomized_fit(hp, hi, hv, m, n, k, p, q, "QR", True, om)
    orc.transform_sparse(hp, hi, hv, m, n, comps, mean, True)
    cpu_ms = (time.perf_counter() - t0) * 1e3
    _, total = alg_bytes(m, n, len(hv), k + p, k, q, tsize=8)
    return {"workload": f"c1: {m} x {n} CSR f64 density {density}, k={k} p={p} q={q} QR (BASELINE.json configs[0])",
            "value": total / (cpu_ms * 1e-3) / 1e9, "unit": "GB/s", "cores": orc.num_threads(), "kind": "port", "cpu_ms": cpu_ms,
            "gpu_ms": gpu_ms, "sample": "the full problem, fit + closed-form transform; restatement of the reference algorithm"}


def e2e_host_ms(args, m, n, density, k, p, q, dev, local_rank):
    """One fit_transform through the HOST entry point the reference signature maps to (usize indices in host memory ->
    result in host memory: index narrowing, PCIe, fit, projection, download); outside the timed loop."""
    import sapca
    import scipy.sparse as sp
    from sapca import synth
    ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device=dev)
    a = sp.csr_matrix((val.cpu().numpy(), idx.cpu().numpy().astype(np.int64), ptr.cpu().numpy()), shape=(m, n))
    del ptr, idx, val
    pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).device(local_rank).collect_timings(True)
           .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
    # what a Rust caller hands over: nalgebra_sparse's usize arrays, already in host memory
    a.indptr = a.indptr.astype(np.uint64)
    a.indices = a.indices.astype(np.uint64)
    a.has_sorted_indices = True
    best, parts = float("inf"), {}
    for _ in range(3):                       # the first call allocates the device buffers and pins the staging ring
        t0 = time.perf_counter()
        pca.fit_transform(a)
        dt = (time.perf_counter() - t0) * 1e3
        if dt < best:
            t = pca.timings()
            best, parts = dt, {"upload_ms": t.upload_ms, "fit_ms": t.fit_total_ms, "transform_ms": t.transform_ms}
    return best, parts


def run_randomized(workload, scaling, steps, warmup, rank, world, local_rank, dev, rdev, shared, spmm_variant):
    """`warmup` untimed + `steps` timed fit_transforms of one randomized-SVD workload, rows range-partitioned over the
    ranks (weak: every rank holds the workload's row count; strong: the workload's rows are split).  The timed region is
    bracketed by a barrier + torch.cuda.synchronize() on both sides; the returned wall time is the MAX over ranks."""
    import sapca
    import torch.distributed as dist
    from sapca import synth
    m_cfg, n, density, k, p, q = WORKLOADS[workload]
    seed = 42
    if scaling == "strong":
        # rows of the one m_cfg-row matrix, split evenly: the generator's rows are identically distributed, so equal row
        # counts are entry-balanced to 0.1 % (sapca_partition_rows does the same from the row offsets of a host matrix)
        r0, r1 = m_cfg * rank // world, m_cfg * (rank + 1) // world
        m_total = m_cfg
    else:
        r0, r1 = rank * m_cfg, (rank + 1) * m_cfg
        m_total = m_cfg * world
    m = r1 - r0
    ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=seed, row_start=r0, dtype=torch.float32, device=dev)
    x = sapca.DeviceCsr(ptr, idx, val, (m, n))
    nnz = x.nnz
    pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).device(local_rank).collect_timings(True)
           .spmm_variant(spmm_variant)
           .svd_method(sapca.SVDMethod.Random(p, q, sapca.PowerIterationNormalizer.QR)).build())
    transport = "none"
    if world > 1:
        from sapca import dist as sdist
        # RCCL inside the library; torch.distributed callback as the fallback (and the only way when ranks share a GPU)
        transport = sdist.init_comm(pca, prefer="torch" if shared else "rccl", stage_through_host=shared)
        if shared:
            transport += " (gloo on host copies: ranks share a GPU)"

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        out = pca.fit_transform(x)
    barrier()
    t0 = time.perf_counter()
    sweep_ms = []
    stage = {}
    for _ in range(steps):
        out = pca.fit_transform(x)
        t = pca.timings()
        sweep_ms += list(t.spmm_sweep_ms[: t.n_spmm]) + list(t.spmmt_sweep_ms[: t.n_spmmt])
        for f in ("prepare_ms", "stats_ms", "spmm_ms", "spmmt_ms", "ortho_ms", "small_svd_ms", "transform_ms", "comm_ms", "fit_total_ms"):
            stage[f] = stage.get(f, 0.0) + getattr(t, f) / steps
    barrier()
    dt = time.perf_counter() - t0
    avg_sweep_ms = float(np.mean(sweep_ms)) if sweep_ms else float("nan")
    per_rank_sweep = [avg_sweep_ms]
    t_last = pca.timings()
    per_rank_comm = [{"rank": rank, "transport": transport, "comm_ms": stage.get("comm_ms", 0.0),
                      "at_sweep_pieces": int(t_last.at_sweep_pieces), "side_lane": bool(pca.comm_has_side_lane()) if world > 1 else False}]
    if world > 1:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank_comm[0])
        per_rank_comm = gathered
        if len({g["transport"] for g in gathered}) != 1 or len({g["at_sweep_pieces"] for g in gathered}) != 1:
            # (ranks that disagree on the transport or on the piece count would have mismatched collectives: say so and leave,
            #  on every rank, instead of reporting a number from a run that only happened not to hang)
            if rank == 0:
                print(f"bench.py: ranks disagree on the collective transport / sweep pieces: {gathered}", file=sys.stderr, flush=True)
            raise SystemExit(3)
        tt = torch.tensor([dt], dtype=torch.float64, device=rdev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        nn = torch.tensor([float(nnz)], dtype=torch.float64, device=rdev)
        dist.all_reduce(nn)
        nnz_total = float(nn.item())
        sw = torch.zeros(world, dtype=torch.float64, device=rdev)
        sw[rank] = avg_sweep_ms
        dist.all_reduce(sw)
        per_rank_sweep = [float(v) for v in sw.tolist()]
    else:
        nnz_total = float(nnz)
    assert out.shape == (m, k) and bool(torch.isfinite(out).all())
    l = k + p
    sweep_bytes, _ = alg_bytes(m, n, nnz, l, k, q)                      # per rank (one launch)
    _, total_bytes = alg_bytes(m_total, n, nnz_total, l, k, q)          # whole job
    t = pca.timings()
    res = {"workload": workload, "scaling": scaling, "m_total": m_total, "m": m, "n": n, "density": density, "k": k, "p": p, "q": q,
           "seed": seed, "nnz": nnz, "nnz_total": nnz_total, "dt": dt, "steps": steps, "ms_per_step": dt / steps * 1e3,
           "value": total_bytes / (dt / steps) / 1e9, "stage": stage, "transport": transport, "sweep_ms": sweep_ms,
           "avg_sweep_ms": avg_sweep_ms, "per_rank_sweep": per_rank_sweep, "per_rank_comm": per_rank_comm, "sweep_bytes": sweep_bytes,
           "achieved": sweep_bytes / (avg_sweep_ms * 1e-3) / 1e9, "sweep_kernel": int(t.sweep_kernel),
           "slots": 0.5 * (t.sweep_slots_a + t.sweep_slots_at)}
    del out, x, ptr, idx, val, pca
    torch.cuda.empty_cache()
    return res


def sub_record(r):
    """A secondary workload in the line: what was run, its step time and the sweep's place against the HBM roofline."""
    return {"workload": f"{r['workload']}: {r['m_total']} x {r['n']} CSR f32, density {r['density']}, k={r['k']} p={r['p']} q={r['q']} QR, "
                        f"{r['scaling']} scaling, inputs resident in HBM",
            "steps": r["steps"], "ms_per_step": r["ms_per_step"], "value": r["value"], "unit": "GB/s", "nnz": int(r["nnz_total"]),
            "sweep_ms": r["avg_sweep_ms"], "prepare_ms": r["stage"].get("prepare_ms"), "stage_ms": r["stage"], "collectives": r["transport"],
            "slots_per_stored_entry": (r["slots"] / r["nnz"]) if r["slots"] else None,
            "roofline": {"bound": "hbm", "achieved": r["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["achieved"] / HBM_PEAK_GBS,
                         "algorithmic_bytes_per_launch": r["sweep_bytes"], "avg_launch_ms": r["avg_sweep_ms"],
                         "launches_timed": len(r["sweep_ms"])}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="strong: the workload's rows are split over the ranks (the default for --gpus N > 1, workload c4: 1M x 30k, "
                         "BASELINE.json configs[3]); weak: every rank holds the workload's rows (the default at one GPU, workload c2: "
                         "BASELINE.json configs[1])")
    ap.add_argument("--spmm-variant", type=int, default=0)
    ap.add_argument("--weak-c2", action="store_true",
                    help="N > 1: after the strong-scaled C4 headline also run 3 steps of weak-scaled C2 shards (a `weak_c2` record)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic with two rocprofv3 --pmc child runs (use the committed figure)")
    ap.add_argument("--cpu-sample-rows", type=int, default=0,
                    help="time the CPU restatement on the first ROWS rows of the workload instead of all of it (C2 whole: about a minute)")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the measured copy rate, the host-path run, the C1 comparison and the secondary workloads "
                         "(c4_1gpu at one GPU)")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    # the reference's own binary, before this process touches a GPU (the probe starts child processes)
    ref_probe = reference_binary_baseline() if (world == 1 and not args.no_cpu_baseline) else None
    if args.scaling is None:
        # the driver runs `bench.py --gpus N`: N > 1 must measure BASELINE.json configs[3] (1M x 30k split by rows over
        # the ranks: north_star's ">= 6x at 8 GPUs"), one GPU configs[1]
        args.scaling = "strong" if (world > 1 and args.workload in (None, "c4", "c5")) else "weak"
    if args.workload is None:
        args.workload = "c4" if args.scaling == "strong" else "c2"
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    shared = world > ndev                     # fewer GPUs than ranks (a one-GPU box rehearsing the launcher): ranks share
    local_rank = local_rank % max(ndev, 1)    # devices and the collectives go through gloo on host copies
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    rdev = torch.device("cpu") if shared else dev     # where this script's own small reductions live

    if args.workload == "c3":
        return bench_lanczos(args, rank, local_rank, world, dev)
    m_cfg, n, density, k, p, q = WORKLOADS[args.workload]
    # the host-path measurement first, on a process that has done nothing else yet (what a caller's process looks like;
    # measured after the timed loop the same upload took 33-49 ms instead of 21: host memory placement, not the library)
    e2e = None
    if world == 1 and not args.no_extras and args.workload in ("c2", "small"):
        try:
            e2e = e2e_host_ms(args, m_cfg, n, density, k, p, q, dev, local_rank)
        except Exception as e:   # never lose the line to an extra
            e2e = repr(e)
        torch.cuda.empty_cache()
    r = run_randomized(args.workload, args.scaling, args.steps, args.warmup, rank, world, local_rank, dev, rdev, shared, args.spmm_variant)
    # secondary workloads (every rank takes part): BASELINE configs[3] on one GPU beside the C2 headline (north_star's
    # roofline target is quoted on 1M x 30k at one GPU); the weak-scaled C2 shards beside the strong-scaled C4 headline
    extra = {}
    if not args.no_extras:
        try:
            if world == 1 and args.workload == "c2":
                # every other BASELINE config on this one GPU: configs[2] (masked f64 Lanczos), [3] and [4] (their matrices whole)
                for key, fn in (("c3", lambda: run_lanczos(3, 1, local_rank, dev)),
                                ("c4_1gpu", lambda: sub_record(run_randomized("c4", "strong", 3, 1, rank, world, local_rank, dev, rdev, shared, args.spmm_variant))),
                                ("c5_1gpu", lambda: sub_record(run_randomized("c5", "strong", 2, 1, rank, world, local_rank, dev, rdev, shared, args.spmm_variant)))):
                    try:
                        extra[key] = fn()
                    except Exception as e:
                        extra[key + "_error"] = repr(e)
                        torch.cuda.empty_cache()
            elif world > 1 and args.workload == "c4" and args.scaling == "strong" and args.weak_c2:
                extra["weak_c2"] = sub_record(run_randomized("c2", "weak", 3, 1, rank, world, local_rank, dev, rdev, shared, args.spmm_variant))
        except Exception as e:   # never lose the line to an extra (on every rank the same way: the workload is collective)
            extra["secondary_error"] = repr(e)

    traffic, traffic_source = None, None
    if rank == 0 and world == 1 and not args.no_extras and not args.no_traffic:
        traffic, traffic_source = measure_traffic(args.workload, 2 * q + 3)      # 2q + 2 sweeps of the fit + the projection's
        if traffic is None:
            traffic_source = "not measured in this run (" + traffic_source + "); "
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if traffic is None and os.path.exists(tpath):
        try:
            tj = json.load(open(tpath)).get(args.workload, {})
            traffic = tj.get("hbm_bytes_per_launch")
            if traffic is not None:
                # not measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same command (tools/pmc_traffic.sh)
                traffic_source = (traffic_source or "") + "profiles/traffic.json (committed rocprofv3 --pmc passes of this command; not measured in this run)"
        except Exception:
            traffic = None
    if rank == 0:
        kernel_names = {0: "spmm_rowgather_kernel", 1: "spmm_quad_kernel (entries staged in LDS)", 2: "spmm_dq_kernel (DPP-fed quad sweep)"}
        slots = r["slots"]
        lds_bytes = slots * 4 * 64                                      # one 256-byte panel row gathered from LDS per entry slot
        avg_sweep_ms, achieved = r["avg_sweep_ms"], r["achieved"]
        line = {
            "metric": "sparse_pca_fit_transform_algorithmic_throughput", "value": r["value"],
            "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"],
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {r['m_total']} x {n} CSR f32, density {density}, gapped generator seed {r['seed']}, "
                                   f"SparsePCA fit_transform, SVDMethod::Random k={k} p={p} q={q} QR, rows range-partitioned "
                                   f"over {world} GPU(s) ({args.scaling} scaling), inputs resident in HBM, collectives: {r['transport']}",
                       "nnz": int(r["nnz_total"]), "rows_per_gpu": r["m"], "sweeps_per_fit": 2 * q + 2, "stage_ms": r["stage"],
                       "collectives": r["transport"], "comm_ms": r["stage"].get("comm_ms", 0.0), "sweep_ms_per_rank": r["per_rank_sweep"],
                       "comm_per_rank": r["per_rank_comm"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": f"{kernel_names.get(r['sweep_kernel'], '?')}: spmm sweep (A*X and A^T*Y launches, HIP events on the library stream)",
                         "algorithmic_bytes_per_launch": r["sweep_bytes"], "avg_launch_ms": avg_sweep_ms,
                         "launches_timed": len(r["sweep_ms"])},
        }
        if slots > 0:
            # what binds the sweep in practice: every entry slot gathers a 256-byte panel row from LDS and costs four
            # wave-level VALU instructions per four slots; the guide's aggregate ds_read_b128 rate is ~150 TB/s
            line["roofline"]["secondary"] = {"bound": "lds", "achieved": lds_bytes / (avg_sweep_ms * 1e-3) / 1e9, "peak": 150000.0,
                                             "unit": "GB/s", "frac": lds_bytes / (avg_sweep_ms * 1e-3) / 1e9 / 150000.0,
                                             "lds_gather_bytes_per_launch": lds_bytes, "entry_slots_per_launch": slots,
                                             "stored_entries_per_launch": int(r["nnz"]),
                                             # the same on STORED entries only (padding slots are not work): the honest figure
                                             "frac_stored": int(r["nnz"]) * 256.0 / (avg_sweep_ms * 1e-3) / 1e9 / 150000.0}
        line.update(extra)
        if not args.no_extras:
            peak_meas = measured_copy_gbs(local_rank)
            line["roofline"]["peak_measured_by"] = "sapca_measure_copy_gbs: 1 GiB, 16 B per lane, read + write"
            line["roofline"]["peak_measured"] = peak_meas
            line["roofline"]["frac_of_measured"] = achieved / peak_meas
        if isinstance(e2e, tuple):
            line["e2e_host_ms"], line["e2e_host_parts"] = e2e
        elif e2e is not None:
            line["e2e_host_ms"] = None
            line["e2e_host_error"] = e2e
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(args.workload, m_cfg, n, density, k, p, q, r["seed"], dev, args.cpu_sample_rows)
            line["cpu_baseline_reference"] = ref_probe
            if not args.no_extras:
                line["cpu_baseline_c1"] = c1_comparison(dev, local_rank)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
