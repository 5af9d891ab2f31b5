"""Where the waves of the DPP-fed sweep wait: reads the s_memtime stamps of ONE sweep launch of a stamp build of the library
(tools/dq2_variant.sh stamps DQ2_STAMPS=1; the records are described in tools/gen_spmm_dq2.py) and prints the share of wave
time spent at the tile barrier / waiting for the wave's own LDS-DMA pieces / for entry loads / in the chunks, overall and by the
rank of a wave's stream length among the four waves of its SIMD, plus the sampled LDS round trip.

  SAPCA_LIB_PATH=single-algebra_amd/lib/exp/libsapca_stamps.so python3 tools/dq_stamps.py c2|c4|c5 [launch index: 0 = first A sweep, 1 = first A^T sweep]
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))
import numpy as np
import torch
import sapca
from sapca import synth
from sapca import _lib as L

SHAPES = {"c2": (200_000, 20_000, 0.03, 50), "c4": (1_000_000, 30_000, 0.03, 50), "c5": (2_000_000, 50_000, 0.01, 100),
          "small": (20_000, 4_000, 0.03, 20)}
wl = sys.argv[1] if len(sys.argv) > 1 else "c2"
which = int(sys.argv[2]) if len(sys.argv) > 2 else 0
m, n, density, k = SHAPES[wl]
dev = torch.device("cuda", 0)
ptr, idx, val = synth.gapped_csr(m, n, density, k, seed=42, dtype=torch.float32, device=dev)
x = sapca.DeviceCsr(ptr, idx, val, (m, n))
pca = (sapca.SparsePCABuilder.new().n_components(k).random_seed(42).device(0).collect_timings(True)
       .svd_method(sapca.SVDMethod.Random(10, 1, sapca.PowerIterationNormalizer.NONE)).build())
lib = L.load()
if not hasattr(lib, "sapca_debug_dq_stamps"):
    raise SystemExit("this library carries no stamps: build one with tools/dq2_variant.sh stamps DQ2_STAMPS=1 and point SAPCA_LIB_PATH at it")
lib.sapca_debug_dq_stamps.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.sapca_debug_dq_stamps.restype = None
pca.fit(x)   # warm-up: formats, buffers
torch.cuda.synchronize()
WGS = 4096
buf = torch.zeros((WGS, 16, 12, 64), dtype=torch.int32, device=dev)
lib.sapca_debug_dq_stamps(C.c_void_p(buf.data_ptr()), WGS, which)
pca.fit(x)
torch.cuda.synchronize()
lib.sapca_debug_dq_stamps(None, 0, -1)
t = pca.timings()
sweeps = list(t.spmm_sweep_ms[: t.n_spmm]) + list(t.spmmt_sweep_ms[: t.n_spmmt])
r = buf.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
used = np.where(r[:, :, 0, :].any(axis=(1, 2)))[0]
if used.size == 0:
    raise SystemExit("no stamps recorded (did launch %d run the DPP-fed sweep?)" % which)
r = r[: used.max() + 1]
R = [r[:, :, j, :] for j in range(12)]
T0, T1, T2, T3, T4, T5, VM, LRT, _, NCH, OWN, P4 = R
valid = T5 != 0                      # (wg, wave, tile slot)
d = lambda a, b: (a - b) & 0xFFFFFFFF
setup1, land, barrier, setup2, stream = d(T1, T0), d(T2, T1), d(T3, T2), d(T4, T3), d(T5, T4)
# the tail of a tile (pieces the chunks did not issue, bookkeeping) = next tile's top - this tile's end, inside one wave
tail = np.zeros_like(T0)
tail[:, :, :-1] = d(T0[:, :, 1:], T5[:, :, :-1])
tail_valid = valid.copy(); tail_valid[:, :, -1] = False; tail_valid[:, :, :-1] &= valid[:, :, 1:] & (T0[:, :, 1:] > 0)
tail = np.where(tail_valid & (tail < 1 << 24), tail, 0)
tot = lambda a: float(a[valid].sum())
chunks = tot(NCH)
total = sum(tot(a) for a in (setup1, land, barrier, setup2, stream)) + float(tail.sum())
print(f"workload {wl}  launch {which}  workgroups {r.shape[0]}  (wave, tile) records {int(valid.sum())}  chunks {int(chunks)}  "
      f"sweep times of the stamped fit (ms): {' '.join('%.3f' % v for v in sweeps)}")
smem = np.percentile((VM[valid & (NCH > 0)] / np.maximum(NCH[valid & (NCH > 0)], 1)), 5)
print(f"stamp floor (5th percentile of a chunk's entry wait = one SMEM round trip): {smem:.0f} cycles")
print("share of wave time                      cycles/(wave,tile)   share")
rows = [("tile top -> pieces wait (scalar set-up, issue)", setup1), ("waiting for the wave's own LDS-DMA pieces", land),
        ("at the tile barrier", barrier), ("barrier -> first chunk (step counts)", setup2),
        ("chunks: entry-load waits (vmcnt)", VM), ("chunks: everything else (issue + LDS waits)", stream - VM),
        ("tile end -> next tile top (late pieces, bookkeeping)", tail)]
for name, a in rows:
    v = tot(a) if a is not tail else float(tail.sum())
    print(f"  {name:52s} {v / valid.sum():10.0f}   {100 * v / total:5.1f} %")
nz = valid & (NCH > 0)
per_chunk = tot(np.where(nz, stream, 0)) / chunks
print(f"inside the chunks: {per_chunk:.0f} cycles per chunk of 16 steps = {per_chunk / 16:.1f} per step per wave "
      f"(x 1/16 waves x 1/4 slots = {per_chunk / 16 / 16 / 4:.2f} cycles per slot per CU while streaming)")
print(f"  entry-load wait per chunk             {tot(np.where(nz, VM, 0)) / chunks:7.0f}  (floor {smem:.0f})")
print(f"  LDS round trip, position 0 (C -> D)   {tot(np.where(nz, LRT, 0)) / chunks:7.0f}  of which the wave's own way to the wait {tot(np.where(nz, OWN, 0)) / chunks:.0f}"
      f"  -> waited {tot(np.where(nz, LRT - OWN, 0)) / chunks:.0f}")
print(f"  LDS wait at position 4 (steady state) {tot(np.where(nz, P4, 0)) / chunks:7.0f}  (floor {smem:.0f}: the stamp in front of the wait counts in lgkmcnt)")
# by rank of the wave's chunk count among the four waves of its SIMD (waves w, w+4, w+8, w+12)
nch4 = NCH.reshape(NCH.shape[0], 4, 4, 64)   # [wg][i][simd][tile], wave = 4 i + simd
order = np.argsort(np.argsort(nch4, axis=1, kind="stable"), axis=1).reshape(NCH.shape)
print("by rank of the wave's stream length within its SIMD (0 = shortest):   barrier   pieces   chunks   entry waits   (cycles per (wave, tile))")
for rk in range(4):
    sel = valid & (order == rk)
    f = lambda a: float(a[sel].sum()) / max(sel.sum(), 1)
    print(f"   rank {rk}: chunks/tile {f(NCH):5.2f}   {f(barrier):8.0f} {f(land):8.0f} {f(stream):8.0f} {f(VM):8.0f}")
# by wave index (waves w, w + 4, w + 8, w + 12 share a SIMD; a lower index is an older wave)
print("by wave index:  chunks/tile   stream   per chunk   barrier   start after barrier leave (first chunk entered - earliest leave of the workgroup)")
first_leave = np.where(valid, T3, 1 << 62).min(axis=1, keepdims=True)
for w in range(16):
    sel = valid[:, w, :]
    f = lambda a: float(a[:, w, :][sel].sum()) / max(sel.sum(), 1)
    lag = ((T4 - first_leave) & 0xFFFFFFFF)[:, w, :][sel]
    print(f"   wave {w:2d} (SIMD {w % 4}): {f(NCH):5.2f} {f(stream):9.0f} {f(stream) / max(f(NCH), 1e-9):9.0f} {f(barrier):9.0f} {lag.mean():9.0f}")
# spread of the waves' stream times inside one (workgroup, tile): how much of it follows the step counts
st = np.where(valid, stream, 0).astype(np.float64); nc = np.where(valid, NCH, 0).astype(np.float64)
okt = valid.all(axis=1)
if okt.any():
    a = st.transpose(0, 2, 1)[okt]; b = nc.transpose(0, 2, 1)[okt]      # [(wg, tile)][wave]
    ca = a - a.mean(axis=1, keepdims=True); cb = b - b.mean(axis=1, keepdims=True)
    slope = (ca * cb).sum() / max((cb * cb).sum(), 1e-9)
    resid = ca - slope * cb
    print(f"within a (workgroup, tile): stream time std {ca.std():.0f} cycles; {slope:.0f} cycles per extra chunk explain {100 * (1 - resid.var() / max(ca.var(), 1e-9)):.0f} % of the variance; "
          f"residual std {resid.std():.0f}")
    # systematic part per wave index of the residual
    print("   mean residual by wave index:", " ".join("%+.0f" % v for v in resid.mean(axis=0)))
    # how persistent is a wave's lag from tile to tile (the hand-off only helps what is not persistent)
    dev = (a - a.mean(axis=1, keepdims=True))
    wg_id = np.repeat(np.arange(st.shape[0])[:, None], st.shape[2], axis=1)[okt]
    same = wg_id[1:] == wg_id[:-1]
    if same.any():
        c = (dev[1:][same] * dev[:-1][same]).mean() / max(dev.var(), 1e-9)
        print(f"   correlation of a wave's deviation between consecutive tiles: {c:.2f}")
    tile_max = a.max(axis=1); tile_mean = a.mean(axis=1)
    # what a one-tile run-ahead could recover: sum over tiles of max_w  vs  max_w of the sum over pairs of tiles
    print(f"   sum over tiles of the slowest wave's stream: {tile_max.sum():.3e}; of the mean wave's: {tile_mean.sum():.3e}  (ratio {tile_max.sum() / tile_mean.sum():.3f})")
    for wgi in np.unique(wg_id)[:0]:
        pass
    tot_w = np.zeros((st.shape[0], 16)); ntile = okt.sum(axis=1)
    for g in range(st.shape[0]):
        tot_w[g] = st[g][:, okt[g]].sum(axis=1)
    usable = ntile > 0
    print(f"   with NO barrier at all (every wave only bound by its own total): slowest wave's total / mean wave's total = {(tot_w[usable].max(axis=1).sum() / tot_w[usable].mean(axis=1).sum()):.3f}")
if os.environ.get("STAMPS_DUMP"):
    np.savez_compressed(os.environ["STAMPS_DUMP"], r=r.astype(np.uint32))
# the slowest wave of a (workgroup, tile) decides: its stream against the mean
smax = np.where(valid, stream, 0).max(axis=1)
smean = np.where(valid, stream, 0).sum(axis=1) / np.maximum(valid.sum(axis=1), 1)
ok = valid.any(axis=1)
print(f"slowest wave's stream / mean stream per (workgroup, tile): {smax[ok].sum() / smean[ok].sum():.3f}")
wg_span = np.where(valid, T5, 0).max(axis=(1, 2)) - np.where(valid, T0, 1 << 62).min(axis=(1, 2))
print(f"workgroup life (first tile top -> last tile done), cycles: median {np.median(wg_span):.0f}  max {wg_span.max():.0f}")
