#!/usr/bin/env python3
"""A slots / barrier model of the DPP-fed quad sweep (csrc/spmm_dq.hip), run on the CPU from the bench matrix itself.

What the sweep executes is fixed by its operator format: per (row block, interleaved column tile) a wave walks its 16 quads
(4 consecutive rows in lockstep, one per 16-lane group), a quad's segment padded to its longest row and to an even step
count; the 16 waves of a workgroup meet at a barrier per tile (the double-buffered panel tile is handed on).  Its time is
    T = c_slot * max over workgroups of  sum over tiles of  (max over the block's waves of that wave's steps in the tile) * 4
with c_slot the cycles a CU spends per executed entry slot.  The model rebuilds the step tables of a SAMPLE of row blocks
of a workload (rows are statistically identical: the generator permutes them) and reports, for the current format and for
the alternatives VERDICT r3 asked about:
  slots / entry    executed entry slots per stored entry (lockstep + even-step padding)
  barrier          sum_t max_w steps(w, t) / max_w sum_t steps(w, t): what the per-tile barrier costs over the block's slowest wave
  handoff          the same with a flag-based tile hand-off: a wave may run ONE tile ahead of the slowest wave (the tile it
                   enters must have been loaded into the buffer the slowest wave has left: two buffers)
  T / T_now        predicted sweep time relative to the current format, with c_slot scaled by the VALU + LDS issue cost of
                   the layout's step (tools/ubench: DPP move 4.6, v_pk_fma_f32 5.2 cycles per SIMD; LDS 256 B per clock)

    python tools/sweep_model.py [c2|c4|c5] [--blocks 24] [--transposed]

The calibration point: C2's A sweep executes 1.62e8 slots for 1.2e8 entries (rocprofv3, profiles/r03_c2_dq_pmc.txt) and takes
0.555 ms = 1.81 cycles per slot per CU at 2.07 GHz; the model reproduces the slot count from the matrix alone (printed).
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "single-algebra_amd", "python"))

WORKLOADS = {   # m, n, density, k, panel columns l
    "c2": (200_000, 20_000, 0.03, 50, 60),
    "c4": (1_000_000, 30_000, 0.03, 50, 60),
    "c5": (2_000_000, 50_000, 0.01, 100, 110),
}
CUS = 256
TILE_BYTES = 80 * 1024


def block_rows_of(op_rows):
    """natural_partition() of spmm_tiled.hip: 1024-row blocks, the block count rounded up to a multiple of the CUs"""
    nrb = -(-op_rows // 1024)
    if nrb >= 192:
        nrb = -(-nrb // CUS) * CUS
    return -(-op_rows // nrb), nrb


def counts_per_row_tile(ptr, idx, rows, nct):
    """cnt[r, t] = entries of row r in interleaved tile t (columns c with c % nct == t)"""
    cnt = np.zeros((rows, nct), dtype=np.int32)
    r = np.repeat(np.arange(rows), np.diff(ptr))
    np.add.at(cnt, (r, idx % nct), 1)
    return cnt


def steps_of(cnt, lock, even=True, sort_rows=False):
    """steps[q, t] of lockstep groups of `lock` consecutive rows: the longest row of the group, rounded to even"""
    rows, nct = cnt.shape
    if sort_rows:
        cnt = cnt[np.argsort(-cnt.sum(1), kind="stable")]
    pad = (-rows) % lock
    if pad:
        cnt = np.vstack([cnt, np.zeros((pad, nct), cnt.dtype)])
    st = cnt.reshape(-1, lock, nct).max(1)
    if even:
        st = st + (st & 1)
    return st


def analyse(cnt, lock, quads_per_wave, *, even=True, sort_rows=False, deal=False):
    st = steps_of(cnt, lock, even, sort_rows)                # [quads, tiles]
    nq, nct = st.shape
    # the kernel's split of a block's quads over its 16 waves: wave w walks quads [w nq / 16, (w + 1) nq / 16) (q_first() in
    # spmm_tiled.hip) -- every wave the same count to within one, whatever the block's row count; `deal`: wave w walks quads
    # w, w + 16, ... instead (same counts, every wave a sample of the whole block)
    nw = 16 * 16 // quads_per_wave if quads_per_wave != 16 else 16
    nw = 16
    wave = np.zeros((nw, nct), dtype=np.int64)
    for w in range(nw):
        sel = np.arange(w, nq, nw) if deal else np.arange(w * nq // nw, (w + 1) * nq // nw)
        wave[w] = st[sel].sum(0)
    slots = int(st.sum()) * lock
    barrier = wave.max(0).sum()                              # a barrier per tile: the slowest wave sets the pace
    mean = wave.sum(1).mean()
    # flag hand-off, one tile ahead: tile t can be entered once every wave has LEFT tile t - 2 (its buffer is refilled
    # behind them; the refill itself is hidden, as it is today)
    nw = wave.shape[0]
    end = np.zeros((nw, nct + 2))
    for t in range(nct):
        ready = end[:, t].max() if t >= 1 else 0.0           # end[:, t] holds the ends of tile t - 2 (offset by two)
        start = np.maximum(end[:, t + 1], ready)
        end[:, t + 2] = start + wave[:, t]
    handoff = end[:, nct + 1].max()
    return slots, float(barrier), float(mean), float(handoff), float(wave.sum(1).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload", nargs="?", default="c2", choices=sorted(WORKLOADS))
    ap.add_argument("--blocks", type=int, default=24, help="row blocks of the operator to sample")
    ap.add_argument("--transposed", action="store_true", help="model A^T's operator (rows = columns of A) instead of A's")
    args = ap.parse_args()
    import torch
    from sapca import synth
    m, n, dens, k, l = WORKLOADS[args.workload]
    ld = 64   # panel columns per pass (l > 64: two column passes over the same format)
    tc = TILE_BYTES // (ld * 4)
    if not args.transposed:
        brows, nrb = block_rows_of(m)
        rows = min(m, brows * args.blocks)
        ptr, idx, _ = synth.gapped_csr(rows, n, dens, k, seed=42, m_global=m, dtype=torch.float32)
        ptr, idx = ptr.numpy(), idx.numpy().astype(np.int64)
        nct = -(-n // tc)
        cnt_all = counts_per_row_tile(ptr, idx, rows, nct)
        op_rows, what = m, "A"
    else:
        # A^T: its rows are A's columns; a sample of A's rows gives every column's count per tile of A's rows only in
        # expectation, so generate whole columns from a row sample scaled up: use the first `rows` rows of A as the tile
        # population (tiles of A^T interleave A's rows: r % nct) -- the per-(column, tile) counts are Binomial(rows_in_tile, p_col)
        brows, nrb = block_rows_of(n)
        nct = -(-m // tc)
        sample_tiles = min(nct, 96)
        rows_a = sample_tiles * tc                       # A rows r with r % nct < sample_tiles would be scattered: take contiguous rows instead
        ptr, idx, _ = synth.gapped_csr(rows_a, n, dens, k, seed=42, m_global=m, dtype=torch.float32)
        ptr, idx = ptr.numpy(), idx.numpy().astype(np.int64)
        r = np.repeat(np.arange(rows_a), np.diff(ptr))
        ncols = min(n, brows * args.blocks)
        keep = idx < ncols
        cnt_all = np.zeros((ncols, sample_tiles), dtype=np.int32)
        np.add.at(cnt_all, (idx[keep], r[keep] % sample_tiles), 1)
        rows, op_rows, what = ncols, n, "A^T (%d of %d tiles)" % (sample_tiles, nct)
        nct = sample_tiles
    nnz = int(cnt_all.sum())
    print("%s of %s: %d x %d, %d stored entries in the sample (%d row blocks of %d rows, %d tiles of %d panel rows), %.2f entries per (row, tile)"
          % (what, args.workload, rows, n if not args.transposed else m, nnz, -(-rows // brows), brows, nct, tc, nnz / (rows * nct)))

    layouts = [
        # name, lockstep rows, quads per wave, VALU+LDS cycles per step per SIMD (issue floor), kwargs
        ("current: 4-row lockstep, even steps", 4, 16, 2 * 4.6 + 2 * 5.2, {}),
        ("4-row lockstep, odd steps allowed", 4, 16, 2 * 4.6 + 2 * 5.2, {"even": False}),
        ("odd steps + quads dealt to the waves round-robin", 4, 16, 2 * 4.6 + 2 * 5.2, {"even": False, "deal": True}),
        ("4-row lockstep, rows sorted by length in the block", 4, 16, 2 * 4.6 + 2 * 5.2, {"sort_rows": True}),
        ("2-row lockstep (32 lanes x 2 columns)", 2, 32, 2 * 4.6 + 1 * 5.2, {}),
        ("1 row per wave-step (64 lanes x 1 column, scalar-fed)", 1, 64, 4.0 + 4.0, {"even": False}),
    ]
    base = None
    print("%-58s %8s %8s %8s %9s %9s" % ("layout", "slots/e", "barrier", "handoff", "T/T_now", "T_ho/T_now"))
    for name, lock, qpw, cyc_step, kw in layouts:
        tot_slots = 0
        t_bar = t_ho = t_mean = 0.0
        nb = -(-rows // brows)
        for b in range(nb):
            cnt = cnt_all[b * brows:(b + 1) * brows]
            slots, barrier, mean, handoff, slowest = analyse(cnt, lock, qpw, **kw)
            tot_slots += slots
            t_bar += barrier
            t_ho += handoff
            t_mean += slowest   # (what the block would take without any per-tile synchronisation: its slowest wave's own steps)
        # time per block ~ steps on the critical path x cycles per step (a step = `lock` slots); blocks are dealt one per CU
        t_now = t_bar * cyc_step
        t_hand = t_ho * cyc_step
        if base is None:
            base = t_now
        print("%-58s %8.3f %8.3f %8.3f %9.3f %9.3f" % (name, tot_slots / nnz, t_bar / t_mean, t_ho / t_mean, t_now / base, t_hand / base))
    # a tile twice as long (160 KiB single-buffered): half the tiles, twice the entries per (row, tile); the refill is exposed
    cnt2 = cnt_all[:, : (cnt_all.shape[1] // 2) * 2].reshape(rows, -1, 2).sum(2)
    tot_slots = 0
    t_bar = t_mean = 0.0
    for b in range(-(-rows // brows)):
        slots, barrier, mean, _, slowest = analyse(cnt2[b * brows:(b + 1) * brows], 4, 16)
        tot_slots += slots
        t_bar += barrier
        t_mean += slowest
    refill = 160 * 1024 / 64.0 / 4.0   # SIMD-cycles a 160 KiB refill at 64 B per clock per CU keeps the CU waiting
    t = t_bar * (2 * 4.6 + 2 * 5.2) + cnt2.shape[1] * (-(-rows // brows)) * refill
    print("%-58s %8.3f %8.3f %8s %9.3f %9s" % ("640-row tile, single buffer (refill exposed)", tot_slots / int(cnt2.sum()), t_bar / t_mean, "-", t / base, "-"))
    print("\n(slots/e: executed entry slots per stored entry; barrier / handoff: critical-path steps over the slowest wave's own steps;\n"
          " T/T_now: predicted sweep time against today's format at the issue floor of each layout's step)")


if __name__ == "__main__":
    main()
