#!/usr/bin/env python3
"""Emits single-algebra_amd/csrc/spmm_dq_gen.h: the inline-asm main loop of the DPP-fed quad sweep (spmm_dq.hip).

The kernel around it (C++ prologue and epilogue) is in spmm_dq.hip; this file only writes the part that cannot be
expressed in HIP C++: accumulators in fixed VGPRs addressed through the VGPR index mode (M0[7:0] = 4 * row slot,
SRC2 + DST relative), entries broadcast to their 16-lane row group with DPP row_newbcast, the panel tile brought into
LDS by LDS-DMA (global_load_lds_dwordx4) into two 80 KiB buffers, counted vmcnt / lgkmcnt waits.

Per workgroup (16 waves, one (row block, tile range)), for every tile t of the range:
  wait for this wave's LDS-DMA pieces of tile t; s_barrier;
  chunks of 16 steps: a chunk is one 512-byte coalesced load (lane (g, i) = step i of lane group g) into one of three
  rotating register pairs, then eight groups of two steps each {2 v_add_u32_dpp (address), 2 ds_read_b128,
  2 v_mov_b32_dpp (value), 4 v_pk_fma_f32 under the index}.  Every chunk also issues, in this order, one LDS-DMA piece
  of tile t+1 (chunks 0..4; the other LDS buffer), the descriptor load of the next tile (once) and the entry load of
  the chunk three ahead -- of this tile, or of the first three chunks of the next tile when both tiles have at least
  three chunks ("linked": the next tile then starts with its entries and descriptors already in registers).

Vector-memory operations return in order, so a wait for a chunk's entries is written as the number of operations
issued after them: the entries of chunk g were the last operation of chunk slot g-3, hence the count of what slots
g-2 and g-1 issued (S_N2 + S_N1, kept at run time, 0..3 each); the barrier wait likewise counts what was issued after
the last LDS-DMA piece (S_ND).  A tile that was not linked loads its first three chunks and descriptors itself and
waits for everything before its first chunk.

Run:  python3 tools/gen_spmm_dq.py   (writes the header next to the kernel; the header is committed)
"""
import os
import re
import sys

ACC = 56          # first accumulator register; 4 per row slot, + 4 for the trash slot (steps past the end of a stream)
RG = 8            # row slots per lane group (rows per wave = 4 * RG): set per variant in main()
TILE_B = 81920    # bytes of one LDS tile buffer (320 panel rows of 256 bytes)
EB = [(10, 11), (12, 13), (14, 15)]
VDESC, VLB, VT, VDESC2 = 16, 17, 18, 19
VA64 = 30         # v[30:31]: 64-bit source address of an LDS-DMA piece
DEPTH = 2         # two-step groups in flight per wave (LDS reads issued ahead of their FMAs); 3 needs 12 more VGPRs
A = [[20, 21], [22, 23]]
B = [24, 26]
VINFO = (28, 29)
W = [[32, 36], [40, 44]]


def set_depth(d, rg):
    """register map for `d` groups in flight (d = 3 only fits the 8-slot variant: v10..v55 + two registers behind the accumulators)"""
    global DEPTH, A, B, VINFO, W
    DEPTH = d
    if d == 2:
        A, B, VINFO, W = [[20, 21], [22, 23]], [24, 26], (28, 29), [[32, 36], [40, 44]]
    else:
        top = ACC + 4 * rg + 4
        A, B, VINFO, W = [[20, 21], [22, 23], [24, 25]], [26, 28, 30], (top, top + 1), [[32, 36], [40, 44], [48, 52]]
# scalars
S_D0, S_D1, S_REM, S_DL = 36, 37, 38, 39
S_PTR = 40        # s[40:41] entry stream pointer of the current chunk
S_DP = 42         # s[42:43] scratch pointer
S_T, S_NT, S_CW, S_TABS, S_BUF, S_NCH, S_OFF8 = 44, 45, 46, 47, 48, 49, 50
S_A, S_B2, S_C, S_D, S_E = 51, 52, 53, 54, 55
S_INFO = 56       # s[56:57] info pointer of the current 64-tile window
S_4NCT, S_TLAST, S_DL1 = 58, 59, 60
S_CI, S_DROW, S_DLDS = 61, 62, 63
S_N1, S_N2, S_ND, S_NS = 64, 65, 66, 67      # operations issued in the previous / second previous chunk slot, since the last piece, in this slot
S_LINK, S_PRE, S_BASE = 68, 69, 70           # this tile preloads the next one; this tile was preloaded; entry buffer of its chunk 0
S_LAST = 71
S_LASTG, S_EARLY = 76, 77                    # valid two-step groups of the tile's last chunk; the chunk was left early
S_DADDR = 78      # s[78:79] scalar source address of the next LDS-DMA piece
S_PSTEP, S_LIM, S_ROWBW = 80, 81, 82         # bytes between pieces; last S_DROW whose four rows are all inside the panel; 20 * wave * nct
S_PTRN = 72       # s[72:73] entry stream pointer of the next tile
S_DPN = 74        # s[74:75] descriptor pointer of the next tile

_uid = [0]


def uid(prefix):
    _uid[0] += 1
    return f"{prefix}{_uid[0]}"


B64 = os.environ.get("DQ_B64", "0") != "0"   # experiment (1.5 % slower at C2): one 64-bit row_newbcast move per step + a plain add
P64 = [[20, 22], [24, 26]]                      # .. into these even-aligned pairs: lo becomes the LDS address, hi is the value


def grp_a(k, e, L):
    p = k % DEPTH
    ex, ey = e
    if B64 and DEPTH == 2:
        for t in range(2):
            L.append(f"v_mov_b64_dpp v[{P64[p][t]}:{P64[p][t] + 1}], v[{ex}:{ey}] row_newbcast:{2 * k + t} row_mask:0xf bank_mask:0xf")
        for t in range(2):
            L.append(f"v_add_u32 v{P64[p][t]}, v{P64[p][t]}, v{VLB}")
        for t in range(2):
            L.append(f"ds_read_b128 v[{W[p][t]}:{W[p][t] + 3}], v{P64[p][t]}")
        return
    for t in range(2):
        L.append(f"v_add_u32_dpp v{A[p][t]}, v{ex}, v{VLB} row_newbcast:{2 * k + t} row_mask:0xf bank_mask:0xf")
    for t in range(2):
        L.append(f"ds_read_b128 v[{W[p][t]}:{W[p][t] + 3}], v{A[p][t]}")
    for t in range(2):
        L.append(f"v_mov_b32_dpp v{B[p] + t}, v{ey} row_newbcast:{2 * k + t} row_mask:0xf bank_mask:0xf")


def grp_b(k, L, wait):
    p = k % DEPTH
    sd = S_D0 if k < 4 else S_D1
    L.append(f"s_waitcnt lgkmcnt({wait})")
    L.append(f"s_set_gpr_idx_on s{sd}, gpr_idx(SRC2,DST)")
    for t in range(2):
        w = W[p][t]
        if B64 and DEPTH == 2:
            b, sel = P64[p][t], "op_sel:[0,1,0] op_sel_hi:[1,1,1]"    # the value is the pair's upper half
        else:
            b, sel = B[p], ("op_sel_hi:[1,0,1]" if t == 0 else "op_sel:[0,1,0] op_sel_hi:[1,1,1]")
        L.append(f"v_pk_fma_f32 v[{ACC}:{ACC + 1}], v[{w}:{w + 1}], v[{b}:{b + 1}], v[{ACC}:{ACC + 1}] {sel}")
        L.append(f"v_pk_fma_f32 v[{ACC + 2}:{ACC + 3}], v[{w + 2}:{w + 3}], v[{b}:{b + 1}], v[{ACC + 2}:{ACC + 3}] {sel}")
    L.append("s_set_gpr_idx_off")
    if k not in (3, 7):
        L.append(f"s_lshr_b32 s{sd}, s{sd}, 8")


def wait_vmcnt(L, sreg, maxn):
    """s_waitcnt vmcnt(min(s[sreg], maxn)): the count is a run-time value, the instruction takes an immediate"""
    end = uid("wv_end")
    labels = [uid("wv") for _ in range(maxn + 1)]
    for n in range(maxn, 0, -1):
        L += [f"s_cmp_ge_u32 s{sreg}, {n}", f"s_cbranch_scc1 {labels[n]}"]
    L += ["s_waitcnt vmcnt(0)", f"s_branch {end}"]
    for n in range(1, maxn + 1):
        L += [f"{labels[n]}:", f"s_waitcnt vmcnt({n})"]
        if n < maxn:
            L.append(f"s_branch {end}")
    L.append(f"{end}:")


def dma_tile_setup(L):
    """scalar source address of this wave's first piece of tile S_DROW: X + (S_DROW + 20 * wave * nct) * stride"""
    L += [f"s_add_u32 s{S_A}, s{S_DROW}, s{S_ROWBW}",
          f"s_mul_hi_u32 s{S_DADDR + 1}, s{S_A}, %[stride]", f"s_mul_i32 s{S_DADDR}, s{S_A}, %[stride]",
          f"s_add_u32 s{S_DADDR}, s{S_DADDR}, %[xlo]", f"s_addc_u32 s{S_DADDR + 1}, s{S_DADDR + 1}, %[xhi]"]


def dma_piece(L, O=None):
    """one 1 KiB LDS-DMA piece of the next tile: 4 panel rows x 256 bytes (lane group g: row S_DROW + (20 wave + g) nct,
    lane: 16 bytes of it).  The source address is scalar (s[S_DADDR] + the lane's constant offset) unless one of the four
    rows lies past the panel's end: those pieces (the last one or two of a tile) clamp the row per lane."""
    skip, slow, done = uid("dmaskip"), uid("dmaslow"), uid("dmadone")
    L += ["s_bitcmp1_b32 %[mode], 1", f"s_cbranch_scc1 {skip}"]   # diagnostics (SAPCA_DQ_MODE & 2): no refills, timing only
    L += [f"s_cmp_gt_i32 s{S_DROW}, s{S_LIM}", f"s_cbranch_scc1 {slow}"]
    L.append(f"s_mov_b32 m0, s{S_DLDS}")
    L.append("s_nop 0")
    L.append(f"global_load_lds_dwordx4 %[voff], s[{S_DADDR}:{S_DADDR + 1}]")
    S = [] if O is None else O
    if O is None:
        L.append(f"s_branch {done}")
    S.append(f"{slow}:")
    # 64-bit per-lane source address: panels above 4 GiB exist (10M rows x 128 columns)
    S.append(f"v_add_u32 v{VT}, s{S_DROW}, %[rowb0]")
    S.append(f"v_min_u32 v{VT}, %[prm1], v{VT}")
    S.append(f"v_mul_hi_u32 v{VA64 + 1}, v{VT}, %[stride]")
    S.append(f"v_mul_lo_u32 v{VA64}, v{VT}, %[stride]")
    S.append(f"v_add_co_u32 v{VA64}, vcc, %[col16], v{VA64}")
    S.append(f"v_addc_co_u32 v{VA64 + 1}, vcc, 0, v{VA64 + 1}, vcc")
    S.append(f"v_mov_b32 v{VT}, %[xhi]")                      # (an SGPR operand beside vcc would be a second constant-bus read)
    S.append(f"v_add_co_u32 v{VA64}, vcc, %[xlo], v{VA64}")
    S.append(f"v_addc_co_u32 v{VA64 + 1}, vcc, v{VT}, v{VA64 + 1}, vcc")
    S.append(f"s_mov_b32 m0, s{S_DLDS}")
    S.append("s_nop 0")
    S.append(f"global_load_lds_dwordx4 v[{VA64}:{VA64 + 1}], off")
    if O is None:
        L += S
    else:
        S.append(f"s_branch {done}")
    L.append(f"{done}:")
    L.append(f"{skip}:")
    L.append(f"s_add_u32 s{S_DROW}, s{S_DROW}, s{S_4NCT}")
    L.append(f"s_add_u32 s{S_DLDS}, s{S_DLDS}, 0x400")
    L += [f"s_add_u32 s{S_DADDR}, s{S_DADDR}, s{S_PSTEP}", f"s_addc_u32 s{S_DADDR + 1}, s{S_DADDR + 1}, 0"]


def ptr_from_off8(L, s_off8, dst, base):
    """s[dst:dst+1] = base (64-bit operand name) + 64 * s_off8   (s_off8: entry offset in units of 8 entries)"""
    L.append(f"s_lshl_b32 s{S_C}, s{s_off8}, 6")
    L.append(f"s_lshr_b32 s{S_D}, s{s_off8}, 26")
    L.append(f"s_mov_b64 s[{dst}:{dst + 1}], %[{base}]")
    L.append(f"s_add_u32 s{dst}, s{dst}, s{S_C}")
    L.append(f"s_addc_u32 s{dst + 1}, s{dst + 1}, s{S_D}")


def desc_ptr(L, s_off8, s_cw, dst):
    """s[dst:dst+1] = desc + 8 * (off8 / 8 + cw)"""
    L += [f"s_lshr_b32 s{S_E}, s{s_off8}, 3", f"s_add_u32 s{S_E}, s{S_E}, s{s_cw}", f"s_lshl_b32 s{S_C}, s{S_E}, 3", f"s_lshr_b32 s{S_D}, s{S_E}, 29",
          f"s_mov_b64 s[{dst}:{dst + 1}], %[desc]", f"s_add_u32 s{dst}, s{dst}, s{S_C}", f"s_addc_u32 s{dst + 1}, s{dst + 1}, s{S_D}"]


def chunk_body(buf, L, O):
    """chunk S_CI of the tile (16 steps) from entry buffer `buf`.  The steady state (chunk >= 5, more than three chunks
    left, two reloads behind this chunk's entries) falls through; everything else is out of line (O)."""
    e = EB[buf]
    depth = DEPTH
    lbl = f"c{buf}"
    # ---- wait for this chunk's entries: s_waitcnt vmcnt(S_N1 + S_N2); chunk 0 of a tile that was not preloaded: vmcnt(0)
    L += [f"s_cmp_eq_u32 s{S_CI}, 0", f"s_cbranch_scc1 {lbl}_first", f"s_add_u32 s{S_A}, s{S_N1}, s{S_N2}", f"s_cmp_lg_u32 s{S_A}, 2",
          f"s_cbranch_scc1 {lbl}_slow", "s_waitcnt vmcnt(2)", f"{lbl}_ready:"]
    O += [f"{lbl}_slow:"]
    wait_vmcnt(O, S_A, 6)
    O += [f"s_branch {lbl}_ready"]
    # chunk 0: a preloaded tile takes its descriptors from the second register (loaded before its first entries)
    O += [f"{lbl}_first:", f"s_cmp_lg_u32 s{S_PRE}, 0", f"s_cbranch_scc1 {lbl}_pre", "s_waitcnt vmcnt(0)", f"s_mov_b32 s{S_N1}, 0", f"s_mov_b32 s{S_N2}, 0",
          f"s_branch {lbl}_ready", f"{lbl}_pre:", f"s_add_u32 s{S_A}, s{S_N1}, s{S_N2}"]
    wait_vmcnt(O, S_A, 6)
    O += [f"v_mov_b32 v{VDESC}, v{VDESC2}", f"s_branch {lbl}_ready"]
    # a descriptor register covers 32 chunks: longer streams (dense tiles) reload it, everything waited for
    L += [f"s_cmp_ge_u32 s{S_DL}, 64", f"s_cbranch_scc1 {lbl}_dreload", f"{lbl}_dok:"]
    # pattern mode (mode bit 3; MaskedSparsePCA's projection, quirk Q3): every stored non-zero value counts as 1, zeros
    # (padding, and stored zeros, which the caller handles) as 0 -- rewritten once in the freshly loaded entry registers
    L += ["s_bitcmp1_b32 %[mode], 3", f"s_cbranch_scc1 {lbl}_pat", f"{lbl}_patdone:"]
    O += [f"{lbl}_pat:", f"v_cmp_neq_f32 vcc, 0, v{e[1]}", f"v_cndmask_b32_e64 v{e[1]}, 0, 1.0, vcc", f"s_branch {lbl}_patdone"]
    O += [f"{lbl}_dreload:", f"s_add_u32 s{S_DP}, s{S_DP}, 0x100", f"s_addc_u32 s{S_DP + 1}, s{S_DP + 1}, 0",
          f"global_load_dword v{VDESC}, %[l4], s[{S_DP}:{S_DP + 1}]", "s_waitcnt vmcnt(0)", f"s_mov_b32 s{S_N1}, 0", f"s_mov_b32 s{S_N2}, 0",
          f"s_mov_b32 s{S_DL}, 0", f"s_branch {lbl}_dok"]
    L.append(f"s_add_u32 s{S_DL1}, s{S_DL}, 1")
    L.append(f"v_readlane_b32 s{S_D0}, v{VDESC}, s{S_DL}")
    L.append(f"v_readlane_b32 s{S_D1}, v{VDESC}, s{S_DL1}")
    for k in range(depth):
        grp_a(k, e, L)
    for k in range(8):
        ahead = min(depth, 8 - k) - 1
        grp_b(k, L, 2 * ahead)
        if k + depth < 8:
            grp_a(k + depth, e, L)
            if k + depth == 7:      # the buffer's last DPP reads are issued: this slot's memory operations
                L += [f"s_mov_b32 s{S_NS}, 0", f"s_cmp_lt_u32 s{S_CI}, 5", f"s_cbranch_scc1 {lbl}_dma", f"{lbl}_nodma:"]
                O += [f"{lbl}_dma:"]
                dma_piece(O)
                O += [f"s_mov_b32 s{S_NS}, 1", f"s_mov_b32 s{S_ND}, 0", f"s_branch {lbl}_nodma"]
                # rem counts this chunk: the chunk three ahead is in this tile when rem > 3
                L += [f"s_cmp_le_u32 s{S_REM}, 3", f"s_cbranch_scc1 {lbl}_tail",
                      f"global_load_dwordx2 v[{e[0]}:{e[1]}], %[eoff], s[{S_PTR}:{S_PTR + 1}] offset:1536 nt",
                      f"s_add_u32 s{S_NS}, s{S_NS}, 1", f"s_add_u32 s{S_ND}, s{S_ND}, 1", f"{lbl}_issued:",
                      f"s_mov_b32 s{S_N2}, s{S_N1}", f"s_mov_b32 s{S_N1}, s{S_NS}"]
                # linked: chunk 3 - rem of the next tile; its descriptors go first (rem == 3)
                O += [f"{lbl}_tail:", f"s_cmp_eq_u32 s{S_LINK}, 0", f"s_cbranch_scc1 {lbl}_issued",
                      f"s_cmp_lg_u32 s{S_REM}, 3", f"s_cbranch_scc1 {lbl}_nodesc",
                      f"global_load_dword v{VDESC2}, %[l4], s[{S_DPN}:{S_DPN + 1}]",
                      f"s_add_u32 s{S_NS}, s{S_NS}, 1", f"s_add_u32 s{S_ND}, s{S_ND}, 1", f"{lbl}_nodesc:",
                      f"s_sub_u32 s{S_A}, 3, s{S_REM}", f"s_lshl_b32 s{S_A}, s{S_A}, 9", f"v_add_u32 v{VT}, s{S_A}, %[eoff]",
                      f"global_load_dwordx2 v[{e[0]}:{e[1]}], v{VT}, s[{S_PTRN}:{S_PTRN + 1}] nt",
                      f"s_add_u32 s{S_NS}, s{S_NS}, 1", f"s_add_u32 s{S_ND}, s{S_ND}, 1", f"s_branch {lbl}_issued"]
    L.append(f"s_add_u32 s{S_PTR}, s{S_PTR}, 0x200")
    L.append(f"s_addc_u32 s{S_PTR + 1}, s{S_PTR + 1}, 0")
    L.append(f"s_add_u32 s{S_DL}, s{S_DL}, 2")
    L.append(f"s_sub_u32 s{S_REM}, s{S_REM}, 1")
    L.append(f"s_add_u32 s{S_CI}, s{S_CI}, 1")


def body():
    L = []
    nreg = 4 * RG + 4
    # accumulators and the trash slot <- 0
    L += [f"s_mov_b32 s{S_A}, 0", f"s_set_gpr_idx_on s{S_A}, gpr_idx(DST)", "1:", f"v_mov_b32 v{ACC}, 0",
          f"s_add_u32 s{S_A}, s{S_A}, 1", f"s_set_gpr_idx_idx s{S_A}", f"s_cmp_lt_u32 s{S_A}, {nreg}", "s_cbranch_scc1 1b",
          "s_set_gpr_idx_off"]
    L += [f"s_mov_b32 s{S_T}, 0", f"s_mov_b32 s{S_NT}, %[ntiles]", f"s_mov_b32 s{S_CW}, %[cw]", f"s_mov_b32 s{S_TABS}, %[t0]",
          f"s_mov_b32 s{S_BUF}, 0", f"s_mov_b64 s[{S_INFO}:{S_INFO + 1}], %[info]", f"s_lshl_b32 s{S_4NCT}, %[nct], 2",
          f"s_add_u32 s{S_TLAST}, %[t0], %[ntiles]", f"s_sub_u32 s{S_TLAST}, s{S_TLAST}, 1",
          f"s_mov_b32 s{S_N1}, 0", f"s_mov_b32 s{S_N2}, 0", f"s_mov_b32 s{S_ND}, 0", f"s_mov_b32 s{S_PRE}, 0", f"s_mov_b32 s{S_BASE}, 0", f"s_mov_b32 s{S_EARLY}, 0"]
    L += [f"v_readfirstlane_b32 s{S_ROWBW}, %[rowb0]", f"s_mul_i32 s{S_PSTEP}, s{S_4NCT}, %[stride]",
          f"s_mul_i32 s{S_LIM}, %[nct], 3", f"s_add_u32 s{S_LIM}, s{S_LIM}, s{S_ROWBW}", f"s_sub_u32 s{S_LIM}, %[prm1], s{S_LIM}"]
    # info window: lane t' holds {entry offset / 8, chunk count} of tile t0 + 64 * window + t'
    L += [f"global_load_dwordx2 v[{VINFO[0]}:{VINFO[1]}], %[l8], s[{S_INFO}:{S_INFO + 1}]"]
    # first tile: this wave's five pieces into buffer 0, synchronously
    L += [f"s_mov_b32 s{S_DROW}, s{S_TABS}", f"s_mov_b32 s{S_DLDS}, %[wdma]", f"s_mov_b32 s{S_CI}, 0"]
    dma_tile_setup(L)
    L += ["13:"]
    dma_piece(L)
    L += [f"s_add_u32 s{S_CI}, s{S_CI}, 1", f"s_cmp_lt_u32 s{S_CI}, 5", "s_cbranch_scc1 13b", "s_waitcnt vmcnt(0)"]
    L += ["10:"]   # ---- tile loop
    # a new 64-tile window of the info table (tiles are never linked across windows)
    L += [f"s_and_b32 s{S_A}, s{S_T}, 63", f"s_cmp_lg_u32 s{S_A}, 0", "s_cbranch_scc1 11f", f"s_cmp_eq_u32 s{S_T}, 0",
          "s_cbranch_scc1 11f", f"s_add_u32 s{S_INFO}, s{S_INFO}, 0x200", f"s_addc_u32 s{S_INFO + 1}, s{S_INFO + 1}, 0",
          f"global_load_dwordx2 v[{VINFO[0]}:{VINFO[1]}], %[l8], s[{S_INFO}:{S_INFO + 1}]", "s_waitcnt vmcnt(0)", "11:"]
    L += [f"s_and_b32 s{S_A}, s{S_T}, 63", f"v_readlane_b32 s{S_OFF8}, v{VINFO[0]}, s{S_A}", f"v_readlane_b32 s{S_NCH}, v{VINFO[1]}, s{S_A}",
          f"s_lshr_b32 s{S_LASTG}, s{S_NCH}, 16", f"s_and_b32 s{S_NCH}, s{S_NCH}, 0xffff",   # {chunks, valid groups of the last one}
          "s_bitcmp1_b32 %[mode], 0", f"s_cselect_b32 s{S_NCH}, 0, s{S_NCH}"]   # diagnostics (SAPCA_DQ_MODE & 1): tiles without compute
    ptr_from_off8(L, S_OFF8, S_PTR, "ent")
    desc_ptr(L, S_OFF8, S_CW, S_DP)
    # link to the next tile?  (same info window, both with at least three chunks)
    L += [f"s_mov_b32 s{S_LINK}, 0", f"s_add_u32 s{S_B2}, s{S_T}, 1", f"s_cmp_ge_u32 s{S_B2}, s{S_NT}", "s_cbranch_scc1 14f",
          f"s_and_b32 s{S_B2}, s{S_B2}, 63", f"s_cmp_eq_u32 s{S_B2}, 0", "s_cbranch_scc1 14f", f"s_cmp_lt_u32 s{S_NCH}, 3", "s_cbranch_scc1 14f",
          f"v_readlane_b32 s{S_LAST}, v{VINFO[1]}, s{S_B2}", f"s_and_b32 s{S_LAST}, s{S_LAST}, 0xffff", f"s_cmp_lt_u32 s{S_LAST}, 3", "s_cbranch_scc1 14f",
          f"v_readlane_b32 s{S_LAST}, v{VINFO[0]}, s{S_B2}", f"s_mov_b32 s{S_LINK}, 1"]
    ptr_from_off8(L, S_LAST, S_PTRN, "ent")
    L += [f"s_add_u32 s{S_B2}, s{S_CW}, 16"]
    desc_ptr(L, S_LAST, S_B2, S_DPN)
    L += ["14:"]
    # a tile that was not preloaded: its first three chunks and its descriptors
    L += [f"s_cmp_lg_u32 s{S_PRE}, 0", "s_cbranch_scc1 15f"]
    for i, e in enumerate(EB):
        L.append(f"global_load_dwordx2 v[{e[0]}:{e[1]}], %[eoff], s[{S_PTR}:{S_PTR + 1}] offset:{512 * i} nt")
    L.append(f"global_load_dword v{VDESC}, %[l4], s[{S_DP}:{S_DP + 1}]")
    L += [f"s_add_u32 s{S_ND}, s{S_ND}, 4", f"s_mov_b32 s{S_BASE}, 0", "15:"]
    # this wave's pieces of the tile have landed: wait for all but what was issued after the last of them
    wait_vmcnt(L, S_ND, 6)
    L += ["s_bitcmp1_b32 %[mode], 2", "s_cbranch_scc1 nobar", "s_barrier", "nobar:"]   # (SAPCA_DQ_MODE & 4: timing only)
    # pieces of the next tile go to the other buffer (the last tile reloads itself: harmless)
    L += [f"s_add_u32 s{S_DROW}, s{S_TABS}, 1", f"s_min_u32 s{S_DROW}, s{S_DROW}, s{S_TLAST}", f"s_sub_u32 s{S_DLDS}, {TILE_B}, s{S_BUF}",
          f"s_add_u32 s{S_DLDS}, s{S_DLDS}, %[wdma]"]
    dma_tile_setup(L)
    L += [f"v_add_u32 v{VLB}, s{S_BUF}, %[lb]", f"s_mov_b32 s{S_REM}, s{S_NCH}", f"s_mov_b32 s{S_DL}, 0", f"s_mov_b32 s{S_CI}, 0"]
    # chunk 0 reads entry buffer S_BASE
    L += [f"s_cmp_eq_u32 s{S_BASE}, 1", "s_cbranch_scc1 enter1", f"s_cmp_eq_u32 s{S_BASE}, 2", "s_cbranch_scc1 enter2"]
    L += ["12:"]
    O = []
    for i in range(3):
        if i:
            L.append(f"enter{i}:")
        L += [f"s_cmp_eq_u32 s{S_REM}, 0", "s_cbranch_scc1 19f"]
        chunk_body(i, L, O)
    L += ["s_branch 12b"]
    L += O            # the rare paths of the three chunk bodies
    L += ["19:"]
    # pieces the chunks did not issue (fewer than five chunks): they count as part of the last slot
    L += [f"s_cmp_ge_u32 s{S_CI}, 5", "s_cbranch_scc1 18f", "17:"]
    dma_piece(L)
    L += [f"s_add_u32 s{S_N1}, s{S_N1}, 1", f"s_mov_b32 s{S_ND}, 0",
          f"s_add_u32 s{S_CI}, s{S_CI}, 1", f"s_cmp_lt_u32 s{S_CI}, 5", "s_cbranch_scc1 17b", "18:"]
    # the next tile starts in the buffer after this tile's last chunk when it was preloaded
    L += [f"s_add_u32 s{S_BASE}, s{S_BASE}, s{S_NCH}", "16:", f"s_cmp_lt_u32 s{S_BASE}, 3", "s_cbranch_scc1 20f", f"s_sub_u32 s{S_BASE}, s{S_BASE}, 3",
          "s_branch 16b", "20:", f"s_mov_b32 s{S_PRE}, s{S_LINK}"]
    L += [f"s_add_u32 s{S_T}, s{S_T}, 1", f"s_add_u32 s{S_TABS}, s{S_TABS}, 1", f"s_add_u32 s{S_CW}, s{S_CW}, 16",
          f"s_sub_u32 s{S_BUF}, {TILE_B}, s{S_BUF}", f"s_cmp_lt_u32 s{S_T}, s{S_NT}", "s_cbranch_scc1 10b"]
    L += ["s_waitcnt vmcnt(0)"]
    return L


def uniq_labels(L):
    """inline asm may be emitted more than once per module: named labels get the %= suffix"""
    names = set()
    for ln in L:
        m = re.match(r"^([a-z][a-z0-9_]*):$", ln)
        if m:
            names.add(m.group(1))
    out = []
    for ln in L:
        for n in sorted(names, key=len, reverse=True):
            ln = re.sub(rf"\b{n}\b", n + "_%=", ln)
        out.append(ln)
    return out


def clobbers():
    v = [f"v{i}" for i in range(10, 48 if DEPTH == 2 else 56)] + [f"v{i}" for i in range(ACC + 4 * RG, ACC + 4 * RG + (4 if DEPTH == 2 else 6))]
    s = [f"s{i}" for i in range(36, 83)]
    return v + s + ["memory", "scc", "m0", "vcc"]


def main():
    global RG
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "single-algebra_amd", "csrc", "spmm_dq_gen.h")
    with open(path, "w") as out:
        out.write("// generated by tools/gen_spmm_dq.py -- do not edit; the generator documents the structure\n")
        out.write(f"#define DQ_ACC_BASE {ACC}\n#define DQ_TILE_BYTES {TILE_B}\n")
        depth8 = int(os.environ.get("DQ_DEPTH8", "2"))
        for rg in (8, 16):   # 512-row and 1024-row blocks
            RG = rg
            set_depth(depth8 if rg == 8 else 2, rg)
            _uid[0] = 0
            L = uniq_labels(body())
            out.write(f"#define DQ_MAIN_ASM_{rg} \\\n")
            for ln in L:
                out.write(f'  "{ln}\\n" \\\n')
            out.write("\n")
            out.write(f"#define DQ_MAIN_CLOBBERS_{rg} " + ", ".join(f'"{c}"' for c in clobbers()) + "\n")
            print(f"wrote {path}: RG {rg}, {len(L)} lines")


if __name__ == "__main__":
    main()
