#!/usr/bin/env python3
"""Emits single-algebra_amd/csrc/spmm_dq2_gen.h: the inline-asm main loop of the DPP-fed quad sweep, second formulation
(spmm_dq.hip, spmm_dq2_kernel).

Same operator format, same entry feed (one coalesced 512-byte load per 16 steps of a wave's four rows, step s broadcast to
its 16-lane group by DPP row_newbcast:s), same LDS-DMA double-buffered panel tile as the first formulation
(tools/gen_spmm_dq.py).  What changed is how a step finds its accumulators.  There the row slot of every two-step group
came from a descriptor byte through the VGPR index mode: s_set_gpr_idx_on / off around the FMAs of every group (DPP
instructions are not exempt from the destination index), a shift per group, two v_readlane per chunk, a descriptor table
beside the format, and steps past the end of a wave's stream executed into a trash slot (7 % of all steps at C2).  Here the
row slot is a matter of control flow:

  * the main loop exists once per row slot (RG copies of the eight two-step groups of a chunk), with the slot's four
    accumulator registers written into the FMAs;
  * a scalar counter holds the two-step groups the current quad still has in this tile; every group ends with
    s_sub_u32 + s_cbranch_scc1, and the branch (taken once per quad and tile) leads through a short stub that fetches the
    next non-empty quad's count (v_readlane from a register holding the tile's sixteen counts: the format's own `steps`
    table, no descriptors) into the same position of the next slot's copy;
  * the stream of a (wave, tile) ends exactly where its last quad ends: nothing is executed past it;
  * per group: 8 VALU + 2 LDS + 3 scalar instructions instead of 8 + 2 + 4, per chunk about a dozen control
    instructions instead of thirty (no run-time wait counts: every chunk slot issues exactly one entry load, last, so
    the entries of chunk g are complete at vmcnt(2); one shared chunk routine per entry buffer, entered and left by
    s_setpc_b64).

Chunk control is shared by all slots: BODY[s][7] jumps to the routine of the current entry buffer (S_CTL), which copies
the buffer into the register pair the bodies read, re-issues the buffer's load three chunks ahead (this tile's, or the next
tile's first three chunks when the two tiles are linked), issues one LDS-DMA piece of the next tile while there are any,
starts the first two groups and returns to position 0 of the current slot's copy (S_RET).

Run:  python3 tools/gen_spmm_dq2.py   (writes the header next to the kernel; the header is committed)
"""
import os
import re
import sys

ACC = 56          # first accumulator register; 4 per row slot
RG = 16           # row slots per lane group (rows per wave = 4 * RG): set per variant in main()
TILE_B = 81920    # bytes of one LDS tile buffer (320 panel rows of 256 bytes)
EB = [(10, 11), (12, 13), (14, 15)]
ECUR = (16, 17)   # the chunk being executed: {LDS byte offset, value} of lane (g, i) = step i of lane group g
VT, VLB = 18, 19
A = [[20, 21], [22, 23]]
B = [24, 26]
VINFO = (28, 29)
VA64 = 30         # v[30:31]: 64-bit source address of an LDS-DMA piece
W = [[32, 36], [40, 44]]
VCNT, VCNT2 = 48, 49   # two-step groups - 1 of this wave's quads in the current tile (lane j = quad j; -1: none); raw step counts of the next
DEPTH = int(os.environ.get("DQ2_DEPTH", "2"))   # two-step groups in flight per wave (3: a third register set behind the accumulators)
ODD = os.environ.get("DQ2_ODD", "1") != "0"     # quads of any step count (the format does not round a quad's steps up to even: the default since round 4): the counter runs per step; DQ2_ODD=0 + -DSAPCA_EVEN_STEPS: rounds 2-3
PRIO = os.environ.get("DQ2_PRIO", "0") != "0"   # per-tile issue priority from the info table
# Issue priority per chunk (round 5; profiles/r05_dq_stamps_*.txt).  A SIMD issues oldest-wave-first among equal priorities: the
# four waves of a SIMD drift apart by +-20 % per tile and the oldest waits a fifth of its time at the tile barrier while the
# youngest, alone at the end, cannot fill the SIMD.  1: the priority rotates, (position of the wave among its SIMD's four + chunks
# started) mod 4; 2: by the chunks the wave still has in this tile (most remaining first), min(3, remaining >> DQ2_ROT_SHIFT).
# One counter check per POSITION (two steps) instead of one per step: `s_sub 2` borrows exactly when the quad ends inside the
# position (1 or 2 steps left), and that case runs out of line (end_s_k: once per quad and tile).  The straight path of a position
# is 13 instructions instead of 15, with one conditional branch instead of two.
# The tile barrier as late as it can be: everything between "my pieces have landed" and the first chunk that touches neither the
# LDS tile nor the other buffer (the wait for the step counts, their four VALU instructions, the next tile's DMA set-up, the
# chunk routine's choice) runs BEFORE s_barrier -- the waves that arrive early do it while they would wait anyway, and all sixteen
# start streaming right behind the barrier instead of queueing 50 instructions each behind it (the youngest wave of a SIMD
# entered its first chunk 550 cycles after the oldest: profiles/r05_dq_stamps_final.txt).
LATEBAR = os.environ.get("DQ2_LATEBAR", "1") != "0"   # (the default since round 5)
ONEBR = os.environ.get("DQ2_ONEBR", "1") != "0"   # (the default since round 5; DQ2_ONEBR=0: a check per step)
ROT = int(os.environ.get("DQ2_ROT", "2"))        # (the default since round 5: 2; DQ2_ROT=0: none)
ROT_SHIFT = int(os.environ.get("DQ2_ROT_SHIFT", "1"))   # -1: per tile, 1 for streams of more than DQ2_ROT_LONG chunks, else 0 (short streams: a chunk is a big share of the tile)
ROT_LONG = int(os.environ.get("DQ2_ROT_LONG", "7"))
ROT_SUB = int(os.environ.get("DQ2_ROT_SUB", "0"))     # 2: min(3, max(0, remaining - SUB) >> SHIFT)
ROT_AGE = os.environ.get("DQ2_ROT_AGE", "0") != "0"   # 2: the wave's place among its SIMD's four (0 = oldest) is added to `remaining`, and is
                                                      # its priority between the tile barrier and its first chunk (the youngest leaves the barrier last)
# 3: most-remaining-first only where it decides who reaches the barrier last -- the last three chunks of a (wave, tile) stream
# run at priority 2, 1, 0 and everything before at 3; set on the out-of-line path those chunks take anyway (tail_r), so the
# straight path of a chunk carries no priority code at all.  DQ2_ROT_AGE=1: the two younger waves of a SIMD step down one chunk later.
ILV = os.environ.get("DQ2_ILV", "0") != "0"     # FMAs of a group interleaved with the next group's DPP instructions, reads last
# experiment switches (environment, read when the header is generated)
B64 = os.environ.get("DQ2_B64", "0") != "0"     # one 64-bit row_newbcast move per step ({offset, value}) + a plain add, instead of add_dpp + mov_dpp
FMAC = os.environ.get("DQ2_FMAC", "0") != "0"   # four v_fmac_f32 per step instead of two v_pk_fma_f32
P64 = [[20, 22], [24, 26]]                        # B64: even-aligned pairs, lo becomes the LDS address, hi is the value
# DQ2_STAMPS=1 (debug builds only: tools/dq2_variant.sh stamps DQ2_STAMPS=1, tools/dq_stamps.py): s_memtime stamps where a wave
# can wait, kept in lanes of spare VGPRs (lane = tile & 63) and handed to the kernel's epilogue, which writes them to a side
# buffer.  Per (wave, tile): R0 tile top, R1 before the wait for this wave's LDS-DMA pieces, R2 pieces landed = barrier arrive,
# R3 barrier leave, R4 first chunk entered, R5 tile done; sums over the tile's chunks: R6 the time spent at `s_waitcnt vmcnt(2)`
# (entry loads), R7 one LDS round trip per chunk (C: group 0's reads issued -> D: position 0's wait for them over), R9 the
# wave's own time between the two (C -> E: it arrives at that wait; R7 - R9 = what position 0 waited), R10 the wait of
# position 4 (steady state: its reads were issued one position earlier); R8 = chunks.  A stamp issued in front of a wait
# counts in lgkmcnt itself, so a measured LDS wait is at least one SMEM latency (the floor shows in R6 of chunks whose
# entries had long arrived).  f32 sweeps only.
STAMPS = os.environ.get("DQ2_STAMPS", "0") != "0"
ST_A, ST_B = 98, 100           # s[98:99], s[100:101]: stamp pairs (free in the plain kernel)
ST_E, ST_E4, ST_D4 = 74, 64, 76   # s[74:75], s[64:65], s[76:77]: stamps before position 0's wait, before / after position 4's
ST_SVM, ST_SL, ST_SOWN, ST_SP4 = 37, 39, 46, 60   # scalar accumulators of a tile
ST_R = 120                     # v[120:127] = R0..R7
ST_R8, ST_R9, ST_R10 = 53, 54, 55   # (+ v51, v52 unused; v[52:55] leaves the block as one operand)


F64 = False       # set per variant in main(): the f64 sweep (16-byte entries, 512-byte panel rows, four f64 columns per lane)
ACCW = 4          # accumulator registers per row slot (8 for f64)
CHUNK_B = 512     # bytes of one 16-step chunk of a wave's stream (1024 for f64)
RPP = 4           # panel rows per 1 KiB LDS-DMA piece (2 for f64)


def set_f64(on):
    """register map and sizes of the f64 variant: entries {u32 offset, u32 pad, f64 value}, a lane holds columns 2q, 2q+1,
    32+2q, 32+2q+1 of its row (two ds_read_b128, 256 bytes apart), four v_fma_f64 per step"""
    global F64, ACC, ACCW, CHUNK_B, RPP, EB, ECUR, VT, VLB, A, B, VINFO, VA64, W, VCNT, VCNT2
    F64 = on
    if on:
        ACC, ACCW, CHUNK_B, RPP = 80, 8, 1024, 2
        EB = [(10, 13), (14, 17), (18, 21)]
        ECUR = (22, 25)                 # x = v22 (offset), value in v[24:25]
        VT, VLB = 26, 27
        A = [[28, 29], [30, 31]]
        B = [[32, 34], [36, 38]]       # 64-bit value pairs
        VINFO, VA64 = (40, 41), 42
        W = [[44, 52], [60, 68]]       # 8 registers per step
        VCNT, VCNT2 = 76, 77
    else:
        ACC, ACCW, CHUNK_B, RPP = 56, 4, 512, 4
        EB = [(10, 11), (12, 13), (14, 15)]
        ECUR = (16, 17)
        VT, VLB = 18, 19
        A = [[20, 21], [22, 23]]
        B = [24, 26]
        VINFO, VA64 = (28, 29), 30
        W = [[32, 36], [40, 44]]
        VCNT, VCNT2 = 48, 49


# scalars
S_C = 36                      # groups the current quad still has after this one
S_REM = 38                    # chunks left in this tile, the one about to start included
S_PTR = 40                    # s[40:41] entry stream pointer of the chunk about to start
S_STP = 42                    # s[42:43] steps table of the current tile (this wave's 16 quads)
S_T, S_NT, S_TABS, S_BUF, S_NCH, S_OFF8 = 44, 45, 47, 48, 49, 50
S_A, S_B2, S_CC, S_D, S_E = 51, 52, 53, 54, 55
S_INFO = 56                   # s[56:57] info pointer of the current 64-tile window
S_4NCT, S_TLAST = 58, 59
S_NP, S_DROW, S_DLDS = 61, 62, 63        # LDS-DMA pieces of the next tile still to issue; their next panel row / LDS address
S_ND = 66                     # vector-memory operations issued since the last piece
S_LINK, S_PRE, S_BASE = 68, 69, 70       # this tile preloads the next one; this tile was preloaded; entry buffer of its chunk 0
S_LAST = 71
S_PTRN = 72                   # s[72:73] entry stream pointer of the next tile
S_DADDR = 78                  # s[78:79] scalar source address of the next LDS-DMA piece
S_PSTEP, S_LIM, S_ROWBW = 80, 81, 82
S_RET = 84                    # s[84:85] position 0 of the current slot's copy of the main loop
S_CTL = 86                    # s[86:87] chunk routine of the next chunk's entry buffer
S_CTLA = [88, 90, 92]         # s[88:93] the three chunk routines
S_BODY0 = 94                  # s[94:95] position 0 of slot 0's copy
S_QM = 96                     # s[96:97] lanes that hold one of this wave's quads
S_ROT = 83                    # DQ2_ROT=1: chunks started + the wave's place among the four of its SIMD
S_SH = 67                     # DQ2_ROT_SHIFT=-1: this tile's shift of the remaining-chunks priority

_uid = [0]


def uid(prefix):
    _uid[0] += 1
    return f"{prefix}{_uid[0]}"


def stamp(L, reg):
    """s_memtime -> lane (tile & 63) of v[reg]; only at points where no LDS read is in flight"""
    if not STAMPS:
        return
    L += [f"s_memtime s[{ST_A}:{ST_A + 1}]", f"s_and_b32 s{ST_B}, s{S_T}, 63", "s_waitcnt lgkmcnt(0)", f"s_mov_b32 m0, s{ST_B}", "s_nop 3",
          f"v_writelane_b32 v{reg}, s{ST_A}, m0"]


def stamp_lds_sample_close(L):
    """the samples of the chunk that just ended (every pair is zero when there was none): D - C, E - C, D4 - E4"""
    L += [f"s_sub_u32 s{ST_B}, s{ST_B}, s{ST_A}", f"s_add_u32 s{ST_SL}, s{ST_SL}, s{ST_B}",
          f"s_sub_u32 s{ST_E}, s{ST_E}, s{ST_A}", f"s_add_u32 s{ST_SOWN}, s{ST_SOWN}, s{ST_E}",
          f"s_sub_u32 s{ST_D4}, s{ST_D4}, s{ST_E4}", f"s_add_u32 s{ST_SP4}, s{ST_SP4}, s{ST_D4}",
          f"s_mov_b64 s[{ST_E4}:{ST_E4 + 1}], 0", f"s_mov_b64 s[{ST_D4}:{ST_D4 + 1}], 0"]   # (a chunk may end before position 4)


def stamp_pairs_clear(L):
    for pr in (ST_A, ST_B, ST_E, ST_E4, ST_D4):
        L.append(f"s_mov_b64 s[{pr}:{pr + 1}], 0")
    for a in (ST_SVM, ST_SL, ST_SOWN, ST_SP4):
        L.append(f"s_mov_b32 s{a}, 0")


def setprio_tree(L, sreg, tag):
    """s_setprio takes an immediate: s[sreg] in 0..3 through a two-level branch"""
    L += [f"s_cmp_lt_u32 s{sreg}, 2", f"s_cbranch_scc1 rotlo_{tag}", f"s_cmp_eq_u32 s{sreg}, 2", f"s_cbranch_scc1 rot2_{tag}", "s_setprio 3",
          f"s_branch rotdone_{tag}", f"rot2_{tag}:", "s_setprio 2", f"s_branch rotdone_{tag}", f"rotlo_{tag}:", f"s_cmp_eq_u32 s{sreg}, 0",
          f"s_cbranch_scc1 rot0_{tag}", "s_setprio 1", f"s_branch rotdone_{tag}", f"rot0_{tag}:", "s_setprio 0", f"rotdone_{tag}:"]


def grp_a_parts(k):
    """two steps issued: (address instructions, value instructions, panel row reads)"""
    p = k % DEPTH
    ex, ey = ECUR
    adds, movs, reads = [], [], []
    if F64:
        for t in range(2):
            adds.append(f"v_add_u32_dpp v{A[p][t]}, v{ex}, v{VLB} row_newbcast:{2 * k + t} row_mask:0xf bank_mask:0xf")
        for t in range(2):
            w = W[p][t]
            reads.append(f"ds_read_b128 v[{w}:{w + 3}], v{A[p][t]}")
            reads.append(f"ds_read_b128 v[{w + 4}:{w + 7}], v{A[p][t]} offset:256")
        for t in range(2):
            movs.append(f"v_mov_b64_dpp v[{B[p][t]}:{B[p][t] + 1}], v[{ex + 2}:{ex + 3}] row_newbcast:{2 * k + t} row_mask:0xf bank_mask:0xf")
        return adds, movs, reads
    if B64:
        for t in range(2):
            adds.append(f"v_mov_b64_dpp v[{P64[p][t]}:{P64[p][t] + 1}], v[{ex}:{ey}] row_newbcast:{2 * k + t} row_mask:0xf bank_mask:0xf")
        for t in range(2):
            movs.append(f"v_add_u32 v{P64[p][t]}, v{P64[p][t]}, v{VLB}")
        for t in range(2):
            reads.append(f"ds_read_b128 v[{W[p][t]}:{W[p][t] + 3}], v{P64[p][t]}")
        return adds, movs, reads
    for t in range(2):
        adds.append(f"v_add_u32_dpp v{A[p][t]}, v{ex}, v{VLB} row_newbcast:{2 * k + t} row_mask:0xf bank_mask:0xf")
    for t in range(2):
        reads.append(f"ds_read_b128 v[{W[p][t]}:{W[p][t] + 3}], v{A[p][t]}")
    for t in range(2):
        movs.append(f"v_mov_b32_dpp v{B[p] + t}, v{ey} row_newbcast:{2 * k + t} row_mask:0xf bank_mask:0xf")
    return adds, movs, reads


def grp_a(k, L):
    adds, movs, reads = grp_a_parts(k)
    if B64 or ILV:
        L += adds + movs + reads      # (B64: the add sits between the move and the read; ILV: the reads as far behind their addresses as they go)
    else:
        L += adds + reads + movs


def grp_fma_parts(s, k):
    """the FMAs of group k into slot s's accumulators: (first step, second step)"""
    p = k % DEPTH
    a = ACC + ACCW * s
    out = []
    for t in range(2):
        w = W[p][t]
        L = []
        if F64:
            b = B[p][t]
            for c in range(4):
                L.append(f"v_fma_f64 v[{a + 2 * c}:{a + 2 * c + 1}], v[{w + 2 * c}:{w + 2 * c + 1}], v[{b}:{b + 1}], v[{a + 2 * c}:{a + 2 * c + 1}]")
        elif FMAC:
            val = P64[p][t] + 1 if B64 else B[p] + t
            for c in range(4):
                L.append(f"v_fmac_f32 v{a + c}, v{val}, v{w + c}")
        else:
            if B64:
                b, sel = P64[p][t], "op_sel:[0,1,0] op_sel_hi:[1,1,1]"    # the value is the pair's upper half
            else:
                b, sel = B[p], ("op_sel_hi:[1,0,1]" if t == 0 else "op_sel:[0,1,0] op_sel_hi:[1,1,1]")
            L.append(f"v_pk_fma_f32 v[{a}:{a + 1}], v[{w}:{w + 1}], v[{b}:{b + 1}], v[{a}:{a + 1}] {sel}")
            L.append(f"v_pk_fma_f32 v[{a + 2}:{a + 3}], v[{w + 2}:{w + 3}], v[{b}:{b + 1}], v[{a + 2}:{a + 3}] {sel}")
        out.append(L)
    return out


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


def dma_piece(L):
    """one 1 KiB LDS-DMA piece of the next tile: 4 panel rows x 256 bytes (lane group g: row S_DROW + (20 wave + g) nct,
    lane: 16 bytes of it).  The source address is scalar (s[S_DADDR] + the lane's constant offset) unless one of the four
    rows lies past the panel's end: those pieces (the last one or two of a tile) clamp the row per lane."""
    slow, done = uid("dmaslow"), uid("dmadone")
    L += [f"s_cmp_gt_i32 s{S_DROW}, s{S_LIM}", f"s_cbranch_scc1 {slow}"]
    L.append(f"s_mov_b32 m0, s{S_DLDS}")
    L.append("s_nop 0")
    L.append(f"global_load_lds_dwordx4 %[voff], s[{S_DADDR}:{S_DADDR + 1}]")
    L.append(f"s_branch {done}")
    L.append(f"{slow}:")
    # 64-bit per-lane source address: panels above 4 GiB exist (10M rows x 128 columns)
    L.append(f"v_add_u32 v{VT}, s{S_DROW}, %[rowb0]")
    L.append(f"v_min_u32 v{VT}, %[prm1], v{VT}")
    L.append(f"v_mul_hi_u32 v{VA64 + 1}, v{VT}, %[stride]")
    L.append(f"v_mul_lo_u32 v{VA64}, v{VT}, %[stride]")
    L.append(f"v_add_co_u32 v{VA64}, vcc, %[col16], v{VA64}")
    L.append(f"v_addc_co_u32 v{VA64 + 1}, vcc, 0, v{VA64 + 1}, vcc")
    L.append(f"v_mov_b32 v{VT}, %[xhi]")                      # (an SGPR operand beside vcc would be a second constant-bus read)
    L.append(f"v_add_co_u32 v{VA64}, vcc, %[xlo], v{VA64}")
    L.append(f"v_addc_co_u32 v{VA64 + 1}, vcc, v{VT}, v{VA64 + 1}, vcc")
    L.append(f"s_mov_b32 m0, s{S_DLDS}")
    L.append("s_nop 0")
    L.append(f"global_load_lds_dwordx4 v[{VA64}:{VA64 + 1}], off")
    L.append(f"{done}:")
    L.append(f"s_add_u32 s{S_DROW}, s{S_DROW}, s{S_4NCT}")
    L.append(f"s_add_u32 s{S_DLDS}, s{S_DLDS}, 0x400")
    L += [f"s_add_u32 s{S_DADDR}, s{S_DADDR}, s{S_PSTEP}", f"s_addc_u32 s{S_DADDR + 1}, s{S_DADDR + 1}, 0"]
    L += [f"s_sub_u32 s{S_NP}, s{S_NP}, 1", f"s_mov_b32 s{S_ND}, 0"]


def ptr_from_off8(L, s_off8, dst, base):
    """s[dst:dst+1] = base (64-bit operand name) + 64 * s_off8   (s_off8: entry offset in units of 8 entries)"""
    sh = (5 if ODD else 6) + (1 if F64 else 0)   # (ODD: streams start at multiples of 4 entries; f64: 16-byte entries)
    L.append(f"s_lshl_b32 s{S_CC}, s{s_off8}, {sh}")
    L.append(f"s_lshr_b32 s{S_D}, s{s_off8}, {32 - sh}")
    L.append(f"s_mov_b64 s[{dst}:{dst + 1}], %[{base}]")
    L.append(f"s_add_u32 s{dst}, s{dst}, s{S_CC}")
    L.append(f"s_addc_u32 s{dst + 1}, s{dst + 1}, s{S_D}")


def set_ret(L, s):
    """S_RET = position 0 of slot s's copy"""
    if s == 0:
        L.append(f"s_mov_b64 s[{S_RET}:{S_RET + 1}], s[{S_BODY0}:{S_BODY0 + 1}]")
    else:
        L += [f"s_add_u32 s{S_RET}, s{S_BODY0}, body_{s}_0-body_0_0", f"s_addc_u32 s{S_RET + 1}, s{S_BODY0 + 1}, 0"]


def bodies(L):
    """RG copies of a chunk's eight two-step groups.  Position k of slot s: the FMAs of group k into slot s's accumulators,
    then group k + 2 issued; the quad's counter; fall through to position k + 1 (position 7: the chunk routine).
    ODD: the counter runs per step, so a quad may hand over between the two steps of a group (body_s_k_mid)."""
    tails = []
    for s in range(RG):
        for k in range(8):
            L.append(f"body_{s}_{k}:")
            ahead = min(DEPTH, 8 - k) - 1
            if STAMPS and k in (0, 4):
                L.append(f"s_memtime s[{ST_E if k == 0 else ST_E4}:{(ST_E if k == 0 else ST_E4) + 1}]")
            L.append(f"s_waitcnt lgkmcnt({(4 if F64 else 2) * ahead})")
            if STAMPS and k in (0, 4):
                L.append(f"s_memtime s[{ST_B if k == 0 else ST_D4}:{(ST_B if k == 0 else ST_D4) + 1}]")
            f0, f1 = grp_fma_parts(s, k)
            if ODD and ONEBR:
                # S_C = steps the quad has left, this position's first included, minus one
                L += [f"s_sub_u32 s{S_C}, s{S_C}, 2", f"s_cbranch_scc1 end_{s}_{k}"]
                L += f0
                L.append(f"body_{s}_{k}_mid:")
                L += f1
                nxt = []
                if k + DEPTH < 8:
                    grp_a(k + DEPTH, nxt)
                L += nxt
                # out of line: the quad ends inside this position
                T = [f"end_{s}_{k}:", f"s_cmp_eq_u32 s{S_C}, -1", f"s_cbranch_scc1 end2_{s}_{k}"]
                T += f0 + [f"s_branch landm_{s + 1}_{k}"]                      # one step left: the second step is the next quad's first
                T += [f"end2_{s}_{k}:"] + f0 + [f"end2_{s}_{k}_mid:"] + f1 + nxt + [f"s_branch land_{s + 1}_{k}"]   # two left
                tails.append(T)
                continue
            if ODD:
                L += f0
                L += [f"s_sub_u32 s{S_C}, s{S_C}, 1", f"s_cbranch_scc1 landm_{s + 1}_{k}", f"body_{s}_{k}_mid:"]
                L += f1
                if k + DEPTH < 8:
                    grp_a(k + DEPTH, L)
            elif k + DEPTH < 8 and ILV and not B64:
                # the two steps' FMAs go to the same accumulators: the next group's address / value instructions between them
                adds, movs, reads = grp_a_parts(k + DEPTH)
                L += f0 + adds + f1 + movs + reads
            else:
                L += f0 + f1
                if k + DEPTH < 8:
                    grp_a(k + DEPTH, L)
            L.append(f"s_sub_u32 s{S_C}, s{S_C}, 1")
            L.append(f"s_cbranch_scc1 land_{s + 1}_{k}")          # the quad's last group: on to the next non-empty quad
        L.append(f"s_setpc_b64 s[{S_CTL}:{S_CTL + 1}]")
    for T in tails:
        L += T


def stubs(L):
    """land_s_k: slot s takes over behind position k (k = 7: at the next chunk's position 0, through the chunk routine).
    Empty quads (no step in this tile) pass the turn on; behind the last slot the tile is done."""
    for k in range(8):
        for s in range(1, RG + 1):
            L.append(f"land_{s}_{k}:")
            if s == RG:
                L.append("s_branch tile_done")
                continue
            L += [f"v_readlane_b32 s{S_C}, v{VCNT}, {s}", f"s_cmp_lt_i32 s{S_C}, 0", f"s_cbranch_scc1 land_{s + 1}_{k}"]
            set_ret(L, s)
            if k < 7:
                L.append(f"s_branch body_{s}_{k + 1}")
            else:
                L.append(f"s_setpc_b64 s[{S_CTL}:{S_CTL + 1}]")
    if ODD:   # the same hand-over between the two steps of group k
        for k in range(8):
            for s in range(1, RG + 1):
                L.append(f"landm_{s}_{k}:")
                if s == RG:
                    L.append("s_branch tile_done")
                    continue
                L += [f"v_readlane_b32 s{S_C}, v{VCNT}, {s}", f"s_cmp_lt_i32 s{S_C}, 0", f"s_cbranch_scc1 landm_{s + 1}_{k}"]
                set_ret(L, s)
                if ONEBR:   # the quad's first step is the position's second: one step fewer at the next position -- or none at all
                    L += [f"s_sub_u32 s{S_C}, s{S_C}, 1", f"s_cbranch_scc1 end2_{s}_{k}_mid"]
                L.append(f"s_branch body_{s}_{k}_mid")
    # tile entry: the first non-empty quad, then the first chunk's routine
    L.append("enter_0:")
    L += [f"v_readlane_b32 s{S_C}, v{VCNT}, 0", f"s_cmp_lt_i32 s{S_C}, 0", "s_cbranch_scc1 land_1_7"]
    set_ret(L, 0)
    L.append(f"s_setpc_b64 s[{S_CTL}:{S_CTL + 1}]")


def LOAD():
    """the chunk load: 8 bytes per lane (f32 entries) or 16"""
    return "global_load_dwordx4" if F64 else "global_load_dwordx2"


def chunk_routines(L, pattern):
    """ctl_r: the chunk in entry buffer r is about to run.  Its entries are complete at vmcnt(2): every chunk slot issues
    exactly one entry load, as its last vector-memory operation, so two younger loads (and whatever came with them) may
    still be out."""
    for r in range(3):
        e = EB[r]
        L.append(f"ctl_{r}:")
        if STAMPS:   # (no LDS read is in flight here: position 7 waited for the chunk's last)
            stamp_lds_sample_close(L)
            L.append(f"s_memtime s[{ST_A}:{ST_A + 1}]")
        L.append("s_waitcnt vmcnt(2)")
        if STAMPS:
            L += [f"s_memtime s[{ST_B}:{ST_B + 1}]", "s_waitcnt lgkmcnt(0)", f"s_sub_u32 s{ST_B}, s{ST_B}, s{ST_A}", f"s_add_u32 s{ST_SVM}, s{ST_SVM}, s{ST_B}"]
        if F64:
            L += [f"v_mov_b32 v{ECUR[0]}, v{e[0]}", f"v_mov_b32 v{ECUR[0] + 2}, v{e[0] + 2}", f"v_mov_b32 v{ECUR[0] + 3}, v{e[0] + 3}"]
        else:
            L += [f"v_mov_b32 v{ECUR[0]}, v{e[0]}", f"v_mov_b32 v{ECUR[1]}, v{e[1]}"]
        if pattern:
            # pattern mode (MaskedSparsePCA's projection, quirk Q3): every stored non-zero value counts as 1, zeros
            # (padding, and stored zeros, which the caller handles) as 0
            L += ["s_nop 0", f"v_cmp_neq_f32 vcc, 0, v{ECUR[1]}", f"v_cndmask_b32 v{ECUR[1]}, 0, 1.0, vcc"]
        if ROT:
            if ROT == 1:
                L += [f"s_add_u32 s{S_ROT}, s{S_ROT}, 1", f"s_and_b32 s{S_E}, s{S_ROT}, 3"]
            else:
                L.append(f"s_mov_b32 s{S_E}, s{S_REM}")
                if ROT_AGE:
                    L.append(f"s_add_u32 s{S_E}, s{S_E}, s{S_ROT}")
                if ROT_SUB:
                    L += [f"s_max_u32 s{S_E}, s{S_E}, {ROT_SUB}", f"s_sub_u32 s{S_E}, s{S_E}, {ROT_SUB}"]
                if ROT_SHIFT > 0:
                    L.append(f"s_lshr_b32 s{S_E}, s{S_E}, {ROT_SHIFT}")
                elif ROT_SHIFT < 0:
                    L.append(f"s_lshr_b32 s{S_E}, s{S_E}, s{S_SH}")
                L.append(f"s_min_u32 s{S_E}, s{S_E}, 3")
            setprio_tree(L, S_E, f"c{r}")
        # one LDS-DMA piece of the next tile while there are any (out of line)
        L += [f"s_cmp_lg_u32 s{S_NP}, 0", f"s_cbranch_scc1 dma_{r}", f"dmaback_{r}:"]
        # the buffer's next load: three chunks ahead in this tile; in the last three slots the next tile's first chunks (linked)
        L += [f"s_cmp_le_u32 s{S_REM}, 3", f"s_cbranch_scc1 tail_{r}",
              f"{LOAD()} v[{e[0]}:{e[1]}], %[eoff], s[{S_PTR}:{S_PTR + 1}] offset:{3 * CHUNK_B} nt", f"issued_{r}:"]
        L += [f"s_add_u32 s{S_ND}, s{S_ND}, 1", f"s_add_u32 s{S_PTR}, s{S_PTR}, {CHUNK_B}", f"s_addc_u32 s{S_PTR + 1}, s{S_PTR + 1}, 0",
              f"s_sub_u32 s{S_REM}, s{S_REM}, 1",
              f"s_mov_b64 s[{S_CTL}:{S_CTL + 1}], s[{S_CTLA[(r + 1) % 3]}:{S_CTLA[(r + 1) % 3] + 1}]"]
        for k in range(DEPTH):
            grp_a(k, L)
            if STAMPS and k == 0:
                L.append(f"s_memtime s[{ST_A}:{ST_A + 1}]")
        L.append(f"s_setpc_b64 s[{S_RET}:{S_RET + 1}]")
    for r in range(3):
        e = EB[r]
        L.append(f"dma_{r}:")
        dma_piece(L)
        L.append(f"s_branch dmaback_{r}")
        # last three slots of a tile: chunk 3 - rem of the next tile when linked (its step counts go first, rem == 3);
        # otherwise a load nobody reads, so that the slot still issues one (the wait counts rely on it)
        L.append(f"tail_{r}:")
        if ROT == 3:
            src = S_REM
            if ROT_AGE:   # S_ROT = 1 for the two younger waves of the SIMD
                L += [f"s_add_u32 s{S_E}, s{S_REM}, s{S_ROT}"]
                src = S_E
            L += [f"s_cmp_ge_u32 s{src}, 4", f"s_cbranch_scc1 tp3_{r}", f"s_cmp_eq_u32 s{src}, 3", f"s_cbranch_scc1 tp2_{r}", f"s_cmp_eq_u32 s{src}, 2",
                  f"s_cbranch_scc1 tp1_{r}", "s_setprio 0", f"s_branch tpd_{r}", f"tp1_{r}:", "s_setprio 1", f"s_branch tpd_{r}", f"tp2_{r}:", "s_setprio 2",
                  f"s_branch tpd_{r}", f"tp3_{r}:", "s_setprio 3", f"tpd_{r}:"]
        L += [f"s_cmp_eq_u32 s{S_LINK}, 0", f"s_cbranch_scc1 dummy_{r}",
              f"s_cmp_lg_u32 s{S_REM}, 3", f"s_cbranch_scc1 nocnt_{r}",
              f"global_load_ushort v{VCNT2}, %[l2c], s[{S_STP}:{S_STP + 1}] offset:512",
              f"s_add_u32 s{S_ND}, s{S_ND}, 1", f"nocnt_{r}:",
              f"s_sub_u32 s{S_A}, 3, s{S_REM}", f"s_lshl_b32 s{S_A}, s{S_A}, {10 if F64 else 9}", f"v_add_u32 v{VT}, s{S_A}, %[eoff]",
              f"{LOAD()} v[{e[0]}:{e[1]}], v{VT}, s[{S_PTRN}:{S_PTRN + 1}] nt", f"s_branch issued_{r}",
              f"dummy_{r}:", f"{LOAD()} v[{e[0]}:{e[1]}], %[eoff], s[{S_PTR}:{S_PTR + 1}] nt", f"s_branch issued_{r}"]


def body(pattern):
    L = []
    # accumulators <- 0
    for i in range(ACCW * RG):
        L.append(f"v_mov_b32 v{ACC + i}, 0")
    if STAMPS:
        for v in list(range(ST_R, ST_R + 8)) + [52, ST_R8, ST_R9, ST_R10]:
            L.append(f"v_mov_b32 v{v}, 0")
    L += [f"s_mov_b32 s{S_T}, 0", f"s_mov_b32 s{S_NT}, %[ntiles]", f"s_mov_b32 s{S_TABS}, %[t0]",
          f"s_mov_b32 s{S_BUF}, 0", f"s_mov_b64 s[{S_INFO}:{S_INFO + 1}], %[info]", f"s_lshl_b32 s{S_4NCT}, %[nct], {2 if RPP == 4 else 1}",   # panel rows a piece advances by: RPP * nct
          f"s_add_u32 s{S_TLAST}, %[t0], %[ntiles]", f"s_sub_u32 s{S_TLAST}, s{S_TLAST}, 1",
          f"s_mov_b32 s{S_ND}, 0", f"s_mov_b32 s{S_PRE}, 0", f"s_mov_b32 s{S_BASE}, 0", f"s_mov_b64 s[{S_STP}:{S_STP + 1}], %[stp]"]
    # code addresses: the three chunk routines and slot 0's copy of the main loop
    L += [f"s_getpc_b64 s[{S_BODY0}:{S_BODY0 + 1}]", "pcref:"]
    for r in range(3):
        L += [f"s_add_u32 s{S_CTLA[r]}, s{S_BODY0}, ctl_{r}-pcref", f"s_addc_u32 s{S_CTLA[r] + 1}, s{S_BODY0 + 1}, 0"]
    L += [f"s_add_u32 s{S_BODY0}, s{S_BODY0}, body_0_0-pcref", f"s_addc_u32 s{S_BODY0 + 1}, s{S_BODY0 + 1}, 0"]
    # lanes that hold a quad of this wave: lane < my_quads  (l2 = 2 * lane)
    L += [f"s_lshl_b32 s{S_A}, %[myq], 1", f"v_cmp_gt_u32 vcc, s{S_A}, %[l2]", f"s_mov_b64 s[{S_QM}:{S_QM + 1}], vcc"]
    L += [f"v_readfirstlane_b32 s{S_ROWBW}, %[rowb0]", f"s_mul_i32 s{S_PSTEP}, s{S_4NCT}, %[stride]",
          f"s_mul_i32 s{S_LIM}, %[nct], {RPP - 1}", f"s_add_u32 s{S_LIM}, s{S_LIM}, s{S_ROWBW}", f"s_sub_u32 s{S_LIM}, %[prm1], s{S_LIM}"]
    if ROT == 1 or (ROT and ROT_AGE):
        # the stream slot this wave walks is (wdma - LDS base) / 5120; hardware wave w walks slot (w & 3) * 4 + 3 - w / 4
        # (spmm_dq.hip, stream_of_wave), so 3 - (slot & 3) = w / 4 is the wave's place among the four of its SIMD (0: the oldest)
        L += [f"v_readfirstlane_b32 s{S_ROT}, %[lb]", f"s_sub_u32 s{S_ROT}, %[wdma], s{S_ROT}", f"s_lshr_b32 s{S_ROT}, s{S_ROT}, 10",
              f"s_mul_i32 s{S_ROT}, s{S_ROT}, 205", f"s_lshr_b32 s{S_ROT}, s{S_ROT}, 10", f"s_and_b32 s{S_ROT}, s{S_ROT}, 3",
              f"s_sub_u32 s{S_ROT}, 3, s{S_ROT}"]
        if ROT == 3:
            L.append(f"s_lshr_b32 s{S_ROT}, s{S_ROT}, 1")
    # info window: lane t' holds {entry offset / 8, chunk count} of tile t0 + 64 * window + t'
    L += [f"global_load_dwordx2 v[{VINFO[0]}:{VINFO[1]}], %[l8], s[{S_INFO}:{S_INFO + 1}]"]
    # first tile: this wave's five pieces into buffer 0, synchronously
    L += [f"s_mov_b32 s{S_DROW}, s{S_TABS}", f"s_mov_b32 s{S_DLDS}, %[wdma]", f"s_mov_b32 s{S_NP}, 5"]
    dma_tile_setup(L)
    L += ["first_pieces:"]
    dma_piece(L)
    L += [f"s_cmp_lg_u32 s{S_NP}, 0", "s_cbranch_scc1 first_pieces", "s_waitcnt vmcnt(0)"]
    L += ["tile_top:"]   # ---- tile loop
    stamp(L, ST_R + 0)
    # a new 64-tile window of the info table (tiles are never linked across windows)
    L += [f"s_and_b32 s{S_A}, s{S_T}, 63", f"s_cmp_lg_u32 s{S_A}, 0", "s_cbranch_scc1 same_window", f"s_cmp_eq_u32 s{S_T}, 0",
          "s_cbranch_scc1 same_window", f"s_add_u32 s{S_INFO}, s{S_INFO}, 0x200", f"s_addc_u32 s{S_INFO + 1}, s{S_INFO + 1}, 0",
          f"global_load_dwordx2 v[{VINFO[0]}:{VINFO[1]}], %[l8], s[{S_INFO}:{S_INFO + 1}]", "s_waitcnt vmcnt(0)", "same_window:"]
    L += [f"s_and_b32 s{S_A}, s{S_T}, 63", f"v_readlane_b32 s{S_OFF8}, v{VINFO[0]}, s{S_A}", f"v_readlane_b32 s{S_NCH}, v{VINFO[1]}, s{S_A}"]
    if PRIO:
        # issue priority for this tile: the rank of this wave's step count among the four waves of its SIMD (bits 16..17 of the
        # info word): the waves of a SIMD then reach the tile's barrier together instead of leaving the longest to finish alone
        L += [f"s_bfe_u32 s{S_B2}, s{S_NCH}, 0x20010", f"s_cmp_eq_u32 s{S_B2}, 0", "s_cbranch_scc1 prio0", f"s_cmp_eq_u32 s{S_B2}, 1", "s_cbranch_scc1 prio1",
              f"s_cmp_eq_u32 s{S_B2}, 2", "s_cbranch_scc1 prio2", "s_setprio 3", "s_branch prio_set", "prio2:", "s_setprio 2", "s_branch prio_set",
              "prio1:", "s_setprio 1", "s_branch prio_set", "prio0:", "s_setprio 0", "prio_set:"]
    L += [f"s_and_b32 s{S_NCH}, s{S_NCH}, 0xffff"]
    ptr_from_off8(L, S_OFF8, S_PTR, "ent")
    # link to the next tile?  (same info window, both with at least three chunks)
    L += [f"s_mov_b32 s{S_LINK}, 0", f"s_add_u32 s{S_B2}, s{S_T}, 1", f"s_cmp_ge_u32 s{S_B2}, s{S_NT}", "s_cbranch_scc1 no_link",
          f"s_and_b32 s{S_B2}, s{S_B2}, 63", f"s_cmp_eq_u32 s{S_B2}, 0", "s_cbranch_scc1 no_link", f"s_cmp_lt_u32 s{S_NCH}, 3", "s_cbranch_scc1 no_link",
          f"v_readlane_b32 s{S_LAST}, v{VINFO[1]}, s{S_B2}", f"s_and_b32 s{S_LAST}, s{S_LAST}, 0xffff", f"s_cmp_lt_u32 s{S_LAST}, 3", "s_cbranch_scc1 no_link",
          f"v_readlane_b32 s{S_LAST}, v{VINFO[0]}, s{S_B2}", f"s_mov_b32 s{S_LINK}, 1"]
    ptr_from_off8(L, S_LAST, S_PTRN, "ent")
    L += ["no_link:"]
    # a tile that was not preloaded: its first three chunks and its step counts
    L += [f"s_cmp_lg_u32 s{S_PRE}, 0", "s_cbranch_scc1 preloaded"]
    for i, e in enumerate(EB):
        L.append(f"{LOAD()} v[{e[0]}:{e[1]}], %[eoff], s[{S_PTR}:{S_PTR + 1}] offset:{CHUNK_B * i} nt")
    L.append(f"global_load_ushort v{VCNT2}, %[l2c], s[{S_STP}:{S_STP + 1}]")
    L += [f"s_add_u32 s{S_ND}, s{S_ND}, 4", f"s_mov_b32 s{S_BASE}, 0", "preloaded:"]
    # this wave's pieces of the tile have landed: wait for all but what was issued after the last of them
    stamp(L, ST_R + 1)
    wait_vmcnt(L, S_ND, 6)
    def tile_barrier(L):
        stamp(L, ST_R + 2)
        L += ["s_barrier"]
        if ROT == 2 and ROT_AGE:
            setprio_tree(L, S_ROT, "bar")
        if ROT == 3:
            L.append("s_setprio 3")
        stamp(L, ST_R + 3)
    if not LATEBAR:
        tile_barrier(L)
    # the step counts (issued before the last three entry loads of a linked predecessor); everything of a tile that loaded its own
    L += [f"s_cmp_lg_u32 s{S_PRE}, 0", "s_cbranch_scc1 cnt_pre", "s_waitcnt vmcnt(0)", "s_branch cnt_ok", "cnt_pre:", "s_waitcnt vmcnt(3)", "cnt_ok:"]
    L += [f"v_mov_b32 v{VCNT}, v{VCNT2}" if ODD else f"v_lshrrev_b32 v{VCNT}, 1, v{VCNT2}", f"v_add_u32 v{VCNT}, -1, v{VCNT}", f"s_mov_b64 vcc, s[{S_QM}:{S_QM + 1}]",
          f"v_cndmask_b32 v{VCNT}, -1, v{VCNT}, vcc"]
    # pieces of the next tile go to the other buffer (the last tile reloads itself: harmless)
    L += [f"s_add_u32 s{S_DROW}, s{S_TABS}, 1", f"s_min_u32 s{S_DROW}, s{S_DROW}, s{S_TLAST}", f"s_sub_u32 s{S_DLDS}, {TILE_B}, s{S_BUF}",
          f"s_add_u32 s{S_DLDS}, s{S_DLDS}, %[wdma]", f"s_mov_b32 s{S_NP}, 5"]
    dma_tile_setup(L)
    L += [f"v_add_u32 v{VLB}, s{S_BUF}, %[lb]", f"s_mov_b32 s{S_REM}, s{S_NCH}"]
    if ROT == 2 and ROT_SHIFT < 0:
        L += [f"s_cmp_gt_u32 s{S_NCH}, {ROT_LONG}", f"s_cselect_b32 s{S_SH}, 1, 0"]
    # the routine of chunk 0's entry buffer
    L += [f"s_mov_b64 s[{S_CTL}:{S_CTL + 1}], s[{S_CTLA[0]}:{S_CTLA[0] + 1}]", f"s_cmp_eq_u32 s{S_BASE}, 0", "s_cbranch_scc1 ctl_set",
          f"s_mov_b64 s[{S_CTL}:{S_CTL + 1}], s[{S_CTLA[1]}:{S_CTLA[1] + 1}]", f"s_cmp_eq_u32 s{S_BASE}, 1", "s_cbranch_scc1 ctl_set",
          f"s_mov_b64 s[{S_CTL}:{S_CTL + 1}], s[{S_CTLA[2]}:{S_CTLA[2] + 1}]", "ctl_set:"]
    if LATEBAR:
        tile_barrier(L)
    stamp(L, ST_R + 4)
    if STAMPS:
        stamp_pairs_clear(L)
    L += [f"s_cmp_eq_u32 s{S_NCH}, 0", "s_cbranch_scc1 tile_done", "s_branch enter_0"]
    bodies(L)
    stubs(L)
    chunk_routines(L, pattern)
    L += ["tile_done:", "s_waitcnt lgkmcnt(0)"]
    if STAMPS:
        stamp_lds_sample_close(L)
        stamp(L, ST_R + 5)   # (leaves the tile's lane in m0)
        for acc, rec in ((ST_SVM, ST_R + 6), (ST_SL, ST_R + 7), (ST_SOWN, ST_R9), (ST_SP4, ST_R10)):
            L.append(f"v_writelane_b32 v{rec}, s{acc}, m0")
        L.append(f"v_writelane_b32 v{ST_R8}, s{S_NCH}, m0")
    # pieces the chunks did not issue (fewer than five chunks)
    L += [f"s_cmp_eq_u32 s{S_NP}, 0", "s_cbranch_scc1 pieces_done", "more_pieces:"]
    dma_piece(L)
    L += [f"s_cmp_lg_u32 s{S_NP}, 0", "s_cbranch_scc1 more_pieces", "pieces_done:"]
    # the next tile starts in the buffer after this tile's last chunk when it was preloaded
    L += [f"s_add_u32 s{S_BASE}, s{S_BASE}, s{S_NCH}", "base_mod:", f"s_cmp_lt_u32 s{S_BASE}, 3", "s_cbranch_scc1 base_ok", f"s_sub_u32 s{S_BASE}, s{S_BASE}, 3",
          "s_branch base_mod", "base_ok:", f"s_mov_b32 s{S_PRE}, s{S_LINK}"]
    L += [f"s_add_u32 s{S_T}, s{S_T}, 1", f"s_add_u32 s{S_TABS}, s{S_TABS}, 1",
          f"s_add_u32 s{S_STP}, s{S_STP}, 0x200", f"s_addc_u32 s{S_STP + 1}, s{S_STP + 1}, 0",
          f"s_sub_u32 s{S_BUF}, {TILE_B}, s{S_BUF}", f"s_cmp_lt_u32 s{S_T}, s{S_NT}", "s_cbranch_scc1 tile_top"]
    L += ["s_waitcnt vmcnt(0)"]
    return L


def uniq_labels(L):
    """inline asm may be emitted more than once per module: named labels get the %= suffix"""
    names = set()
    for ln in L:
        m = re.match(r"^([a-z][a-z0-9_]*):$", ln)
        if m:
            names.add(m.group(1))
    pat = re.compile(r"\b(" + "|".join(sorted(names, key=len, reverse=True)) + r")\b")
    return [pat.sub(lambda m: m.group(1) + "_%=", ln) for ln in L]


def set_depth_regs():
    global A, B, W
    if DEPTH == 3:
        top = ACC + 4 * RG
        A = [[20, 21], [22, 23], [50, 51]]
        B = [24, 26, 52]
        W = [[32, 36], [40, 44], [top, top + 4]]
    assert not (DEPTH == 3 and B64)


def clobbers():
    if F64:
        return [f"v{i}" for i in range(10, 78)] + [f"s{i}" for i in range(36, 98)] + ["memory", "scc", "m0", "vcc"]
    v = [f"v{i}" for i in range(10, 50)]
    if DEPTH == 3:
        v += [f"v{i}" for i in range(50, 54)] + [f"v{i}" for i in range(ACC + 4 * RG, ACC + 4 * RG + 8)]
    s = [f"s{i}" for i in range(36, 98)]
    if STAMPS:   # (v[120:127] and v[52:55] are outputs of the block, not clobbers)
        s += [f"s{i}" for i in range(98, 102)]
    return v + s + ["memory", "scc", "m0", "vcc"]


def main():
    global RG, STAMPS, B64, FMAC, ILV, DEPTH
    here = os.path.dirname(os.path.abspath(__file__))
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(here, "..", "single-algebra_amd", "csrc", "spmm_dq2_gen.h")
    with open(path, "w") as out:
        out.write("// generated by tools/gen_spmm_dq2.py -- do not edit; the generator documents the structure\n")
        out.write(f"#define DQ2_ACC_BASE {ACC}\n#define DQ2_TILE_BYTES {TILE_B}\n#define DQ2_ODD_STEPS {1 if ODD else 0}\n")
        if STAMPS:
            out.write("#define DQ2_STAMPS 1\n")
        # the pattern variant (quirk Q3) differs from the plain one by three instructions in each chunk routine: one macro body with
        # a parameter at those places instead of two copies of the loop
        out.write('#define DQ2_PAT_LINES "s_nop 0\\n" "v_cmp_neq_f32 vcc, 0, v%d\\n" "v_cndmask_b32 v%d, 0, 1.0, vcc\\n"\n' % (ECUR[1], ECUR[1]))
        for rg in (8, 16):   # 512-row and 1024-row blocks
            RG = rg
            set_depth_regs()
            _uid[0] = 0
            plain = uniq_labels(body(False))
            _uid[0] = 0
            pat = uniq_labels(body(True))
            out.write(f"#define DQ2_MAIN_ASM_{rg}_(PAT) \\\n")
            i = j = 0
            marks = 0
            while i < len(plain):
                if plain[i] == pat[j]:
                    out.write(f'  "{plain[i]}\\n" \\\n')
                    i += 1
                    j += 1
                else:   # the pattern variant's three extra lines stand here
                    assert pat[j] == "s_nop 0" and pat[j + 1].startswith("v_cmp_neq_f32") and pat[j + 2].startswith("v_cndmask_b32"), pat[j:j + 3]
                    out.write("  PAT \\\n")
                    j += 3
                    marks += 1
            assert j == len(pat) and marks == 3, (j, len(pat), marks)
            out.write("\n")
            out.write(f'#define DQ2_MAIN_ASM_{rg} DQ2_MAIN_ASM_{rg}_("")\n#define DQ2_MAIN_ASM_{rg}_PAT DQ2_MAIN_ASM_{rg}_(DQ2_PAT_LINES)\n')
            print(f"wrote DQ2_MAIN_ASM_{rg}(_PAT): {len(plain)} lines")
            out.write(f"#define DQ2_MAIN_CLOBBERS_{rg} " + ", ".join(f'"{c}"' for c in clobbers()) + "\n")
        # the f64 sweep: four row slots per lane group (blocks of <= 256 rows, the f64 quad format's), accumulators from v80;
        # the f32 experiment switches (B64, FMAC, ILV, DEPTH, stamps) do not apply to it
        keep = (STAMPS, B64, FMAC, ILV, DEPTH)
        STAMPS, B64, FMAC, ILV, DEPTH = False, False, False, False, 2
        RG = 4
        set_f64(True)
        _uid[0] = 0
        L = uniq_labels(body(False))
        out.write("#define DQ2_F64_ACC_BASE 80\n#define DQ2_MAIN_ASM_F64 \\\n")
        for ln in L:
            out.write(f'  "{ln}\\n" \\\n')
        out.write("\n")
        out.write("#define DQ2_MAIN_CLOBBERS_F64 " + ", ".join(f'"{c}"' for c in clobbers()) + "\n")
        print(f"wrote DQ2_MAIN_ASM_F64: {len(L)} lines")
        set_f64(False)
        STAMPS, B64, FMAC, ILV, DEPTH = keep


if __name__ == "__main__":
    main()
