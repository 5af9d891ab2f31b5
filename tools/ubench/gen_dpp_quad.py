#!/usr/bin/env python3
"""Emits dpp_quad_gen.h: the inline-asm body of the DPP-fed quad sweep micro-benchmark (dpp_quad.hip).

One wave = 4 lane groups of 16 lanes; a group walks one output row at a time, a lane holds 4 of the 64 panel
columns.  Entries {u32 tile byte offset, f32 value} are NOT staged in LDS: a wave-wide global_load_dwordx2 puts
16 consecutive steps of every group into one VGPR pair (lane (g, i) = step i of group g), and step s is broadcast
to the 16 lanes of its group by DPP row_newbcast:s, folded into the address add (v_add_u32_dpp) and a v_mov_b32_dpp
for the value.  The accumulators sit in v[ACC + 4 j ..] and are addressed through the VGPR index mode
(s_set_gpr_idx_on, M0[7:0] = 4 j) because the row j a step belongs to is wave-uniform but not static.
Steps come in groups of two that share a row (rows are padded to an even step count).

Variants (name -> options): pipeline depth in groups, groups of the next chunk issued before the last groups of this
one are consumed (xchunk), and ablations (no LDS reads / no FMAs) that show which unit bounds the loop.
"""
import sys

ACC = 64
EB = [(10, 11), (12, 13), (14, 15)]
VDESC = 16
A = [[20, 21], [22, 23], [24, 25]]
B = [26, 28, 30]
W = [[32, 36], [40, 44], [48, 52]]


class Gen:
    def __init__(self, fma="pk", rg=8, depth=2, xchunk=False, nolds=False, nofma=False, alwayson=False):
        self.fma, self.rg, self.depth, self.xchunk, self.nolds, self.nofma = fma, rg, depth, xchunk, nolds, nofma
        self.alwayson = alwayson   # index mode stays on: are DPP instructions exempt from the destination index?
        self.L = []
        self.slot = 0   # next register set for an issued group (round robin over `depth`)
        self.cons = 0   # register set of the next group to consume

    def grp_a(self, k, e):
        p = self.slot
        self.slot = (self.slot + 1) % self.depth
        ex, ey = e
        L = self.L
        for t in range(2):
            L.append(f"v_add_u32_dpp v{A[p][t]}, v{ex}, %[lb] row_newbcast:{2 * k + t} row_mask:0xf bank_mask:0xf")
        if not self.nolds:
            for t in range(2):
                L.append(f"ds_read_b128 v[{W[p][t]}:{W[p][t] + 3}], v{A[p][t]}")
        for t in range(2):
            L.append(f"v_mov_b32_dpp v{B[p] + t}, v{ey} row_newbcast:{2 * k + t} row_mask:0xf bank_mask:0xf")

    def grp_b(self, k, wait):
        p = self.cons
        self.cons = (self.cons + 1) % self.depth
        L = self.L
        sd = 40 if k < 4 else 41
        if not self.nolds:
            L.append(f"s_waitcnt lgkmcnt({wait})")
        if not self.nofma:
            L.append(f"s_set_gpr_idx_idx s{sd}" if self.alwayson else f"s_set_gpr_idx_on s{sd}, gpr_idx(SRC2,DST)")
            for t in range(2):
                w = W[p][t]
                if self.fma == "pk":
                    sel = "op_sel_hi:[1,0,1]" if t == 0 else "op_sel:[0,1,0] op_sel_hi:[1,1,1]"
                    L.append(f"v_pk_fma_f32 v[{ACC}:{ACC + 1}], v[{w}:{w + 1}], v[{B[p]}:{B[p] + 1}], v[{ACC}:{ACC + 1}] {sel}")
                    L.append(f"v_pk_fma_f32 v[{ACC + 2}:{ACC + 3}], v[{w + 2}:{w + 3}], v[{B[p]}:{B[p] + 1}], v[{ACC + 2}:{ACC + 3}] {sel}")
                else:
                    for c in range(4):
                        L.append(f"v_fma_f32 v{ACC + c}, v{w + c}, v{B[p] + t}, v{ACC + c}")
            if not self.alwayson:
                L.append("s_set_gpr_idx_off")
        if k not in (3, 7):
            L.append(f"s_lshr_b32 s{sd}, s{sd}, 8")

    def chunk(self, buf):
        """16 steps from entry buffer `buf`; its reload (3 chunks ahead) is issued after its last DPP read"""
        L, d = self.L, self.depth
        e, en = EB[buf], EB[(buf + 1) % 3]
        if not self.xchunk:
            L.append("s_waitcnt vmcnt(2)")
        L += ["s_add_u32 s47, s46, 1", f"v_readlane_b32 s40, v{VDESC}, s46", f"v_readlane_b32 s41, v{VDESC}, s47"]
        if self.alwayson:
            L.append("s_set_gpr_idx_on s40, gpr_idx(SRC2,DST)")
        if not self.xchunk:
            for k in range(d):
                self.grp_a(k, e)
        for k in range(8):
            if self.xchunk:
                self.grp_b(k, 2 * (d - 1))
                if k + d < 8:
                    self.grp_a(k + d, e)
                    if k + d == 7:
                        L.append(f"global_load_dwordx2 v[{e[0]}:{e[1]}], %[eoff], s[44:45] offset:1536")
                        L.append("s_waitcnt vmcnt(2)")       # the next chunk's buffer
                else:
                    self.grp_a(k + d - 8, en)
            else:
                ahead = min(d, 8 - k) - 1
                self.grp_b(k, 2 * ahead)
                if k + d < 8:
                    self.grp_a(k + d, e)
                    if k + d == 7:
                        L.append(f"global_load_dwordx2 v[{e[0]}:{e[1]}], %[eoff], s[44:45] offset:1536")
        if self.alwayson:
            L.append("s_set_gpr_idx_off")
        L += ["s_add_u32 s44, s44, 0x200", "s_addc_u32 s45, s45, 0", "s_add_u32 s46, s46, 2"]

    def body(self):
        L, rg = self.L, self.rg
        L += ["s_mov_b32 s40, 0", "s_set_gpr_idx_on s40, gpr_idx(DST)", "1:", f"v_mov_b32 v{ACC}, 0", "s_add_u32 s40, s40, 1",
              "s_set_gpr_idx_idx s40", f"s_cmp_lt_u32 s40, {4 * rg}", "s_cbranch_scc1 1b", "s_set_gpr_idx_off"]
        L += ["s_mov_b64 s[44:45], %[ptr]", "s_mov_b64 s[48:49], %[dptr]"]
        for i, e in enumerate(EB):
            L.append(f"global_load_dwordx2 v[{e[0]}:{e[1]}], %[eoff], s[44:45] offset:{512 * i}")
        if self.xchunk:
            L.append("s_waitcnt vmcnt(2)")
            for k in range(self.depth):
                self.grp_a(k, EB[0])
        L.append("4:")                                   # block of 30 chunks sharing one descriptor load
        L += [f"global_load_dword v{VDESC}, %[doff], s[48:49]", "s_add_u32 s48, s48, 0x100", "s_addc_u32 s49, s49, 0", "s_waitcnt vmcnt(0)",
              "s_mov_b32 s46, 0", "s_mov_b32 s42, 10", "3:"]
        slot0, cons0 = self.slot, self.cons
        for buf in range(3):
            self.chunk(buf)
        assert (self.slot, self.cons) == (slot0, cons0) or not self.xchunk or self.depth == 2, "register rotation must close over the unrolled loop"
        L += ["s_sub_u32 s42, s42, 1", "s_cmp_lg_u32 s42, 0", "s_cbranch_scc1 3b"]
        L += ["s_sub_u32 %[n], %[n], 1", "s_cmp_lg_u32 %[n], 0", "s_cbranch_scc1 4b"]
        L += ["s_waitcnt vmcnt(0)", "s_waitcnt lgkmcnt(0)"]
        L += ["s_mov_b32 s40, 0", "2:", "s_set_gpr_idx_on s40, gpr_idx(SRC0)", f"v_mov_b32 %[tmp], v{ACC}", "s_set_gpr_idx_off",
              "global_store_dword %[ooff], %[tmp], %[optr]", "v_add_u32 %[ooff], 0x100, %[ooff]", "s_add_u32 s40, s40, 1",
              f"s_cmp_lt_u32 s40, {4 * rg}", "s_cbranch_scc1 2b", "s_waitcnt vmcnt(0)"]
        return L


def clobbers(rg):
    v = [f"v{i}" for i in range(10, 56)] + [f"v{i}" for i in range(ACC, ACC + 4 * rg)]
    s = [f"s{i}" for i in range(40, 50)]
    return v + s + ["memory", "scc", "m0"]


VARIANTS = (
    ("PK8", dict()),
    ("F8", dict(fma="fma")),
    ("PK16", dict(rg=16)),
    ("PK8D3", dict(depth=3)),
    ("PK8X", dict(xchunk=True)),
    ("PK8XD3", dict(xchunk=True, depth=3)),
    ("PK8_NOLDS", dict(nolds=True)),
    ("PK8_NOFMA", dict(nofma=True)),
    ("PK8_ON", dict(alwayson=True)),
)


def main():
    out = open(sys.argv[1] if len(sys.argv) > 1 else "dpp_quad_gen.h", "w")
    out.write("// generated by gen_dpp_quad.py -- do not edit\n")
    for name, opts in VARIANTS:
        g = Gen(**opts)
        out.write(f"#define DQ_ASM_{name} \\\n")
        for ln in g.body():
            out.write(f'  "{ln}\\n" \\\n')
        out.write("\n")
        out.write(f"#define DQ_CLOB_{name} " + ", ".join(f'"{c}"' for c in clobbers(g.rg)) + "\n\n")
    out.close()


if __name__ == "__main__":
    main()
