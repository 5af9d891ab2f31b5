// DPP-fed quad sweep: the default f32 sparse x dense-panel sweep for panels of 64 columns (two passes for 128).
//
// Same operator format as spmm_tiled.hip's quad sweep (row blocks of <= 512 or 1024 rows, interleaved column tiles of 320
// panel rows, four rows of a quad advancing in lockstep, entries {u32 tile byte offset, f32 value} stored step by step),
// read differently.  Measured on the staged-entry kernel: per step it pays two dependent LDS round trips
// (entry, then panel row) and ran at 3.5 cycles per entry slot per CU.  Here
//   * entries never touch LDS: a wave loads 16 steps of its four rows with ONE coalesced 512-byte global_load_dwordx2
//     (lane (g, i) = step i of lane group g) and step s reaches the 16 lanes of its group through DPP row_newbcast:s,
//     folded into the address add (v_add_u32_dpp) and one v_mov_b32_dpp for the value;
//   * the panel tile is double-buffered in the whole 160 KiB of LDS (2 x 80 KiB) and filled by LDS-DMA
//     (global_load_lds_dwordx4) while the previous tile is being used: no register staging, no ds_write phase;
//   * the accumulators of a wave's 64 (32) rows sit in v[56..119] (v[56..87]); the main loop exists once per row slot with
//     the slot's registers written into its FMAs, and which slot a step belongs to is control flow: a scalar counter of
//     the two-step groups the current quad still has in this tile, one conditional branch per group, a stub that fetches
//     the next quad's count from the format's own `steps` table.  A wave's stream in a tile ends exactly at its last step.
//     (Round 2 addressed the accumulators through the VGPR index mode from a descriptor byte per two steps: two more scalar
//     instructions per group, a descriptor table, 7 % of the steps executed past stream ends into a trash slot.)
// The main loop is inline assembly (fixed registers, counted waits), written by tools/gen_spmm_dq2.py into
// spmm_dq2_gen.h; prologue and epilogue are HIP C++.
//
// Replaces: the sweeps Y = Ac Omega / Z = Ac^T Y inside single_svdlib::randomized::randomized_svd (call sites
// /root/reference/src/dimred/pca/sparse/mod.rs:170-180, sparse_masked/mod.rs:341-351) and the projection of transform.
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include <algorithm>

#include "spmm_dq.h"
#ifdef DQ2_GEN_H   // experiment variants of the generated main loop (tools/dq2_variant.sh)
#include DQ2_GEN_H
#else
#include "spmm_dq2_gen.h"
#endif

namespace sapca {
namespace k {

namespace {

constexpr int WAVE = 64;
constexpr int DQ_WAVES = 16, DQ_THREADS = DQ_WAVES * WAVE;
constexpr int DQ_BLOCK_QUADS = 256;           // stride of the per-chunk quad step table (= Q_BLOCK_QUADS of spmm_tiled.hip)
constexpr int DQ_TILE_BYTES = DQ2_TILE_BYTES;
constexpr int DQ_LDS = 2 * DQ_TILE_BYTES;
static_assert((DQ2_ODD_STEPS != 0) == kOddSteps, "the generated main loop and the format disagree on odd step counts (DQ2_ODD / -DSAPCA_ODD_STEPS)");
constexpr int DQ_OFF_UNIT = kOddSteps ? 4 : 8;   // entries per unit of a stream's offset in the info table
static_assert(DQ2_ACC_BASE == 56, "the accumulator operands below are written out for row slots starting at v56");

__host__ __device__ inline int dq_first(int wave, int nquads) { return wave * nquads / DQ_WAVES; }
// Which of the sixteen streams of a row block a hardware wave walks.  Waves w, w + 4, w + 8, w + 12 share a SIMD, and a
// block's quads are dealt to the streams s as [s nq / 16, (s + 1) nq / 16): when nq is not a multiple of 16 the longer streams
// recur with a period that divides 16 (every fourth one for nq = 16 a + 4) -- walked by wave s they would all land on one SIMD
// (measured, round 5: SIMD 3 carried 8 % more chunks than the others at C2).  Walked in transposed order they spread over
// the four SIMDs, and the longer ones (s & 3 == 3 first) go to the OLDEST wave of each SIMD, which the issue arbiter serves
// first (profiles/r05_dq_stamps_c2.txt).
__host__ __device__ inline int stream_of_wave(int hw_wave) { return (hw_wave & 3) * 4 + 3 - (hw_wave >> 2); }

// Which (row block, tile range) a workgroup takes when the tile range of an operator is split over workgroups (A^T of a tall
// matrix: few row blocks, each streaming the whole m x 64 panel).  Workgroups are dealt to the eight XCDs round-robin
// (workgroup i runs on XCD i mod 8) and every XCD has its own L2: with (block, range) = (i / nsplit, i mod nsplit) the
// workgroups that walk ONE tile range -- and could share its refills -- sit on eight different L2s, and at C5 (49 row blocks,
// a 1 GB panel) 15.5 of the 25 GB of refills a pass requests leave the L2 (profiles/r05_c5_traffic.txt).  Here XCD x takes a
// contiguous run of the list ordered by (tile range, row block): its 32 workgroups walk one or two tile ranges together.
__device__ inline int xcd_run_index(int i, int total) {
  const int x = i & 7, y = i >> 3, q = total >> 3, r = total & 7;
  return x * q + min(x, r) + y;   // XCD x holds q workgroups, one more for x < total mod 8
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float v8f __attribute__((ext_vector_type(8)));
typedef unsigned v8u __attribute__((ext_vector_type(8)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// One wave per (row block, tile): lane w < 16 computes the entry offset and chunk count of sweep wave w's stream, and the rank
// of its step count among the four streams walked on one SIMD (stream_of_wave: streams 4 i .. 4 i + 3; the longest gets 3) --
// the sweep's issue priority for that tile under DQ2_PRIO.
__global__ void __launch_bounds__(256)
dq_info_kernel(const int32_t* __restrict__ blk_row0, int nct, const int64_t* __restrict__ chunk_off,
               const uint32_t* __restrict__ wave_off, const uint16_t* __restrict__ steps, uint32_t* __restrict__ info, int64_t nchunks) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wpb = blockDim.x / WAVE;
  for (int64_t cidx = (int64_t)blockIdx.x * wpb + threadIdx.x / WAVE; cidx < nchunks; cidx += (int64_t)gridDim.x * wpb) {
    const int rb = (int)(cidx / nct), t = (int)(cidx % nct);
    const int nrows = blk_row0[rb + 1] - blk_row0[rb];
    const int nquads = (nrows + 3) / 4;
    const int wave = lane & (DQ_WAVES - 1);
    const int quad0 = dq_first(wave, nquads), my_quads = dq_first(wave + 1, nquads) - quad0;   // <= 16
    int n = 0;
    if (lane < DQ_WAVES)
      for (int j = 0; j < my_quads; ++j) n += (int)steps[cidx * DQ_BLOCK_QUADS + quad0 + j];
    int rank = 0;
#pragma unroll
    for (int d = 1; d < 4; ++d) {   // (streams 4 i .. 4 i + 3 are walked by the hardware waves i, i + 4, i + 8, i + 12: one SIMD)
      const int other = (wave & 12) | ((wave + d) & 3);
      const int no = __shfl(n, other);
      rank += (no < n || (no == n && other < wave)) ? 1 : 0;
    }
    if (lane < DQ_WAVES) {
      const int64_t s = chunk_off[cidx] + (int64_t)wave_off[cidx * DQ_WAVES + wave];
      uint32_t* w = info + (((int64_t)rb * DQ_WAVES + wave) * nct + t) * 2;
      w[0] = (uint32_t)(s / DQ_OFF_UNIT);
      w[1] = (uint32_t)((n + 15) / 16) | ((uint32_t)rank << 16);
    }
  }
}

template <int RG, bool PAT>   // row slots per lane group: 8 (blocks of <= 512 rows) or 16 (<= 1024); pattern mode (quirk Q3)
__global__ void __launch_bounds__(DQ_THREADS)
spmm_dq_kernel(const int32_t* __restrict__ blk_row0, const uint32_t* __restrict__ perm, int nct,
                const uint64_t* __restrict__ ent, const uint16_t* __restrict__ steps, const uint32_t* __restrict__ info,
                int64_t panel_rows, const float* __restrict__ X, int ldx, int nsplit, int tiles_per_split,
                float* __restrict__ out, int64_t out_rows_total, int ldo, int ncols, const float* __restrict__ cvec, int rb0
#ifdef DQ2_STAMPS
                , uint32_t* __restrict__ stamps, int stamp_wgs
#endif
                ) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int rb = rb0 + blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  if (nsplit > 1) {
    const int nrb_here = (int)gridDim.x / nsplit, j = xcd_run_index((int)blockIdx.x, (int)gridDim.x);
    sp = j / nrb_here;
    rb = rb0 + j % nrb_here;
  }
  const int ct0 = sp * tiles_per_split, ct1 = min(nct, ct0 + tiles_per_split);
  const int wave = stream_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x / WAVE)), lane = threadIdx.x & (WAVE - 1);   // (the stream this wave walks)
  const int g = lane / 16, q = lane % 16;
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  const int nquads = (nrows + 3) / 4;
  const int quad0 = dq_first(wave, nquads), my_quads = dq_first(wave + 1, nquads) - quad0;   // <= RG
  const int my_rows = min(nrows, 4 * (quad0 + my_quads)) - 4 * quad0;

  v8f a0, a1, a2, a3, a4, a5, a6, a7;   // row slots 2i, 2i+1 (RG = 8: a0..a3)
#ifdef DQ2_STAMPS
  v8u st8 = v8u(0u);   // R0..R7 of tools/gen_spmm_dq2.py's stamp mode, lane = tile & 63
  v4u st4 = v4u(0u);   // {-, R8, R10, -}
#define DQ2_STAMP_OUTS , "=&{v[120:127]}"(st8), "=&{v[52:55]}"(st4)
#else
#define DQ2_STAMP_OUTS
#endif
  if (ct0 < ct1) {
    const unsigned lds_base = (unsigned)(size_t)lds;
    unsigned lb = lds_base + q * 16, eoff = q * 32 + g * 8, l8 = lane * 8, l2 = lane * 2, l2c = min(lane, 15) * 2, col16 = q * 16;
    unsigned rowb0 = (unsigned)(4 * (wave * 5) + g) * (unsigned)nct;
    const unsigned wdma = __builtin_amdgcn_readfirstlane(lds_base + wave * 5 * 1024);
    const unsigned myq = __builtin_amdgcn_readfirstlane((unsigned)my_quads);
    auto uniform64 = [](unsigned long long p) {
      return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) << 32) |
             (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)p);
    };
    const unsigned long long ip = uniform64((unsigned long long)(info + (((int64_t)rb * DQ_WAVES + wave) * nct + ct0) * 2));
    const unsigned long long stp = uniform64((unsigned long long)(steps + ((int64_t)rb * nct + ct0) * DQ_BLOCK_QUADS + quad0));
    const unsigned xlo = (unsigned)(uintptr_t)X, xhi = (unsigned)((uintptr_t)X >> 32);
    const unsigned voff = (unsigned)g * (unsigned)nct * ((unsigned)ldx * 4u) + col16;   // lane part of an LDS-DMA piece's source address
    const unsigned stride = (unsigned)ldx * 4u, prm1 = (unsigned)(panel_rows - 1), ntiles = (unsigned)(ct1 - ct0), t0 = (unsigned)ct0;
#define DQ2_INPUTS                                                                                                      \
  [ent] "s"(ent), [stp] "s"(stp), [xlo] "s"(xlo), [xhi] "s"(xhi), [info] "s"(ip), [lb] "v"(lb), [eoff] "v"(eoff), [l8] "v"(l8), [l2] "v"(l2), [l2c] "v"(l2c), \
      [col16] "v"(col16), [rowb0] "v"(rowb0), [nct] "s"(nct), [stride] "s"(stride), [prm1] "s"(prm1),                     \
      [ntiles] "s"(ntiles), [t0] "s"(t0), [wdma] "s"(wdma), [myq] "s"(myq), [voff] "v"(voff)
    if constexpr (RG == 8) {
      if constexpr (PAT)
        asm volatile(DQ2_MAIN_ASM_8_PAT
                     : "=&{v[56:63]}"(a0), "=&{v[64:71]}"(a1), "=&{v[72:79]}"(a2), "=&{v[80:87]}"(a3) DQ2_STAMP_OUTS
                     : DQ2_INPUTS
                     : DQ2_MAIN_CLOBBERS_8);
      else
        asm volatile(DQ2_MAIN_ASM_8
                     : "=&{v[56:63]}"(a0), "=&{v[64:71]}"(a1), "=&{v[72:79]}"(a2), "=&{v[80:87]}"(a3) DQ2_STAMP_OUTS
                     : DQ2_INPUTS
                     : DQ2_MAIN_CLOBBERS_8);
      a4 = a5 = a6 = a7 = v8f(0.f);
    } else {
      if constexpr (PAT)
        asm volatile(DQ2_MAIN_ASM_16_PAT
                     : "=&{v[56:63]}"(a0), "=&{v[64:71]}"(a1), "=&{v[72:79]}"(a2), "=&{v[80:87]}"(a3), "=&{v[88:95]}"(a4),
                       "=&{v[96:103]}"(a5), "=&{v[104:111]}"(a6), "=&{v[112:119]}"(a7) DQ2_STAMP_OUTS
                     : DQ2_INPUTS
                     : DQ2_MAIN_CLOBBERS_16);
      else
        asm volatile(DQ2_MAIN_ASM_16
                     : "=&{v[56:63]}"(a0), "=&{v[64:71]}"(a1), "=&{v[72:79]}"(a2), "=&{v[80:87]}"(a3), "=&{v[88:95]}"(a4),
                       "=&{v[96:103]}"(a5), "=&{v[104:111]}"(a6), "=&{v[112:119]}"(a7) DQ2_STAMP_OUTS
                     : DQ2_INPUTS
                     : DQ2_MAIN_CLOBBERS_16);
    }
#undef DQ2_INPUTS
  } else {
    a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = v8f(0.f);
  }
  const v8f acc2[8] = {a0, a1, a2, a3, a4, a5, a6, a7};
#ifdef DQ2_STAMPS
  if (stamps && (int)blockIdx.x < stamp_wgs) {   // [workgroup][wave][12 records][64 tiles]
    uint32_t* w = stamps + (((size_t)blockIdx.x * DQ_WAVES + threadIdx.x / WAVE) * 12) * WAVE + lane;   // (by hardware wave)
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j * WAVE] = st8[j];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[(8 + j) * WAVE] = st4[j];
  }
#endif

  // lane (g, q) holds columns 4q..4q+3 of row 4j+g of its wave
  float* dst_base = out + (nsplit > 1 ? (int64_t)sp * out_rows_total * ldo : 0);
  const int col = q * 4;
  float cv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) cv[i] = (cvec && nsplit == 1 && col + i < ncols) ? cvec[col + i] : 0.f;
  const bool vec16 = (ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(dst_base) & 15) == 0);
#pragma unroll
  for (int j = 0; j < RG; ++j) {
    const v4f accj = (j & 1) ? __builtin_shufflevector(acc2[j / 2], acc2[j / 2], 4, 5, 6, 7) : __builtin_shufflevector(acc2[j / 2], acc2[j / 2], 0, 1, 2, 3);
    const int r = 4 * j + g;
    if (r < my_rows) {
      const int64_t slot = (int64_t)row0 + 4 * quad0 + r;   // slot position -> output row
      float* y = dst_base + (perm ? (int64_t)perm[slot] : slot) * ldo + col;
      if (col + 3 < ncols) {
        v4f o = accj;
        o.x -= cv[0]; o.y -= cv[1]; o.z -= cv[2]; o.w -= cv[3];
        if (vec16) *reinterpret_cast<v4f*>(y) = o;
        else *reinterpret_cast<v4f_a4*>(y) = o;   // the projection's leading dimension is k: rows are only 4-byte aligned
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (col + i < ncols) y[i] = accj[i] - cv[i];
      }
    }
  }
}


// The same sweep for T = f64 (generic over T like the reference, /root/reference/src/dimred/pca/sparse/mod.rs:33-36): the f64
// quad format of spmm_tiled.hip (16-byte entries {u32 tile byte offset, pad, f64 value}, blocks of <= 256 rows = four row
// slots per lane group, tiles of 160 panel rows of 512 bytes), a lane holds columns 2q, 2q+1, 32+2q, 32+2q+1 of its row
// (two ds_read_b128 256 bytes apart), four v_fma_f64 per step, the value broadcast as one 64-bit DPP move; an LDS-DMA piece
// is two panel rows.  Main loop: DQ2_MAIN_ASM_F64 (tools/gen_spmm_dq2.py).
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));
static_assert(DQ2_F64_ACC_BASE == 80, "the accumulator operands below are written out for row slots starting at v80");
constexpr int DQ_F64_RG = 4;

__global__ void __launch_bounds__(DQ_THREADS)
spmm_dq_f64_kernel(const int32_t* __restrict__ blk_row0, const uint32_t* __restrict__ perm, int nct, const void* __restrict__ ent,
                   const uint16_t* __restrict__ steps, const uint32_t* __restrict__ info, int64_t panel_rows, const double* __restrict__ X,
                   int ldx, int nsplit, int tiles_per_split, double* __restrict__ out, int64_t out_rows_total, int ldo, int ncols,
                   const double* __restrict__ cvec, int rb0) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  int rb = rb0 + blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  if (nsplit > 1) {
    const int nrb_here = (int)gridDim.x / nsplit, j = xcd_run_index((int)blockIdx.x, (int)gridDim.x);
    sp = j / nrb_here;
    rb = rb0 + j % nrb_here;
  }
  const int ct0 = sp * tiles_per_split, ct1 = min(nct, ct0 + tiles_per_split);
  const int wave = stream_of_wave(__builtin_amdgcn_readfirstlane(threadIdx.x / WAVE)), lane = threadIdx.x & (WAVE - 1);   // (the stream this wave walks)
  const int g = lane / 16, q = lane % 16;
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  const int nquads = (nrows + 3) / 4;
  const int quad0 = dq_first(wave, nquads), my_quads = dq_first(wave + 1, nquads) - quad0;   // <= 4
  const int my_rows = min(nrows, 4 * (quad0 + my_quads)) - 4 * quad0;

  v4d a0, a1, a2, a3, a4, a5, a6, a7;   // slot j: a(2j) = columns 2q, 2q+1 ; a(2j+1) = columns 32+2q, 32+2q+1
  if (ct0 < ct1) {
    const unsigned lds_base = (unsigned)(size_t)lds;
    const int half = lane / 32;   // which of a piece's two panel rows this lane brings
    unsigned lb = lds_base + q * 16, eoff = q * 64 + g * 16, l8 = lane * 8, l2 = lane * 2, l2c = min(lane, 15) * 2, col16 = (lane & 31) * 16;
    unsigned rowb0 = (unsigned)(2 * (wave * 5) + half) * (unsigned)nct;
    const unsigned wdma = __builtin_amdgcn_readfirstlane(lds_base + wave * 5 * 1024);
    const unsigned myq = __builtin_amdgcn_readfirstlane((unsigned)my_quads);
    auto uniform64 = [](unsigned long long p) {
      return ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(p >> 32)) << 32) |
             (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)p);
    };
    const unsigned long long ip = uniform64((unsigned long long)(info + (((int64_t)rb * DQ_WAVES + wave) * nct + ct0) * 2));
    const unsigned long long stp = uniform64((unsigned long long)(steps + ((int64_t)rb * nct + ct0) * DQ_BLOCK_QUADS + quad0));
    const unsigned xlo = (unsigned)(uintptr_t)X, xhi = (unsigned)((uintptr_t)X >> 32);
    const unsigned stride = (unsigned)ldx * 8u;
    const unsigned voff = (unsigned)half * (unsigned)nct * stride + col16;   // lane part of an LDS-DMA piece's source address
    const unsigned prm1 = (unsigned)(panel_rows - 1), ntiles = (unsigned)(ct1 - ct0), t0 = (unsigned)ct0;
    asm volatile(DQ2_MAIN_ASM_F64
                 : "=&{v[80:87]}"(a0), "=&{v[88:95]}"(a2), "=&{v[96:103]}"(a4), "=&{v[104:111]}"(a6)
                 : [ent] "s"(ent), [stp] "s"(stp), [xlo] "s"(xlo), [xhi] "s"(xhi), [info] "s"(ip), [lb] "v"(lb), [eoff] "v"(eoff), [l8] "v"(l8),
                   [l2] "v"(l2), [l2c] "v"(l2c), [col16] "v"(col16), [rowb0] "v"(rowb0), [nct] "s"(nct), [stride] "s"(stride), [prm1] "s"(prm1),
                   [ntiles] "s"(ntiles), [t0] "s"(t0), [wdma] "s"(wdma), [myq] "s"(myq), [voff] "v"(voff)
                 : DQ2_MAIN_CLOBBERS_F64);
  } else {
    a0 = a2 = a4 = a6 = v4d(0.0);
  }
  (void)a1; (void)a3; (void)a5; (void)a7;
  // slot j sits in a(2j) as {col 2q, col 2q+1, col 32+2q, col 32+2q+1} (eight VGPRs)
  const v4d accs[DQ_F64_RG] = {a0, a2, a4, a6};
  double* dst_base = out + (nsplit > 1 ? (int64_t)sp * out_rows_total * ldo : 0);
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int col = v * 32 + q * 2;
    double cv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) cv[i] = (cvec && nsplit == 1 && col + i < ncols) ? cvec[col + i] : 0.0;
#pragma unroll
    for (int j = 0; j < DQ_F64_RG; ++j) {
      const int r = 4 * j + g;
      if (r < my_rows) {
        const int64_t spos = (int64_t)row0 + 4 * quad0 + r;   // slot position -> output row
        double* y = dst_base + (perm ? (int64_t)perm[spos] : spos) * ldo + col;
        const double x0 = v ? accs[j].z : accs[j].x, x1 = v ? accs[j].w : accs[j].y;
        if (col + 1 < ncols) {
          v2d o;
          o.x = x0 - cv[0]; o.y = x1 - cv[1];
          *reinterpret_cast<v2d_a8*>(y) = o;
        } else if (col < ncols) {
          y[0] = x0 - cv[0];
        }
      }
    }
  }
}

}  // namespace

bool dq_build_tables(TiledOp& op, TiledBuffers& buf, hipStream_t s) {
  op.dq = false;
  static const bool off = dbg_env("SAPCA_NO_DQ") != nullptr;
  static const bool off64 = dbg_env("SAPCA_NO_DQ_F64") != nullptr;   // f64 fits on round 1's staged-entry sweep (A/B runs)
  if (off || !op.valid || op.fmt != 1 || op.ldp != 64 || op.tile_bytes != DQ_TILE_BYTES) return false;
  if (op.elem == 8) {
    if (off64 || op.block_rows != 64 * DQ_F64_RG) return false;
  } else if (op.elem != 4 || (op.block_rows != 512 && op.block_rows != 1024)) {
    return false;
  }
  if (op.total_entries / DQ_OFF_UNIT >= (int64_t)1 << 32) return false;
  const int64_t nchunks = (int64_t)op.nrb * op.nct;
  const size_t info_words = ((size_t)nchunks * DQ_WAVES + 64) * 2;   // (+ a 64-tile window of slack: the sweep loads whole windows)
  uint32_t* d_info = buf.dq_info.as<uint32_t>(info_words);
  SAPCA_HIP(hipMemsetAsync(d_info + (size_t)nchunks * DQ_WAVES * 2, 0, 64 * 2 * sizeof(uint32_t), s));
  hipLaunchKernelGGL(dq_info_kernel, dim3((unsigned)std::min<int64_t>(nchunks / 4 + 1, 8192)), dim3(256), 0, s, op.blk_row0, op.nct,
                     op.chunk_off, op.wave_off, reinterpret_cast<const uint16_t*>(op.steps), d_info, nchunks);
  SAPCA_HIP(hipGetLastError());
  // (the chunk count of a wave's stream in one tile: at most 16 quads x 65535 steps / 16 = 65535, the 16 bits the sweep reads)
  op.dq_info = d_info;
  op.dq = true;
  return true;
}

bool dq_usable(const TiledOp& op, int ldx) { return op.dq && ldx >= 64 && ldx % 64 == 0; }   // (wider panels: column passes of 64)

void launch_dq_f64(const TiledOp& op, const double* X, int ldx, double* out, int ldo, int ncols, const double* cvec, hipStream_t s) {
  static LdsAttrState attr;
  ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_dq_f64_kernel), DQ_LDS, attr);
  hipLaunchKernelGGL(spmm_dq_f64_kernel, dim3((unsigned)(op.nrb * op.nsplit)), dim3(DQ_THREADS), DQ_LDS, s, op.blk_row0, op.row_perm, op.nct, op.ent,
                     reinterpret_cast<const uint16_t*>(op.steps), op.dq_info, op.cols, X, ldx, op.nsplit, op.tiles_per_split, out, op.rows, ldo, ncols,
                     cvec, 0);
  SAPCA_HIP(hipGetLastError());
}

#ifdef DQ2_STAMPS
// stamp builds only (tools/dq_stamps.sh): the side buffer of the launch to instrument, counted from the call below
static uint32_t* g_stamp_buf = nullptr;
static int g_stamp_wgs = 0, g_stamp_which = -1, g_stamp_launch = 0;
extern "C" __attribute__((visibility("default"))) void sapca_debug_dq_stamps(void* dev_buf, int max_wgs, int which_launch) {
  g_stamp_buf = static_cast<uint32_t*>(dev_buf);
  g_stamp_wgs = max_wgs;
  g_stamp_which = which_launch;
  g_stamp_launch = 0;
}
#endif

template <int RG>
static void launch_dq_t(const TiledOp& op, const float* X, int ldx, float* out, int ldo, int ncols, const float* cvec, int flags,
                        hipStream_t s, int rb0, int rb1, int nsplit, int tiles_per_split) {
  static LdsAttrState attr[2];
  const bool pat = (flags & 8) != 0;
  const auto launch = [&](auto kern, LdsAttrState& a) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(kern), DQ_LDS, a);
    hipLaunchKernelGGL(kern, dim3((unsigned)((rb1 - rb0) * nsplit)), dim3(DQ_THREADS), DQ_LDS, s, op.blk_row0, op.row_perm, op.nct,
                       reinterpret_cast<const uint64_t*>(op.ent), reinterpret_cast<const uint16_t*>(op.steps), op.dq_info, op.cols, X, ldx,
                       nsplit, tiles_per_split, out, op.rows, ldo, ncols, cvec, rb0
#ifdef DQ2_STAMPS
                       , (g_stamp_launch++ == g_stamp_which) ? g_stamp_buf : nullptr, g_stamp_wgs
#endif
                       );
  };
  if (pat) launch(&spmm_dq_kernel<RG, true>, attr[1]);
  else launch(&spmm_dq_kernel<RG, false>, attr[0]);
}

void launch_dq(const TiledOp& op, const float* X, int ldx, float* out, int ldo, int ncols, const float* cvec, hipStream_t s, int flags) {
  launch_dq_blocks(op, 0, op.nrb, op.nsplit, op.tiles_per_split, X, ldx, out, ldo, ncols, cvec, s, flags);
}

// row blocks [rb0, rb1) only, their tile range cut into `nsplit` pieces of `tiles_per_split` tiles: the split is a matter of the
// launch (the per-(block, wave, tile) tables do not depend on it), so a caller that sweeps the operator in halves -- the
// row-sharded A^T sweep, whose first half is all-reduced while the second runs -- can keep every CU busy in each half
void launch_dq_blocks(const TiledOp& op, int rb0, int rb1, int nsplit, int tiles_per_split, const float* X, int ldx, float* out, int ldo,
                      int ncols, const float* cvec, hipStream_t s, int flags) {
  if (rb1 <= rb0) return;
  if (op.block_rows == 1024) launch_dq_t<16>(op, X, ldx, out, ldo, ncols, cvec, flags, s, rb0, rb1, nsplit, tiles_per_split);
  else launch_dq_t<8>(op, X, ldx, out, ldo, ncols, cvec, flags, s, rb0, rb1, nsplit, tiles_per_split);
}

// ---- MaskedSparsePCA::transform (quirk Q3, /root/reference/src/dimred/pca/sparse_masked/mod.rs:488-529) through the sweep ---------
// t_ik = sum over the stored, kept entries of row i of (a_ij - mu_j) W_jk  =  (A W)_ik - (P diag(mu) W)_ik, P the pattern of the
// stored entries.  The second product is the same sweep with every stored non-zero value read as 1 (mode bit 3); stored zeros look
// like the format's padding there, so a pass over the values finds them (rare) and adds their -mu_j W_j by hand.
namespace {
__global__ void scale_rows_kernel(const float* __restrict__ W, int64_t n, int ld, const float* __restrict__ mu, float* __restrict__ out) {
  const int64_t total = n * ld, stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) out[i] = W[i] * mu[i / ld];
}

__global__ void subtract_kernel(float* __restrict__ out, const float* __restrict__ sub, int64_t count) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) out[i] -= sub[i];
}

// one wave per row: out[i][:] -= mu_j W[j][:] for every stored entry of the row whose value is zero
__global__ void __launch_bounds__(256)
stored_zero_fix_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const float* __restrict__ val, int64_t rows,
                       const float* __restrict__ mu, const float* __restrict__ W, int ldw, int k, float* __restrict__ out, int ldo) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE, nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e1 = ptr[r + 1];
    for (int64_t e0 = ptr[r]; e0 < e1; e0 += WAVE) {
      const int64_t e = e0 + lane;
      unsigned long long zeros = __ballot(e < e1 && val[e] == 0.f);
      while (zeros) {
        const int b = __builtin_ctzll(zeros);
        zeros &= zeros - 1;
        const int64_t j = idx[e0 + b];
        const float m = mu[j];
        for (int c = lane; c < k; c += WAVE) out[r * ldo + c] -= m * W[j * ldw + c];
      }
    }
  }
}
}  // namespace

bool q3_projection_dq(const CsrView<float>& A, const TiledOp& op, const float* W, int ldw, const float* mu, float* W2, float* tmp,
                      float* out, int k, hipStream_t s) {
  if (!op.valid || op.nsplit != 1 || !dq_usable(op, ldw) || op.rows != A.rows || op.cols != A.cols || k > ldw) return false;
  const int64_t n = A.cols, m = A.rows;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)std::min<int64_t>((n * ldw + 255) / 256, 4096)), dim3(256), 0, s, W, n, ldw, mu, W2);
  for (int c0 = 0; c0 < k; c0 += op.ldp) {   // 64 columns of the panel per pass
    const int nc = std::min(k - c0, op.ldp);
    launch_dq(op, W + c0, ldw, out + c0, k, nc, nullptr, s, 0);
    launch_dq(op, W2 + c0, ldw, tmp + c0, k, nc, nullptr, s, 8);
  }
  hipLaunchKernelGGL(subtract_kernel, dim3((unsigned)std::min<int64_t>((m * k + 255) / 256, 4096)), dim3(256), 0, s, out, tmp, m * (int64_t)k);
  hipLaunchKernelGGL(stored_zero_fix_kernel, dim3((unsigned)std::min<int64_t>((m + 3) / 4, 8192)), dim3(256), 0, s, A.ptr, A.idx, A.val, m,
                     mu, W, ldw, k, out, k);
  SAPCA_HIP(hipGetLastError());
  return true;
}

}  // namespace k
}  // namespace sapca
