// LDS-staged sparse x dense-panel sweep (f32; f64 in the quad formulation) and the tile-major operator formats it reads.
//
// Why: per stored entry the sweep reads 8 B of A but a whole panel row (256 B at l = 60).  Served
// from L2 that gather caps the sweep far below the HBM roofline (measured 5.8 % with
// the row kernel of spmm.hip).  Here the panel rows of one column tile are staged in LDS once per
// workgroup and every gather is an LDS read, while A streams from HBM exactly once, contiguously.
//
// Two formulations live in this file (DESIGN.md §5 has the measurements that led from one to the other):
//
//  * "quad" (default; second half of the file): rows in blocks of <= 512 (one workgroup of 16 waves),
//    columns in `nct` INTERLEAVED tiles (tile t = columns == t mod nct, 80 KiB of LDS).  A wave is four
//    groups of 16 lanes and every group walks its own row with ds_read_b128; four consecutive rows (a
//    quad) advance in lockstep, so the format stores a quad's segment step by step, 4 entries per step,
//    padded with {0, 0.0f} to its longest row.  8 quads per wave, accumulators in VGPRs across all tiles,
//    no cross-lane reduction.  Builders: histogram/index + count + fill (LDS-staged for A, streaming for
//    the tile-major rows of A^T, direct scatter as the fallback), or A^T straight from A: through per-chunk
//    buckets (atd_*, round 2: the default for unmasked f32 fits; no transposed CSR, no sort) or round 1's
//    tquad_* (short runs of A per chunk).  The production sweep over this format is spmm_dq.hip's.
//
//  * "pair" (SAPCA_TILED_FMT=0; first half): contiguous column tiles of 96 KiB; the two half-waves of a
//    wave take two consecutive entries of ONE row per step (ds_read_b64), accumulators duplicated in the
//    two halves and summed at the end.  1.47 ms per C2 sweep against 0.89 ms for the quad kernel; kept
//    for A/B runs and because its ablation switches (SAPCA_ABL) document how the numbers were obtained.
//
//    Panels wider than 64 columns take two column passes over the same format (spmm_tiled); f64 values and
//    panels use the same format with 16-byte entries and 512-byte panel rows (spmm_quad_f64_kernel).
//
// Common to both: an entry is {u32 byte offset of its panel row inside the LDS tile, f32 value}; per
// column tile the workgroup does  barrier; panel tile + the block's entry chunk -> LDS (both prefetched
// into registers during the previous tile's compute); barrier; compute.  Tile ranges can be split over
// workgroups (A^T has few rows): partial sums go to a slab that a second kernel adds in fixed order.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.h"
#include "spmm_dq.h"

namespace sapca {
namespace k {

namespace {

constexpr int WAVE = 64;
constexpr int RW = 32;            // max rows per wave (register accumulators)
#ifndef SAPCA_RW2
#define SAPCA_RW2 32
#endif
#ifndef SAPCA_PADSTEPS
#define SAPCA_PADSTEPS 1   // row segments of the two-lane-group format are padded to this many steps
#endif
#ifndef SAPCA_ABL
#define SAPCA_ABL 0   // compile-time ablations of the sweep's inner loop (tools/abl_build.sh); 0 in the product
#endif
constexpr int RW2 = SAPCA_RW2;   // rows per wave of the default (2 lane groups, LDP 64) configuration
constexpr int BLOCK_ROWS = 512;   // stride of the per-(block, tile) step table; max rows per block
// lane groups ("slots") per wave: 2 half-waves (16 waves/workgroup, 512 rows) or 4 quarter-waves
// (8 waves/workgroup, 256 rows, half the LDS instructions per entry)
constexpr int waves_for(int slots) { return slots == 2 ? 16 : 8; }
constexpr int TILE_BYTES = 96 * 1024;
constexpr int LDS_TOTAL = 160 * 1024;
constexpr int STAGE_BYTES = LDS_TOTAL - TILE_BYTES - 1024;   // entry staging capacity
constexpr int STAGE_ENTRIES = STAGE_BYTES / 8 - WAVE;        // keep one chunk of slack for read-ahead

struct Ent { uint32_t off; float val; };
struct EntD { uint32_t off; uint32_t pad; double val; };   // the same entry for f64 values (16 bytes)
template <typename VT> struct EntOf { typedef Ent type; };
template <> struct EntOf<double> { typedef EntD type; };

// ---------------------------------------------------------------------------------- builder
// seg[r][t] (t = 0..nct) = number of entries of row r with col < t*TC  (prep.hip: tile_index_kernel)

// one block per (row block, column tile): batch counts, wave offsets, chunk size
template <int WAVES, int PAD>
__global__ void __launch_bounds__(BLOCK_ROWS)
tiled_count_kernel(const int32_t* __restrict__ seg, const int32_t* __restrict__ blk_row0, int nct,
                   uint8_t* __restrict__ steps, uint32_t* __restrict__ wave_off, int64_t* __restrict__ chunk_size) {
  __shared__ uint32_t scan[BLOCK_ROWS];
  const int rb = blockIdx.x / nct, ct = blockIdx.x % nct;
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  const int rpw = (nrows + WAVES - 1) / WAVES;
  const int lr = threadIdx.x;
  uint32_t padded = 0, nb = 0;
  if (lr < nrows) {
    const int64_t r = row0 + lr;
    const int len = seg[r * (nct + 1) + ct + 1] - seg[r * (nct + 1) + ct];
    nb = (uint32_t)((len + PAD - 1) / PAD);
    padded = nb * PAD;
  }
  steps[(int64_t)blockIdx.x * BLOCK_ROWS + lr] = (uint8_t)(WAVES == 16 ? nb * SAPCA_PADSTEPS : nb);
  scan[lr] = padded;
  __syncthreads();
  for (int off = 1; off < BLOCK_ROWS; off <<= 1) {   // inclusive prefix (Hillis-Steele)
    uint32_t v = lr >= off ? scan[lr - off] : 0;
    __syncthreads();
    scan[lr] += v;
    __syncthreads();
  }
  const uint32_t excl = scan[lr] - padded;
  if (lr < nrows && lr % rpw == 0) wave_off[(int64_t)blockIdx.x * WAVES + lr / rpw] = excl;
  if (lr == BLOCK_ROWS - 1) chunk_size[blockIdx.x] = scan[lr];
  if (lr < WAVES && lr * rpw >= nrows) wave_off[(int64_t)blockIdx.x * WAVES + lr] = scan[BLOCK_ROWS - 1];
}

// one single-wave workgroup per (row block, wave): it copies that wave's rows' entries into every
// column tile's chunk (one wave per workgroup so that operators with few row blocks -- A^T -- still
// spread over all CUs)
template <int WAVES, int PAD>
__global__ void __launch_bounds__(WAVE)
tiled_fill_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const float* __restrict__ val,
                  const int32_t* __restrict__ seg, const int32_t* __restrict__ blk_row0, int nct, int tc,
                  int ldp_bytes, const int64_t* __restrict__ chunk_off, const uint32_t* __restrict__ wave_off,
                  Ent* __restrict__ ent, uint32_t* __restrict__ run_global) {
  extern __shared__ uint32_t run[];  // [nct] next free slot of this wave in every tile's chunk,
                                     // relative to the block's first chunk (one LDS read per entry)
  const int rb = blockIdx.x / WAVES;
  const int wave = blockIdx.x % WAVES, lane = threadIdx.x;
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  const int rpw = (nrows + WAVES - 1) / WAVES;
  // many column tiles (A^T of a tall matrix): the table does not fit LDS and lives in HBM/L2 instead
  uint32_t* mybase = run_global ? run_global + (size_t)blockIdx.x * nct : run;
  const int64_t block_base = chunk_off[(int64_t)rb * nct];
  for (int t = lane; t < nct; t += WAVE)
    mybase[t] = (uint32_t)(chunk_off[(int64_t)rb * nct + t] - block_base) + wave_off[((int64_t)rb * nct + t) * WAVES + wave];
  __builtin_amdgcn_wave_barrier();
  Ent* __restrict__ out = ent + block_base;
  const int lr0 = wave * rpw, lr1 = min(nrows, lr0 + rpw);
  for (int lr = lr0; lr < lr1; ++lr) {
    const int64_t r = row0 + lr;
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    const int32_t* sg = seg + r * (nct + 1);
    for (int64_t e = e0 + lane; e < e1; e += WAVE) {
      const int c = idx[e];
      const int t = c / tc;
      Ent x;
      x.off = (uint32_t)(c - t * tc) * (uint32_t)ldp_bytes;
      x.val = val[e];
      out[mybase[t] + (uint32_t)((e - e0) - sg[t])] = x;
    }
    __builtin_amdgcn_wave_barrier();  // the wave's reads of mybase (above) precede its update (below)
    for (int t = lane; t < nct; t += WAVE) {
      const int len = sg[t + 1] - sg[t];
      mybase[t] += (uint32_t)((len + PAD - 1) / PAD * PAD);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

inline int grid_for(int64_t work_items, int block, int cap = 8192) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// ---------------------------------------------------------------------------------- sweep
typedef float v2f __attribute__((ext_vector_type(2)));
template <int VPL> struct Lane;
template <> struct Lane<2> {
  using V = v2f;
  __device__ static inline V load(const char* p) { return *reinterpret_cast<const v2f*>(p); }
};
template <> struct Lane<4> {
  typedef float V __attribute__((ext_vector_type(4)));
  __device__ static inline V load(const char* p) { return *reinterpret_cast<const V*>(p); }
};

// U consecutive steps of one row: entry reads first, panel gathers next, FMAs last, so U gathers
// per wave are in flight.
template <int VPL, int PAD, int U>
__device__ __forceinline__ void steps_batch(typename Lane<VPL>::V& acc, const char* stage_lane, const char* tile_lane) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  u2 e[U];
#if SAPCA_ABL & 2   // ablation: no entry reads (one fixed entry held in registers, no extra VALU)
  {
    u2 f;
    f.x = (unsigned)(size_t)stage_lane & 0xff00u;
    f.y = 0x3f800000u;
    asm volatile("" : "+v"(f));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      e[u] = f;
      e[u].x += u * 256;   // folds into the ds_read offset field
    }
  }
#else
#pragma unroll
  for (int u = 0; u < U; ++u) e[u] = *reinterpret_cast<const u2*>(stage_lane + u * (PAD * 8));
#endif
  typename Lane<VPL>::V w[U];
#if SAPCA_ABL & 1   // ablation: no panel gathers
#pragma unroll
  for (int u = 0; u < U; ++u) w[u] = typename Lane<VPL>::V(__uint_as_float(e[u].x));
#else
#pragma unroll
  for (int u = 0; u < U; ++u) w[u] = Lane<VPL>::load(tile_lane + e[u].x);
#endif
#if SAPCA_ABL & 4   // ablation: no FMAs
#pragma unroll
  for (int u = 0; u < U; ++u) asm volatile("" ::"v"(w[u]), "v"(e[u].y));
#else
#pragma unroll
  for (int u = 0; u < U; ++u) acc += __uint_as_float(e[u].y) * w[u];
#endif
}

typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v4f_a4 __attribute__((ext_vector_type(4), aligned(4)));   // output rows are only element-aligned (ldo = k)
constexpr int np_tile(int threads) { return TILE_BYTES / (threads * 16); }
constexpr int np_stage(int threads) { return (STAGE_BYTES + threads * 16 - 1) / (threads * 16); }

// loads are unconditional (addresses clamped into the valid range) so the arrays stay in VGPRs
template <int N, int THREADS>
__device__ __forceinline__ void load_regs(v4f (&r)[N], const char* src, int bytes) {
#pragma unroll
  for (int i = 0; i < N; ++i)
    r[i] = *reinterpret_cast<const v4f*>(src + min((i * THREADS + (int)threadIdx.x) * 16, bytes - 16));
}
template <int N, int THREADS>
__device__ __forceinline__ void store_regs(const v4f (&r)[N], char* dst, int capacity) {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int b = (i * THREADS + (int)threadIdx.x) * 16;
    if ((i + 1) * THREADS * 16 <= capacity || b + 16 <= capacity) *reinterpret_cast<v4f*>(dst + b) = r[i];
  }
}

#define SAPCA_PREFETCH(CT)                                                                          \
  {                                                                                                 \
    const int64_t cidx_ = (int64_t)rb * nct + (CT);                                                 \
    const int64_t c_lo_ = chunk_off[cidx_];                                                         \
    const int64_t first_ = (int64_t)(CT) * tc;                                                      \
    load_regs<NP_TILE, THREADS>(pt, reinterpret_cast<const char*>(X + first_ * LDP),                         \
                       (int)min<int64_t>(tc, panel_rows - first_) * LDP * 4);                       \
    load_regs<NP_STAGE, THREADS>(ps, reinterpret_cast<const char*>(ent + c_lo_),                             \
                        max(16, (int)(chunk_off[cidx_ + 1] - c_lo_) * 8));                          \
  }

template <int LDP, int SLOTS, bool PREFETCH>  // LDP: panel leading dimension in floats (64 or 128)
__global__ void __launch_bounds__(waves_for(SLOTS) * WAVE)
spmm_tiled_kernel(const int32_t* __restrict__ blk_row0, int nct, int tc, const int64_t* __restrict__ chunk_off,
                  const uint32_t* __restrict__ wave_off, const uint8_t* __restrict__ steps,
                  const Ent* __restrict__ ent, int64_t panel_rows, const float* __restrict__ X, int nsplit,
                  int tiles_per_split, float* __restrict__ out, int64_t out_rows_total, int ldo, int ncols,
                  const float* __restrict__ cvec, int mode) {
  constexpr int WAVES = waves_for(SLOTS), THREADS = WAVES * WAVE, PAD = SLOTS;
  constexpr int LPE = WAVE / SLOTS;   // lanes that cover one panel row
  constexpr int VPL = LDP / LPE;      // panel values per lane
  constexpr int RWK = (SLOTS == 2 && LDP == 128) ? RW / 2 : (SLOTS == 2 ? RW2 : RW);   // rows per wave: the accumulators must fit the VGPR budget
  constexpr int NP_TILE = np_tile(THREADS), NP_STAGE = np_stage(THREADS);
  using LN = Lane<VPL>;
  using V = typename LN::V;
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* tile = lds;
  char* stage = lds + TILE_BYTES;
  const int rb = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int ct0 = sp * tiles_per_split, ct1 = min(nct, ct0 + tiles_per_split);
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const int half = lane / LPE, q = lane % LPE;   // half = lane group (slot)
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  const int rpw = (nrows + WAVES - 1) / WAVES;
  const int my_rows = max(0, min(nrows - wave * rpw, rpw));
  const char* tl = tile + q * (VPL * 4);

  V acc[RWK];
#pragma unroll
  for (int i = 0; i < RWK; ++i) acc[i] = V(0.f);

  v4f pt[NP_TILE], ps[NP_STAGE];
  // per-tile bookkeeping of this wave (step counts of its rows, start of its entries), fetched one
  // tile ahead and BEFORE the bulk prefetch: vmcnt retires in order, so a small load issued after
  // the prefetch would make its first use wait for the whole prefetch
  int cnt_next = 0;
  unsigned woff_next = 0;
#define SAPCA_BOOKKEEPING(CT)                                                                      \
  {                                                                                                \
    const int64_t cidx_ = (int64_t)rb * nct + (CT);                                                \
    cnt_next = lane < my_rows ? (int)steps[cidx_ * BLOCK_ROWS + wave * rpw + lane] : 0;            \
    woff_next = wave_off[cidx_ * WAVES + wave];                                                    \
  }
  if (ct0 < ct1) SAPCA_BOOKKEEPING(ct0)
  if (PREFETCH && ct0 < ct1) SAPCA_PREFETCH(ct0)
  for (int ct = ct0; ct < ct1; ++ct) {
    if (!(mode & 8) || ct == ct0) {
      __syncthreads();  // the previous tile's readers are done
      if (!PREFETCH && (!(mode & 2) || ct == ct0)) SAPCA_PREFETCH(ct)
      store_regs<NP_TILE, THREADS>(pt, tile, TILE_BYTES);
      store_regs<NP_STAGE, THREADS>(ps, stage, STAGE_BYTES);
      __syncthreads();
    }
    const int cnt_v = cnt_next;
    const unsigned woff = woff_next;
    if (ct + 1 < ct1 && !(mode & 8)) SAPCA_BOOKKEEPING(ct + 1)
    if (PREFETCH && ct + 1 < ct1 && !(mode & 10)) SAPCA_PREFETCH(ct + 1)
    if (mode & 1) continue;
    if (my_rows > 0) {
      const char* sl = stage + (size_t)woff * 8 + half * 8;
#pragma unroll
      for (int rr = 0; rr < RWK; ++rr) {
        int n = __builtin_amdgcn_readlane(cnt_v, rr);
        while (n >= 4) {
          steps_batch<VPL, PAD, 4>(acc[rr], sl, tl);
          sl += 4 * PAD * 8;
          n -= 4;
        }
        if (n >= 2) {
          steps_batch<VPL, PAD, 2>(acc[rr], sl, tl);
          sl += 2 * PAD * 8;
          n -= 2;
        }
        if ((SLOTS != 2 || SAPCA_PADSTEPS < 2) && n) {
          steps_batch<VPL, PAD, 1>(acc[rr], sl, tl);
          sl += PAD * 8;
        }
      }
    }
  }

  // sum the two half-waves; half 0 writes columns VPL*q .. of every row of this wave
  float* dst_base = out + (nsplit > 1 ? (int64_t)sp * out_rows_total * ldo : 0);
  const int col = q * VPL;
  float cv[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) cv[i] = (cvec && nsplit == 1) ? cvec[col + i] : 0.f;
#pragma unroll
  for (int rr = 0; rr < RWK; ++rr) {
    V a = acc[rr];
#pragma unroll
    for (int off = LPE; off < WAVE; off <<= 1)
#pragma unroll
      for (int i = 0; i < VPL; ++i) a[i] += __shfl_xor(a[i], off);
    if (rr < my_rows && half == 0) {
      float* y = dst_base + (int64_t)(row0 + wave * rpw + rr) * ldo + col;
#pragma unroll
      for (int i = 0; i < VPL; ++i)
        if (col + i < ncols) y[i] = a[i] - cv[i];
    }
  }
}

// out[r][j] = sum_sp part[sp][r][j] - cvec[j]   (fixed order)
template <typename VT>
__global__ void split_reduce_kernel(const VT* __restrict__ part, int nsplit, int64_t rows, int ldo, int ncols,
                                    const VT* __restrict__ cvec, VT* __restrict__ out, int ldy) {
  const int64_t total = rows * ldo;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int j = (int)(i % ldo);
    if (j >= ncols) continue;
    VT s = 0;
    for (int sp = 0; sp < nsplit; ++sp) s += part[(int64_t)sp * total + i];
    out[(i / ldo) * ldy + j] = s - (cvec ? cvec[j] : (VT)0);
  }
}

// the same for a run of rows: slab sp starts `slab_stride` elements behind slab sp - 1
__global__ void split_reduce_rows_kernel(const float* __restrict__ part, int nsplit, int64_t slab_stride, int64_t rows, int ldo, int ncols,
                                         float* __restrict__ out, int ldy) {
  const int64_t total = rows * ldo;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int j = (int)(i % ldo);
    if (j >= ncols) continue;
    float s = 0;
    for (int sp = 0; sp < nsplit; ++sp) s += part[(int64_t)sp * slab_stride + i];
    out[(i / ldo) * ldy + j] = s;
  }
}

#undef SAPCA_PREFETCH
#undef SAPCA_BOOKKEEPING

// ------------------------------------------------------------------------ "quad" format and sweep
// Second formulation of the staged sweep.  Measured on the first one (profiles/): a
// SIMD retires about one instruction per 4-6 cycles whatever its kind, so the sweep time follows
// the number of instructions per stored entry.  Here a wave is four groups of 16 lanes and every
// group walks its OWN row (ds_read_b128: a lane holds 4 of the 64 panel columns, 8 of 128), so
// one instruction stream advances four entries, there is no duplicate accumulator to reduce at
// the end, and the accumulators of a 512-row block take 32 VGPRs instead of 64.  The four rows of
// a quad advance in lockstep: the format pads every quad's segment in a tile to its longest row
// (zero entries: offset 0, value 0).
#ifndef SAPCA_QWAVES
#define SAPCA_QWAVES 16   // waves per workgroup of the quad sweep: 16 x 8 quads, or 8 x 16 quads with 8-step batches
#endif
constexpr int QGROUPS = 4, QLANES = WAVE / QGROUPS, QWAVES = SAPCA_QWAVES, QTHREADS = QWAVES * WAVE;
constexpr int Q_TILE_BYTES = 80 * 1024;          // default split of the 160 KiB: 80 KiB panel tile + 79 KiB entry staging
constexpr int Q_MAX_TILES_RUNS = 16384;          // tile-major builder (bounded by the tile arithmetic's float reciprocal and the index tables' size)
constexpr int Q_TILE_BYTES_BIG = 96 * 1024;      // for operators whose chunks leave room: fewer, longer tile steps
constexpr int q_stage_bytes(int tile_bytes) { return LDS_TOTAL - tile_bytes - 1024; }
constexpr int QBLOCK_ROWS = 1024;                // most rows of a quad-format block (the DPP-fed sweep with 16 row slots per lane group)
constexpr int Q_BLOCK_QUADS = QBLOCK_ROWS / 4;   // stride of the per-chunk quad step table
constexpr int ENT_SLACK = 4 * WAVE;             // zero entries behind the last chunk: the sweeps read whole 16-step chunks (and three ahead)
constexpr int q_rows_per_group(int ldp) { return (ldp == 64 ? 128 : 64) / QWAVES; }
// steps of a quad in a tile: its longest row segment, rounded up to an even count (the DPP-fed sweep of spmm_dq.hip
// switches row slots every two steps)
__host__ __device__ inline int q_steps(int longest) { return kOddSteps ? longest : (longest + 1) & ~1; }
// quads (4 consecutive rows) of a block are dealt to its 16 waves in contiguous, balanced ranges
__host__ __device__ inline int q_first(int wave, int nquads) { return wave * nquads / QWAVES; }

// Column tiles of this format are INTERLEAVED: tile t holds the panel rows {c : c mod nct == t}, at
// position c / nct.  Contiguous column ranges with their own density (gene modules, a dense band of
// a cluster) are thereby spread over all tiles, every (row, tile) segment has about the same length,
// the waves of a workgroup reach the per-tile barrier together and quads pad little.
__device__ __forceinline__ void divmod_small(int c, int d, float inv, int& q, int& r) {   // c < 2^24
  q = (int)((float)c * inv);
  r = c - q * d;
  if (r < 0) { r += d; --q; }
  if (r >= d) { r -= d; ++q; }
}

// seg[r][t] = number of entries of row r in tiles < t (t = 0..nct): LDS histogram per row, one wave per row
__global__ void __launch_bounds__(256)
tile_hist_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, int64_t rows, int nct, float inv_nct,
                 int32_t* __restrict__ seg) {
  extern __shared__ uint32_t hist_all[];
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  uint32_t* hist = hist_all + (size_t)wave * nct;
  for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
    for (int t = lane; t < nct; t += WAVE) hist[t] = 0;
    __builtin_amdgcn_wave_barrier();
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    for (int64_t eb = e0 + lane; eb < e1; eb += 8 * WAVE) {   // eight loads in flight per lane
      int c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = eb + u * WAVE < e1 ? idx[eb + u * WAVE] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c[u] >= 0) {
          int q, t;
          divmod_small(c[u], nct, inv_nct, q, t);
          atomicAdd(&hist[t], 1u);
        }
    }
    __builtin_amdgcn_wave_barrier();
    uint32_t carry = 0;
    int32_t* out = seg + r * (nct + 1);
    for (int t0 = 0; t0 < nct; t0 += WAVE) {
      const int t = t0 + lane;
      const uint32_t v = t < nct ? hist[t] : 0u;
      uint32_t x = v;
#pragma unroll
      for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if (lane >= off) x += y;
      }
      if (t < nct) out[t] = (int32_t)(carry + x - v);
      carry += __shfl(x, WAVE - 1);
    }
    if (lane == 0) out[nct] = (int32_t)carry;
    __builtin_amdgcn_wave_barrier();
  }
}

// one block per (row block, group of QC_TILES column tiles), one thread per quad: steps of every quad (= its longest row
// segment), entry offset of every wave, chunk size
constexpr int QC_TILES = 8;   // tiles per workgroup of quad_count_kernel: a thread reads its rows' 9 consecutive segment bounds (one or two lines)
                              // instead of one line per (row, tile) -- a table with a 4 KiB row pitch (A^T of C2 in f64) went at 1.6 ms
__global__ void __launch_bounds__(Q_BLOCK_QUADS)
quad_count_kernel(const int32_t* __restrict__ seg, const int32_t* __restrict__ blk_row0, const uint32_t* __restrict__ perm,
                  int nct, uint16_t* __restrict__ steps, uint32_t* __restrict__ quad_off, uint32_t* __restrict__ wave_off,
                  int64_t* __restrict__ chunk_size, const uint16_t* __restrict__ cnt16 = nullptr, int64_t cnt_stride = 0,
                  int64_t* __restrict__ raw_size = nullptr) {
  __shared__ uint32_t scan[Q_BLOCK_QUADS];
  __shared__ uint32_t wave_total[Q_BLOCK_QUADS / WAVE], raw_part[Q_BLOCK_QUADS / WAVE];
  const int groups = (nct + QC_TILES - 1) / QC_TILES;
  const int rb = blockIdx.x / groups, ct0 = (blockIdx.x % groups) * QC_TILES;
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  const int nquads = (nrows + 3) / 4;
  const int q = threadIdx.x, lane = q & (WAVE - 1);
  int len[4][QC_TILES];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int lr = 4 * q + g;
    const bool have = lr < nrows;
    const int64_t r = have ? (perm ? (int64_t)perm[row0 + lr] : (int64_t)row0 + lr) : 0;   // slot -> row (rows sorted by length)
    if (cnt16) {
      // (the bucket builder of A^T counts entries per (tile, row) instead of indexing a transposed CSR)
#pragma unroll
      for (int j = 0; j < QC_TILES; ++j) len[g][j] = (have && ct0 + j < nct) ? (int)cnt16[(int64_t)(ct0 + j) * cnt_stride + r] : 0;
    } else {
      int bound[QC_TILES + 1];
#pragma unroll
      for (int j = 0; j <= QC_TILES; ++j) bound[j] = (have && ct0 + j <= nct) ? seg[r * (nct + 1) + ct0 + j] : 0;
#pragma unroll
      for (int j = 0; j < QC_TILES; ++j) len[g][j] = (have && ct0 + j < nct) ? bound[j + 1] - bound[j] : 0;
    }
  }
#pragma unroll
  for (int j = 0; j < QC_TILES; ++j) {
    const int ct = ct0 + j;
    if (ct >= nct) break;
    const int64_t chunk = (int64_t)rb * nct + ct;
    const int longest = max(max(len[0][j], len[1][j]), max(len[2][j], len[3][j]));
    int raw = len[0][j] + len[1][j] + len[2][j] + len[3][j];
    const int qmax = q_steps(longest);
    const uint32_t padded = (uint32_t)qmax * 4u;
    steps[chunk * Q_BLOCK_QUADS + q] = (uint16_t)qmax;
    // inclusive scan of the padded quad sizes over the block: inside each wave by shuffles, the wave totals through LDS
    uint32_t inc = padded;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const uint32_t y = __shfl_up(inc, off);
      if (lane >= off) inc += y;
    }
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) raw += __shfl_xor(raw, off);
    __syncthreads();   // (the previous tile's readers are done)
    if (lane == WAVE - 1) wave_total[q / WAVE] = inc;
    if (lane == 0) raw_part[q / WAVE] = (uint32_t)raw;
    __syncthreads();
    for (int w = 0; w < q / WAVE; ++w) inc += wave_total[w];
    scan[q] = inc;
    __syncthreads();
    // [row block][quad][tile]: the builder reads one quad's offsets in all tiles contiguously
    if (quad_off && q < nquads) quad_off[((int64_t)rb * Q_BLOCK_QUADS + q) * nct + ct] = inc - padded;   // (the bucket route rebuilds them from `steps`)
    if (q < QWAVES) {
      const int first_quad = q_first(q, nquads);
      wave_off[chunk * QWAVES + q] = first_quad > 0 ? scan[first_quad - 1] : 0u;
    }
    if (q == Q_BLOCK_QUADS - 1) {
      chunk_size[chunk] = inc;
      if (raw_size) {
        uint32_t total = 0;
        for (int w = 0; w < Q_BLOCK_QUADS / WAVE; ++w) total += raw_part[w];
        raw_size[chunk] = total;
      }
    }
  }
}

// one wave per row: the k-th entry (in column order) that the row has in tile t goes to slot
// (k*4 + g) of its quad's segment in that tile's chunk (g = row mod 4 within the block)
template <typename VT>
__global__ void __launch_bounds__(256)
quad_fill_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const VT* __restrict__ val,
                 int64_t rows, const int32_t* __restrict__ blk_row0, const uint32_t* __restrict__ perm, int nrb, int nct,
                 float inv_nct, int ldp_bytes, const int64_t* __restrict__ chunk_off, const uint32_t* __restrict__ quad_off,
                 typename EntOf<VT>::type* __restrict__ ent) {
  typedef typename EntOf<VT>::type E;
  extern __shared__ uint32_t cnt_all[];   // per wave and tile: slot of the row's next entry (relative to the block's first chunk)
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  uint32_t* cnt = cnt_all + (size_t)wave * nct;
  const int64_t sp = (int64_t)blockIdx.x * 4 + wave;   // slot position; its row is perm[sp]
  if (sp >= rows) return;
  const int64_t r = perm ? (int64_t)perm[sp] : sp;
  int lo = 0, hi = nrb;   // row block: the last b with blk_row0[b] <= sp
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)blk_row0[mid] <= sp) lo = mid; else hi = mid;
  }
  const int rb = lo;
  const int lr = (int)(sp - blk_row0[rb]);
  const int qd = lr >> 2;
  const uint32_t g = (uint32_t)(lr & 3);
  const int64_t* __restrict__ coff = chunk_off + (int64_t)rb * nct;
  const uint32_t* __restrict__ qoff = quad_off + ((int64_t)rb * Q_BLOCK_QUADS + qd) * nct;
  const int64_t block_base = coff[0];
  for (int t = lane; t < nct; t += WAVE) cnt[t] = (uint32_t)(coff[t] - block_base) + qoff[t] + g;
  __builtin_amdgcn_wave_barrier();
  const int64_t e0 = ptr[r], e1 = ptr[r + 1];
  E* __restrict__ out = ent + block_base;
  // batches of 64 entries in column order; within a batch the LDS atomic hands out the ranks of
  // equal tiles (a fixed function of the input: the format is reproducible run to run)
  for (int64_t eb = e0; eb < e1; eb += 8 * WAVE) {   // 8 batches loaded ahead: one round trip per 512 entries
    int c[8];
    VT v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int64_t e = eb + u * WAVE + lane;
      c[u] = e < e1 ? idx[e] : -1;
      v[u] = e < e1 ? val[e] : (VT)0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (c[u] >= 0) {
        int i, t;
        divmod_small(c[u], nct, inv_nct, i, t);
        E x = E();
        x.off = (uint32_t)i * (uint32_t)ldp_bytes;
        x.val = v[u];
        out[atomicAdd(&cnt[t], 4u)] = x;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// The same fill with the quad assembled in LDS first: a workgroup owns one quad (a wave per row),
// scatters the entries into an LDS image of the quad's segments (padding pre-zeroed) and then writes
// every tile's segment out as one contiguous run -- the direct version above issues one isolated
// 8-byte store per entry, which is what it spends its time on.  Quads larger than the LDS image
// (very long rows) take the direct route and zero their padding themselves, so the entry buffer
// needs no memset on this path.
constexpr int QF_CAP_MIN = 4096, QF_CAP_MAX = 6144;   // entries of one quad staged in LDS: 32 KiB (more workgroups per CU) .. 48 KiB
// (quad `qi` of the operator: rb = qi / Q_BLOCK_QUADS, qd = qi % Q_BLOCK_QUADS; all 256 threads of the workgroup)
template <typename VT>
__device__ __forceinline__ void quad_fill_staged_one(int qi, uint32_t* qf_lds, const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                                     const VT* __restrict__ val, const int32_t* __restrict__ seg,
                                                     const int32_t* __restrict__ blk_row0, const uint32_t* __restrict__ perm, int nct, int cap,
                                                     float inv_nct, int ldp_bytes, const int64_t* __restrict__ chunk_off,
                                                     const uint32_t* __restrict__ quad_off, typename EntOf<VT>::type* __restrict__ ent) {
  typedef typename EntOf<VT>::type E;
  constexpr int EW = (int)sizeof(E) / 4;   // entry size in LDS words
  E* stage = reinterpret_cast<E*>(qf_lds);         // [cap]
  int64_t* gofs = reinterpret_cast<int64_t*>(qf_lds + EW * cap);   // [nct] where the quad's segment of tile t goes
  uint32_t* lofs = qf_lds + EW * cap + 2 * nct;     // [nct + 1] start of every tile's segment in the image
  uint32_t* cnt_all = lofs + nct + 1;              // [4][nct] entries of row g seen so far in tile t
  const int rb = qi / Q_BLOCK_QUADS, qd = qi % Q_BLOCK_QUADS;
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  if (4 * qd >= nrows) return;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const int qrows = min(4, nrows - 4 * qd);
  const int64_t s0 = (int64_t)row0 + 4 * qd;   // first slot position of the quad
  int64_t rowof[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) rowof[g] = g < qrows ? (perm ? (int64_t)perm[s0 + g] : s0 + g) : 0;
  const int64_t* __restrict__ coff = chunk_off + (int64_t)rb * nct;
  const uint32_t* __restrict__ qoff = quad_off + ((int64_t)rb * Q_BLOCK_QUADS + qd) * nct;
  // segment sizes (4 x the longest of the quad's rows in the tile), then their exclusive scan
  for (int t = threadIdx.x; t < nct; t += 256) {
    int mx = 0;
    for (int g = 0; g < qrows; ++g) {
      const int32_t* sg = seg + rowof[g] * (nct + 1);
      mx = max(mx, sg[t + 1] - sg[t]);
    }
    lofs[t + 1] = (uint32_t)q_steps(mx) * 4u;
    gofs[t] = coff[t] + (int64_t)qoff[t];   // (fetched here, beside the segment bounds: the write-out below waits on LDS only)
    for (int g = 0; g < 4; ++g) cnt_all[g * nct + t] = 0;
  }
  if (threadIdx.x == 0) lofs[0] = 0;
  __syncthreads();
  if (wave == 0) {
    uint32_t carry = 0;
    for (int t0 = 0; t0 < nct; t0 += WAVE) {
      const int t = t0 + lane;
      const uint32_t v = t < nct ? lofs[t + 1] : 0u;
      uint32_t x = v;
#pragma unroll
      for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t y = __shfl_up(x, off);
        if (lane >= off) x += y;
      }
      if (t < nct) lofs[t + 1] = carry + x;
      carry += __shfl(x, WAVE - 1);
    }
  }
  __syncthreads();
  const uint32_t total = lofs[nct];
  const bool staged = total <= (uint32_t)cap;
  if (staged) {
    uint64_t* z = reinterpret_cast<uint64_t*>(stage);
    for (uint32_t i = threadIdx.x; i < total * (EW / 2); i += 256) z[i] = 0;
  } else {
    // direct route: zero the padding slots of this wave's row in global memory
    if (wave < qrows) {
      const int32_t* sg = seg + rowof[wave] * (nct + 1);
      for (int t = lane; t < nct; t += WAVE) {
        const uint32_t steps = (lofs[t + 1] - lofs[t]) / 4u;
        E* dst = ent + gofs[t];
        for (uint32_t k = (uint32_t)(sg[t + 1] - sg[t]); k < steps; ++k) dst[k * 4u + wave] = E();
      }
    } else {
      for (int t = lane; t < nct; t += WAVE) {   // rows past the end of the block: all padding
        const uint32_t steps = (lofs[t + 1] - lofs[t]) / 4u;
        E* dst = ent + gofs[t];
        for (uint32_t k = 0; k < steps; ++k) dst[k * 4u + wave] = E();
      }
    }
  }
  __syncthreads();
  if (wave < qrows) {
    const int64_t r = rowof[wave];
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    uint32_t* cnt = cnt_all + (size_t)wave * nct;
    // batches of 64 entries in column order; within a batch the LDS atomic hands out the ranks of
    // equal tiles (a fixed function of the input: the format is reproducible run to run)
    // 8 batches are loaded ahead so that one memory round trip serves 512 entries
    for (int64_t eb = e0; eb < e1; eb += 8 * WAVE) {
      int c[8];
      VT v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int64_t e = eb + u * WAVE + lane;
        c[u] = e < e1 ? idx[e] : -1;
        v[u] = e < e1 ? val[e] : (VT)0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (c[u] >= 0) {
          int i, t;
          divmod_small(c[u], nct, inv_nct, i, t);
          E x = E();
          x.off = (uint32_t)i * (uint32_t)ldp_bytes;
          x.val = v[u];
          const uint32_t k = atomicAdd(&cnt[t], 1u);
          if (staged) stage[lofs[t] + k * 4u + (uint32_t)wave] = x;
          else ent[gofs[t] + k * 4u + (uint32_t)wave] = x;
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  if (!staged) return;
  __syncthreads();
  for (int t = wave; t < nct; t += 4) {
    const uint32_t lo = lofs[t], n = lofs[t + 1] - lo;
    E* dst = ent + gofs[t];
    for (uint32_t j = lane; j < n; j += WAVE) dst[j] = stage[lo + j];
  }
}

// A workgroup walks quads qi = blockIdx.x, blockIdx.x + gridDim.x, ... (the default grid is one workgroup per quad: qf_grid).
template <typename VT>
__global__ void __launch_bounds__(256)
quad_fill_staged_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const VT* __restrict__ val,
                        const int32_t* __restrict__ seg, const int32_t* __restrict__ blk_row0,
                        const uint32_t* __restrict__ perm, int nct, int cap, float inv_nct,
                        int ldp_bytes, const int64_t* __restrict__ chunk_off, const uint32_t* __restrict__ quad_off,
                        typename EntOf<VT>::type* __restrict__ ent, int nquads_all) {
  extern __shared__ __attribute__((aligned(16))) uint32_t qf_lds[];
  for (int qi = blockIdx.x; qi < nquads_all; qi += gridDim.x) {
    quad_fill_staged_one<VT>(qi, qf_lds, ptr, idx, val, seg, blk_row0, perm, nct, cap, inv_nct, ldp_bytes, chunk_off, quad_off, ent);
    __syncthreads();   // (the image and its tables are reused by the next quad)
  }
}

// ---- rows whose entries are already grouped by tile (transpose_csr(..., tile_major_nct)) ----------
// seg[r][t] = number of entries of row r in tiles < t: the tile of an entry (idx mod nct) is
// non-decreasing along the row, so a binary search per boundary does it
__global__ void tile_index_mod_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                      const uint64_t* __restrict__ packed, int64_t rows, int nct, float inv_nct,
                                      int32_t* __restrict__ seg) {
  const int64_t total = rows * (int64_t)(nct + 1);
  int64_t o = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; o < total; o += stride) {
    const int64_t r = o / (nct + 1);
    const int t = (int)(o - r * (nct + 1));
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    int64_t lo = e0, hi = e1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      int q, tm;
      divmod_small(packed ? (int)(packed[mid] >> 32) : idx[mid], nct, inv_nct, q, tm);
      if (tm < t) lo = mid + 1; else hi = mid;
    }
    seg[o] = (int32_t)(lo - e0);
  }
}

// One pass over the packed tile-major rows: the column statistics of A (row sums of A^T, accumulated in
// exactly the order of prep.hip's row_sums kernels) and seg[r][t] from the places where the tile changes.
__global__ void at_stats_index_kernel(const int64_t* __restrict__ ptr, const uint64_t* __restrict__ packed, int64_t rows,
                                      int nct, float inv_nct, double* __restrict__ sum, double* __restrict__ sumsq,
                                      int32_t* __restrict__ seg) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    int32_t* sg = seg + r * (nct + 1);
    double a = 0, b = 0;
    int carry = -1;   // tile of the entry before this batch
    for (int64_t eb = e0; eb < e1; eb += WAVE) {
      const int64_t e = eb + lane;
      const bool valid = e < e1;
      int t = nct;   // past the end: closes every remaining boundary
      if (valid) {
        const uint64_t pv = packed[e];
        const double v = (double)__uint_as_float((uint32_t)pv);
        a += v;
        b += v * v;
        int q;
        divmod_small((int)(pv >> 32), nct, inv_nct, q, t);
      }
      int tprev = __shfl_up(t, 1);
      if (lane == 0) tprev = carry;
      // entry e is the first one of tiles (tprev, t]; lanes past the end write the closing boundaries once
      if (valid || e == e1)
        for (int tt = tprev + 1; tt <= t; ++tt) sg[tt] = (int32_t)(e - e0);
      carry = __shfl(t, WAVE - 1);
    }
    // rows whose length is a multiple of 64 (or zero) have not closed their boundaries yet
    if (((e1 - e0) & (WAVE - 1)) == 0)
      for (int tt = carry + 1 + lane; tt <= nct; tt += WAVE) sg[tt] = (int32_t)(e1 - e0);
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
      a += __shfl_xor(a, off);
      b += __shfl_xor(b, off);
    }
    if (lane == 0) {
      sum[r] = a;
      if (sumsq) sumsq[r] = b;
    }
  }
}

// rows sorted by length: keys for the descending sort and the identity payload
__global__ void row_len_iota_kernel(const int64_t* __restrict__ ptr, int64_t rows, uint32_t* __restrict__ len,
                                    uint32_t* __restrict__ iota) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) {
    len[r] = (uint32_t)(ptr[r + 1] - ptr[r]);
    iota[r] = (uint32_t)r;
  }
}

// sum over consecutive groups of four rows of 4 * (longest of the four): what the quads would hold if the
// rows kept their natural order (the quad padding estimate that decides whether sorting is worth its cost)
__global__ void __launch_bounds__(256)
natural_quad_slots_kernel(const int64_t* __restrict__ ptr, int64_t rows, unsigned long long* __restrict__ out) {
  __shared__ unsigned long long red[4];
  unsigned long long acc = 0;
  const int64_t nq = (rows + 3) / 4;
  for (int64_t qd = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; qd < nq; qd += (int64_t)gridDim.x * blockDim.x) {
    int64_t mx = 0;
    for (int g = 0; g < 4 && 4 * qd + g < rows; ++g) mx = max(mx, ptr[4 * qd + g + 1] - ptr[4 * qd + g]);
    acc += (unsigned long long)(4 * mx);
  }
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// Streaming fill: a workgroup owns a quad; every (quad, tile) segment is four contiguous source runs
// interleaved step by step ([k][g]) and padded with zero entries, written as one contiguous piece.
// SEG_LDS: the quad's four rows of seg are staged in LDS (few tiles: the table is small and the
// workgroups stay many per CU); otherwise every lane reads its row's bounds from global memory one
// tile step ahead (many tiles: a [4][tiles + 1] table would leave one or two workgroups per CU).
template <bool SEG_LDS, typename VT = float>
__global__ void __launch_bounds__(256)
quad_fill_runs_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const VT* __restrict__ val,
                      const uint64_t* __restrict__ packed, const int32_t* __restrict__ seg, const int32_t* __restrict__ blk_row0,
                      const uint32_t* __restrict__ perm, int nct, float inv_nct, int ldp_bytes,
                      const int64_t* __restrict__ chunk_off, const uint32_t* __restrict__ quad_off,
                      typename EntOf<VT>::type* __restrict__ ent) {
  typedef typename EntOf<VT>::type Ent;   // (f64: 16-byte entries; the packed rows are an f32 route)
  extern __shared__ int32_t sg_lds[];   // [4][nct + 1] the quad's rows of seg
  const int rb = blockIdx.x / Q_BLOCK_QUADS, qd = blockIdx.x % Q_BLOCK_QUADS;
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  if (4 * qd >= nrows) return;
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const int qrows = min(4, nrows - 4 * qd);
  const int64_t s0q = (int64_t)row0 + 4 * qd;   // first slot position of the quad; slot -> row through perm
  if (SEG_LDS) {
    for (int i = threadIdx.x; i < 4 * (nct + 1); i += 256) {
      const int g = i / (nct + 1);
      const int64_t rg = g < qrows ? (perm ? (int64_t)perm[s0q + g] : s0q + g) : 0;
      sg_lds[i] = g < qrows ? seg[rg * (nct + 1) + (i - g * (nct + 1))] : 0;
    }
    __syncthreads();
  }
  const int g = lane & 3, k0 = lane >> 2;   // lane -> (step k0 + 16*pass, row g)
  const int64_t myrow = g < qrows ? (perm ? (int64_t)perm[s0q + g] : s0q + g) : -1;
  const int64_t base = myrow >= 0 ? ptr[myrow] : 0;
  const int32_t* mysg = SEG_LDS ? sg_lds + g * (nct + 1) : seg + (myrow >= 0 ? myrow : 0) * (nct + 1);
  const int64_t* __restrict__ coff = chunk_off + (int64_t)rb * nct;
  const uint32_t* __restrict__ qoff = quad_off + ((int64_t)rb * Q_BLOCK_QUADS + qd) * nct;
  int s_nx = 0, e_nx = 0;
  int64_t d_nx = 0;
  if (!SEG_LDS && wave < nct) {
    s_nx = myrow >= 0 ? mysg[wave] : 0;
    e_nx = myrow >= 0 ? mysg[wave + 1] : 0;
    d_nx = coff[wave] + qoff[wave];
  }
  for (int t = wave; t < nct; t += 4) {
    int s0, len;
    Ent* dst;
    if (SEG_LDS) {
      s0 = mysg[t];
      len = mysg[t + 1] - s0;
      dst = ent + coff[t] + qoff[t];
    } else {
      s0 = s_nx;
      len = e_nx - s_nx;
      dst = ent + d_nx;
      if (t + 4 < nct) {
        s_nx = myrow >= 0 ? mysg[t + 4] : 0;
        e_nx = myrow >= 0 ? mysg[t + 5] : 0;
        d_nx = coff[t + 4] + qoff[t + 4];
      }
    }
    int qmax = max(len, __shfl_xor(len, 1));
    qmax = q_steps(max(qmax, __shfl_xor(qmax, 2)));
    for (int k = k0; k < qmax; k += 16) {
      Ent x{};
      if (k < len) {
        const int64_t e = base + s0 + k;
        int c;
        if constexpr (sizeof(VT) == 4) {
          if (packed) {
            const uint64_t pv = packed[e];
            c = (int)(pv >> 32);
            x.val = __uint_as_float((uint32_t)pv);
          } else {
            c = idx[e];
            x.val = val[e];
          }
        } else {
          c = idx[e];
          x.val = val[e];
        }
        int i, tm;
        divmod_small(c, nct, inv_nct, i, tm);
        x.off = (uint32_t)i * (uint32_t)ldp_bytes;
      }
      dst[k * 4 + g] = x;
    }
  }
}

// ---- the same format for A^T, built straight from A (no transposed CSR, no global sort) ----------
// Operator T = S^T for a source S (m x n CSR): T's rows are S's columns, T's interleaved tiles run
// over S's rows.  The chunk (block b of T rows, tile t) receives, from every source row r = t + i*nct,
// the contiguous run of its entries whose column lies in block b (found through segb), so one
// workgroup per chunk reads ~|block| * density entries per source row and owns the chunk's whole
// output region.  The rank of an entry inside its T row's segment (= the number of earlier source
// rows of this tile holding the same column) comes from per-column bit masks over the tile's rows,
// so the result is byte-identical to what the builder above makes from a transposed CSR.

// segb[r][b] = number of entries of row r with column < bounds[b]   (b = 0..nb)
__global__ void bound_index_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, int64_t rows,
                                   const int32_t* __restrict__ bounds, int nb, int32_t* __restrict__ segb) {
  const int64_t total = rows * (int64_t)(nb + 1);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int64_t r = i / (nb + 1);
    const int b = (int)(i - r * (nb + 1));
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    const int bound = bounds[b];
    int64_t lo = e0, hi = e1;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (idx[mid] < bound) lo = mid + 1; else hi = mid;
    }
    segb[i] = (int32_t)(lo - e0);
  }
}

constexpr int TQ_THREADS = 1024, TQ_GROUPS = TQ_THREADS / 16;
constexpr int TQ_NI = 5;   // source rows per 16-lane group: a tile holds at most 320 = 5 * 64 rows

// the run of source row i of this tile inside the chunk's column block: [e0, e1)
__device__ __forceinline__ void tquad_runs(int64_t (&e0)[TQ_NI], int64_t (&e1)[TQ_NI], const int64_t* __restrict__ ptr,
                                           const int32_t* __restrict__ segb, int nb, int b, int t, int nct, int nr) {
  const int grp = threadIdx.x / 16;
#pragma unroll
  for (int j = 0; j < TQ_NI; ++j) {
    const int i = grp + j * TQ_GROUPS;
    e0[j] = e1[j] = 0;
    if (i < nr) {
      const int64_t r = (int64_t)t + (int64_t)i * nct;
      const int64_t base = ptr[r];
      const int32_t* sb = segb + r * (nb + 1) + b;
      e0[j] = base + sb[0];
      e1[j] = base + sb[1];
    }
  }
}

// Per chunk: bit i of mask[column] <- source row i stores the column; the number of set bits below
// bit i is the entry's rank in its T row's segment (written to rank[], aligned with the source
// entries); the longest of a quad's four columns gives its steps.
__global__ void __launch_bounds__(TQ_THREADS)
tquad_count_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const int32_t* __restrict__ segb,
                   int64_t src_rows, const int32_t* __restrict__ blk_row0, int nb, int nct, int maskw,
                   uint16_t* __restrict__ rank, uint16_t* __restrict__ steps, uint32_t* __restrict__ quad_off,
                   uint32_t* __restrict__ wave_off, int64_t* __restrict__ chunk_size) {
  extern __shared__ uint32_t tq_lds[];
  uint32_t* mask = tq_lds;                                   // [QBLOCK_ROWS][maskw]
  uint32_t* pre = tq_lds + (size_t)QBLOCK_ROWS * maskw;       // [QBLOCK_ROWS][maskw]: set bits in the words before
  __shared__ uint32_t wsum[QBLOCK_ROWS / WAVE];
  __shared__ uint32_t qex[Q_BLOCK_QUADS];
  // launch order: all column blocks of one tile are neighbours, so the source rows they share
  // (and the segb lines) are fetched from HBM once and then hit in L2 / Infinity Cache
  const int t = blockIdx.x / nb, b = blockIdx.x % nb;
  const int64_t chunk = (int64_t)b * nct + t;
  const int c0 = blk_row0[b], nrows = blk_row0[b + 1] - c0;
  const int nquads = (nrows + 3) / 4;
  const int nr = (int)((src_rows - t + nct - 1) / nct);
  const int grp = threadIdx.x / 16, gl = threadIdx.x & 15;
  int64_t e0[TQ_NI], e1[TQ_NI];
  tquad_runs(e0, e1, ptr, segb, nb, b, t, nct, nr);
  for (int i = threadIdx.x; i < QBLOCK_ROWS * maskw; i += TQ_THREADS) mask[i] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < TQ_NI; ++j) {
    const int i = grp + j * TQ_GROUPS;
    for (int64_t e = e0[j] + gl; e < e1[j]; e += 16) atomicOr(&mask[(idx[e] - c0) * maskw + (i >> 5)], 1u << (i & 31));
  }
  __syncthreads();
  const int lr = threadIdx.x;
  int len = 0;
  if (lr < nrows) {
    for (int d = 0; d < maskw; ++d) {
      pre[lr * maskw + d] = (uint32_t)len;
      len += __popc(mask[lr * maskw + d]);
    }
  }
  // steps of every quad, exclusive scan of the padded quad sizes (wave scans + 8 wave totals)
  uint32_t padded = 0, incl = 0;
  const int lane = lr & (WAVE - 1), wv = lr / WAVE;
  if (lr < QBLOCK_ROWS) {
    int qmax = max(len, __shfl_xor(len, 1));
    qmax = q_steps(max(qmax, __shfl_xor(qmax, 2)));
    padded = (lr & 3) == 0 ? (uint32_t)qmax * 4u : 0u;
    if ((lr & 3) == 0) steps[chunk * Q_BLOCK_QUADS + lr / 4] = (uint16_t)qmax;
    incl = padded;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const uint32_t y = __shfl_up(incl, off);
      if (lane >= off) incl += y;
    }
    if (lane == WAVE - 1) wsum[wv] = incl;
  }
  __syncthreads();
  if (lr < QBLOCK_ROWS) {
    uint32_t before = 0;
    for (int w = 0; w < wv; ++w) before += wsum[w];
    incl += before;
    if ((lr & 3) == 0) {
      quad_off[chunk * Q_BLOCK_QUADS + lr / 4] = incl - padded;
      qex[lr / 4] = incl - padded;
    }
    if (lr == QBLOCK_ROWS - 1) chunk_size[chunk] = incl;
  }
  __syncthreads();
  if (lr < QWAVES) wave_off[chunk * QWAVES + lr] = qex[q_first(lr, nquads)];
  // ranks of this chunk's entries (idx is re-read: cache hits)
#pragma unroll
  for (int j = 0; j < TQ_NI; ++j) {
    const int i = grp + j * TQ_GROUPS;
    for (int64_t e = e0[j] + gl; e < e1[j]; e += 16) {
      const int w = (idx[e] - c0) * maskw + (i >> 5);
      rank[e] = (uint16_t)(pre[w] + __popc(mask[w] & ((1u << (i & 31)) - 1u)));
    }
  }
}

__global__ void __launch_bounds__(TQ_THREADS)
tquad_fill_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const float* __restrict__ val,
                  const uint16_t* __restrict__ rank, const int32_t* __restrict__ segb, int64_t src_rows,
                  const int32_t* __restrict__ blk_row0, int nb, int nct, int ldp_bytes,
                  const int64_t* __restrict__ chunk_off, const uint32_t* __restrict__ quad_off, Ent* __restrict__ ent) {
  __shared__ uint32_t qoff[Q_BLOCK_QUADS];
  const int t = blockIdx.x / nb, b = blockIdx.x % nb;
  const int64_t chunk = (int64_t)b * nct + t;
  const int c0 = blk_row0[b];
  const int nr = (int)((src_rows - t + nct - 1) / nct);
  const int grp = threadIdx.x / 16, gl = threadIdx.x & 15;
  int64_t e0[TQ_NI], e1[TQ_NI];
  tquad_runs(e0, e1, ptr, segb, nb, b, t, nct, nr);
  if (threadIdx.x < Q_BLOCK_QUADS) qoff[threadIdx.x] = quad_off[chunk * Q_BLOCK_QUADS + threadIdx.x];
  __syncthreads();
  Ent* __restrict__ out = ent + chunk_off[chunk];
#pragma unroll
  for (int j = 0; j < TQ_NI; ++j) {
    const int i = grp + j * TQ_GROUPS;
    for (int64_t e = e0[j] + gl; e < e1[j]; e += 16) {
      const int lr = idx[e] - c0;
      Ent x;
      x.off = (uint32_t)i * (uint32_t)ldp_bytes;
      x.val = val[e];
      out[qoff[lr >> 2] + (uint32_t)rank[e] * 4u + (uint32_t)(lr & 3)] = x;
    }
  }
}

// U consecutive steps of one quad: four rows advance together, one per lane group
template <int NV, int U>
__device__ __forceinline__ void quad_batch(v4f (&acc)[NV], const char* stage_lane, const char* tile_lane) {
  typedef unsigned int u2 __attribute__((ext_vector_type(2)));
  u2 e[U];
#if SAPCA_ABL & 2   // ablation: no entry reads
  {
    u2 f;
    f.x = (unsigned)(size_t)stage_lane & 0xff00u;
    f.y = 0x3f800000u;
    asm volatile("" : "+v"(f));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      e[u] = f;
      e[u].x += u * 512;
    }
  }
#else
#pragma unroll
  for (int u = 0; u < U; ++u) e[u] = *reinterpret_cast<const u2*>(stage_lane + u * (QGROUPS * 8));
#endif
  v4f w[U][NV];
#if SAPCA_ABL & 1   // ablation: no panel gathers
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int v = 0; v < NV; ++v) w[u][v] = v4f(__uint_as_float(e[u].x));
#else
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int v = 0; v < NV; ++v) w[u][v] = *reinterpret_cast<const v4f*>(tile_lane + e[u].x + v * 256);
#endif
#if SAPCA_ABL & 4   // ablation: no FMAs
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int v = 0; v < NV; ++v) asm volatile("" ::"v"(w[u][v]), "v"(e[u].y));
#else
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] += __uint_as_float(e[u].y) * w[u][v];
#endif
}

// panel rows t, t + nct, t + 2 nct, ... (clamped: slots past the last row are never referenced)
template <int N, int LDP>
__device__ __forceinline__ void load_tile_interleaved(v4f (&r)[N], const float* __restrict__ X, int ldx, int t, int nct,
                                                      int64_t panel_rows) {
  constexpr int CPR = LDP / 4;   // 16-byte chunks per panel row (LDP of the panel's ldx columns)
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int j = i * QTHREADS + (int)threadIdx.x;
    const int64_t grow = min((int64_t)(j / CPR) * nct + t, panel_rows - 1);
    r[i] = *reinterpret_cast<const v4f*>(X + grow * ldx + (j % CPR) * 4);
  }
}

#define SAPCA_QPREFETCH(CT)                                                                         \
  {                                                                                                 \
    const int64_t cidx_ = (int64_t)rb * nct + (CT);                                                 \
    const int64_t c_lo_ = chunk_off[cidx_];                                                         \
    load_tile_interleaved<NP_TILE, LDP>(pt, X, ldx, (CT), nct, panel_rows);                              \
    load_regs<NP_STAGE, QTHREADS>(ps, reinterpret_cast<const char*>(ent + c_lo_),                   \
                                  max(16, (int)(chunk_off[cidx_ + 1] - c_lo_) * 8));                \
  }
#define SAPCA_QBOOKKEEPING(CT)                                                                      \
  {                                                                                                 \
    const int64_t cidx_ = (int64_t)rb * nct + (CT);                                                 \
    cnt_next = lane < my_quads ? (int)steps[cidx_ * Q_BLOCK_QUADS + quad0 + lane] : 0;             \
    woff_next = wave_off[cidx_ * QWAVES + wave];                                                    \
  }

template <int LDP, bool PREFETCH, int TILE_B>
__global__ void __launch_bounds__(QTHREADS)
spmm_quad_kernel(const int32_t* __restrict__ blk_row0, const uint32_t* __restrict__ perm, int nct, int tc,
                 const int64_t* __restrict__ chunk_off,
                 const uint32_t* __restrict__ wave_off, const uint16_t* __restrict__ steps, const Ent* __restrict__ ent,
                 int64_t panel_rows, const float* __restrict__ X, int ldx, int nsplit, int tiles_per_split,
                 float* __restrict__ out, int64_t out_rows_total, int ldo, int ncols, const float* __restrict__ cvec,
                 int mode) {
  constexpr int NV = LDP / 64;            // b128 reads per panel row per lane
  constexpr int RG = q_rows_per_group(LDP);
  constexpr int STAGE_B = q_stage_bytes(TILE_B);
  constexpr int NP_TILE = TILE_B / (QTHREADS * 16), NP_STAGE = (STAGE_B + QTHREADS * 16 - 1) / (QTHREADS * 16);
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* tile = lds;
  char* stage = lds + TILE_B;
  const int rb = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int ct0 = sp * tiles_per_split, ct1 = min(nct, ct0 + tiles_per_split);
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const int g = lane / QLANES, q = lane % QLANES;
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  const int nquads = (nrows + 3) / 4;
  const int quad0 = q_first(wave, nquads), my_quads = q_first(wave + 1, nquads) - quad0;   // <= RG
  const int my_rows = min(nrows, 4 * (quad0 + my_quads)) - 4 * quad0;
  const char* tl = tile + q * 16;

  v4f acc[RG][NV];
#pragma unroll
  for (int i = 0; i < RG; ++i)
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[i][v] = v4f(0.f);

  v4f pt[NP_TILE], ps[NP_STAGE];
  int cnt_next = 0;
  unsigned woff_next = 0;
  if (ct0 < ct1) SAPCA_QBOOKKEEPING(ct0)
  if (PREFETCH && ct0 < ct1) SAPCA_QPREFETCH(ct0)
  for (int ct = ct0; ct < ct1; ++ct) {
    if (!(mode & 8) || ct == ct0) {
      __syncthreads();  // the previous tile's readers are done
      if (!PREFETCH) SAPCA_QPREFETCH(ct)
      store_regs<NP_TILE, QTHREADS>(pt, tile, TILE_B);
      store_regs<NP_STAGE, QTHREADS>(ps, stage, STAGE_B);
      __syncthreads();
    }
    const int cnt_v = cnt_next;
    const unsigned woff = woff_next;
    if (ct + 1 < ct1 && !(mode & 8)) SAPCA_QBOOKKEEPING(ct + 1)
    if (PREFETCH && ct + 1 < ct1 && !(mode & 10)) SAPCA_QPREFETCH(ct + 1)
    if (mode & 1) continue;
    if (my_quads > 0) {
      const char* sl = stage + (size_t)woff * 8 + g * 8;
#pragma unroll
      for (int j = 0; j < RG; ++j) {
        int n = __builtin_amdgcn_readlane(cnt_v, j);
        if constexpr (QWAVES == 8 && NV == 1) {   // 256 VGPRs per lane: 8 steps in flight per wave
          while (n >= 8) {
            quad_batch<NV, 8>(acc[j], sl, tl);
            sl += 8 * QGROUPS * 8;
            n -= 8;
          }
        }
        while (n >= 4) {
          quad_batch<NV, 4>(acc[j], sl, tl);
          sl += 4 * QGROUPS * 8;
          n -= 4;
        }
        if (n >= 2) {
          quad_batch<NV, 2>(acc[j], sl, tl);
          sl += 2 * QGROUPS * 8;
          n -= 2;
        }
        if (n) {
          quad_batch<NV, 1>(acc[j], sl, tl);
          sl += QGROUPS * 8;
        }
      }
    }
  }

  // lane (g, q) holds columns 4q..4q+3 (and 64+4q.. for 128-wide panels) of row 4j+g of its wave
  float* dst_base = out + (nsplit > 1 ? (int64_t)sp * out_rows_total * ldo : 0);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    const int col = v * 64 + q * 4;
    float cv[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cv[i] = (cvec && nsplit == 1 && col + i < ncols) ? cvec[col + i] : 0.f;
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      const int r = 4 * j + g;
      if (r < my_rows) {
        const int64_t sp = (int64_t)row0 + 4 * quad0 + r;   // slot position -> output row
        float* y = dst_base + (perm ? (int64_t)perm[sp] : sp) * ldo + col;
        if (col + 3 < ncols) {
          v4f o = acc[j][v];
          o.x -= cv[0]; o.y -= cv[1]; o.z -= cv[2]; o.w -= cv[3];
          *reinterpret_cast<v4f_a4*>(y) = o;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (col + i < ncols) y[i] = acc[j][v][i] - cv[i];
        }
      }
    }
  }
}
#undef SAPCA_QPREFETCH
#undef SAPCA_QBOOKKEEPING

template <int LDP, bool PREFETCH, int TILE_B>
void launch_quad(const TiledOp& op, const float* X, int ldx, float* out, int ldo, int ncols, const float* cvec, int mode,
                 hipStream_t s) {
  static LdsAttrState attr;
  ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_quad_kernel<LDP, PREFETCH, TILE_B>), LDS_TOTAL, attr);
  hipLaunchKernelGGL((spmm_quad_kernel<LDP, PREFETCH, TILE_B>), dim3((unsigned)(op.nrb * op.nsplit)), dim3(QTHREADS), LDS_TOTAL, s,
                     op.blk_row0, op.row_perm, op.nct, op.tc, op.chunk_off, op.wave_off, reinterpret_cast<const uint16_t*>(op.steps),
                     reinterpret_cast<const Ent*>(op.ent), op.cols, X, ldx, op.nsplit, op.tiles_per_split, out, op.rows, ldo,
                     ncols, cvec, mode);
}

// ---- the same sweep for f64 panels and values ------------------------------------------------------
// A 64-column f64 panel row is 512 bytes: the tile geometry of the 128-float panels (160 rows per
// 80 KiB tile, 4 rows per lane group, 256 rows per workgroup).  A lane holds the doubles 2q, 2q+1 and
// 32+2q, 32+2q+1 of its row (two ds_read_b128); entries are 16 bytes {offset, pad, f64 value}.
typedef double v2d __attribute__((ext_vector_type(2)));
typedef double v2d_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef unsigned int u4v __attribute__((ext_vector_type(4)));

template <int U>
__device__ __forceinline__ void quad_batch_f64(v2d (&acc)[2], const char* stage_lane, const char* tile_lane) {
  u4v e[U];
#pragma unroll
  for (int u = 0; u < U; ++u) e[u] = *reinterpret_cast<const u4v*>(stage_lane + u * (QGROUPS * 16));
  v2d w[U][2];
#pragma unroll
  for (int u = 0; u < U; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v) w[u][v] = *reinterpret_cast<const v2d*>(tile_lane + e[u].x + v * 256);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const double val = __hiloint2double((int)e[u].w, (int)e[u].z);
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[v] += val * w[u][v];
  }
}

template <int TILE_B>
__global__ void __launch_bounds__(QTHREADS)
spmm_quad_f64_kernel(const int32_t* __restrict__ blk_row0, const uint32_t* __restrict__ perm, int nct,
                     const int64_t* __restrict__ chunk_off, const uint32_t* __restrict__ wave_off,
                     const uint16_t* __restrict__ steps, const EntD* __restrict__ ent, int64_t panel_rows,
                     const double* __restrict__ X, int ldx, int nsplit, int tiles_per_split, double* __restrict__ out,
                     int64_t out_rows_total, int ldo, int ncols, const double* __restrict__ cvec) {
  constexpr int RG = q_rows_per_group(128);
  constexpr int STAGE_B = q_stage_bytes(TILE_B);
  constexpr int NP_TILE = TILE_B / (QTHREADS * 16), NP_STAGE = (STAGE_B + QTHREADS * 16 - 1) / (QTHREADS * 16);
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* tile = lds;
  char* stage = lds + TILE_B;
  const int rb = blockIdx.x / nsplit, sp = blockIdx.x % nsplit;
  const int ct0 = sp * tiles_per_split, ct1 = min(nct, ct0 + tiles_per_split);
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  const int g = lane / QLANES, q = lane % QLANES;
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  const int nquads = (nrows + 3) / 4;
  const int quad0 = q_first(wave, nquads), my_quads = q_first(wave + 1, nquads) - quad0;   // <= RG
  const int my_rows = min(nrows, 4 * (quad0 + my_quads)) - 4 * quad0;
  const char* tl = tile + q * 16;

  v2d acc[RG][2];
#pragma unroll
  for (int i = 0; i < RG; ++i)
#pragma unroll
    for (int v = 0; v < 2; ++v) acc[i][v] = v2d(0.0);

  for (int ct = ct0; ct < ct1; ++ct) {
    const int64_t cidx = (int64_t)rb * nct + ct;
    const int64_t c_lo = chunk_off[cidx];
    const int cnt_v = lane < my_quads ? (int)steps[cidx * Q_BLOCK_QUADS + quad0 + lane] : 0;
    const unsigned woff = wave_off[cidx * QWAVES + wave];
    {
      v4f pt[NP_TILE], ps[NP_STAGE];
      // the panel as 128 floats per row: the same interleaved 16-byte chunks
      load_tile_interleaved<NP_TILE, 128>(pt, reinterpret_cast<const float*>(X), 2 * ldx, ct, nct, panel_rows);
      load_regs<NP_STAGE, QTHREADS>(ps, reinterpret_cast<const char*>(ent + c_lo), max(16, (int)(chunk_off[cidx + 1] - c_lo) * 16));
      __syncthreads();  // the previous tile's readers are done
      store_regs<NP_TILE, QTHREADS>(pt, tile, TILE_B);
      store_regs<NP_STAGE, QTHREADS>(ps, stage, STAGE_B);
      __syncthreads();
    }
    if (my_quads > 0) {
      const char* sl = stage + (size_t)woff * 16 + g * 16;
#pragma unroll
      for (int j = 0; j < RG; ++j) {
        int n = __builtin_amdgcn_readlane(cnt_v, j);
        while (n >= 4) {
          quad_batch_f64<4>(acc[j], sl, tl);
          sl += 4 * QGROUPS * 16;
          n -= 4;
        }
        if (n >= 2) {
          quad_batch_f64<2>(acc[j], sl, tl);
          sl += 2 * QGROUPS * 16;
          n -= 2;
        }
        if (n) {
          quad_batch_f64<1>(acc[j], sl, tl);
          sl += QGROUPS * 16;
        }
      }
    }
  }

  // lane (g, q) holds columns 2q, 2q+1 and 32+2q, 32+2q+1 of row 4j+g of its wave
  double* dst_base = out + (nsplit > 1 ? (int64_t)sp * out_rows_total * ldo : 0);
#pragma unroll
  for (int v = 0; v < 2; ++v) {
    const int col = v * 32 + q * 2;
    double cv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) cv[i] = (cvec && nsplit == 1 && col + i < ncols) ? cvec[col + i] : 0.0;
#pragma unroll
    for (int j = 0; j < RG; ++j) {
      const int r = 4 * j + g;
      if (r < my_rows) {
        const int64_t spos = (int64_t)row0 + 4 * quad0 + r;   // slot position -> output row
        double* y = dst_base + (perm ? (int64_t)perm[spos] : spos) * ldo + col;
        if (col + 1 < ncols) {
          v2d o = acc[j][v];
          o.x -= cv[0]; o.y -= cv[1];
          *reinterpret_cast<v2d_a8*>(y) = o;
        } else if (col < ncols) {
          y[0] = acc[j][v].x - cv[0];
        }
      }
    }
  }
}

template <int TILE_B>
void launch_quad_f64(const TiledOp& op, const double* X, int ldx, double* out, int ldo, int ncols, const double* cvec,
                     hipStream_t s) {
  static LdsAttrState attr;
  ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_quad_f64_kernel<TILE_B>), LDS_TOTAL, attr);
  hipLaunchKernelGGL((spmm_quad_f64_kernel<TILE_B>), dim3((unsigned)(op.nrb * op.nsplit)), dim3(QTHREADS), LDS_TOTAL, s,
                     op.blk_row0, op.row_perm, op.nct, op.chunk_off, op.wave_off, reinterpret_cast<const uint16_t*>(op.steps),
                     reinterpret_cast<const EntD*>(op.ent), op.cols, X, ldx, op.nsplit, op.tiles_per_split, out, op.rows, ldo,
                     ncols, cvec);
}

template <int LDP, int SLOTS, bool PREFETCH>
void launch_tiled(const TiledOp& op, const float* X, float* out, int ldo, int ncols, const float* cvec, int mode,
                  hipStream_t s) {
  static LdsAttrState attr;
  ensure_dynamic_lds(reinterpret_cast<const void*>(&spmm_tiled_kernel<LDP, SLOTS, PREFETCH>), LDS_TOTAL, attr);
  hipLaunchKernelGGL((spmm_tiled_kernel<LDP, SLOTS, PREFETCH>), dim3((unsigned)(op.nrb * op.nsplit)), dim3(waves_for(SLOTS) * WAVE), LDS_TOTAL,
                     s, op.blk_row0, op.nct, op.tc, op.chunk_off, op.wave_off, op.steps,
                     reinterpret_cast<const Ent*>(op.ent), op.cols, X, op.nsplit, op.tiles_per_split, out, op.rows, ldo,
                     ncols, cvec, mode);
}


// Exclusive scans of one or two int64 arrays of count + 1 elements (the last input element is ignored; the last output is
// the total) by ONE workgroup, plus the maximum of the first array: the tables here have 1e4 .. 1e6 elements, and one
// launch replaces six of the library's (histogram / lookback / scan kernels for each of reduce and scan).
// Rounds of 1024 x PER elements: a thread loads its PER consecutive elements of the round together and keeps them in
// registers -- one round trip to memory per round instead of one per element, which is what the kernel's time is at
// these sizes; the running totals carry from round to round.
template <int PER, bool HAS_B>
__global__ void __launch_bounds__(1024)
small_scan_kernel(int64_t* __restrict__ a, int64_t* __restrict__ b, int64_t count, int64_t* __restrict__ out_max_total) {
  __shared__ int64_t wsum[2][16], wmax[16];
  const int tid = threadIdx.x, lane = tid & (WAVE - 1), wave = tid / WAVE;
  int64_t carry_a = 0, carry_b = 0, m = 0;
  for (int64_t base = 0; base < count; base += 1024 * PER) {
    const int64_t lo = min(count, base + (int64_t)tid * PER), hi = min(count, lo + PER);
    int64_t sa = 0, sb = 0, mx = 0;
    int64_t ra_[PER], rb_[HAS_B ? PER : 1];
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      ra_[u] = lo + u < hi ? a[lo + u] : 0;
      if constexpr (HAS_B) rb_[u] = lo + u < hi ? b[lo + u] : 0;
    }
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      sa += ra_[u];
      mx = max(mx, ra_[u]);
      if constexpr (HAS_B) sb += rb_[u];
    }
    // exclusive scan of the 1024 per-thread sums: inside a wave by shuffles, across the 16 waves through LDS
    int64_t ia = sa, ib = sb;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const int64_t ya = __shfl_up(ia, off), yb = __shfl_up(ib, off);
      if (lane >= off) { ia += ya; ib += yb; }
    }
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) mx = max(mx, __shfl_xor(mx, off));
    __syncthreads();   // (the previous round's readers of wsum are done)
    if (lane == WAVE - 1) { wsum[0][wave] = ia; wsum[1][wave] = ib; }
    if (lane == 0) wmax[wave] = mx;
    __syncthreads();
    int64_t ra = carry_a + ia - sa, rb = carry_b + ib - sb, ta = 0, tb = 0;
#pragma unroll 2
    for (int w = 0; w < 16; ++w) {
      if (w < wave) { ra += wsum[0][w]; rb += wsum[1][w]; }
      ta += wsum[0][w];
      tb += wsum[1][w];
      m = max(m, wmax[w]);
    }
    carry_a += ta;
    carry_b += tb;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      if (lo + u < hi) {
        a[lo + u] = ra;
        if constexpr (HAS_B) b[lo + u] = rb;
      }
      ra += ra_[u];
      if constexpr (HAS_B) rb += rb_[u];
    }
  }
  if (tid == 0) {
    a[count] = carry_a;
    if (HAS_B && b) b[count] = carry_b;
    if (out_max_total) { out_max_total[0] = m; out_max_total[1] = carry_a; }
  }
}

void launch_small_scan(int64_t* a, int64_t* b, int64_t count, int64_t* out_max_total, hipStream_t s) {
  const int64_t per = (count + 1023) / 1024;
  if (b && per <= 8) hipLaunchKernelGGL((small_scan_kernel<8, true>), dim3(1), dim3(1024), 0, s, a, b, count, out_max_total);
  else if (b) hipLaunchKernelGGL((small_scan_kernel<16, true>), dim3(1), dim3(1024), 0, s, a, b, count, out_max_total);
  else hipLaunchKernelGGL((small_scan_kernel<24, false>), dim3(1), dim3(1024), 0, s, a, b, count, out_max_total);
}

// ---- A^T's format straight from A through per-chunk buckets (no transposed CSR, no sort) -----------------------
// The transposition route moves every entry five times (pack, two radix passes, statistics, fill).  Here a histogram
// pass counts the entries of every (tile of A rows, column), which is all the quad counting needs; a scatter pass drops
// every entry of A into the bucket of its chunk (block of A^T rows, tile) as {slot in the block, row in the tile, value};
// one workgroup per chunk then ranks the entries of every A^T row by their row in the tile with per-slot bit masks (the
// ranks of the column-sorted transposed row) and writes the chunk's region of the format, padding included.  The bytes
// are those of the other routes.  The column statistics come from the finished chunks: every (tile, A^T row) segment is
// summed in its stored order, the per-tile partial sums are added in tile order (a fixed order: reproducible).
constexpr int ATD_MAX_COLS = 65536;       // the histogram keeps two 16-bit counters per LDS word
constexpr int ATD_THREADS = 1024;
constexpr int ATD_MASK_WORDS = 10;        // 320 rows of a tile

constexpr int ATD_HIST_THREADS = 512;   // (a tile per workgroup; eight waves: three or four tiles per CU at once, no second round at C2)
constexpr int ATD_HIST_THREADS_WIDE = 1024;   // histograms above 64 KiB leave one workgroup per CU: sixteen waves then (C5: 100 KB)

// With `bnd` (the gather fill below, natural row order of A^T): A^T's row blocks are contiguous column ranges -- block b
// holds the columns c with (int)((float)c * blk_scale) == b -- and bnd[t * tc + i][b] becomes the position in A's arrays of the first entry of row
// t + i * nct whose column lies in block b or behind (b = 0..nrb; the row's end for the blocks it does not reach), counted
// from the row's first entry (uint32: half the table of round 4's absolute int64 positions, and no memset of it).  A row is
// sorted by column, so these are the places where the block of the column changes: found in the registers that hold the
// row's indices for the histogram anyway.
__global__ void __launch_bounds__(ATD_HIST_THREADS_WIDE)
atd_hist_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, int64_t m, int nct, int tc, int64_t n2,
                uint16_t* __restrict__ cnt16, float blk_scale, int nrb, uint32_t* __restrict__ bnd, int64_t* __restrict__ disorder,
                unsigned long long* __restrict__ slots_to_clear) {
  if (blockIdx.x == 0 && threadIdx.x == 0 && slots_to_clear) *slots_to_clear = 0ull;   // (atd_rowlen_kernel, next on this stream, adds to it)
  // disorder: set when a row's entries leave the order the run ends rely on (a block after a later block: the caller handed
  // over rows whose columns do not ascend) -- the host then takes the bucket route, which maps every column through a table
  extern __shared__ uint32_t atd_h32[];   // n2 / 2 words: counters of columns 2w, 2w + 1
  const int t = blockIdx.x;
  const int nw = (int)(n2 / 2);
  const int nthreads = (int)blockDim.x, nwaves = nthreads / WAVE;
  for (int i = threadIdx.x; i < nw; i += nthreads) atd_h32[i] = 0u;
  __syncthreads();
  const int wave = threadIdx.x / WAVE, lane = threadIdx.x & (WAVE - 1);
  // (the next row's offsets are fetched while this row's indices are in flight: a row is one batch at C5's 500 entries)
  int64_t r = (int64_t)t + (int64_t)wave * nct;
  int64_t e0 = (wave < tc && r < m) ? ptr[r] : 0, e1 = (wave < tc && r < m) ? ptr[r + 1] : 0;
  for (int i = wave; i < tc; i += nwaves) {
    if (r >= m) {   // tile rows past the last row of A: empty runs (the table is not cleared beforehand)
      if (bnd)
        for (int j = lane; j <= nrb; j += WAVE) bnd[((int64_t)t * tc + i) * (nrb + 1) + j] = 0u;
      continue;     // (r only grows: every later row of this wave is past the end too)
    }
    const int64_t rn = r + (int64_t)nwaves * nct;
    const bool more = i + nwaves < tc && rn < m;
    const int64_t n0 = more ? ptr[rn] : 0, n1 = more ? ptr[rn + 1] : 0;
    int lastb = -1;   // block of the last entry seen in this row (wave-uniform)
    // a row's run ends side by side, as offsets from the row's first entry (32 bits: a row of A holds fewer than 2^32 entries)
    uint32_t* __restrict__ bnd_row = bnd ? bnd + ((int64_t)t * tc + i) * (nrb + 1) : nullptr;
    const int64_t row_e0 = e0;
    for (int64_t base = e0; base < e1; base += 8 * WAVE) {   // eight loads in flight per lane (a wave-uniform trip count: the
      const int64_t eb = base + lane;                         //  boundary search below talks across lanes)
      int c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = eb + u * WAVE < e1 ? idx[eb + u * WAVE] : -1;
      if (bnd) {   // (ahead of the LDS atomics: nothing here waits on the LDS queue)
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const bool valid = c[u] >= 0;   // (valid lanes are a prefix of the wave)
          const unsigned long long valids = __ballot(valid);
          if (valids == 0ull) break;
          // block of column c: (int)((float)c * scale) -- the partition is DEFINED by this expression (the host derives the
          // block table from the same single-precision product: build_tiled_at_direct), so three instructions decide it
          const int b = (int)((float)max(c[u], 0) * blk_scale);
          const int up = __builtin_amdgcn_update_dpp(0, b, 0x138, 0xf, 0xf, false);   // wave_shr:1 -- lane l reads lane l - 1
          const int prev = lane == 0 ? lastb : up;
          if (valid && b != prev)
            for (int j = prev + 1; j <= b; ++j) bnd_row[j] = (uint32_t)(eb + u * WAVE - row_e0);
          if (valid && b < prev) *disorder = 1;   // (rare, benign race: every writer stores the same value)
          lastb = __builtin_amdgcn_readlane(b, __builtin_popcountll(valids) - 1);
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c[u] >= 0) atomicAdd(&atd_h32[c[u] >> 1], 1u << (16 * (c[u] & 1)));   // (at most 320 rows per tile: no carry into the neighbour)
    }
    if (bnd)
      for (int j = lastb + 1 + lane; j <= nrb; j += WAVE) bnd_row[j] = (uint32_t)(e1 - row_e0);
    r = rn;
    e0 = n0;
    e1 = n1;
  }
  __syncthreads();
  uint32_t* out = reinterpret_cast<uint32_t*>(cnt16 + (int64_t)t * n2);
  for (int i = threadIdx.x; i < nw; i += nthreads) out[i] = atd_h32[i];
}

// len[c] = entries of column c (summed over the tiles); the caller scans it into A^T's row offsets.
// A workgroup takes 64 columns, its 16 waves a sixteenth of the tiles each.
__global__ void __launch_bounds__(1024)
atd_rowlen_kernel(const uint16_t* __restrict__ cnt16, int64_t n, int64_t n2, int nct, int64_t* __restrict__ len,
                  unsigned long long* __restrict__ natural_slots) {
  __shared__ uint32_t part[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  const int per = (nct + 15) / 16;
  uint32_t a = 0;
  if (c < n) {
    // (eight independent loads in flight: one dependent round trip per tile made this small kernel 66 us at C2's 625 tiles)
    const int t1 = min(nct, (grp + 1) * per);
    int t = grp * per;
    for (; t + 8 <= t1; t += 8) {
      uint32_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = cnt16[(int64_t)(t + u) * n2 + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; t < t1; ++t) a += cnt16[(int64_t)t * n2 + c];
  }
  part[grp][lane] = a;
  __syncthreads();
  if (grp == 0) {
    int64_t total = 0;
    if (c < n)
      for (int g = 0; g < 16; ++g) total += part[g][lane];
    if (c <= n) len[c] = total;   // (len[n] = 0: the scan turns it into the total)
    // what the quads would hold if A^T's rows kept their natural order: 4 x the longest of every four consecutive rows
    // (natural_quad_slots_kernel's sum, gathered here: the builder then needs neither that kernel nor the memset in front of it)
    if (natural_slots) {
      int64_t mx = max(total, __shfl_xor(total, 1));
      mx = max(mx, __shfl_xor(mx, 2));
      unsigned long long acc = (lane & 3) == 0 ? (unsigned long long)(4 * mx) : 0ull;
#pragma unroll
      for (int off = WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
      if (lane == 0 && acc) atomicAdd(natural_slots, acc);
    }
  }
}

// colmap[row of A^T] = block << 10 | slot inside the block
__global__ void atd_colmap_kernel(const int32_t* __restrict__ blk, int nrb, const uint32_t* __restrict__ perm, int64_t rows,
                                  uint32_t* __restrict__ colmap) {
  const int64_t sp = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (sp >= rows) return;
  int lo = 0, hi = nrb;   // last block with blk[b] <= sp
  while (lo + 1 < hi) {
    const int mid = (lo + hi) >> 1;
    if ((int64_t)blk[mid] <= sp) lo = mid; else hi = mid;
  }
  const int64_t row = perm ? (int64_t)perm[sp] : sp;
  colmap[row] = ((uint32_t)lo << 10) | (uint32_t)(sp - blk[lo]);
}

// one wave per row of A: its entries go to the buckets (block of their column, tile of the row).  Every contiguous run of
// lanes bound for the same bucket reserves its places with one atomic on the bucket's cursor; the atomics of a whole row
// (ten batches of 64 entries) are in flight together.  The order inside a bucket is whatever the atomics make it: the
// fill ranks entries by their row in the tile, not by their place in the bucket.
__global__ void __launch_bounds__(256)
atd_scatter_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const float* __restrict__ val, int64_t m, int nct,
                   int tc, const uint32_t* __restrict__ colmap, const int64_t* __restrict__ bucket_off, uint32_t* __restrict__ cursor,
                   uint2* __restrict__ bucket) {
  const int lane = threadIdx.x & (WAVE - 1);
  // workgroups are dealt to the eight XCDs round-robin: XCD x takes the tiles t = x (mod 8), so the partial lines of a bucket
  // (all its writers handle rows of one tile) meet in one L2 instead of being written back piecemeal from several
  const int xcd = blockIdx.x & 7;
  const int64_t wave = (int64_t)(blockIdx.x >> 3) * (blockDim.x / WAVE) + threadIdx.x / WAVE;
  const int64_t nwaves = (int64_t)(gridDim.x >> 3) * (blockDim.x / WAVE);
  const int tiles_x = (nct - xcd + 7) / 8;                 // tiles of this XCD
  const int64_t items = (int64_t)tiles_x * tc;             // (tile, row in the tile)
  constexpr int UB = 10;
  for (int64_t item = wave; item < items; item += nwaves) {
    const int i = (int)(item / tiles_x), t = xcd + 8 * (int)(item - (int64_t)i * tiles_x);
    const int64_t r = (int64_t)t + (int64_t)i * nct;
    if (r >= m) continue;
    const int64_t e1 = ptr[r + 1];
    for (int64_t eb = ptr[r]; eb < e1; eb += UB * WAVE) {
      int cc[UB];
      uint32_t cmv[UB];
      float vv[UB];
      int64_t base[UB];
      int mine[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int64_t e = eb + u * WAVE + lane;
        cc[u] = e < e1 ? idx[e] : -1;
        vv[u] = e < e1 ? val[e] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) cmv[u] = cc[u] >= 0 ? colmap[cc[u]] : 0xffffffffu;
      // run leaders reserve: nothing below waits for an atomic before all of them are issued
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const uint32_t rb = cmv[u] >> 10;
        const uint32_t prev = __shfl_up(rb, 1);
        const bool valid = cc[u] >= 0;
        const bool leader = valid && (lane == 0 || rb != prev);
        const unsigned long long leaders = __ballot(leader), valids = __ballot(valid);
        const unsigned long long upto = leaders & ((2ull << lane) - 1ull);   // leaders at or before this lane
        mine[u] = upto ? 63 - __builtin_clzll(upto) : 0;
        base[u] = 0;
        if (leader) {
          const unsigned long long above = leaders & ~((2ull << lane) - 1ull);   // leaders past this lane
          const int next = above ? __builtin_ctzll(above) : __builtin_popcountll(valids);   // (valid lanes are a prefix)
          const int64_t b = (int64_t)rb * nct + t;
          base[u] = bucket_off[b] + (int64_t)atomicAdd(&cursor[b], (uint32_t)(next - lane));
        }
      }
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const int64_t b0 = __shfl(base[u], mine[u]);
        if (cc[u] >= 0)
          bucket[b0 + (lane - mine[u])] = make_uint2(((cmv[u] & 1023u) << 9) | (uint32_t)i, __float_as_uint(vv[u]));
      }
    }
  }
}

// one workgroup per chunk: ranks from per-slot bit masks over the tile's rows, the chunk's region of the format
// assembled in LDS a few dozen quads at a time (written out in whole lines, padding included), the partial column sums
// of this tile from the assembled segments
constexpr int ATD_STAGE_ENT = 4096;   // entries of the LDS image (a quad holds at most 4 x 320)

// GATHER (natural row order of A^T: a block is a contiguous column range): the chunk's entries come straight from A -- in
// every row of the tile the columns of the block are one contiguous run, whose ends atd_hist_kernel left in `bnd` -- and
// there are no buckets and no scatter pass.  A thread takes every 1024th entry of the concatenated runs.
template <bool GATHER>
__global__ void __launch_bounds__(ATD_THREADS, 8)   // 64 VGPRs: two workgroups per CU
atd_fill_kernel(const uint2* __restrict__ bucket, const int64_t* __restrict__ bucket_off, const int32_t* __restrict__ blk_row0,
                const uint32_t* __restrict__ perm, int nct, int ldp_bytes, const int64_t* __restrict__ chunk_off,
                const uint32_t* __restrict__ quad_off, const uint16_t* __restrict__ steps, Ent* __restrict__ ent,
                double* __restrict__ psum, double* __restrict__ psq, int64_t n,
                const int32_t* __restrict__ a_idx, const float* __restrict__ a_val, const uint32_t* __restrict__ bnd,
                const int64_t* __restrict__ a_ptr, int64_t a_rows, int tc, int nrb_all) {
  // one pool: the rows' bit masks (40 KiB) and the image the chunk is assembled in (32 KiB); once the ranks are known the
  // masks are dead and the image takes the whole pool (chunks whose entries the threads hold in registers)
  constexpr int MASK_WORDS_ALL = QBLOCK_ROWS * ATD_MASK_WORDS;
  __shared__ __attribute__((aligned(16))) uint32_t pool[MASK_WORDS_ALL + 2 * ATD_STAGE_ENT];
  uint32_t* mask = pool;
  Ent* stage = reinterpret_cast<Ent*>(pool + MASK_WORDS_ALL);
  __shared__ uint32_t qoff_s[Q_BLOCK_QUADS + 1];
  __shared__ uint32_t wtot[Q_BLOCK_QUADS / WAVE];
  __shared__ uint16_t len_s[QBLOCK_ROWS];
  __shared__ int64_t run_lo[GATHER ? 32 * ATD_MASK_WORDS : 1];        // first entry of every tile row's run
  __shared__ uint32_t run_pre[GATHER ? 32 * ATD_MASK_WORDS + 1 : 1];  // entries of the runs before it
  __shared__ uint32_t rtot[GATHER ? 32 * ATD_MASK_WORDS / WAVE : 1];
  // GATHER: the chunks of one tile read neighbouring runs of the same rows of A -- they run back to back on one XCD (workgroups
  // are dealt to the eight XCDs round-robin), so a row's lines and pages are fetched once while its blocks pass
  int rb, t;
  if constexpr (GATHER) {   // grid: 8 x ceil(nct / 8) x nrb workgroups; XCD x takes the tiles t = x (mod 8), block after block
    const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    t = xcd + 8 * (j / nrb_all);
    rb = j % nrb_all;
    if (t >= nct) return;
  } else {
    rb = (int)(blockIdx.x / nct);
    t = (int)(blockIdx.x % nct);
  }
  const int64_t chunk = (int64_t)rb * nct + t;
  const int row0 = blk_row0[rb], nrows = blk_row0[rb + 1] - row0;
  const int nquads = (nrows + 3) / 4;
  for (int i = threadIdx.x; i < QBLOCK_ROWS * ATD_MASK_WORDS; i += ATD_THREADS) mask[i] = 0u;
  // entry offsets of the quads inside the chunk: the scan of their step counts (4 entries per step)
  if (threadIdx.x < Q_BLOCK_QUADS) {
    const int q = threadIdx.x, lane = q & (WAVE - 1);
    const uint32_t sz = q < nquads ? 4u * (uint32_t)steps[chunk * Q_BLOCK_QUADS + q] : 0u;
    uint32_t inc = sz;
#pragma unroll
    for (int off = 1; off < WAVE; off <<= 1) {
      const uint32_t y = __shfl_up(inc, off);
      if (lane >= off) inc += y;
    }
    qoff_s[q + 1] = inc;                       // inclusive, within the wave
    if (lane == WAVE - 1) wtot[q / WAVE] = inc;
  }
  if constexpr (GATHER) {
    if (threadIdx.x < 32 * ATD_MASK_WORDS) {   // (the same first waves; tc <= 320)
      const int i = threadIdx.x, lane = i & (WAVE - 1);
      // (bnd[row][block]: the chunks of one tile, run back to back on this XCD, read neighbouring words of the same lines)
      const uint32_t* br = bnd + ((int64_t)t * tc + i) * (nrb_all + 1) + rb;
      const int64_t arow = (int64_t)t + (int64_t)i * nct;                 // the row of A behind tile row i
      const int64_t e0 = (i < tc && arow < a_rows) ? a_ptr[arow] : 0;
      const int64_t lo = i < tc ? e0 + br[0] : 0;
      const int64_t hi = i < tc ? e0 + br[1] : 0;
      run_lo[i] = lo;
      uint32_t inc = hi > lo ? (uint32_t)(hi - lo) : 0u;
#pragma unroll
      for (int off = 1; off < WAVE; off <<= 1) {
        const uint32_t y = __shfl_up(inc, off);
        if (lane >= off) inc += y;
      }
      run_pre[i + 1] = inc;
      if (lane == WAVE - 1) rtot[i / WAVE] = inc;
    }
  }
  __syncthreads();
  if (threadIdx.x < Q_BLOCK_QUADS) {
    uint32_t add = 0;
    for (int w = 0; w < (int)threadIdx.x / WAVE; ++w) add += wtot[w];
    qoff_s[threadIdx.x + 1] += add;
    if (threadIdx.x == 0) qoff_s[0] = 0u;
  }
  if constexpr (GATHER) {
    if (threadIdx.x < 32 * ATD_MASK_WORDS) {
      uint32_t add = 0;
      for (int w = 0; w < (int)threadIdx.x / WAVE; ++w) add += rtot[w];
      run_pre[threadIdx.x + 1] += add;
      if (threadIdx.x == 0) run_pre[0] = 0u;
    }
  }
  __syncthreads();
  const int64_t b0 = GATHER ? 0 : bucket_off[chunk];
  const int64_t b1 = GATHER ? (int64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)run_pre[32 * ATD_MASK_WORDS]) : bucket_off[chunk + 1];
  constexpr int HOLD = 10;   // entries a thread keeps in registers (chunks of up to 10240 entries: one read of the bucket)
  // GATHER: the tile row of each of the first 10240 concatenated entries, written run by run into the (still unused) LDS
  // image -- a thread then finds its entries with one LDS read each instead of a search over the runs
  uint16_t* row_of = reinterpret_cast<uint16_t*>(stage);
  if constexpr (GATHER) {
    if (threadIdx.x < 3 * 32 * ATD_MASK_WORDS) {   // three threads per run
      const uint32_t i = threadIdx.x / 3u;
      const uint32_t k1 = min(run_pre[i + 1], (uint32_t)(HOLD * ATD_THREADS));
      for (uint32_t k = run_pre[i] + threadIdx.x % 3u; k < k1; k += 3u) row_of[k] = (uint16_t)i;
    }
    __syncthreads();
  }
  // entry f of the chunk as {slot in the block << 9 | row in the tile, value}
  auto fetch = [&](int64_t f, bool tabled) -> uint2 {
    if constexpr (GATHER) {
      int lo = 0;
      if (tabled) {
        lo = row_of[f];
      } else {
        int hi = 32 * ATD_MASK_WORDS - 1;   // the last row with run_pre[row] <= f
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if ((int64_t)run_pre[mid] <= f) lo = mid; else hi = mid - 1;
        }
      }
      const int64_t e = run_lo[lo] + (f - (int64_t)run_pre[lo]);
      return make_uint2(((uint32_t)(a_idx[e] - row0) << 9) | (uint32_t)lo, __float_as_uint(a_val[e]));
    } else {
      return bucket[f];
    }
  };
  Ent* dst = ent + chunk_off[chunk];
  uint2 kv[HOLD];
#pragma unroll
  for (int u = 0; u < HOLD; ++u) {
    const int64_t e = b0 + threadIdx.x + (int64_t)u * ATD_THREADS;
    kv[u] = e < b1 ? fetch(e, true) : make_uint2(0xffffffffu, 0u);
  }
#pragma unroll
  for (int u = 0; u < HOLD; ++u)
    if (kv[u].x != 0xffffffffu) {
      const uint32_t slot = kv[u].x >> 9, i = kv[u].x & 511u;
      atomicOr(&mask[slot * ATD_MASK_WORDS + (i >> 5)], 1u << (i & 31u));
    }
  for (int64_t e = b0 + threadIdx.x + (int64_t)HOLD * ATD_THREADS; e < b1; e += ATD_THREADS) {   // (longer chunks: the rest from memory)
    const uint32_t key = fetch(e, false).x;
    const uint32_t slot = key >> 9, i = key & 511u;
    atomicOr(&mask[slot * ATD_MASK_WORDS + (i >> 5)], 1u << (i & 31u));
  }
  __syncthreads();
  // Set bits in the mask words before each word, per slot (round 5: a rank was up to ten LDS reads and popcounts; now two reads).
  // The table takes the place of `row_of`, which nobody reads once the held entries are fetched; chunks too long to be held
  // assemble their image there later and keep the loop over the words.
  const bool held = b1 - b0 <= (int64_t)HOLD * ATD_THREADS;   // no entry is ranked again below: the masks' LDS joins the image
  uint16_t* pre = reinterpret_cast<uint16_t*>(stage);        // [QBLOCK_ROWS][ATD_MASK_WORDS]
  for (int slot = threadIdx.x; slot < QBLOCK_ROWS; slot += ATD_THREADS) {   // (and the entries of every row of the block in this tile)
    uint32_t run = 0;
#pragma unroll
    for (int w = 0; w < ATD_MASK_WORDS; ++w) {
      if (held) pre[slot * ATD_MASK_WORDS + w] = (uint16_t)run;
      run += __builtin_popcount(mask[slot * ATD_MASK_WORDS + w]);
    }
    len_s[slot] = (uint16_t)run;
  }
  __syncthreads();
  // rank of an entry inside its (A^T row, tile) segment = the rows of the tile before its own that hold the column
  auto rank_of = [&](uint32_t slot, uint32_t i) {
    const uint32_t* mk = mask + slot * ATD_MASK_WORDS;
    uint32_t rank = __builtin_popcount(mk[i >> 5] & ((1u << (i & 31u)) - 1u));
    if (held) return rank + (uint32_t)pre[slot * ATD_MASK_WORDS + (i >> 5)];
    for (uint32_t w = 0; w < (i >> 5); ++w) rank += __builtin_popcount(mk[w]);
    return rank;
  };
#pragma unroll
  for (int u = 0; u < HOLD; ++u) {   // the key becomes {place in the chunk (23 bits), row in the tile << 23}: the loop over the image below only places
    if (kv[u].x != 0xffffffffu) {
      const uint32_t slot = kv[u].x >> 9, i = kv[u].x & 511u;
      kv[u].x = (qoff_s[slot >> 2] + rank_of(slot, i) * 4u + (slot & 3u)) | (i << 23);
    }
    asm volatile("" ::: "memory");   // (one entry's LDS reads at a time: twelve unrolled copies in flight cost 128 VGPRs)
  }
  // The image is two halves: while one run of quads is written out (and summed), the next is assembled in the other half --
  // one barrier per run.  Slots are written exactly once: entries by the threads that hold them, the padding behind a
  // row's last entry by a thread per row.
  const uint32_t HALF = held ? (uint32_t)(MASK_WORDS_ALL / 2 + ATD_STAGE_ENT) / 2u : (uint32_t)ATD_STAGE_ENT / 2u;   // (a quad holds at most 4 x 320 entries)
  Ent* const image = held ? reinterpret_cast<Ent*>(pool) : stage;
  __syncthreads();   // (every rank is known: the image may take the masks' and the table's LDS)
  auto row_len = [&](int slot) { return (int)len_s[slot]; };
  int half = 0;
  for (int q0 = 0; q0 < nquads; half ^= 1) {
    // the next run of quads whose segments fit half the image
    int q1 = q0 + 1;
    {
      int lo = q0 + 1, hi = nquads;   // largest q1 with qoff[q1] - qoff[q0] <= HALF
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (qoff_s[mid] - qoff_s[q0] <= HALF) lo = mid; else hi = mid - 1;
      }
      q1 = __builtin_amdgcn_readfirstlane(lo);   // (every thread finds the same run: keep it in scalar registers)
    }
    Ent* st = image + (half ? HALF : 0u);
    const uint32_t o0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)qoff_s[q0]);
    const uint32_t size = (uint32_t)__builtin_amdgcn_readfirstlane((int)qoff_s[q1]) - o0;
    for (int slot = 4 * q0 + (int)threadIdx.x; slot < 4 * q1; slot += ATD_THREADS) {   // padding
      const uint32_t qo = qoff_s[slot >> 2], nsteps = (qoff_s[(slot >> 2) + 1] - qo) / 4u;
      const uint32_t o = qo - o0 + (uint32_t)(slot & 3);
#pragma clang loop vectorize(disable) unroll(disable)
      for (uint32_t k = slot < nrows ? (uint32_t)row_len(slot) : 0u; k < nsteps; ++k) st[o + 4u * k] = Ent{0u, 0.f};
    }
    auto place = [&](uint32_t slot, uint32_t i, uint32_t rank, uint32_t vbits) {
      const int q = (int)(slot >> 2);
      if (q < q0 || q >= q1) return;
      Ent x;
      x.off = i * (uint32_t)ldp_bytes;
      x.val = __uint_as_float(vbits);
      st[qoff_s[q] - o0 + rank * 4u + (slot & 3u)] = x;
    };
#pragma unroll
    for (int u = 0; u < HOLD; ++u) {   // (an empty hold has the place 2^23 - 1: never inside a run)
      const uint32_t at = (kv[u].x & 0x7fffffu) - o0;
      if (at < size) {
        Ent x;
        x.off = (kv[u].x >> 23) * (uint32_t)ldp_bytes;
        x.val = __uint_as_float(kv[u].y);
        st[at] = x;
      }
    }
    for (int64_t e = b0 + threadIdx.x + (int64_t)HOLD * ATD_THREADS; e < b1; e += ATD_THREADS) {
      const uint2 x = fetch(e, false);
      const uint32_t slot = x.x >> 9, i = x.x & 511u;
      const int q = (int)(slot >> 2);
      if (q >= q0 && q < q1) place(slot, i, rank_of(slot, i), x.y);
    }
    __syncthreads();   // (this half is complete; the other one was read out before the previous barrier)
    // (segments are multiples of 4 entries: 16-byte pieces)
    {
      const uint4* src4 = reinterpret_cast<const uint4*>(st);
      uint4* dst4 = reinterpret_cast<uint4*>(dst + o0);
      for (uint32_t i = threadIdx.x; i < size / 2; i += ATD_THREADS) dst4[i] = src4[i];
    }
    if (psum) {
      // two threads per row: the entries at even and at odd places of its segment, added in stored order, then the two halves
      const int pairs = 2 * (min(4 * q1, nrows) - 4 * q0);
      for (int pr = (int)threadIdx.x; pr < pairs; pr += ATD_THREADS) {
        const int slot = 4 * q0 + (pr >> 1), part = pr & 1;
        const int len = row_len(slot);
        const uint32_t o = qoff_s[slot >> 2] - o0 + (uint32_t)(slot & 3);
        double a = 0, b = 0;
#pragma clang loop vectorize(disable) unroll(disable)
        for (int k = part; k < len; k += 2) {
          const double v = (double)st[o + 4u * k].val;
          a += v;
          b += v * v;
        }
        a += __shfl_xor(a, 1);
        b += __shfl_xor(b, 1);
        if (part == 0) {
          const int64_t row = perm ? (int64_t)perm[row0 + slot] : (int64_t)row0 + slot;
          psum[(int64_t)t * n + row] = a;
          psq[(int64_t)t * n + row] = b;
        }
      }
    }
    q0 = q1;
  }
}

// sum[c] = the per-tile partial sums of column c added in a fixed order: sixteen runs of consecutive tiles, then the runs
__global__ void __launch_bounds__(1024)
atd_stats_reduce_kernel(const double* __restrict__ psum, const double* __restrict__ psq, int64_t n, int nct,
                        double* __restrict__ sum, double* __restrict__ sumsq) {
  __shared__ double pa[16][64], pb[16][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  const int per = (nct + 15) / 16;
  double a = 0, b = 0;
  if (c < n)
    for (int t = grp * per; t < min(nct, (grp + 1) * per); ++t) {
      a += psum[(int64_t)t * n + c];
      b += psq[(int64_t)t * n + c];
    }
  pa[grp][lane] = a;
  pb[grp][lane] = b;
  __syncthreads();
  if (grp == 0 && c < n) {
    double x = 0, y = 0;
    for (int g = 0; g < 16; ++g) {
      x += pa[g][lane];
      y += pb[g][lane];
    }
    sum[c] = x;
    sumsq[c] = y;
  }
}

}  // namespace

// ---------------------------------------------------------------------------------- host side
namespace {
// VT = float: every route below.  VT = double: the quad format through the direct fill only (the tile-major /
// packed-row routes and the LDS-staged fill carry f32 values); its 64-column panel rows are 512 bytes, so
// the geometry is that of the 128-float panels -- `ldp` below is the row length in FLOAT units.
// the bucket route to A^T's format: S describes A^T (row offsets only), the entries come from A itself
struct AtDirectSrc {
  const CsrView<float>* A;
  const uint16_t* cnt16;   // [tile][column] entry counts, row stride n2
  int64_t n2;
  DevBuf* scratch;         // buckets, bucket offsets, column map, partial sums
  double* stats;           // out: sum | sumsq per column of A (may be null)
  const uint32_t* bnd;     // gather fill: ends of every (A^T row block, A row) run for `nrb_nat` natural blocks (or null)
  int64_t nrb_nat;
  const std::vector<int32_t>* blk_nat;   // ... and where those blocks start (float_blocks below)
  bool slots_done = false;               // buf.misc holds the natural quads' slot count already (atd_rowlen_kernel)
};

// The natural blocks of the gather fill: block_of(c) = (int)((float)c * scale), with the scale taken down from nrb / n until
// the last column lands in block nrb - 1.  Host and device evaluate the same IEEE single-precision product, so the table
// below IS the device's partition; blocks differ from n / nrb columns by one at most and none is empty (scale <= 1).
void float_blocks(int64_t n, int64_t nrb, std::vector<int32_t>& blk, float& scale) {
  scale = (float)nrb / (float)n;
  auto block_of = [&](int64_t c) { return (int64_t)(int)((float)(int)c * scale); };
  while (block_of(n - 1) > nrb - 1) scale = std::nextafterf(scale, 0.0f);
  blk.assign((size_t)nrb + 1, (int32_t)n);
  int64_t b = 0;
  blk[0] = 0;
  for (int64_t c = 0; c < n; ++c) {
    const int64_t bc = block_of(c);
    while (b < bc) blk[(size_t)++b] = (int32_t)c;
  }
  while (b < nrb) blk[(size_t)++b] = (int32_t)n;   // (blocks past the last column: empty; cannot happen while nrb <= n)
}

// rows per block of the DPP-fed sweep's operators, and the natural (unsorted) partition of `op_rows` rows: block count and
// the split of the tile range over workgroups
int dq_block_rows(int64_t op_rows) {
  static const int dq_rows_env = dbg_env("SAPCA_DQ_BLOCK_ROWS") ? atoi(dbg_env("SAPCA_DQ_BLOCK_ROWS")) : 0;
  return dq_rows_env == 512 || dq_rows_env == 1024 ? dq_rows_env : (op_rows >= 1024 * 16 ? 1024 : 512);
}
void natural_partition(int64_t op_rows, int nct, int block_rows, int64_t& nrb, int& nsplit) {
  nrb = (op_rows + block_rows - 1) / block_rows;
  nsplit = 1;
  if (nrb >= 192) {
    nrb = round_up(nrb, 256);
  } else {
    // few row blocks (A^T): split the tile range so that (blocks x splits) lands just under a
    // multiple of the 256 CUs -- one workgroup per CU per round, no half-empty last round
    static const int split_wgs_env = dbg_env("SAPCA_SPLIT_WGS") ? atoi(dbg_env("SAPCA_SPLIT_WGS")) : 0;
    // 1024-row blocks fill a CU's LDS and registers alone: one workgroup per CU; the others run two per CU
    const int64_t split_wgs = split_wgs_env > 0 ? split_wgs_env : (block_rows > 512 ? 256 : 512);
    nsplit = (int)std::min<int64_t>(nct, std::max<int64_t>(1, split_wgs / nrb));
    const int64_t nrb_fit = split_wgs / nsplit;
    if (nrb_fit >= nrb && nrb_fit <= op_rows) nrb = nrb_fit;
  }
}

// workgroups of the staged fill: one per quad.  (The kernel can walk several quads per workgroup -- -DSAPCA_QF_WGS=n caps the
// grid: measured at C2 / C4 with 1024, 2048, 4096 workgroups, round 4: the same 2.0 ms preparation at C2 and +0.8 ms at C4
// (gpurun_out/r4_ab_qf.txt, r4_ab_c4.txt) -- the fill runs beside A^T's builder, which holds every wave slot of a CU while
// its workgroups are resident, so what this kernel gets are the gaps, and short-lived workgroups fill gaps best.)
inline unsigned qf_grid(int64_t nquads) {
#ifdef SAPCA_QF_WGS
  const int64_t wgs = SAPCA_QF_WGS;
#else
  const int64_t wgs = nquads;
#endif
  return (unsigned)std::max<int64_t>(1, std::min<int64_t>(nquads, wgs));
}

template <typename VT>
bool build_tiled_t(const CsrView<VT>& S, bool transposed, int ldp_elems, TiledOp& op, TiledBuffers& buf, hipStream_t s,
                   bool rows_tile_major, const uint64_t* packed_rows, bool allow_big_tile, bool seg_ready,
                   const AtDirectSrc* direct = nullptr) {
  typedef typename EntOf<VT>::type E;
  constexpr bool f32 = sizeof(VT) == 4;
  const int ldp = ldp_elems * (int)sizeof(VT) / 4;
  SAPCA_CHECK(ldp == 64 || ldp == 128, SAPCA_ERR_ARG, "tiled sweep: panel rows must be 256 or 512 bytes");
  op = TiledOp();
  if (!f32 && (transposed || packed_rows)) return false;
  if (S.rows == 0 || S.cols == 0 || S.nnz == 0) return false;
  // the operator is S, or S^T built straight from S (quad format only)
  const int64_t op_rows = transposed ? S.cols : S.rows, op_cols = transposed ? S.rows : S.cols;
  static const int fmt_env = dbg_env("SAPCA_TILED_FMT") ? atoi(dbg_env("SAPCA_TILED_FMT")) : 1;
  const bool quad = fmt_env == 1 || !f32;   // 1: a row per 16-lane group (default); 0: two half-waves per row
  int tile_bytes = quad ? Q_TILE_BYTES : TILE_BYTES;
  int tc = tile_bytes / (ldp * 4);
  if (transposed && (!quad || (tc + 63) / 64 > TQ_NI)) return false;
  if (rows_tile_major && !quad) return false;
  if (packed_rows && !(quad && !transposed && rows_tile_major)) return false;   // only the tile-major builders read packed rows
  int nct = (int)((op_cols + tc - 1) / tc);
  const int maskw = (Q_TILE_BYTES / (ldp * 4) + 31) / 32;   // only the transposed builder uses it (default split)
  // row blocks of <= 512 rows.  With enough rows the block count is a multiple of the 256 CUs (every
  // CU runs the same number of workgroups); with few rows (A^T) the tile range is split instead.
  static const int slots_env = dbg_env("SAPCA_TILED_SLOTS") ? atoi(dbg_env("SAPCA_TILED_SLOTS")) : 2;
  const int slots = (ldp == 64 && slots_env == 4) ? 4 : 2;
  const int waves = quad ? QWAVES : waves_for(slots);
  // (the DPP-fed sweep double-buffers the default 80 KiB tile and holds 8 or 16 row slots per lane group: f32 operators
  // with 64-column tiles keep that split, and take 1024-row blocks -- half the tile refills and barriers per entry --
  // when the operator has at least 16 of them (A^T of a tall matrix: the tile range is split over workgroups instead))
  const bool dq_candidate = f32 && quad && ldp == 64 && dbg_env("SAPCA_NO_DQ") == nullptr;
  const int block_rows = dq_candidate ? dq_block_rows(op_rows)
                         : quad ? QWAVES * QGROUPS * q_rows_per_group(ldp)
                                : waves * ((slots == 2 && ldp == 128) ? RW / 2 : (slots == 2 ? RW2 : RW));
  int stage_cap = quad ? q_stage_bytes(tile_bytes) / (int)sizeof(E) - WAVE : STAGE_ENTRIES;
  int64_t nrb = 0;
  int nsplit = 1;
  natural_partition(op_rows, nct, block_rows, nrb, nsplit);
  // the bigger tile (fewer, longer tile steps, less quad padding) when the chunks are expected to leave room
  // in the smaller staging area; operators fed by the tile-major transposition keep the default split
  // (their tile count is fixed before the transposition runs)
  // (f64 operators keep the default split too: the f64 DPP-fed sweep double-buffers the 80 KiB tile)
  const bool dq64_candidate = !f32 && quad && dbg_env("SAPCA_NO_DQ") == nullptr && dbg_env("SAPCA_NO_DQ_F64") == nullptr;
  if (quad && !transposed && !rows_tile_major && allow_big_tile && !dq_candidate && !dq64_candidate && dbg_env("SAPCA_TILE_DEFAULT") == nullptr) {
    const int tcb = Q_TILE_BYTES_BIG / (ldp * 4);
    const double est = 1.3 * (double)S.nnz / ((double)nrb * std::ceil((double)op_cols / tcb));
    if (est <= 0.78 * (q_stage_bytes(Q_TILE_BYTES_BIG) / (int)sizeof(E) - WAVE)) {
      tile_bytes = Q_TILE_BYTES_BIG;
      tc = tcb;
      nct = (int)((op_cols + tc - 1) / tc);
      stage_cap = q_stage_bytes(tile_bytes) / (int)sizeof(E) - WAVE;
      if (nsplit > 1) nsplit = (int)std::min<int64_t>(nct, nsplit);
    }
  }
  if (quad && !transposed && (op_cols >= (1 << 24) || nct > (rows_tile_major ? Q_MAX_TILES_RUNS : 4096))) return false;   // float-reciprocal tile arithmetic, LDS tables of the builders
  const float inv_nct = 1.0f / (float)nct;
  int32_t* d_seg = nullptr;
  if (direct && !(f32 && quad && !transposed && rows_tile_major && dq_candidate && tc <= 32 * ATD_MASK_WORDS && block_rows <= QBLOCK_ROWS))
    return false;
  if (!transposed && !direct) {
    d_seg = buf.seg.as<int32_t>((size_t)S.rows * (nct + 1));
    if (quad && rows_tile_major && seg_ready) {
      // at_stats_index() filled it in the statistics pass
    } else if (quad && rows_tile_major) {
      hipLaunchKernelGGL(tile_index_mod_kernel, dim3(grid_for(S.rows * (int64_t)(nct + 1), 256, 16384)), dim3(256), 0, s,
                         S.ptr, S.idx, packed_rows, S.rows, nct, inv_nct, d_seg);
    } else if (quad) {
      hipLaunchKernelGGL(tile_hist_kernel, dim3(grid_for(S.rows, 4, 8192)), dim3(256), (size_t)4 * nct * sizeof(uint32_t), s,
                         S.ptr, S.idx, S.rows, nct, inv_nct, d_seg);
    } else {
      if constexpr (f32) build_tile_index(S, tc, nct, d_seg, s);
    }
  }
  // Rows sorted by length, longest first (quad format built from a CSR): the four rows of a quad and the
  // quads of a wave then have similar lengths in every (interleaved) tile, which is what keeps the quad
  // padding and the per-tile barrier wait small on matrices with skewed row lengths (cell depth, gene
  // detection rate).  Blocks are cut from the sorted order with about equal entry counts (at most 512 rows).
  uint32_t* d_perm = nullptr;
  std::vector<uint32_t> sorted_len;
  bool sort_rows = quad && !transposed && dbg_env("SAPCA_NO_ROWSORT") == nullptr;
  // page-locked staging of this builder: [0] slots of the natural quads, [1..2] largest chunk | total, then the block table
  int64_t* pinned = static_cast<int64_t*>(buf.host.ensure((size_t)(8 + 65536 + 2) * sizeof(int64_t)));
  bool speculate = false;
  bool rows_in_disorder = false;   // (bucket route: the source's rows were found unsorted by the histogram pass)
  if (sort_rows && dbg_env("SAPCA_ROWSORT_ALWAYS") == nullptr) {
    // homogeneous rows pad little in their natural order: skip the sort (0.2 ms per operator at C2) unless the
    // natural quads would hold 10 % more slots than entries.  The count comes back with the chunk sizes below -- the
    // natural order is assumed until then (one wait for the device instead of two); a matrix that needs the sort
    // pays for a second round of counting.
    unsigned long long* d_slots = reinterpret_cast<unsigned long long*>(buf.misc.as<int64_t>(8)) + 4;
    if (!(direct && direct->slots_done)) {
      SAPCA_HIP(hipMemsetAsync(d_slots, 0, sizeof(unsigned long long), s));
      hipLaunchKernelGGL(natural_quad_slots_kernel, dim3(grid_for((S.rows + 3) / 4, 256, 1024)), dim3(256), 0, s, S.ptr, S.rows, d_slots);
    }
    SAPCA_HIP(hipMemcpyAsync(pinned, d_slots, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    speculate = true;
    sort_rows = false;
  }
  auto sort_now = [&] {
    uint32_t* d_len = buf.lens.as<uint32_t>((size_t)2 * S.rows);
    uint32_t* d_len_sorted = d_len + S.rows;
    uint32_t* d_iota = buf.perm.as<uint32_t>((size_t)2 * S.rows);
    d_perm = d_iota + S.rows;
    hipLaunchKernelGGL(row_len_iota_kernel, dim3(grid_for(S.rows, 256, 1 << 30)), dim3(256), 0, s, S.ptr, S.rows, d_len, d_iota);
    size_t sb = 0;
    SAPCA_HIP(rocprim::radix_sort_pairs_desc(nullptr, sb, d_len, d_len_sorted, d_iota, d_perm, (size_t)S.rows, 0u, 32u, s));
    char* tmp = static_cast<char*>(buf.tmp.ensure(sb + 256));
    SAPCA_HIP(rocprim::radix_sort_pairs_desc(tmp, sb, d_len, d_len_sorted, d_iota, d_perm, (size_t)S.rows, 0u, 32u, s));
    sorted_len.resize((size_t)S.rows);
    SAPCA_HIP(hipMemcpyAsync(sorted_len.data(), d_len_sorted, sorted_len.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipStreamSynchronize(s));
  };
  if (sort_rows) sort_now();
  // The entries of one (row block, tile) must fit the LDS staging.  Skewed inputs (a dense cluster
  // inside one tile) can exceed it: halve the rows per block and recount, a few times at most.
  int tiles_per_split = 0;
  int64_t nchunks = 0, max_chunk = 0, total = 0;
  int32_t* d_blk = nullptr;
  int64_t mid_row0 = -1;
  uint8_t* d_steps = nullptr;
  uint32_t* d_wave_off = nullptr;
  uint32_t* d_quad_off = nullptr;
  uint16_t* d_rank = nullptr;
  int64_t* d_chunk = nullptr;
  int64_t* d_raw = nullptr;   // (bucket route) stored entries per chunk, then their exclusive scan
  for (int attempt = 0;; ++attempt) {
    tiles_per_split = (nct + nsplit - 1) / nsplit;
    nsplit = (nct + tiles_per_split - 1) / tiles_per_split;
    std::vector<int32_t> blk;
    if (sort_rows) {
      // greedy cut of the sorted rows: close a block at block_rows rows or at the per-block entry budget
      const double budget = (double)S.nnz / (double)nrb;
      blk.push_back(0);
      double acc = 0;
      int in_block = 0;
      for (int64_t r = 0; r < op_rows; ++r) {
        acc += sorted_len[(size_t)r];
        ++in_block;
        const bool last = r + 1 == op_rows;
        if (last || in_block == block_rows || acc >= budget * (double)blk.size()) {
          blk.push_back((int32_t)(r + 1));
          in_block = 0;
        }
      }
      nrb = (int64_t)blk.size() - 1;
    } else if (direct && direct->bnd && nrb == direct->nrb_nat) {
      blk = *direct->blk_nat;   // (the partition the histogram pass recorded the run ends for)
    } else {
      blk.resize((size_t)nrb + 1);
      for (int64_t b = 0; b <= nrb; ++b) blk[(size_t)b] = (int32_t)(op_rows * b / nrb);
    }
    nchunks = nrb * nct;
    mid_row0 = blk[(size_t)(nrb / 2)];
    d_blk = buf.blk.as<int32_t>((size_t)nrb + 1);
    d_steps = buf.steps.as<uint8_t>((size_t)nchunks * (quad ? (size_t)Q_BLOCK_QUADS * 2 : (size_t)BLOCK_ROWS));
    d_wave_off = buf.wave_off.as<uint32_t>((size_t)nchunks * waves);
    if (quad) d_quad_off = buf.run.as<uint32_t>((size_t)nchunks * Q_BLOCK_QUADS);
    d_chunk = buf.chunk_off.as<int64_t>((size_t)nchunks + 1);
    if (blk.size() <= 2 * 65536) {   // (through the page-locked staging: the copy does not wait on a bounce buffer)
      std::memcpy(pinned + 8, blk.data(), blk.size() * sizeof(int32_t));
      SAPCA_HIP(hipMemcpyAsync(d_blk, pinned + 8, blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    } else {
      SAPCA_HIP(hipMemcpyAsync(d_blk, blk.data(), blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    }
    if (transposed) {
      d_seg = buf.seg.as<int32_t>((size_t)S.rows * (nrb + 1));
      hipLaunchKernelGGL(bound_index_kernel, dim3(grid_for(S.rows * (nrb + 1), 256, 16384)), dim3(256), 0, s, S.ptr, S.idx,
                         S.rows, d_blk, (int)nrb, d_seg);
      d_rank = buf.rank.as<uint16_t>((size_t)S.nnz);
      static LdsAttrState tq_attr;
      ensure_dynamic_lds(reinterpret_cast<const void*>(&tquad_count_kernel), (size_t)2 * QBLOCK_ROWS * maskw * sizeof(uint32_t), tq_attr);
      hipLaunchKernelGGL(tquad_count_kernel, dim3((unsigned)nchunks), dim3(TQ_THREADS),
                         (size_t)2 * QBLOCK_ROWS * maskw * sizeof(uint32_t), s, S.ptr, S.idx, d_seg, S.rows, d_blk, (int)nrb, nct,
                         maskw, d_rank, reinterpret_cast<uint16_t*>(d_steps), d_quad_off, d_wave_off, d_chunk);
    } else if (quad && direct) {
      d_raw = buf.rank.as<int64_t>((size_t)nchunks + 1);
      hipLaunchKernelGGL(quad_count_kernel, dim3((unsigned)(nrb * ((nct + QC_TILES - 1) / QC_TILES))), dim3(Q_BLOCK_QUADS), 0, s, d_seg, d_blk, d_perm, nct,
                         reinterpret_cast<uint16_t*>(d_steps), (uint32_t*)nullptr, d_wave_off, d_chunk, direct->cnt16, direct->n2, d_raw);
    } else if (quad)
      hipLaunchKernelGGL(quad_count_kernel, dim3((unsigned)(nrb * ((nct + QC_TILES - 1) / QC_TILES))), dim3(Q_BLOCK_QUADS), 0, s, d_seg, d_blk, d_perm, nct,
                         reinterpret_cast<uint16_t*>(d_steps), d_quad_off, d_wave_off, d_chunk);
    else if (slots == 2)
      hipLaunchKernelGGL((tiled_count_kernel<16, 2 * SAPCA_PADSTEPS>), dim3((unsigned)nchunks), dim3(BLOCK_ROWS), 0, s, d_seg, d_blk, nct,
                         d_steps, d_wave_off, d_chunk);
    else
      hipLaunchKernelGGL((tiled_count_kernel<8, 4>), dim3((unsigned)nchunks), dim3(BLOCK_ROWS), 0, s, d_seg, d_blk, nct,
                         d_steps, d_wave_off, d_chunk);
    // maximum chunk size (staging capacity check) and the exclusive scans of the chunk sizes (and, on the bucket route, of
    // the stored-entry counts) in one launch
    int64_t* d_max = buf.misc.as<int64_t>(8);
    launch_small_scan(d_chunk, d_raw, nchunks, d_max, s);
    int64_t* host = pinned + 1;
    SAPCA_HIP(hipMemcpyAsync(host, d_max, (direct ? 7 : 2) * sizeof(int64_t), hipMemcpyDeviceToHost, s));   // (direct: + the histogram's disorder flag)
    SAPCA_HIP(hipStreamSynchronize(s));  // blk goes out of scope; sizes needed on the host
    if (direct) rows_in_disorder = host[6] != 0;
    if (speculate) {
      speculate = false;
      if ((double)(unsigned long long)pinned[0] > 1.10 * (double)S.nnz) {   // the natural quads pad too much after all: sort, count again
        sort_rows = true;
        sort_now();
        attempt = -1;
        continue;
      }
    }
    max_chunk = host[0];
    total = host[1];
    if (dbg_env("SAPCA_DEBUG"))
      fprintf(stderr, "sapca: build_tiled%s rows %lld cols %lld nrb %lld nct %d split %d max_chunk %lld (cap %d) total %lld\n",
              transposed ? " (transposed source)" : "", (long long)op_rows, (long long)op_cols, (long long)nrb, nct, nsplit,
              (long long)max_chunk, stage_cap, (long long)total);
    if (max_chunk <= stage_cap || dq_candidate) break;   // the DPP-fed sweep stages no entries: nothing to fit
    if (tile_bytes == Q_TILE_BYTES_BIG)   // the estimate was too optimistic: take the default split instead of halving the row blocks
      return build_tiled_t<VT>(S, transposed, ldp_elems, op, buf, s, rows_tile_major, packed_rows, false, seg_ready);
    if (attempt == 3 || nrb * 2 > op_rows) return false;  // does not fit: the caller stays on the row kernel
    nrb *= 2;
    if (nsplit > 1) nsplit = std::max(1, nsplit / 2);
  }
  E* d_ent = reinterpret_cast<E*>(buf.ent.ensure((size_t)(total + ENT_SLACK) * sizeof(E)));
  bool aux_pending = false;
  op.rows = op_rows; op.cols = op_cols; op.ldp = ldp_elems; op.elem = (int)sizeof(VT); op.tc = tc; op.nct = nct; op.nrb = (int)nrb; op.block_rows = block_rows; op.max_chunk = max_chunk;
  op.nsplit = nsplit; op.tiles_per_split = tiles_per_split; op.total_entries = total; op.slots = slots; op.fmt = quad ? 1 : 0; op.tile_bytes = tile_bytes;
  op.mid_row0 = mid_row0;
  op.blk_row0 = d_blk; op.row_perm = d_perm; op.chunk_off = d_chunk; op.wave_off = d_wave_off; op.steps = d_steps; op.ent = d_ent;
  if constexpr (f32) {
    // the DPP-fed sweep's tables depend on the counts only: queued ahead of the fill (a small kernel that would otherwise wait
    // behind the other stream's fill for a free CU)
    op.valid = true;   // (dq_build_tables looks at it)
    if (!buf.aux) {
      SAPCA_HIP(hipStreamCreateWithFlags(&buf.aux, hipStreamNonBlocking));
      SAPCA_HIP(hipEventCreateWithFlags(&buf.aux_fork, hipEventDisableTiming));
      SAPCA_HIP(hipEventCreateWithFlags(&buf.aux_join, hipEventDisableTiming));
    }
    SAPCA_HIP(hipEventRecord(buf.aux_fork, s));              // counts, offsets and the block table are final here
    SAPCA_HIP(hipStreamWaitEvent(buf.aux, buf.aux_fork, 0));
    const bool dq_ok = dq_build_tables(op, buf, buf.aux);
    SAPCA_HIP(hipEventRecord(buf.aux_join, buf.aux));
    aux_pending = true;
    op.valid = false;
    if (!dq_ok && (block_rows > 512 || max_chunk > stage_cap)) {   // only the DPP-fed sweep reads such operators: the caller stays on the row kernel
      SAPCA_HIP(hipStreamWaitEvent(s, buf.aux_join, 0));
      return false;
    }
  }
  if constexpr (!f32) {
    if (quad) {   // the f64 DPP-fed sweep reads the same tables (an operator it cannot take stays on the staged-entry sweep)
      op.valid = true;
      (void)dq_build_tables(op, buf, s);
      op.valid = false;
    }
  }
  // quads that fit the LDS image on average: staged fill (coalesced stores, pads itself); otherwise
  // the direct fill over a zeroed buffer
  // f64 entries are 16 bytes: the LDS image holds QF_CAP_MIN of them (64 KiB), two workgroups per CU
  const int qf_cap_max = f32 ? QF_CAP_MAX : QF_CAP_MIN;
  const bool staged_fill = quad && !transposed && !rows_tile_major && dbg_env("SAPCA_FILL_DIRECT") == nullptr &&
                           (double)total <= 0.93 * qf_cap_max * ((double)op_rows / 4.0) && nct <= 768;   // (a quad above the image takes the direct route inside the kernel)
  // The image is sized to the operator's average quad + 12 % (round 5; rounds 1-4: 4096 or 6144 entries): a workgroup's LDS is
  // what limits the fill's residency (its waves sit out two memory round trips each), and C2's quads of 3 100 slots fit five
  // workgroups per CU instead of four.  A quad above the image takes the direct route inside the kernel, as before.
  static const bool qf_cap_fixed = dbg_env("SAPCA_QF_CAP_FIXED") != nullptr;
  const double quad_avg = (double)total / std::max(1.0, (double)op_rows / 4.0);
  const int qf_cap_fit = (int)std::min<int64_t>(qf_cap_max, std::max<int64_t>(2048, round_up((int64_t)(1.12 * quad_avg) + 64, 256)));
  const int qf_cap = (!f32 || qf_cap_fixed) ? ((double)total <= 0.85 * QF_CAP_MIN * ((double)op_rows / 4.0) ? QF_CAP_MIN : qf_cap_max) : qf_cap_fit;
  const bool runs_fill = quad && !transposed && rows_tile_major && nct <= Q_MAX_TILES_RUNS;
  if ((packed_rows || direct) && !runs_fill) {
    if (aux_pending) SAPCA_HIP(hipStreamWaitEvent(s, buf.aux_join, 0));
    return false;
  }
  if (staged_fill || runs_fill || direct) SAPCA_HIP(hipMemsetAsync(d_ent + total, 0, (size_t)ENT_SLACK * sizeof(E), s));
  else SAPCA_HIP(hipMemsetAsync(d_ent, 0, (size_t)(total + ENT_SLACK) * sizeof(E), s));
  size_t lds = (size_t)nct * sizeof(uint32_t);
  uint32_t* run_global = nullptr;
  if (!quad && lds > 48 * 1024) {
    run_global = buf.run.as<uint32_t>((size_t)nrb * waves * nct);
    lds = 0;
  }
  if constexpr (f32) {
  if (direct) {
    // bucket offsets, column -> (block, slot), the scatter of A's entries, one workgroup per chunk for the format
    const CsrView<float>& A = *direct->A;
    const size_t scan_bytes = 0;   // (d_raw was scanned together with the chunk sizes)
    // SAPCA_AT_BUCKETS=1: the bucket route also where the gather fill applies (A/B runs; the two write the same bytes)
    // (rows of A whose columns do not ascend -- include/sapca.h promises wrong numbers for them, not stray accesses: the run
    //  ends the histogram recorded are meaningless then, the bucket route needs none)
    const bool gather = direct->bnd != nullptr && d_perm == nullptr && nrb == direct->nrb_nat && !rows_in_disorder &&
                        dbg_env("SAPCA_AT_BUCKETS") == nullptr;
    const size_t a_col = round_up((size_t)op_rows * sizeof(uint32_t), 256), a_bucket = gather ? 0 : round_up((size_t)A.nnz * sizeof(uint2), 256);
    const size_t a_part = direct->stats ? round_up((size_t)nct * op_rows * sizeof(double), 256) : 0;
    const size_t a_scan = round_up(scan_bytes + 256, 256);
    char* base = static_cast<char*>(direct->scratch->ensure(a_col + a_bucket + 2 * a_part + a_scan + (size_t)nchunks * sizeof(uint32_t) + 256));
    uint32_t* d_colmap = reinterpret_cast<uint32_t*>(base);
    uint2* d_bucket = reinterpret_cast<uint2*>(base + a_col);
    double* d_psum = direct->stats ? reinterpret_cast<double*>(base + a_col + a_bucket) : nullptr;
    double* d_psq = direct->stats ? reinterpret_cast<double*>(base + a_col + a_bucket + a_part) : nullptr;
    if (gather) {
      // natural blocks are column ranges of A: every chunk reads its runs of A's rows itself (no buckets, no scatter pass)
      hipLaunchKernelGGL(atd_fill_kernel<true>, dim3((unsigned)(8 * ((nct + 7) / 8) * nrb)), dim3(ATD_THREADS), 0, s, (const uint2*)nullptr,
                         (const int64_t*)nullptr, d_blk, d_perm, nct, ldp * 4, d_chunk, d_quad_off, reinterpret_cast<const uint16_t*>(d_steps), d_ent,
                         d_psum, d_psq, op_rows, A.idx, A.val, direct->bnd, A.ptr, A.rows, tc, (int)nrb);
    } else {
      hipLaunchKernelGGL(atd_colmap_kernel, dim3((unsigned)((op_rows + 255) / 256)), dim3(256), 0, s, d_blk, (int)nrb, d_perm, op_rows,
                         d_colmap);
      uint32_t* d_cursor = reinterpret_cast<uint32_t*>(base + a_col + a_bucket + 2 * a_part + a_scan);
      SAPCA_HIP(hipMemsetAsync(d_cursor, 0, (size_t)nchunks * sizeof(uint32_t), s));
      hipLaunchKernelGGL(atd_scatter_kernel, dim3((unsigned)std::min<int64_t>(round_up((A.rows + 3) / 4, 8), 4096)), dim3(256), 0, s, A.ptr,
                         A.idx, A.val, A.rows, nct, tc, d_colmap, d_raw, d_cursor, d_bucket);
      hipLaunchKernelGGL(atd_fill_kernel<false>, dim3((unsigned)nchunks), dim3(ATD_THREADS), 0, s, d_bucket, d_raw, d_blk, d_perm, nct, ldp * 4,
                         d_chunk, d_quad_off, reinterpret_cast<const uint16_t*>(d_steps), d_ent, d_psum, d_psq, op_rows,
                         (const int32_t*)nullptr, (const float*)nullptr, (const uint32_t*)nullptr, (const int64_t*)nullptr, (int64_t)0, tc, (int)nrb);
    }
    if (direct->stats)
      hipLaunchKernelGGL(atd_stats_reduce_kernel, dim3((unsigned)((op_rows + 63) / 64)), dim3(1024), 0, s, d_psum, d_psq, op_rows, nct,
                         direct->stats, direct->stats + op_rows);
  } else if (transposed)
    hipLaunchKernelGGL(tquad_fill_kernel, dim3((unsigned)nchunks), dim3(TQ_THREADS), 0, s, S.ptr, S.idx, S.val, d_rank, d_seg,
                       S.rows, d_blk, (int)nrb, nct, ldp * 4, d_chunk, d_quad_off, d_ent);
  else if (runs_fill) {
    const int seg_lds_max = dbg_env("SAPCA_RUNS_SEG_LDS_MAX") ? atoi(dbg_env("SAPCA_RUNS_SEG_LDS_MAX")) : 1024;   // tiles; above: bounds from global memory
    if (nct <= seg_lds_max)
      hipLaunchKernelGGL(quad_fill_runs_kernel<true>, dim3((unsigned)(nrb * Q_BLOCK_QUADS)), dim3(256),
                         (size_t)4 * (nct + 1) * sizeof(int32_t), s, S.ptr, S.idx, S.val, packed_rows, d_seg, d_blk, d_perm, nct,
                         inv_nct, ldp * 4, d_chunk, d_quad_off, d_ent);
    else
      hipLaunchKernelGGL(quad_fill_runs_kernel<false>, dim3((unsigned)(nrb * Q_BLOCK_QUADS)), dim3(256), 0, s, S.ptr, S.idx, S.val,
                         packed_rows, d_seg, d_blk, d_perm, nct, inv_nct, ldp * 4, d_chunk, d_quad_off, d_ent);
  } else if (staged_fill)
    hipLaunchKernelGGL(quad_fill_staged_kernel<float>, dim3(qf_grid((int64_t)nrb * Q_BLOCK_QUADS)), dim3(256),
                       (size_t)qf_cap * sizeof(Ent) + ((size_t)7 * nct + 1) * sizeof(uint32_t), s, S.ptr, S.idx, S.val, d_seg, d_blk,
                       d_perm, nct, qf_cap, inv_nct, ldp * 4, d_chunk, d_quad_off, d_ent, (int)(nrb * Q_BLOCK_QUADS));
  else if (quad)
    hipLaunchKernelGGL(quad_fill_kernel<float>, dim3((unsigned)((S.rows + 3) / 4)), dim3(256), (size_t)4 * nct * sizeof(uint32_t), s,
                       S.ptr, S.idx, S.val, S.rows, d_blk, d_perm, (int)nrb, nct, inv_nct, ldp * 4, d_chunk, d_quad_off, d_ent);
  else if (slots == 2)
    hipLaunchKernelGGL((tiled_fill_kernel<16, 2 * SAPCA_PADSTEPS>), dim3((unsigned)(nrb * 16)), dim3(WAVE), lds, s, S.ptr, S.idx, S.val, d_seg,
                       d_blk, nct, tc, ldp * 4, d_chunk, d_wave_off, d_ent, run_global);
  else
    hipLaunchKernelGGL((tiled_fill_kernel<8, 4>), dim3((unsigned)(nrb * 8)), dim3(WAVE), lds, s, S.ptr, S.idx, S.val, d_seg,
                       d_blk, nct, tc, ldp * 4, d_chunk, d_wave_off, d_ent, run_global);
  } else {
    (void)lds; (void)run_global; (void)d_rank;
    if (runs_fill) {
      // rows grouped by tile (the tile-major transposition): a quad's run in a tile is contiguous in its rows
      const int seg_lds_max = dbg_env("SAPCA_RUNS_SEG_LDS_MAX") ? atoi(dbg_env("SAPCA_RUNS_SEG_LDS_MAX")) : 1024;
      if (nct <= seg_lds_max)
        hipLaunchKernelGGL((quad_fill_runs_kernel<true, double>), dim3((unsigned)(nrb * Q_BLOCK_QUADS)), dim3(256),
                           (size_t)4 * (nct + 1) * sizeof(int32_t), s, S.ptr, S.idx, S.val, (const uint64_t*)nullptr, d_seg, d_blk, d_perm, nct,
                           inv_nct, ldp * 4, d_chunk, d_quad_off, d_ent);
      else
        hipLaunchKernelGGL((quad_fill_runs_kernel<false, double>), dim3((unsigned)(nrb * Q_BLOCK_QUADS)), dim3(256), 0, s, S.ptr, S.idx, S.val,
                           (const uint64_t*)nullptr, d_seg, d_blk, d_perm, nct, inv_nct, ldp * 4, d_chunk, d_quad_off, d_ent);
    } else if (staged_fill) {
      const size_t fill_lds = (size_t)qf_cap * sizeof(E) + ((size_t)7 * nct + 1) * sizeof(uint32_t);
      static LdsAttrState attr;
      ensure_dynamic_lds(reinterpret_cast<const void*>(&quad_fill_staged_kernel<double>), fill_lds, attr);
      hipLaunchKernelGGL(quad_fill_staged_kernel<double>, dim3(qf_grid((int64_t)nrb * Q_BLOCK_QUADS)), dim3(256), fill_lds, s, S.ptr, S.idx,
                         S.val, d_seg, d_blk, d_perm, nct, qf_cap, inv_nct, ldp * 4, d_chunk, d_quad_off, d_ent, (int)(nrb * Q_BLOCK_QUADS));
    } else
    hipLaunchKernelGGL(quad_fill_kernel<double>, dim3((unsigned)((S.rows + 3) / 4)), dim3(256), (size_t)4 * nct * sizeof(uint32_t), s,
                       S.ptr, S.idx, S.val, S.rows, d_blk, d_perm, (int)nrb, nct, inv_nct, ldp * 4, d_chunk, d_quad_off, d_ent);
  }
  SAPCA_HIP(hipGetLastError());
  if (aux_pending) SAPCA_HIP(hipStreamWaitEvent(s, buf.aux_join, 0));
  op.valid = true;
  return true;
}
}  // namespace

bool build_tiled(const CsrView<float>& S, bool transposed, int ldp, TiledOp& op, TiledBuffers& buf, hipStream_t s,
                 bool rows_tile_major, const uint64_t* packed_rows, bool allow_big_tile, bool seg_ready) {
  return build_tiled_t<float>(S, transposed, ldp, op, buf, s, rows_tile_major, packed_rows, allow_big_tile, seg_ready);
}
bool build_tiled(const CsrView<double>& S, int ldp, TiledOp& op, TiledBuffers& buf, hipStream_t s, bool rows_tile_major) {
  return build_tiled_t<double>(S, false, ldp, op, buf, s, rows_tile_major, nullptr, true, false);
}

bool build_tiled_at_direct(const CsrView<float>& A, int ldp, TiledOp& op, TiledBuffers& buf, int64_t* at_ptr, double* stats,
                           DevBuf& scratch, hipStream_t s) {
  op = TiledOp();
  const bool off = dbg_env("SAPCA_AT_SORT") != nullptr;   // A/B: the transposition (radix sort) route
  if (off || ldp != 64 || A.rows == 0 || A.cols == 0 || A.nnz == 0 || A.cols > ATD_MAX_COLS || A.rows >= (1 << 24)) return false;
  const int64_t m = A.rows, n = A.cols;
  const int tc = Q_TILE_BYTES / (ldp * 4);
  const int nct = tiled_tile_count(m, ldp);
  if (tc > 32 * ATD_MASK_WORDS || nct > Q_MAX_TILES_RUNS) return false;
  const int64_t n2 = round_up(n, 2);
  // entry counts per (tile of A rows, column); their column totals are A^T's row lengths
  uint16_t* cnt16 = buf.seg.as<uint16_t>((size_t)nct * n2);   // (takes the place of the per-row tile index of the other routes)
  // the natural partition of A^T's rows (the one the format takes unless its rows have to be sorted by length): the
  // histogram pass leaves the ends of every (block, A row) run for the gather fill
  int64_t nrb_nat = 0;
  int nsplit_nat = 1;
  natural_partition(n, nct, dq_block_rows(n), nrb_nat, nsplit_nat);
  uint32_t* bnd = nullptr;
  if (dbg_env("SAPCA_AT_BUCKETS") == nullptr && nrb_nat <= 4096) {
    // (4 (nrb + 1) bytes per row of A: 132 MB at C4 -- round 4: 264 MB of int64, cleared before every fit; the histogram pass
    //  now writes every word itself.  A table that cannot be allocated is not a failed fit: the bucket route needs none)
    try {
      bnd = buf.bounds.as<uint32_t>((size_t)(nrb_nat + 1) * (size_t)nct * tc);
    } catch (const Error&) {
      (void)hipGetLastError();
      bnd = nullptr;
    }
  }
  const size_t hist_lds = (size_t)n2 * 2;
  static LdsAttrState hist_attr;
  ensure_dynamic_lds(reinterpret_cast<const void*>(&atd_hist_kernel), hist_lds, hist_attr);
  std::vector<int32_t> blk_nat;
  float blk_scale = 0.f;
  if (bnd) float_blocks(n, nrb_nat, blk_nat, blk_scale);
  int64_t* d_disorder = buf.misc.as<int64_t>(8) + 6;   // (read back with the builder's one host exchange)
  unsigned long long* d_slots = reinterpret_cast<unsigned long long*>(buf.misc.as<int64_t>(8)) + 4;   // (the natural quads' slots: build_tiled_t reads them back)
  SAPCA_HIP(hipMemsetAsync(d_disorder, 0, sizeof(int64_t), s));
  hipLaunchKernelGGL(atd_hist_kernel, dim3((unsigned)nct), dim3((size_t)n2 * 2 > 52 * 1024 ? ATD_HIST_THREADS_WIDE : ATD_HIST_THREADS),   // (above 52 KiB two workgroups share a CU: sixteen waves each)
                     hist_lds, s, A.ptr, A.idx, m, nct, tc, n2, cnt16, blk_scale, (int)nrb_nat, bnd, d_disorder, d_slots);
  hipLaunchKernelGGL(atd_rowlen_kernel, dim3((unsigned)((n + 64) / 64)), dim3(1024), 0, s, cnt16, n, n2, nct, at_ptr, d_slots);
  launch_small_scan(at_ptr, nullptr, n, nullptr, s);
  SAPCA_HIP(hipGetLastError());
  CsrView<float> At;
  At.rows = n; At.cols = m; At.nnz = A.nnz; At.ptr = at_ptr; At.idx = nullptr; At.val = nullptr;
  AtDirectSrc src{&A, cnt16, n2, &scratch, stats, bnd, nrb_nat, &blk_nat, true};
  return build_tiled_t<float>(At, false, ldp, op, buf, s, true, nullptr, true, false, &src);
}

void at_stats_index(const int64_t* ptr, const uint64_t* packed, int64_t rows, int64_t cols, int ldp, TiledBuffers& buf,
                    double* sum, double* sumsq, hipStream_t s) {
  if (rows == 0) return;
  const int nct = tiled_tile_count(cols, ldp);
  int32_t* d_seg = buf.seg.as<int32_t>((size_t)rows * (nct + 1));
  hipLaunchKernelGGL(at_stats_index_kernel, dim3(grid_for(rows * WAVE, 256, 4096)), dim3(256), 0, s, ptr, packed, rows, nct,
                     1.0f / (float)nct, sum, sumsq, d_seg);
  SAPCA_HIP(hipGetLastError());
}

int tiled_tile_count(int64_t cols, int ldp) {
  const int tc = Q_TILE_BYTES / (ldp * 4);
  return (int)((cols + tc - 1) / tc);
}

void spmm_tiled(const TiledOp& op, const float* X, int ldx, float* Y, int ldy, int ncols, const float* cvec, DevBuf& scratch,
                hipStream_t s, PanelSource<float>* keep) {
  SAPCA_CHECK(op.valid && op.elem == 4, SAPCA_ERR_ARG, "tiled sweep: operator not built");
  SAPCA_CHECK(ldx == op.ldp || (op.fmt == 1 && op.ldp == 64 && ldx % 64 == 0), SAPCA_ERR_ARG,
              "tiled sweep: panel leading dimension does not match the operator's tile geometry");
  static const int mode = dbg_env("SAPCA_TILED_MODE") ? atoi(dbg_env("SAPCA_TILED_MODE")) : 0;  // ablation switches (debug)
  // A 128-wide panel over the 64-wide tile geometry goes through in two column passes: twice the entry
  // traffic, but a tile holds twice the panel rows of the 128-wide geometry (half the tiles, less quad
  // padding, chunks that amortise their refill) -- what keeps wide panels on sparse operators (C5) staged.
  const int passes = ldx / op.ldp;
  float* part = op.nsplit > 1 ? scratch.as<float>((size_t)op.nsplit * op.rows * op.ldp) : nullptr;
  const bool pf = !(mode & 4);
  for (int pass = 0; pass < passes; ++pass) {
    const int c0 = pass * op.ldp;
    if (c0 >= ncols && pass > 0) break;
    const float* Xp = X + c0;
    const float* cv = cvec ? cvec + c0 : nullptr;
    const int ncp = std::min(ncols - c0, op.ldp);
    float* out = Y + c0;
    int ldo = ldy, nc = ncp;
    if (op.nsplit > 1) {
      out = part;
      ldo = op.ldp;
      nc = op.ldp;
    }
    static const bool force_staged = dbg_env("SAPCA_SWEEP_STAGED") != nullptr;   // A/B: the staged-entry quad sweep
    // (blocks of more than 512 rows exist only for the DPP-fed sweep: the switches below do not apply to them)
    const bool staged_ok = op.block_rows <= 512 && op.max_chunk <= (int64_t)(q_stage_bytes(op.tile_bytes) / 8 - WAVE);
    SAPCA_CHECK(op.fmt != 1 || staged_ok || dq_usable(op, ldx), SAPCA_ERR_ARG, "tiled sweep: this operator needs the DPP-fed sweep");
    if (op.fmt == 1 && dq_usable(op, ldx) && (!staged_ok || (!force_staged && mode == 0))) {
      launch_dq(op, Xp, ldx, out, ldo, nc, cv, s);
    } else if (op.fmt == 1) {
      const bool big = op.tile_bytes == Q_TILE_BYTES_BIG;
      if (op.ldp == 64 && pf && big) launch_quad<64, true, Q_TILE_BYTES_BIG>(op, Xp, ldx, out, ldo, nc, cv, mode, s);
      else if (op.ldp == 64 && pf) launch_quad<64, true, Q_TILE_BYTES>(op, Xp, ldx, out, ldo, nc, cv, mode, s);
      else if (op.ldp == 64 && big) launch_quad<64, false, Q_TILE_BYTES_BIG>(op, Xp, ldx, out, ldo, nc, cv, mode, s);
      else if (op.ldp == 64) launch_quad<64, false, Q_TILE_BYTES>(op, Xp, ldx, out, ldo, nc, cv, mode, s);
      else if (big) launch_quad<128, false, Q_TILE_BYTES_BIG>(op, Xp, ldx, out, ldo, nc, cv, mode, s);
      else launch_quad<128, false, Q_TILE_BYTES>(op, Xp, ldx, out, ldo, nc, cv, mode, s);
    } else if (op.ldp == 64 && op.slots == 4) {
      if (pf) launch_tiled<64, 4, true>(op, Xp, out, ldo, nc, cv, mode, s);
      else launch_tiled<64, 4, false>(op, Xp, out, ldo, nc, cv, mode, s);
    } else if (op.ldp == 64) {
      if (pf) launch_tiled<64, 2, true>(op, Xp, out, ldo, nc, cv, mode, s);
      else launch_tiled<64, 2, false>(op, Xp, out, ldo, nc, cv, mode, s);
    } else {
      launch_tiled<128, 2, false>(op, Xp, out, ldo, nc, cv, mode, s);
    }
    if (op.nsplit > 1) {
      const int64_t total = op.rows * (int64_t)op.ldp;
      if (keep && passes == 1 && !cv && ldy == op.ldp && ncp == op.ldp) {   // the caller's next pass over Y sums the slabs
        keep->parts = part;
        keep->nsplit = op.nsplit;
        keep->slab_stride = total;
        break;
      }
      hipLaunchKernelGGL(split_reduce_kernel<float>, dim3(grid_for(total, 256, 4096)), dim3(256), 0, s, part, op.nsplit, op.rows,
                         op.ldp, ncp, cv, Y + c0, ldy);
    }
  }
  SAPCA_HIP(hipGetLastError());
}

// Output rows [first, first + count) of the DPP-fed sweep, for callers that sweep an operator in pieces (the row-sharded
// A^T sweep of a multi-rank fit: a piece's all-reduce runs while the next piece is swept).  `piece` of `npieces`: the row
// blocks are cut into npieces runs; every piece cuts its tile range finely enough to occupy `wgs` workgroups, its partial
// slabs are summed in fixed order.  Requires natural row order (no row_perm: a block's rows are then a contiguous range of
// the output) and a 64-column panel.  Returns false when the operator cannot be swept this way (nothing is launched).
bool spmm_tiled_pieces_ok(const TiledOp& op, int npieces, int ldx) {
  return op.valid && op.elem == 4 && op.fmt == 1 && dq_usable(op, ldx) && ldx == op.ldp && op.row_perm == nullptr && npieces >= 1 && op.nrb >= npieces;
}

// bounds[p] = first output row of piece p (bounds[npieces] = rows): one small copy from the device, synchronous
void spmm_tiled_piece_bounds(const TiledOp& op, int npieces, std::vector<int64_t>& bounds, hipStream_t s) {
  if (npieces == 2 && op.mid_row0 >= 0) {   // (the builder kept the one boundary on the host: no copy, no wait)
    bounds = {0, op.mid_row0, op.rows};
    return;
  }
  std::vector<int32_t> b((size_t)npieces + 1);
  for (int p = 0; p <= npieces; ++p)
    SAPCA_HIP(hipMemcpyAsync(&b[(size_t)p], op.blk_row0 + (int64_t)op.nrb * p / npieces, sizeof(int32_t), hipMemcpyDeviceToHost, s));
  SAPCA_HIP(hipStreamSynchronize(s));
  bounds.assign(b.begin(), b.end());
}

void spmm_tiled_piece(const TiledOp& op, int piece, int npieces, int wgs, int64_t first_row, int64_t row_count, const float* X, int ldx, float* Y,
                      int ldy, int ncols, DevBuf& scratch, hipStream_t s) {
  SAPCA_CHECK(spmm_tiled_pieces_ok(op, npieces, ldx), SAPCA_ERR_ARG, "tiled sweep: operator cannot be swept in pieces");
  const int rb0 = (int)((int64_t)op.nrb * piece / npieces), rb1 = (int)((int64_t)op.nrb * (piece + 1) / npieces);
  int nsplit = std::max(1, std::min(op.nct, wgs / std::max(1, rb1 - rb0)));
  const int tps = (op.nct + nsplit - 1) / nsplit;
  nsplit = (op.nct + tps - 1) / tps;
  const int64_t* first = &first_row;
  const int64_t* count = &row_count;
  float* part = nsplit > 1 ? scratch.as<float>((size_t)nsplit * op.rows * op.ldp) : nullptr;
  if (nsplit > 1) {
    launch_dq_blocks(op, rb0, rb1, nsplit, tps, X, ldx, part, op.ldp, op.ldp, nullptr, s);
    const int64_t total = *count * (int64_t)op.ldp;
    // (slabs keep absolute row positions: the piece's rows start at first * ldp in every slab, slabs are op.rows * ldp apart)
    hipLaunchKernelGGL(split_reduce_rows_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, s, part + *first * op.ldp, nsplit,
                       op.rows * (int64_t)op.ldp, *count, op.ldp, ncols, Y + *first * ldy, ldy);
  } else {
    launch_dq_blocks(op, rb0, rb1, 1, op.nct, X, ldx, Y, ldy, ncols, nullptr, s);
  }
  SAPCA_HIP(hipGetLastError());
}

void spmm_tiled(const TiledOp& op, const double* X, int ldx, double* Y, int ldy, int ncols, const double* cvec, DevBuf& scratch,
                hipStream_t s, PanelSource<double>* keep) {
  SAPCA_CHECK(op.valid && op.elem == 8 && op.fmt == 1, SAPCA_ERR_ARG, "tiled sweep: no f64 operator built");
  SAPCA_CHECK(ldx >= op.ldp && ldx % op.ldp == 0, SAPCA_ERR_ARG,
              "tiled sweep: panel leading dimension does not match the operator's tile geometry");
  double* part = op.nsplit > 1 ? scratch.as<double>((size_t)op.nsplit * op.rows * op.ldp) : nullptr;
  const int passes = ldx / op.ldp;   // 128-column f64 panels: two column passes, as for f32
  for (int pass = 0; pass < passes; ++pass) {
    const int c0 = pass * op.ldp;
    if (c0 >= ncols && pass > 0) break;
    const double* Xp = X + c0;
    const double* cv = cvec ? cvec + c0 : nullptr;
    const int ncp = std::min(ncols - c0, op.ldp);
    double* out = Y + c0;
    int ldo = ldy, nc = ncp;
    if (op.nsplit > 1) {
      out = part;
      ldo = op.ldp;
      nc = op.ldp;
    }
    if (op.dq) launch_dq_f64(op, Xp, ldx, out, ldo, nc, cv, s);   // the DPP-fed sweep (spmm_dq.hip)
    else if (op.tile_bytes == Q_TILE_BYTES_BIG) launch_quad_f64<Q_TILE_BYTES_BIG>(op, Xp, ldx, out, ldo, nc, cv, s);
    else launch_quad_f64<Q_TILE_BYTES>(op, Xp, ldx, out, ldo, nc, cv, s);
    if (op.nsplit > 1) {
      const int64_t total = op.rows * (int64_t)op.ldp;
      if (keep && passes == 1 && !cv && ldy == op.ldp && ncp == op.ldp) {   // the caller's next pass over Y sums the slabs
        keep->parts = part;
        keep->nsplit = op.nsplit;
        keep->slab_stride = total;
        break;
      }
      hipLaunchKernelGGL(split_reduce_kernel<double>, dim3(grid_for(total, 256, 4096)), dim3(256), 0, s, part, op.nsplit, op.rows,
                         op.ldp, ncp, cv, Y + c0, ldy);
    }
  }
  SAPCA_HIP(hipGetLastError());
}

int tiled_geometry(int l) {
  static const bool wide = dbg_env("SAPCA_TILED_GEOM128") != nullptr;   // the 128-wide tile geometry for l > 64 (one pass)
  return (l > 64 && wide) ? 128 : 64;
}

}  // namespace k
}  // namespace sapca
