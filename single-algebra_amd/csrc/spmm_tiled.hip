// LDS-staged sparse x dense-panel sweep (f32) and the tile-major operator format it reads.
//
// Why: per stored entry the sweep reads 8 B of A but a whole panel row (256 B at l = 60).  Served
// from L2 that gather caps the sweep far below the HBM roofline (measured 5.8 % with
// the row kernel of spmm.hip).  Here the panel rows of one column tile are staged in LDS once per
// workgroup and every gather is an LDS read, while A streams from HBM exactly once, contiguously.
//
// Operator format ("tile-major", built once per fit by build_tiled()):
//   rows are cut into `nrb` blocks of <= 512 rows (one workgroup: 16 waves x <= 32 rows), columns
//   into `nct` tiles of TC panel rows (96 KiB of LDS).  For every (row block, column tile) the
//   entries are stored contiguously -- wave 0's rows first, row after row, each row's segment
//   padded with {0, 0.0f} to an even number of entries (one wave-step consumes two).  An entry is
//   {u32 byte offset of its panel row inside the LDS tile, f32 value}; a u8 table holds the number
//   of steps of every (row, tile) segment.
//
// Kernel (1024 threads, one workgroup per CU because of the 160 KiB of LDS):
//   per column tile:  barrier; panel tile + the block's entry chunk -> LDS (both contiguous in HBM,
//   prefetched into registers during the previous tile's compute); barrier; every wave walks its
//   rows.  The two half-waves take two consecutive entries of the row per step: each lane reads
//   its half's entry from the staged chunk (ds_read_b64, two distinct addresses per wave), then
//   its LDP/32 columns of that entry's panel row (ds_read_b64 / b128, conflict-free: a panel row
//   spans all 64 banks) and FMAs them into the row's accumulator (LDP/32 VGPRs).  Half-waves rather
//   than four quarter-waves keep the accumulator redundancy at 2x, so 512 rows share one staged
//   tile; wave-uniform entries (v_readlane, measured) cost ~6x more than this per-lane read.
//   Accumulators stay in VGPRs across all column tiles; the halves are summed at the end.  Tile ranges can be split over workgroups
//   (A^T has few rows): partial sums go to a slab that a second kernel adds in fixed order.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.h"

namespace sapca {
namespace k {

namespace {

constexpr int WAVE = 64;
constexpr int RW = 32;            // max rows per wave (register accumulators)
#ifndef SAPCA_RW2
#define SAPCA_RW2 32
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
  steps[(int64_t)blockIdx.x * BLOCK_ROWS + lr] = (uint8_t)nb;
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
#pragma unroll
  for (int u = 0; u < U; ++u) e[u] = *reinterpret_cast<const u2*>(stage_lane + u * (PAD * 8));
  typename Lane<VPL>::V w[U];
#pragma unroll
  for (int u = 0; u < U; ++u) w[u] = Lane<VPL>::load(tile_lane + e[u].x);
#pragma unroll
  for (int u = 0; u < U; ++u) acc += __uint_as_float(e[u].y) * w[u];
}

typedef float v4f __attribute__((ext_vector_type(4)));
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
    __syncthreads();  // the previous tile's readers are done
    if (!PREFETCH && (!(mode & 2) || ct == ct0)) SAPCA_PREFETCH(ct)
    store_regs<NP_TILE, THREADS>(pt, tile, TILE_BYTES);
    store_regs<NP_STAGE, THREADS>(ps, stage, STAGE_BYTES);
    __syncthreads();
    const int cnt_v = cnt_next;
    const unsigned woff = woff_next;
    if (ct + 1 < ct1) SAPCA_BOOKKEEPING(ct + 1)
    if (PREFETCH && ct + 1 < ct1) SAPCA_PREFETCH(ct + 1)
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
        if (n) {
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
__global__ void split_reduce_kernel(const float* __restrict__ part, int nsplit, int64_t rows, int ldo, int ncols,
                                    const float* __restrict__ cvec, float* __restrict__ out) {
  const int64_t total = rows * ldo;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int j = (int)(i % ldo);
    if (j >= ncols) continue;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += part[(int64_t)sp * total + i];
    out[i] = s - (cvec ? cvec[j] : 0.f);
  }
}

#undef SAPCA_PREFETCH
#undef SAPCA_BOOKKEEPING

template <int LDP, int SLOTS, bool PREFETCH>
void launch_tiled(const TiledOp& op, const float* X, float* out, int ldo, int ncols, const float* cvec, int mode,
                  hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    SAPCA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&spmm_tiled_kernel<LDP, SLOTS, PREFETCH>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL));
    attr = true;
  }
  hipLaunchKernelGGL((spmm_tiled_kernel<LDP, SLOTS, PREFETCH>), dim3((unsigned)(op.nrb * op.nsplit)), dim3(waves_for(SLOTS) * WAVE), LDS_TOTAL,
                     s, op.blk_row0, op.nct, op.tc, op.chunk_off, op.wave_off, op.steps,
                     reinterpret_cast<const Ent*>(op.ent), op.cols, X, op.nsplit, op.tiles_per_split, out, op.rows, ldo,
                     ncols, cvec, mode);
}

}  // namespace

// ---------------------------------------------------------------------------------- host side
bool build_tiled(const CsrView<float>& A, int ldp, TiledOp& op, TiledBuffers& buf, hipStream_t s) {
  SAPCA_CHECK(ldp == 64 || ldp == 128, SAPCA_ERR_ARG, "tiled sweep: panel leading dimension must be 64 or 128");
  op = TiledOp();
  if (A.rows == 0 || A.nnz == 0) return false;
  const int tc = TILE_BYTES / (ldp * 4);
  const int nct = (int)((A.cols + tc - 1) / tc);
  // row blocks of <= 512 rows.  With enough rows the block count is a multiple of the 256 CUs (every
  // CU runs the same number of workgroups); with few rows (A^T) the tile range is split instead.
  static const int slots_env = getenv("SAPCA_TILED_SLOTS") ? atoi(getenv("SAPCA_TILED_SLOTS")) : 2;
  const int slots = (ldp == 64 && slots_env == 4) ? 4 : 2;
  const int waves = waves_for(slots);
  const int block_rows = waves * ((slots == 2 && ldp == 128) ? RW / 2 : (slots == 2 ? RW2 : RW));
  int64_t nrb = (A.rows + block_rows - 1) / block_rows;
  int nsplit = 1;
  if (nrb >= 192) {
    nrb = round_up(nrb, 256);
  } else {
    // few row blocks (A^T): split the tile range so that (blocks x splits) lands just under a
    // multiple of the 256 CUs -- one workgroup per CU per round, no half-empty last round
    nsplit = (int)std::min<int64_t>(nct, std::max<int64_t>(1, 512 / nrb));
    const int64_t nrb_fit = 512 / nsplit;
    if (nrb_fit >= nrb && nrb_fit <= A.rows) nrb = nrb_fit;
  }
  int32_t* d_seg = buf.seg.as<int32_t>((size_t)A.rows * (nct + 1));
  build_tile_index(A, tc, nct, d_seg, s);
  // The entries of one (row block, tile) must fit the LDS staging.  Skewed inputs (a dense cluster
  // inside one tile) can exceed it: halve the rows per block and recount, a few times at most.
  int tiles_per_split = 0;
  int64_t nchunks = 0, max_chunk = 0, total = 0;
  int32_t* d_blk = nullptr;
  uint8_t* d_steps = nullptr;
  uint32_t* d_wave_off = nullptr;
  int64_t* d_chunk = nullptr;
  for (int attempt = 0;; ++attempt) {
    tiles_per_split = (nct + nsplit - 1) / nsplit;
    nsplit = (nct + tiles_per_split - 1) / tiles_per_split;
    std::vector<int32_t> blk((size_t)nrb + 1);
    for (int64_t b = 0; b <= nrb; ++b) blk[(size_t)b] = (int32_t)(A.rows * b / nrb);
    nchunks = nrb * nct;
    d_blk = buf.blk.as<int32_t>((size_t)nrb + 1);
    d_steps = buf.steps.as<uint8_t>((size_t)nchunks * BLOCK_ROWS);
    d_wave_off = buf.wave_off.as<uint32_t>((size_t)nchunks * waves);
    d_chunk = buf.chunk_off.as<int64_t>((size_t)nchunks + 1);
    SAPCA_HIP(hipMemcpyAsync(d_blk, blk.data(), blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, s));
    if (slots == 2)
      hipLaunchKernelGGL((tiled_count_kernel<16, 2>), dim3((unsigned)nchunks), dim3(BLOCK_ROWS), 0, s, d_seg, d_blk, nct,
                         d_steps, d_wave_off, d_chunk);
    else
      hipLaunchKernelGGL((tiled_count_kernel<8, 4>), dim3((unsigned)nchunks), dim3(BLOCK_ROWS), 0, s, d_seg, d_blk, nct,
                         d_steps, d_wave_off, d_chunk);
    SAPCA_HIP(hipMemsetAsync(d_chunk + nchunks, 0, sizeof(int64_t), s));
    // maximum chunk size (staging capacity check), then exclusive scan of the sizes
    size_t tmp_bytes = 0, tmp2 = 0;
    SAPCA_HIP(rocprim::exclusive_scan(nullptr, tmp_bytes, d_chunk, d_chunk, (int64_t)0, (size_t)nchunks + 1,
                                      rocprim::plus<int64_t>(), s));
    int64_t* d_max = buf.misc.as<int64_t>(8);
    SAPCA_HIP(rocprim::reduce(nullptr, tmp2, d_chunk, d_max, (int64_t)0, (size_t)nchunks, rocprim::maximum<int64_t>(), s));
    char* tmp = static_cast<char*>(buf.tmp.ensure(std::max(tmp_bytes, tmp2) + 256));
    SAPCA_HIP(rocprim::reduce(tmp, tmp2, d_chunk, d_max, (int64_t)0, (size_t)nchunks, rocprim::maximum<int64_t>(), s));
    SAPCA_HIP(rocprim::exclusive_scan(tmp, tmp_bytes, d_chunk, d_chunk, (int64_t)0, (size_t)nchunks + 1,
                                      rocprim::plus<int64_t>(), s));
    int64_t host[2] = {0, 0};
    SAPCA_HIP(hipMemcpyAsync(&host[0], d_max, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipMemcpyAsync(&host[1], d_chunk + nchunks, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    SAPCA_HIP(hipStreamSynchronize(s));  // blk goes out of scope; sizes needed on the host
    max_chunk = host[0];
    total = host[1];
    if (getenv("SAPCA_DEBUG"))
      fprintf(stderr, "sapca: build_tiled rows %lld cols %lld nrb %lld nct %d split %d max_chunk %lld (cap %d) total %lld\n",
              (long long)A.rows, (long long)A.cols, (long long)nrb, nct, nsplit, (long long)max_chunk, STAGE_ENTRIES,
              (long long)total);
    if (max_chunk <= STAGE_ENTRIES) break;
    if (attempt == 3 || nrb * 2 > A.rows) return false;  // does not fit: the caller stays on the row kernel
    nrb *= 2;
    if (nsplit > 1) nsplit = std::max(1, nsplit / 2);
  }
  Ent* d_ent = reinterpret_cast<Ent*>(buf.ent.ensure((size_t)(total + 2 * WAVE) * sizeof(Ent)));
  SAPCA_HIP(hipMemsetAsync(d_ent, 0, (size_t)(total + 2 * WAVE) * sizeof(Ent), s));
  size_t lds = (size_t)nct * sizeof(uint32_t);
  uint32_t* run_global = nullptr;
  if (lds > 48 * 1024) {
    run_global = buf.run.as<uint32_t>((size_t)nrb * waves * nct);
    lds = 0;
  }
  if (slots == 2)
    hipLaunchKernelGGL((tiled_fill_kernel<16, 2>), dim3((unsigned)(nrb * 16)), dim3(WAVE), lds, s, A.ptr, A.idx, A.val, d_seg,
                       d_blk, nct, tc, ldp * 4, d_chunk, d_wave_off, d_ent, run_global);
  else
    hipLaunchKernelGGL((tiled_fill_kernel<8, 4>), dim3((unsigned)(nrb * 8)), dim3(WAVE), lds, s, A.ptr, A.idx, A.val, d_seg,
                       d_blk, nct, tc, ldp * 4, d_chunk, d_wave_off, d_ent, run_global);
  SAPCA_HIP(hipGetLastError());
  op.rows = A.rows; op.cols = A.cols; op.ldp = ldp; op.tc = tc; op.nct = nct; op.nrb = (int)nrb;
  op.nsplit = nsplit; op.tiles_per_split = tiles_per_split; op.total_entries = total; op.slots = slots;
  op.blk_row0 = d_blk; op.chunk_off = d_chunk; op.wave_off = d_wave_off; op.steps = d_steps; op.ent = d_ent;
  op.valid = true;
  return true;
}

void spmm_tiled(const TiledOp& op, const float* X, float* Y, int ldy, int ncols, const float* cvec, DevBuf& scratch,
                hipStream_t s) {
  SAPCA_CHECK(op.valid, SAPCA_ERR_ARG, "tiled sweep: operator not built");
  static const int mode = getenv("SAPCA_TILED_MODE") ? atoi(getenv("SAPCA_TILED_MODE")) : 0;  // ablation switches (debug)
  float* out = Y;
  int ldo = ldy, nc = ncols;
  float* part = nullptr;
  if (op.nsplit > 1) {
    SAPCA_CHECK(ldy == op.ldp, SAPCA_ERR_ARG, "tiled sweep with a split tile range needs ldy == panel leading dimension");
    part = scratch.as<float>((size_t)op.nsplit * op.rows * op.ldp);
    out = part;
    ldo = op.ldp;
    nc = op.ldp;
  }
  const bool pf = !(mode & 4);
  if (op.ldp == 64 && op.slots == 4) {
    if (pf) launch_tiled<64, 4, true>(op, X, out, ldo, nc, cvec, mode, s);
    else launch_tiled<64, 4, false>(op, X, out, ldo, nc, cvec, mode, s);
  } else if (op.ldp == 64) {
    if (pf) launch_tiled<64, 2, true>(op, X, out, ldo, nc, cvec, mode, s);
    else launch_tiled<64, 2, false>(op, X, out, ldo, nc, cvec, mode, s);
  } else {
    launch_tiled<128, 2, false>(op, X, out, ldo, nc, cvec, mode, s);
  }
  if (op.nsplit > 1) {
    const int64_t total = op.rows * (int64_t)op.ldp;
    hipLaunchKernelGGL(split_reduce_kernel, dim3(grid_for(total, 256, 4096)), dim3(256), 0, s, part, op.nsplit, op.rows,
                       op.ldp, ncols, cvec, Y);
  }
  SAPCA_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace sapca
