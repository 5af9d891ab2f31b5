// One-time preparation of the device-resident operator: index narrowing, transposition,
// column statistics, mask compaction, tile index.  None of this is in the sweep loop.
//
// Reference semantics restated here:
//   row_sums on A^T            == <CsrMatrix as MatrixSum>::sum_col / sum_col_squared
//                                 (src/sparse/csr.rs:259-312, 558-608), f64 accumulation
//   compact_columns/select_rows == MaskedCSRMatrix::new (call site
//                                 src/dimred/pca/sparse_masked/mod.rs:313): kept columns
//                                 renumbered 0..n' in ascending order
#include <cstdlib>
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.h"

namespace sapca {
namespace k {

namespace {

constexpr int WAVE = 64;

__global__ void narrow_ptr_kernel(const uint64_t* __restrict__ in, int64_t* __restrict__ out, int64_t count) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (int64_t)in[i];
}

__global__ void narrow_idx_kernel(const uint64_t* __restrict__ in, int32_t* __restrict__ out, int64_t count,
                                  uint64_t n, int* __restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < count; i += stride) {
    uint64_t c = in[i];
    bad |= c >= n;
    out[i] = (int32_t)c;
  }
  if (bad) atomicOr(flag, 1);
}

// rowid[e] = row owning entry e; one wave per row, grid-stride.
__global__ void expand_rows_kernel(const int64_t* __restrict__ ptr, int64_t rows, int32_t* __restrict__ rowid) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e1 = ptr[r + 1];
    for (int64_t e = ptr[r] + lane; e < e1; e += WAVE) rowid[e] = (int32_t)r;
  }
}

// ptr[j] = first position in the ascending key array with key >= j, j in [0, nkeys].
template <typename K>
__global__ void lower_bound_kernel(const K* __restrict__ keys, int64_t count, int64_t nkeys,
                                   int64_t* __restrict__ ptr) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j > nkeys) return;
  int64_t lo = 0, hi = count;
  while (lo < hi) {
    int64_t mid = (lo + hi) >> 1;
    if ((int64_t)keys[mid] < j) lo = mid + 1; else hi = mid;
  }
  ptr[j] = lo;
}

template <typename T>
__global__ void gather_transposed_kernel(const uint32_t* __restrict__ perm, const int32_t* __restrict__ rowid,
                                         const T* __restrict__ val, int64_t count, int32_t* __restrict__ t_idx,
                                         T* __restrict__ t_val) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    const uint32_t e = perm[i];
    t_idx[i] = rowid[e];
    t_val[i] = val[e];
  }
}

// packed[e] = (row << 32) | bits(val[e]): the payload the radix sort carries for f32 matrices
__global__ void pack_rows_kernel(const int64_t* __restrict__ ptr, const float* __restrict__ val, int64_t rows,
                                 uint64_t* __restrict__ packed) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e1 = ptr[r + 1];
    for (int64_t e = ptr[r] + lane; e < e1; e += WAVE)
      packed[e] = ((uint64_t)(uint32_t)r << 32) | (uint64_t)__float_as_uint(val[e]);
  }
}

// Tile-major order of the source rows: row r of tile t = r mod nct, position i = r / nct, has rank
// rho(r) = (rows in tiles < t) + i.  Feeding the stable sort in that order makes every transposed row
// come out grouped by tile (and by position inside the tile), which turns the A^T format fill into a
// streaming copy.
__device__ __forceinline__ int64_t tile_major_rank(int64_t r, int64_t rows, int nct) {
  const int64_t base = rows / nct, rem = rows % nct;
  const int64_t t = r % nct, i = r / nct;
  return t * base + (t < rem ? t : rem) + i;
}
__global__ void permuted_len_kernel(const int64_t* __restrict__ ptr, int64_t rows, int nct, int64_t* __restrict__ lenp) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < rows) lenp[tile_major_rank(r, rows, nct)] = ptr[r + 1] - ptr[r];
  if (r == 0) lenp[rows] = 0;
}
template <typename K>
__global__ void pack_rows_permuted_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                          const float* __restrict__ val, int64_t rows, int nct,
                                          const int64_t* __restrict__ pptr, K* __restrict__ keys,
                                          uint64_t* __restrict__ packed) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    const int64_t d0 = pptr[tile_major_rank(r, rows, nct)] - e0;
    for (int64_t e = e0 + lane; e < e1; e += WAVE) {
      keys[d0 + e] = (K)idx[e];
      packed[d0 + e] = ((uint64_t)(uint32_t)r << 32) | (uint64_t)__float_as_uint(val[e]);
    }
  }
}

// f64 matrices: the sort carries a 12-byte (row, value) payload instead of a permutation that a random gather
// has to resolve afterwards (6.6 ms of an 9.3 ms transposition at C3); three 4-byte words, so that a sort pass moves
// 14 bytes per entry with 16-bit keys (fewer than 65536 columns) instead of 20
struct RowVal64 {
  uint32_t row, lo, hi;
};
static_assert(sizeof(RowVal64) == 12, "packed payload");
template <typename K>   // K = uint16_t: also writes the keys (the column indices) as 2-byte values; uint32_t: the sort reads A.idx itself
__global__ void pack_rows64_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const double* __restrict__ val,
                                   int64_t rows, K* __restrict__ keys, RowVal64* __restrict__ packed) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e1 = ptr[r + 1];
    for (int64_t e = ptr[r] + lane; e < e1; e += WAVE) {
      const unsigned long long b = (unsigned long long)__double_as_longlong(val[e]);
      packed[e] = RowVal64{(uint32_t)r, (uint32_t)b, (uint32_t)(b >> 32)};
      if constexpr (sizeof(K) == 2) keys[e] = (K)idx[e];
    }
  }
}
// the same with A's rows visited in tile-major order (pack_rows_permuted_kernel): every transposed row comes out grouped by tile
template <typename K>
__global__ void pack_rows64_permuted_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const double* __restrict__ val,
                                            int64_t rows, int nct, const int64_t* __restrict__ pptr, K* __restrict__ keys,
                                            RowVal64* __restrict__ packed) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    const int64_t d0 = pptr[tile_major_rank(r, rows, nct)] - e0;
    for (int64_t e = e0 + lane; e < e1; e += WAVE) {
      const unsigned long long b = (unsigned long long)__double_as_longlong(val[e]);
      packed[d0 + e] = RowVal64{(uint32_t)r, (uint32_t)b, (uint32_t)(b >> 32)};
      keys[d0 + e] = (K)idx[e];
    }
  }
}
__global__ void unpack_rows64_kernel(const RowVal64* __restrict__ packed, int64_t count, int32_t* __restrict__ t_idx,
                                     double* __restrict__ t_val) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    const RowVal64 p = packed[i];
    t_idx[i] = (int32_t)p.row;
    t_val[i] = __longlong_as_double((long long)(((unsigned long long)p.hi << 32) | (unsigned long long)p.lo));
  }
}

__global__ void unpack_rows_kernel(const uint64_t* __restrict__ packed, int64_t count, int32_t* __restrict__ t_idx,
                                   float* __restrict__ t_val) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    const uint64_t p = packed[i];
    t_idx[i] = (int32_t)(p >> 32);
    t_val[i] = __uint_as_float((uint32_t)p);
  }
}

template <typename T>
__global__ void row_sums_kernel(const int64_t* __restrict__ ptr, const T* __restrict__ val, int64_t rows,
                                double* __restrict__ sum, double* __restrict__ sumsq) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    double a = 0, b = 0;
    const int64_t e1 = ptr[r + 1];
    for (int64_t e = ptr[r] + lane; e < e1; e += WAVE) {
      const double v = (double)val[e];
      a += v;
      b += v * v;
    }
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

// The column map as the kernels below hold it in LDS: one bit per column (kept) and, per 32 columns, the number of kept
// columns before them -- o2m[c] = before[c / 32] + popcount(bits[c / 32] below c % 32) -- instead of a 4-byte gather from L2
// per stored entry.  `words` = ceil(n / 32); dynamic LDS: 2 * words * 4 bytes.
__device__ __forceinline__ void load_column_map(const int32_t* __restrict__ o2m, int64_t n, int words, uint32_t* bits, uint32_t* before) {
  for (int w = threadIdx.x; w < words; w += blockDim.x) {
    uint32_t b = 0u;
    int first = -1;
    for (int k = 0; k < 32; ++k) {
      const int64_t c = (int64_t)w * 32 + k;
      const int32_t mi = c < n ? o2m[c] : -1;
      if (mi >= 0) {
        b |= 1u << k;
        if (first < 0) first = mi;
      }
    }
    bits[w] = b;
    before[w] = first >= 0 ? (uint32_t)first : 0u;   // (kept columns are numbered in ascending order: the first kept one of the word)
  }
  __syncthreads();
}

template <typename T>
__global__ void count_kept_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, int64_t rows,
                                  const int32_t* __restrict__ o2m, int64_t n, int words, int64_t* __restrict__ cnt) {
  extern __shared__ uint32_t cmap_lds[];
  uint32_t* bits = cmap_lds;
  uint32_t* before = cmap_lds + words;
  if (words > 0) load_column_map(o2m, n, words, bits, before);   // (words == 0: the map does not fit LDS, gather o2m instead)
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    int c = 0;
    const int64_t e1 = ptr[r + 1];
    for (int64_t eb = ptr[r] + lane; eb < e1; eb += 8 * WAVE) {   // eight loads in flight per lane
      int col[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) col[u] = eb + u * WAVE < e1 ? idx[eb + u * WAVE] : -1;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (col[u] >= 0) c += words > 0 ? (int)((bits[col[u] >> 5] >> (col[u] & 31)) & 1u) : (int)(o2m[col[u]] >= 0);
    }
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if (lane == 0) cnt[r] = c;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) cnt[rows] = 0;
}

// kept entries -> (new_idx, new_val) at new_ptr[row]; optionally the dropped ones -> (drop_col = ORIGINAL column, drop_val),
// row after row (row r's start: the entries before it that were not kept, ptr[r] - new_ptr[r])
template <typename T>
__global__ void write_kept_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                  const T* __restrict__ val, int64_t rows, const int32_t* __restrict__ o2m, int64_t n, int words,
                                  const int64_t* __restrict__ new_ptr, int32_t* __restrict__ new_idx,
                                  T* __restrict__ new_val, int32_t* __restrict__ drop_col, T* __restrict__ drop_val,
                                  unsigned long long* __restrict__ amax_bits) {
  extern __shared__ uint32_t cmap_lds[];
  double vmax = 0.0;   // max |value| over EVERY stored entry the thread sees (kept or not), nan once one is not finite
  uint32_t* bits = cmap_lds;
  uint32_t* before = cmap_lds + words;
  if (words > 0) load_column_map(o2m, n, words, bits, before);
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    int64_t out = new_ptr[r];
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    int64_t dout = e0 - out;
    for (int64_t base = e0; base < e1; base += 4 * WAVE) {   // four batches of loads in flight
      int col[4], mi[4];
      T v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t e = base + u * WAVE + lane;
        col[u] = e < e1 ? idx[e] : -1;
        v[u] = e < e1 ? val[e] : (T)0;
      }
      if (amax_bits) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double a = fabs((double)v[u]);
          vmax = (a <= 1.7976931348623157e308 && vmax == vmax) ? fmax(vmax, a) : __longlong_as_double(0x7ff8000000000000ll);
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        mi[u] = -1;
        if (col[u] >= 0 && words > 0) {
          const uint32_t b = bits[col[u] >> 5], k = (uint32_t)col[u] & 31u;
          if ((b >> k) & 1u) mi[u] = (int)(before[col[u] >> 5] + (uint32_t)__popc(b & ((1u << k) - 1u)));
        } else if (col[u] >= 0) {
          mi[u] = o2m[col[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const unsigned long long keep = __ballot(mi[u] >= 0);
        const int bef = __popcll(keep & ((1ull << lane) - 1ull));
        if (mi[u] >= 0) {
          new_idx[out + bef] = mi[u];
          new_val[out + bef] = v[u];
        }
        out += __popcll(keep);
        if (drop_col) {
          const unsigned long long drop = __ballot(col[u] >= 0 && mi[u] < 0);
          if (col[u] >= 0 && mi[u] < 0) {
            const int64_t o = dout + __popcll(drop & ((1ull << lane) - 1ull));
            drop_col[o] = col[u];
            drop_val[o] = v[u];
          }
          dout += __popcll(drop);
        }
      }
    }
  }
  if (amax_bits) {   // bit patterns of non-negative doubles order like unsigned integers; the quiet nan's lies above all of them
    unsigned long long b = (unsigned long long)__double_as_longlong(vmax);
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) b = max(b, (unsigned long long)__shfl_xor((long long)b, off));
    if (lane == 0) atomicMax(amax_bits, b);
  }
}

__global__ void selected_lengths_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ rows,
                                        int64_t n_sel, int64_t* __restrict__ len) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_sel) {
    const int32_t r = rows[i];
    len[i] = ptr[r + 1] - ptr[r];
  } else if (i == n_sel) {
    len[i] = 0;
  }
}

template <typename T>
__global__ void copy_selected_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx,
                                     const T* __restrict__ val, const int32_t* __restrict__ rows, int64_t n_sel,
                                     const int64_t* __restrict__ new_ptr, int32_t* __restrict__ new_idx,
                                     T* __restrict__ new_val) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t i = wave; i < n_sel; i += nwaves) {
    const int32_t r = rows[i];
    const int64_t src = ptr[r], len = ptr[r + 1] - src, dst = new_ptr[i];
    for (int64_t t = lane; t < len; t += WAVE) {
      new_idx[dst + t] = idx[src + t];
      new_val[dst + t] = val[src + t];
    }
  }
}


// seg[r][t] = number of entries of row r with col < t*tile_cols (t = 0..n_tiles): one lane
// per (row, boundary) binary search; 64 boundaries of a row share a wave.
template <typename T>
__global__ void tile_index_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, int64_t rows,
                                  int tile_cols, int n_tiles, int32_t* __restrict__ seg) {
  const int64_t total = rows * (int64_t)(n_tiles + 1);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int64_t r = i / (n_tiles + 1);
    const int t = (int)(i - r * (n_tiles + 1));
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    const int64_t bound = (int64_t)t * tile_cols;
    int64_t lo = e0, hi = e1;
    while (lo < hi) {
      int64_t mid = (lo + hi) >> 1;
      if ((int64_t)idx[mid] < bound) lo = mid + 1; else hi = mid;
    }
    seg[i] = (int32_t)(lo - e0);
  }
}

__global__ void ptr_diff_kernel(const int64_t* __restrict__ ptr, int64_t rows, double* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < rows) out[i] = (double)(ptr[i + 1] - ptr[i]);
}

__global__ void histogram_kernel(const int32_t* __restrict__ idx, int64_t count, unsigned int* __restrict__ hist) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < count; i += stride) atomicAdd(&hist[idx[i]], 1u);
}

// The same counts through a private histogram per workgroup in LDS (4 bytes per column: up to 38 400 columns), flushed with one
// global integer atomic per non-empty counter: 1.2e8 global atomics on 20 000 addresses took 3 ms of a separate transform at
// C2's size, the LDS version 0.15.  Integer sums: order-free, reproducible.
__global__ void __launch_bounds__(1024)
histogram_lds_kernel(const int32_t* __restrict__ idx, int64_t count, int64_t n, unsigned int* __restrict__ hist) {
  extern __shared__ unsigned int lh[];
  for (int64_t c = threadIdx.x; c < n; c += blockDim.x) lh[c] = 0u;
  __syncthreads();
  // a contiguous slice per workgroup, 4 indices (16 bytes) per lane and load
  const int64_t per = ((count + gridDim.x - 1) / gridDim.x + 3) & ~(int64_t)3;
  const int64_t lo = (int64_t)blockIdx.x * per, hi = min(count, lo + per);
  for (int64_t i = lo + 4 * (int64_t)threadIdx.x; i < hi; i += 4 * (int64_t)blockDim.x) {
    if (i + 3 < hi && ((reinterpret_cast<uintptr_t>(idx + i) & 15) == 0)) {
      const int4 c4 = *reinterpret_cast<const int4*>(idx + i);
      atomicAdd(&lh[c4.x], 1u);
      atomicAdd(&lh[c4.y], 1u);
      atomicAdd(&lh[c4.z], 1u);
      atomicAdd(&lh[c4.w], 1u);
    } else {
      for (int64_t j = i; j < min(hi, i + 4); ++j) atomicAdd(&lh[idx[j]], 1u);
    }
  }
  __syncthreads();
  for (int64_t c = threadIdx.x; c < n; c += blockDim.x) {
    const unsigned int v = lh[c];
    if (v) atomicAdd(&hist[c], v);
  }
}

__global__ void u32_to_f64_kernel(const unsigned int* __restrict__ in, int64_t count, double* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) out[i] = (double)in[i];
}

inline int grid_for(int64_t work_items, int block, int cap = 8192) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

inline void exclusive_scan_i64(int64_t* data, int64_t count, DevBuf& scratch, size_t scratch_offset, hipStream_t s) {
  size_t bytes = 0;
  SAPCA_HIP(rocprim::exclusive_scan(nullptr, bytes, data, data, (int64_t)0, (size_t)count,
                                    rocprim::plus<int64_t>(), s));
  char* base = static_cast<char*>(scratch.ensure(scratch_offset + bytes + 256));
  SAPCA_HIP(rocprim::exclusive_scan(base + scratch_offset, bytes, data, data, (int64_t)0, (size_t)count,
                                    rocprim::plus<int64_t>(), s));
}

}  // namespace

void narrow_indices(const uint64_t* ptr64, const uint64_t* idx64, int64_t m, int64_t nnz, int64_t n, int64_t* ptr,
                    int32_t* idx, int* flag, hipStream_t s) {
  hipLaunchKernelGGL(narrow_ptr_kernel, dim3(grid_for(m + 1, 256, 1 << 30)), dim3(256), 0, s, ptr64, ptr, m + 1);
  if (nnz > 0)
    hipLaunchKernelGGL(narrow_idx_kernel, dim3(grid_for(nnz, 256, 4096)), dim3(256), 0, s, idx64, idx, nnz,
                       (uint64_t)n, flag);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void transpose_csr(const CsrView<T>& A, int64_t* t_ptr, int32_t* t_idx, T* t_val, DevBuf& scratch, hipStream_t s,
                   int tile_major_nct, const uint64_t** packed_rows_out) {
  if (packed_rows_out) *packed_rows_out = nullptr;
  const int64_t nnz = A.nnz;
  SAPCA_CHECK(nnz < (int64_t)0xFFFFFFFFll, SAPCA_ERR_ARG, "more than 2^32-1 stored entries per shard is not supported");
  if (nnz == 0) {
    SAPCA_HIP(hipMemsetAsync(t_ptr, 0, (A.cols + 1) * sizeof(int64_t), s));
    return;
  }
  int bits = 1;
  while ((1ll << bits) < A.cols) ++bits;
  if constexpr (sizeof(T) == 4) {
    // f32: the sort carries (row, value) itself -- no random gather pass afterwards
    size_t sort_bytes = 0;
    SAPCA_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr,
                                        (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)nnz, 0u, (unsigned)bits, s));
    const size_t a4 = (size_t)round_up(nnz * 4, 256), a8 = (size_t)round_up(nnz * 8, 256);
    const bool permuted = tile_major_nct > 1;
    const size_t extra = permuted ? a4 + (size_t)round_up((A.rows + 1) * 8, 256) : 0;   // keys in sort order, permuted row offsets
    size_t scan_bytes = 0;
    if (permuted)
      SAPCA_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0,
                                        (size_t)A.rows + 1, rocprim::plus<int64_t>(), s));
    char* base = static_cast<char*>(scratch.ensure(a4 + 2 * a8 + extra + std::max(sort_bytes, scan_bytes) + 256));
    uint32_t* keys_out = reinterpret_cast<uint32_t*>(base);
    uint64_t* packed = reinterpret_cast<uint64_t*>(base + a4);
    uint64_t* packed_out = reinterpret_cast<uint64_t*>(base + a4 + a8);
    uint32_t* keys_in = reinterpret_cast<uint32_t*>(base + a4 + 2 * a8);
    int64_t* pptr = reinterpret_cast<int64_t*>(base + a4 + 2 * a8 + a4);
    void* tmp = base + a4 + 2 * a8 + extra;
    if (permuted) {
      hipLaunchKernelGGL(permuted_len_kernel, dim3(grid_for(A.rows, 256, 1 << 30)), dim3(256), 0, s, A.ptr, A.rows,
                         tile_major_nct, pptr);
      SAPCA_HIP(rocprim::exclusive_scan(tmp, scan_bytes, pptr, pptr, (int64_t)0, (size_t)A.rows + 1,
                                        rocprim::plus<int64_t>(), s));
      if (bits <= 16) {
        // the keys are written here anyway: 16-bit ones make every sort pass move 10 instead of 12 bytes per entry
        uint16_t* k16_in = reinterpret_cast<uint16_t*>(keys_in);
        uint16_t* k16_out = reinterpret_cast<uint16_t*>(keys_out);
        size_t sort16 = 0;
        SAPCA_HIP(rocprim::radix_sort_pairs(nullptr, sort16, (const uint16_t*)nullptr, (uint16_t*)nullptr,
                                            (const uint64_t*)nullptr, (uint64_t*)nullptr, (size_t)nnz, 0u, (unsigned)bits, s));
        SAPCA_CHECK(sort16 <= std::max(sort_bytes, scan_bytes) + 256, SAPCA_ERR_NOMEM, "transpose: sort scratch too small");
        hipLaunchKernelGGL((pack_rows_permuted_kernel<uint16_t>), dim3(grid_for(A.rows * WAVE, 256, 4096)), dim3(256), 0, s,
                           A.ptr, A.idx, reinterpret_cast<const float*>(A.val), A.rows, tile_major_nct, pptr, k16_in, packed);
        SAPCA_HIP(rocprim::radix_sort_pairs(tmp, sort16, k16_in, k16_out, packed, packed_out, (size_t)nnz, 0u, (unsigned)bits, s));
        hipLaunchKernelGGL((lower_bound_kernel<uint16_t>), dim3(grid_for(A.cols + 1, 256, 1 << 30)), dim3(256), 0, s, k16_out,
                           nnz, A.cols, t_ptr);
      } else {
        hipLaunchKernelGGL((pack_rows_permuted_kernel<uint32_t>), dim3(grid_for(A.rows * WAVE, 256, 4096)), dim3(256), 0, s,
                           A.ptr, A.idx, reinterpret_cast<const float*>(A.val), A.rows, tile_major_nct, pptr, keys_in, packed);
        SAPCA_HIP(rocprim::radix_sort_pairs(tmp, sort_bytes, keys_in, keys_out, packed, packed_out, (size_t)nnz, 0u,
                                            (unsigned)bits, s));
        hipLaunchKernelGGL((lower_bound_kernel<uint32_t>), dim3(grid_for(A.cols + 1, 256, 1 << 30)), dim3(256), 0, s, keys_out,
                           nnz, A.cols, t_ptr);
      }
    } else {
      hipLaunchKernelGGL(pack_rows_kernel, dim3(grid_for(A.rows * WAVE, 256, 4096)), dim3(256), 0, s, A.ptr,
                         reinterpret_cast<const float*>(A.val), A.rows, packed);
      SAPCA_HIP(rocprim::radix_sort_pairs(tmp, sort_bytes, reinterpret_cast<const uint32_t*>(A.idx), keys_out, packed,
                                          packed_out, (size_t)nnz, 0u, (unsigned)bits, s));
      hipLaunchKernelGGL((lower_bound_kernel<uint32_t>), dim3(grid_for(A.cols + 1, 256, 1 << 30)), dim3(256), 0, s, keys_out,
                         nnz, A.cols, t_ptr);
    }
    if (permuted && packed_rows_out) {
      *packed_rows_out = packed_out;   // the caller consumes (row << 32 | value bits) directly; valid while `scratch` is untouched
    } else {
      hipLaunchKernelGGL(unpack_rows_kernel, dim3(grid_for(nnz, 256, 8192)), dim3(256), 0, s, packed_out, nnz, t_idx,
                         reinterpret_cast<float*>(t_val));
    }
    SAPCA_HIP(hipGetLastError());
    return;
  }
  if constexpr (sizeof(T) == 8) {
    static const bool gather_route = dbg_env("SAPCA_TRANSPOSE_GATHER") != nullptr;
    if (!gather_route) {
      const bool k16 = bits <= 16;
      size_t sb = 0;
      if (k16)
        SAPCA_HIP(rocprim::radix_sort_pairs(nullptr, sb, (const uint16_t*)nullptr, (uint16_t*)nullptr, (const RowVal64*)nullptr,
                                            (RowVal64*)nullptr, (size_t)nnz, 0u, (unsigned)bits, s));
      else
        SAPCA_HIP(rocprim::radix_sort_pairs(nullptr, sb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const RowVal64*)nullptr,
                                            (RowVal64*)nullptr, (size_t)nnz, 0u, (unsigned)bits, s));
      const size_t a4 = (size_t)round_up(nnz * 4, 256), a12 = (size_t)round_up(nnz * 12, 256);
      const bool permuted = tile_major_nct > 1;
      size_t scan_bytes = 0;
      if (permuted)
        SAPCA_HIP(rocprim::exclusive_scan(nullptr, scan_bytes, (int64_t*)nullptr, (int64_t*)nullptr, (int64_t)0,
                                          (size_t)A.rows + 1, rocprim::plus<int64_t>(), s));
      const size_t a_pptr = permuted ? (size_t)round_up((A.rows + 1) * 8, 256) : 0;
      char* base = static_cast<char*>(scratch.ensure(2 * a4 + 2 * a12 + a_pptr + std::max(sb, scan_bytes) + 256));
      uint32_t* keys_out = reinterpret_cast<uint32_t*>(base);
      uint32_t* keys_in = reinterpret_cast<uint32_t*>(base + a4);      // (16-bit keys, or the keys in tile-major row order)
      RowVal64* packed = reinterpret_cast<RowVal64*>(base + 2 * a4);
      RowVal64* packed_out = reinterpret_cast<RowVal64*>(base + 2 * a4 + a12);
      int64_t* pptr = reinterpret_cast<int64_t*>(base + 2 * a4 + 2 * a12);
      void* tmp = base + 2 * a4 + 2 * a12 + a_pptr;
      if (permuted) {
        hipLaunchKernelGGL(permuted_len_kernel, dim3(grid_for(A.rows, 256, 1 << 30)), dim3(256), 0, s, A.ptr, A.rows, tile_major_nct, pptr);
        SAPCA_HIP(rocprim::exclusive_scan(tmp, scan_bytes, pptr, pptr, (int64_t)0, (size_t)A.rows + 1, rocprim::plus<int64_t>(), s));
      }
      if (permuted && k16) {
        uint16_t* k_in = reinterpret_cast<uint16_t*>(keys_in);
        uint16_t* k_out = reinterpret_cast<uint16_t*>(keys_out);
        hipLaunchKernelGGL((pack_rows64_permuted_kernel<uint16_t>), dim3(grid_for(A.rows * WAVE, 256, 4096)), dim3(256), 0, s, A.ptr, A.idx,
                           reinterpret_cast<const double*>(A.val), A.rows, tile_major_nct, pptr, k_in, packed);
        SAPCA_HIP(rocprim::radix_sort_pairs(tmp, sb, k_in, k_out, packed, packed_out, (size_t)nnz, 0u, (unsigned)bits, s));
        hipLaunchKernelGGL((lower_bound_kernel<uint16_t>), dim3(grid_for(A.cols + 1, 256, 1 << 30)), dim3(256), 0, s, k_out, nnz,
                           A.cols, t_ptr);
      } else if (permuted) {
        hipLaunchKernelGGL((pack_rows64_permuted_kernel<uint32_t>), dim3(grid_for(A.rows * WAVE, 256, 4096)), dim3(256), 0, s, A.ptr, A.idx,
                           reinterpret_cast<const double*>(A.val), A.rows, tile_major_nct, pptr, keys_in, packed);
        SAPCA_HIP(rocprim::radix_sort_pairs(tmp, sb, keys_in, keys_out, packed, packed_out, (size_t)nnz, 0u, (unsigned)bits, s));
        hipLaunchKernelGGL((lower_bound_kernel<uint32_t>), dim3(grid_for(A.cols + 1, 256, 1 << 30)), dim3(256), 0, s, keys_out,
                           nnz, A.cols, t_ptr);
      } else if (k16) {
        uint16_t* k_in = reinterpret_cast<uint16_t*>(keys_in);
        uint16_t* k_out = reinterpret_cast<uint16_t*>(keys_out);
        hipLaunchKernelGGL((pack_rows64_kernel<uint16_t>), dim3(grid_for(A.rows * WAVE, 256, 4096)), dim3(256), 0, s, A.ptr, A.idx,
                           reinterpret_cast<const double*>(A.val), A.rows, k_in, packed);
        SAPCA_HIP(rocprim::radix_sort_pairs(tmp, sb, k_in, k_out, packed, packed_out, (size_t)nnz, 0u, (unsigned)bits, s));
        hipLaunchKernelGGL((lower_bound_kernel<uint16_t>), dim3(grid_for(A.cols + 1, 256, 1 << 30)), dim3(256), 0, s, k_out, nnz,
                           A.cols, t_ptr);
      } else {
        hipLaunchKernelGGL((pack_rows64_kernel<uint32_t>), dim3(grid_for(A.rows * WAVE, 256, 4096)), dim3(256), 0, s, A.ptr, A.idx,
                           reinterpret_cast<const double*>(A.val), A.rows, (uint32_t*)nullptr, packed);
        SAPCA_HIP(rocprim::radix_sort_pairs(tmp, sb, reinterpret_cast<const uint32_t*>(A.idx), keys_out, packed, packed_out,
                                            (size_t)nnz, 0u, (unsigned)bits, s));
        hipLaunchKernelGGL((lower_bound_kernel<uint32_t>), dim3(grid_for(A.cols + 1, 256, 1 << 30)), dim3(256), 0, s, keys_out,
                           nnz, A.cols, t_ptr);
      }
      hipLaunchKernelGGL(unpack_rows64_kernel, dim3(grid_for(nnz, 256, 8192)), dim3(256), 0, s, packed_out, nnz, t_idx,
                         reinterpret_cast<double*>(t_val));
      SAPCA_HIP(hipGetLastError());
      return;
    }
  }
  size_t sort_bytes = 0;
  rocprim::counting_iterator<uint32_t> iota(0);
  SAPCA_HIP(rocprim::radix_sort_pairs(nullptr, sort_bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, iota,
                                      (uint32_t*)nullptr, (size_t)nnz, 0u, (unsigned)bits, s));
  const size_t a = (size_t)round_up(nnz * 4, 256);
  char* base = static_cast<char*>(scratch.ensure(3 * a + sort_bytes + 256));
  int32_t* rowid = reinterpret_cast<int32_t*>(base);
  uint32_t* keys_out = reinterpret_cast<uint32_t*>(base + a);
  uint32_t* perm = reinterpret_cast<uint32_t*>(base + 2 * a);
  void* tmp = base + 3 * a;
  hipLaunchKernelGGL(expand_rows_kernel, dim3(grid_for(A.rows * WAVE, 256, 4096)), dim3(256), 0, s, A.ptr, A.rows, rowid);
  SAPCA_HIP(rocprim::radix_sort_pairs(tmp, sort_bytes, reinterpret_cast<const uint32_t*>(A.idx), keys_out, iota, perm,
                                      (size_t)nnz, 0u, (unsigned)bits, s));
  hipLaunchKernelGGL((lower_bound_kernel<uint32_t>), dim3(grid_for(A.cols + 1, 256, 1 << 30)), dim3(256), 0, s, keys_out, nnz,
                     A.cols, t_ptr);
  hipLaunchKernelGGL((gather_transposed_kernel<T>), dim3(grid_for(nnz, 256, 8192)), dim3(256), 0, s, perm, rowid,
                     A.val, nnz, t_idx, t_val);
  SAPCA_HIP(hipGetLastError());
}

void unpack_transposed(const uint64_t* packed, int64_t nnz, int32_t* t_idx, float* t_val, hipStream_t s) {
  if (nnz == 0) return;
  hipLaunchKernelGGL(unpack_rows_kernel, dim3(grid_for(nnz, 256, 8192)), dim3(256), 0, s, packed, nnz, t_idx, t_val);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void row_sums(const CsrView<T>& At, double* sum, double* sumsq, hipStream_t s) {
  if (At.rows == 0) return;
  hipLaunchKernelGGL((row_sums_kernel<T>), dim3(grid_for(At.rows * WAVE, 256, 4096)), dim3(256), 0, s, At.ptr, At.val,
                     At.rows, sum, sumsq);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void compact_columns(const CsrView<T>& A, const int32_t* o2m, int64_t* new_ptr, int32_t* new_idx, T* new_val,
                     int64_t* new_nnz_host, DevBuf& scratch, hipStream_t s, int32_t* drop_col, T* drop_val, unsigned long long* amax_bits) {
  if (amax_bits) SAPCA_HIP(hipMemsetAsync(amax_bits, 0, sizeof(unsigned long long), s));
  // (every workgroup rebuilds the column map in LDS from o2m: few, long-lived workgroups)
  const int g = grid_for(A.rows * WAVE, 256, 2048);
  int words = (int)std::min<int64_t>((A.cols + 31) / 32, 1 << 30);
  size_t lds = (size_t)2 * words * sizeof(uint32_t);
  if (lds > 48 * 1024) { words = 0; lds = 0; }   // more than 196608 columns: the kernels gather o2m from memory
  hipLaunchKernelGGL((count_kept_kernel<T>), dim3(g), dim3(256), lds, s, A.ptr, A.idx, A.rows, o2m, A.cols, words, new_ptr);
  exclusive_scan_i64(new_ptr, A.rows + 1, scratch, 0, s);
  hipLaunchKernelGGL((write_kept_kernel<T>), dim3(g), dim3(256), lds, s, A.ptr, A.idx, A.val, A.rows, o2m, A.cols, words, new_ptr,
                     new_idx, new_val, drop_col, drop_val, amax_bits);
  SAPCA_HIP(hipGetLastError());
  SAPCA_HIP(hipMemcpyAsync(new_nnz_host, new_ptr + A.rows, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SAPCA_HIP(hipStreamSynchronize(s));
}

// sum[c], sumsq[c] over the (column, value) pairs, c < n: a stable radix sort by column (the pairs keep their row order
// inside a column: the same summation order as the row sums of a transposed CSR), segment offsets, row sums.
// seg (n + 1), keys_out / vals_out (count) are work arrays.
template <typename T>
void sums_by_column(const int32_t* cols, const T* vals, int64_t count, int64_t n, int64_t* seg, int32_t* keys_out, T* vals_out,
                    double* sum, double* sumsq, DevBuf& scratch, hipStream_t s) {
  if (n <= 0) return;
  if (count <= 0) {
    SAPCA_HIP(hipMemsetAsync(sum, 0, (size_t)n * sizeof(double), s));
    SAPCA_HIP(hipMemsetAsync(sumsq, 0, (size_t)n * sizeof(double), s));
    return;
  }
  int bits = 1;
  while ((1ll << bits) < n) ++bits;
  size_t sb = 0;
  SAPCA_HIP(rocprim::radix_sort_pairs(nullptr, sb, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const T*)nullptr, (T*)nullptr,
                                      (size_t)count, 0u, (unsigned)bits, s));
  void* tmp = scratch.ensure(sb + 256);
  SAPCA_HIP(rocprim::radix_sort_pairs(tmp, sb, reinterpret_cast<const uint32_t*>(cols), reinterpret_cast<uint32_t*>(keys_out), vals, vals_out,
                                      (size_t)count, 0u, (unsigned)bits, s));
  hipLaunchKernelGGL((lower_bound_kernel<uint32_t>), dim3(grid_for(n + 1, 256, 1 << 30)), dim3(256), 0, s,
                     reinterpret_cast<const uint32_t*>(keys_out), count, n, seg);
  CsrView<T> Xt;
  Xt.rows = n; Xt.cols = 0; Xt.nnz = count; Xt.ptr = seg; Xt.idx = keys_out; Xt.val = vals_out;
  row_sums(Xt, sum, sumsq, s);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void select_rows(const CsrView<T>& At, const int32_t* rows, int64_t n_sel, int64_t* new_ptr, int32_t* new_idx,
                 T* new_val, int64_t* new_nnz_host, DevBuf& scratch, hipStream_t s) {
  hipLaunchKernelGGL(selected_lengths_kernel, dim3(grid_for(n_sel + 1, 256, 1 << 30)), dim3(256), 0, s, At.ptr, rows,
                     n_sel, new_ptr);
  exclusive_scan_i64(new_ptr, n_sel + 1, scratch, 0, s);
  if (n_sel > 0)
    hipLaunchKernelGGL((copy_selected_kernel<T>), dim3(grid_for(n_sel * WAVE, 256, 4096)), dim3(256), 0, s, At.ptr,
                       At.idx, At.val, rows, n_sel, new_ptr, new_idx, new_val);
  SAPCA_HIP(hipGetLastError());
  SAPCA_HIP(hipMemcpyAsync(new_nnz_host, new_ptr + n_sel, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  SAPCA_HIP(hipStreamSynchronize(s));
}



namespace {
template <typename T>
__global__ void mean_from_sums_kernel(const double* __restrict__ sum, double count, const int32_t* __restrict__ sel, int64_t n_used,
                                      T* __restrict__ mu) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n_used) mu[j] = (T)(sum[sel ? (int64_t)sel[j] : j] / count);
}
}  // namespace

namespace {
__global__ void scatter_pairs_kernel(const double* __restrict__ a, const double* __restrict__ b, const int32_t* __restrict__ where,
                                     int64_t count, double* __restrict__ out_a, double* __restrict__ out_b) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < count) {
    out_a[where[j]] = a[j];
    out_b[where[j]] = b[j];
  }
}
}  // namespace

namespace {
__global__ void copy_selected_kernel(const double* __restrict__ a, const double* __restrict__ b, const int32_t* __restrict__ where,
                                     int64_t count, double* __restrict__ out_a, double* __restrict__ out_b) {
  const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < count) {
    out_a[where[j]] = a[where[j]];
    out_b[where[j]] = b[where[j]];
  }
}
}  // namespace

// out[where[j]] = in[where[j]], j < count (both arrays full width): the kept columns' sums into the dropped columns' arrays
void copy_selected(const double* a, const double* b, const int32_t* where, int64_t count, double* out_a, double* out_b, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(copy_selected_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, a, b, where, count, out_a, out_b);
  SAPCA_HIP(hipGetLastError());
}

void scatter_pairs(const double* a, const double* b, const int32_t* where, int64_t count, double* out_a, double* out_b, hipStream_t s) {
  if (count <= 0) return;
  hipLaunchKernelGGL(scatter_pairs_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, a, b, where, count, out_a, out_b);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void mean_from_sums(const double* sum, double count, const int32_t* sel, int64_t n_used, T* mu, hipStream_t s) {
  if (n_used <= 0) return;
  hipLaunchKernelGGL((mean_from_sums_kernel<T>), dim3((unsigned)((n_used + 255) / 256)), dim3(256), 0, s, sum, count, sel, n_used, mu);
  SAPCA_HIP(hipGetLastError());
}
template void mean_from_sums<float>(const double*, double, const int32_t*, int64_t, float*, hipStream_t);
template void mean_from_sums<double>(const double*, double, const int32_t*, int64_t, double*, hipStream_t);
void row_lengths_f64(const int64_t* ptr, int64_t rows, double* out, hipStream_t s) {
  if (rows == 0) return;
  hipLaunchKernelGGL(ptr_diff_kernel, dim3(grid_for(rows, 256, 1 << 30)), dim3(256), 0, s, ptr, rows, out);
  SAPCA_HIP(hipGetLastError());
}

void column_counts_f64(const int32_t* idx, int64_t nnz, int64_t n, double* out, DevBuf& scratch, hipStream_t s) {
  if (n == 0) return;
  unsigned int* hist = scratch.as<unsigned int>((size_t)n);
  SAPCA_HIP(hipMemsetAsync(hist, 0, (size_t)n * sizeof(unsigned int), s));
  const size_t lds = (size_t)n * sizeof(unsigned int);
  if (nnz > 0 && lds <= 150 * 1024 && nnz >= (1 << 20)) {
    static LdsAttrState attr;
    if (lds > 48 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(&histogram_lds_kernel), lds, attr);
    const int wgs = lds > 80 * 1024 ? 256 : 512;   // (one or two workgroups per CU by their LDS)
    hipLaunchKernelGGL(histogram_lds_kernel, dim3(wgs), dim3(1024), lds, s, idx, nnz, n, hist);
  } else if (nnz > 0) {
    hipLaunchKernelGGL(histogram_kernel, dim3(grid_for(nnz, 256, 4096)), dim3(256), 0, s, idx, nnz, hist);
  }
  hipLaunchKernelGGL(u32_to_f64_kernel, dim3(grid_for(n, 256, 1 << 30)), dim3(256), 0, s, hist, n, out);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void build_tile_index(const CsrView<T>& A, int tile_cols, int n_tiles, int32_t* seg, hipStream_t s) {
  if (A.rows == 0) return;
  hipLaunchKernelGGL((tile_index_kernel<T>), dim3(grid_for(A.rows * (int64_t)(n_tiles + 1), 256, 16384)), dim3(256), 0,
                     s, A.ptr, A.idx, A.rows, tile_cols, n_tiles, seg);
  SAPCA_HIP(hipGetLastError());
}

#define INSTANTIATE(T)                                                                                              \
  template void transpose_csr<T>(const CsrView<T>&, int64_t*, int32_t*, T*, DevBuf&, hipStream_t, int,              \
                                 const uint64_t**);                                                                 \
  template void row_sums<T>(const CsrView<T>&, double*, double*, hipStream_t);                                      \
  template void compact_columns<T>(const CsrView<T>&, const int32_t*, int64_t*, int32_t*, T*, int64_t*, DevBuf&,    \
                                   hipStream_t, int32_t*, T*, unsigned long long*);                                 \
  template void sums_by_column<T>(const int32_t*, const T*, int64_t, int64_t, int64_t*, int32_t*, T*, double*, double*, DevBuf&,   \
                                  hipStream_t);                                                                     \
  template void select_rows<T>(const CsrView<T>&, const int32_t*, int64_t, int64_t*, int32_t*, T*, int64_t*,        \
                               DevBuf&, hipStream_t);                                                               \
  template void build_tile_index<T>(const CsrView<T>&, int, int, int32_t*, hipStream_t);
INSTANTIATE(float)
INSTANTIATE(double)
#undef INSTANTIATE

}  // namespace k
}  // namespace sapca
