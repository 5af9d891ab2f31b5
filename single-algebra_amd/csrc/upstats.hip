// Column statistics accumulated while the host CSR is still crossing PCIe.
//
// R1/R2 (sum_col, sum_col_squared: /root/reference/src/sparse/csr.rs:259-312, 558-608) and the per-column stored-entry
// count are sums over entries in any order, so they can start on the chunk of (column index, value) pairs that has
// just landed while the next chunk's DMA runs.  Floating-point atomics would make the result depend on that order;
// instead every value (and its exact square) is added as an integer into a per-column long accumulator of 64-bit
// limbs spaced 32 bits apart (a value's mantissa, shifted to its exponent, is split into 32-bit pieces, so a limb
// takes at least 2^31 additions before it can overflow -- more rows than the ABI admits).  Integer addition is
// associative: the accumulated number is the exact sum whatever the order, and the final conversion rounds it once
// (to nearest even).  The result is bit-reproducible and is the correctly rounded sum.
//
//   f32: value = M * 2^(p - 149),   p = max(E, 1) - 1 in [0, 253],  M < 2^24   ->  9 limbs;  squares 18 limbs
//   f64: value = M * 2^(p - 1074),  p in [0, 2045],                  M < 2^53   -> 67 limbs;  squares 133 limbs
//
// Scattered 64-bit global atomics run at 30 G/s on this part (measured: six per entry took 3.3 ms per 16 M-entry
// chunk, twice the chunk's DMA; per-XCD copies with workgroup-scope atomics changed nothing), so they are not issued
// per entry.  A workgroup owns a tile of columns and a group of the chunk's rows, finds each row's segment inside the
// tile by bisection (rows are sorted by column) and adds into an LDS copy of the tile's accumulators -- restricted to
// the window of limbs the upload's largest value can reach (4 for sums, 8 for squares: 70 binades below the largest
// magnitude; anything smaller goes straight to memory).  One global atomic per non-zero LDS word flushes the tile.
#include <algorithm>

#include "kernels.h"

namespace sapca {
namespace k {
namespace {

template <typename T> struct Exact;
template <> struct Exact<float> {
  static constexpr int kSum = 9, kSq = 18, kBase = -149, kMant = 23, kEmax = 255;
  using Bits = uint32_t;
};
template <> struct Exact<double> {
  static constexpr int kSum = 67, kSq = 133, kBase = -1074, kMant = 52, kEmax = 2047;
  using Bits = uint64_t;
};

constexpr int kWinSum = 4, kWinSq = 8, kWin = kWinSum + kWinSq;   // limbs of a column kept in LDS
constexpr int kTileCols = 1280;                                    // 1280 x (12 x 8 + 4) bytes = 125 KiB
constexpr int kThreads = 1024;
constexpr size_t kTileLds = (size_t)kTileCols * (kWin * sizeof(unsigned long long) + sizeof(unsigned));

// what a piece lands in: the tile's LDS window when it covers the limb, memory otherwise
struct Sink {
  unsigned long long* lds;   // this column's kWin words
  unsigned long long* mem;   // this column's kSum + kSq limbs
};

// accumulator += (+-) mag * 2^pos   (limb j has weight 2^(32 j)); the accumulator's limbs start at mem[limb_off], its LDS
// window covers limbs [win0, win0 + win_len) at lds[lds_off]
__device__ inline void add_shifted(const Sink& k, int limb_off, int win0, int win_len, int lds_off, unsigned __int128 mag, int pos,
                                   bool neg) {
  const int j = pos >> 5, sh = pos & 31;
  uint32_t piece = (uint32_t)((uint32_t)mag << sh);
  unsigned __int128 rest = mag >> (32 - sh);
  for (int i = 0;; ++i) {
    if (piece) {
      const unsigned long long v = neg ? (0ull - (unsigned long long)piece) : (unsigned long long)piece;
      const int w = j + i - win0;
      if (w >= 0 && w < win_len) atomicAdd(&k.lds[lds_off + w], v);   // ds_add_u64
      else atomicAdd(&k.mem[limb_off + j + i], v);
    }
    if (rest == 0) break;
    piece = (uint32_t)rest;
    rest >>= 32;
  }
}

// largest exponent field among the values (they all arrive before the first index chunk)
template <typename T>
__global__ void __launch_bounds__(256) max_exponent_kernel(const T* __restrict__ val, int64_t count, int* __restrict__ emax) {
  using E = Exact<T>;
  using Bits = typename E::Bits;
  int best = 0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride) {
    Bits b;
    const T v = val[e];
    __builtin_memcpy(&b, &v, sizeof(T));
    const int ef = (int)((b >> E::kMant) & (Bits)E::kEmax);
    if (ef != E::kEmax) best = max(best, ef);
  }
  for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
  if ((threadIdx.x & 63) == 0 && best > 0) atomicMax(emax, best);
}

// one workgroup: columns [tile * kTileCols, ..) of rows [r_lo + group * rows_per_group, ..), entries inside [e_lo, e_hi)
template <typename T>
__global__ void __launch_bounds__(kThreads)
colstats_tile_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const T* __restrict__ val, int64_t r_lo,
                     int64_t r_hi, int64_t rows_per_group, int64_t e_lo, int64_t e_hi, int64_t n, const int* __restrict__ emax,
                     unsigned long long* __restrict__ limbs, unsigned* __restrict__ cnt, int* __restrict__ nonfinite) {
  using E = Exact<T>;
  using Bits = typename E::Bits;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  unsigned long long* acc = reinterpret_cast<unsigned long long*>(lds_raw);              // [kTileCols][kWin]
  unsigned* lcnt = reinterpret_cast<unsigned*>(acc + (size_t)kTileCols * kWin);          // [kTileCols]
  const int64_t c0 = (int64_t)blockIdx.x * kTileCols, c1 = min(n, c0 + (int64_t)kTileCols);
  const int64_t g_lo = r_lo + (int64_t)blockIdx.y * rows_per_group, g_hi = min(r_hi, g_lo + rows_per_group);
  for (int i = threadIdx.x; i < kTileCols * kWin; i += kThreads) acc[i] = 0ull;
  for (int i = threadIdx.x; i < kTileCols; i += kThreads) lcnt[i] = 0u;
  // windows: the top limb a piece of the largest value can reach, and the limbs below it
  const int pmax = max(*emax - 1, 0);
  constexpr int kSumPieces = sizeof(T) == 4 ? 2 : 3, kSqPieces = sizeof(T) == 4 ? 3 : 5;
  const int ws = max(0, (pmax >> 5) + kSumPieces - kWinSum), wq = max(0, ((2 * pmax) >> 5) + kSqPieces - kWinSq);
  __syncthreads();
  // one wave per row: a 64-ary search for the first entry with column >= c0 (two coalesced probes for rows up to 4096
  // entries), then the segment 64 entries at a time
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int64_t r = g_lo + wave; r < g_hi; r += kThreads / 64) {
    const int64_t hi = min(ptr[r + 1], e_hi);
    int64_t a = max(ptr[r], e_lo), b = hi;
    while (a < b) {
      const int64_t step = (b - a + 63) >> 6;
      const int64_t pos = a + (int64_t)lane * step;
      const bool ge = pos >= b || (int64_t)idx[pos] >= c0;
      const unsigned long long mask = __ballot(ge);
      const int f = mask ? __builtin_ctzll(mask) : 64;
      if (step == 1) {   // the probes were the entries themselves
        a = min(b, a + f);
        break;
      }
      const int64_t na = f == 0 ? a : a + (int64_t)(f - 1) * step + 1;
      b = f == 0 ? a : min(b, a + (int64_t)f * step);
      a = na;
    }
    for (int64_t e0 = a; e0 < hi; e0 += 64) {
      const int64_t e = e0 + lane;
      const int64_t c = e < hi ? (int64_t)idx[e] : c1;
      const bool in = c >= c0 && c < c1;   // (an unsorted or out-of-range row: refused by upload() once the range flag arrives)
      if (in) {
        Bits bits;
        const T v = val[e];
        __builtin_memcpy(&bits, &v, sizeof(T));
        const bool neg = (bits >> (sizeof(T) * 8 - 1)) != 0;
        const int ef = (int)((bits >> E::kMant) & (Bits)E::kEmax);
        Bits mant = bits & (((Bits)1 << E::kMant) - 1);
        const int lc = (int)(c - c0);
        atomicAdd(&lcnt[lc], 1u);
        if (ef == E::kEmax) {   // inf / nan: the statistics pass of prepare() takes over (and propagates them as before)
          atomicOr(nonfinite, 1);
        } else {
          if (ef) mant |= (Bits)1 << E::kMant;
          if (mant != 0) {      // (a stored zero only counts)
            const int p = ef ? ef - 1 : 0;
            const Sink k{acc + (size_t)lc * kWin, limbs + (size_t)c * (E::kSum + E::kSq)};
            add_shifted(k, 0, ws, kWinSum, 0, (unsigned __int128)mant, p, neg);
            add_shifted(k, E::kSum, wq, kWinSq, kWinSum, (unsigned __int128)mant * (unsigned __int128)mant, 2 * p, false);
          }
        }
      }
      if (__ballot(in) != ~0ull) break;   // the segment ended inside these 64 entries
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < (int)(c1 - c0) * kWin; i += kThreads) {
    const unsigned long long v = acc[i];
    if (v == 0ull) continue;
    const int lc = i / kWin, w = i % kWin;
    unsigned long long* L = limbs + (size_t)(c0 + lc) * (E::kSum + E::kSq);
    atomicAdd(w < kWinSum ? &L[ws + w] : &L[E::kSum + wq + (w - kWinSum)], v);
  }
  for (int i = threadIdx.x; i < (int)(c1 - c0); i += kThreads)
    if (lcnt[i]) atomicAdd(&cnt[c0 + i], lcnt[i]);
}

// the long accumulator as a double, rounded once (to nearest, ties to even)
template <int NL>
__device__ double limbs_to_double(const unsigned long long* L, int base) {
  uint32_t dig[NL + 2];
  __int128 carry = 0;
  for (int j = 0; j < NL; ++j) {
    const __int128 t = (__int128)(long long)L[j] + carry;
    dig[j] = (uint32_t)t;
    carry = t >> 32;
  }
  dig[NL] = (uint32_t)carry;
  carry >>= 32;
  dig[NL + 1] = (uint32_t)carry;
  carry >>= 32;
  const bool neg = carry < 0;
  if (neg) {   // magnitude: two's complement over the digits
    uint64_t c = 1;
    for (int j = 0; j < NL + 2; ++j) {
      const uint64_t t = (uint64_t)(uint32_t)~dig[j] + c;
      dig[j] = (uint32_t)t;
      c = t >> 32;
    }
  }
  int t = NL + 1;
  while (t >= 0 && dig[t] == 0) --t;
  if (t < 0) return 0.0;
  const uint32_t d1 = t >= 1 ? dig[t - 1] : 0u, d2 = t >= 2 ? dig[t - 2] : 0u;
  const unsigned __int128 v = ((unsigned __int128)dig[t] << 64) | ((unsigned __int128)d1 << 32) | (unsigned __int128)d2;
  const int lz = __clz((int)dig[t]);
  uint64_t w = (uint64_t)(v >> (32 - lz));
  bool sticky = (v & (((unsigned __int128)1 << (32 - lz)) - 1)) != 0;
  for (int j = t - 3; j >= 0 && !sticky; --j) sticky = dig[j] != 0;
  if (sticky) w |= 1ull;   // 64 -> 53 bits below: the sticky bit only has to make a tie not a tie
  const double r = ldexp((double)w, 32 * (t - 1) - lz + base);
  return neg ? -r : r;
}

template <typename T>
__global__ void __launch_bounds__(64)
colstats_finish_kernel(const unsigned long long* __restrict__ limbs, const unsigned* __restrict__ cnt, int64_t n,
                       double* __restrict__ out) {
  using E = Exact<T>;
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const unsigned long long* L = limbs + (size_t)c * (E::kSum + E::kSq);
  out[c] = limbs_to_double<E::kSum>(L, E::kBase);
  out[n + c] = limbs_to_double<E::kSq>(L + E::kSum, 2 * E::kBase);
  out[2 * n + c] = (double)cnt[c];
}

inline size_t cnt_stride(int64_t n) { return (size_t)((n + 63) / 64 * 64); }

template <typename T>
void split(void* work, int64_t n, unsigned long long*& limbs, unsigned*& cnt, int*& flag, int*& emax) {
  using E = Exact<T>;
  limbs = static_cast<unsigned long long*>(work);
  cnt = reinterpret_cast<unsigned*>(limbs + (size_t)n * (E::kSum + E::kSq));
  flag = reinterpret_cast<int*>(cnt + cnt_stride(n));
  emax = flag + 1;
}

}  // namespace

template <typename T>
size_t exact_colstats_bytes(int64_t n) {
  using E = Exact<T>;
  return (size_t)n * (E::kSum + E::kSq) * sizeof(unsigned long long) + cnt_stride(n) * sizeof(unsigned) + 256;
}

template <typename T>
void exact_colstats_reset(void* work, int64_t n, hipStream_t s) {
  SAPCA_HIP(hipMemsetAsync(work, 0, exact_colstats_bytes<T>(n), s));
}

template <typename T>
void exact_colstats_scan_values(const T* val, int64_t count, int64_t n, void* work, hipStream_t s) {
  if (count <= 0 || n <= 0) return;
  unsigned long long* limbs;
  unsigned* cnt;
  int *flag, *emax;
  split<T>(work, n, limbs, cnt, flag, emax);
  hipLaunchKernelGGL((max_exponent_kernel<T>), dim3((unsigned)std::min<int64_t>((count + 255) / 256, 2048)), dim3(256), 0, s, val, count,
                     emax);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void exact_colstats_add(const int64_t* ptr, const int32_t* idx, const T* val, int64_t r_lo, int64_t r_hi, int64_t e_lo, int64_t e_hi,
                        int64_t n, void* work, hipStream_t s) {
  if (e_hi <= e_lo || r_hi <= r_lo || n <= 0) return;
  unsigned long long* limbs;
  unsigned* cnt;
  int *flag, *emax;
  split<T>(work, n, limbs, cnt, flag, emax);
  static LdsAttrState attr;
  ensure_dynamic_lds(reinterpret_cast<const void*>(&colstats_tile_kernel<T>), kTileLds, attr);
  const int64_t tiles = (n + kTileCols - 1) / kTileCols;
  // about 1024 workgroups per chunk, at least 64 rows (four per wave) in each
  const int64_t rows = r_hi - r_lo;
  int64_t groups = std::max<int64_t>(1, std::min<int64_t>((1024 + tiles - 1) / tiles, (rows + 63) / 64));
  groups = std::min<int64_t>(groups, 65535);
  const int64_t rows_per_group = (rows + groups - 1) / groups;
  hipLaunchKernelGGL((colstats_tile_kernel<T>), dim3((unsigned)tiles, (unsigned)groups), dim3(kThreads), kTileLds, s, ptr, idx, val, r_lo,
                     r_hi, rows_per_group, e_lo, e_hi, n, emax, limbs, cnt, flag);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void exact_colstats_finish(void* work, int64_t n, double* out, int* nonfinite_host, hipStream_t s) {
  unsigned long long* limbs;
  unsigned* cnt;
  int *flag, *emax;
  split<T>(work, n, limbs, cnt, flag, emax);
  if (n > 0) {
    hipLaunchKernelGGL((colstats_finish_kernel<T>), dim3((unsigned)((n + 63) / 64)), dim3(64), 0, s, limbs, cnt, n, out);
    SAPCA_HIP(hipGetLastError());
  }
  SAPCA_HIP(hipMemcpyAsync(nonfinite_host, flag, sizeof(int), hipMemcpyDeviceToHost, s));
}

#define INST(T)                                                                                                          \
  template size_t exact_colstats_bytes<T>(int64_t);                                                                      \
  template void exact_colstats_reset<T>(void*, int64_t, hipStream_t);                                                    \
  template void exact_colstats_scan_values<T>(const T*, int64_t, int64_t, void*, hipStream_t);                           \
  template void exact_colstats_add<T>(const int64_t*, const int32_t*, const T*, int64_t, int64_t, int64_t, int64_t, int64_t, void*, \
                                      hipStream_t);                                                                      \
  template void exact_colstats_finish<T>(void*, int64_t, double*, int*, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace k
}  // namespace sapca
