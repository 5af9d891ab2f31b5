// Column-wise reductions of a CSR matrix WITHOUT its transpose: z = A^T y and the column statistics (R1 / R2 + counts),
// accumulated per workgroup in LDS and summed over the workgroups.
//
// Where the transposed side has at most ~19k columns (the masked C3 operator keeps 18k) a whole output vector fits the
// 160 KiB of LDS, so a workgroup can scatter its rows' contributions into a private copy instead of gathering along the
// rows of a transposed matrix that first has to be built (a radix sort of every stored entry: 4.4 of the 6.4 ms a C3
// preparation took).  Floating-point scatter would make the result depend on the order in which the waves' atomics land;
// the contributions are therefore added as 64-bit FIXED-POINT integers: the scale is a power of two chosen from a bound of
// the column sums (rows x max|a| x max|y|), integer addition is associative, and the sum of all workgroups' copies is
// converted back once.  The result is bit-for-bit reproducible and independent of how rows are dealt to workgroups; its
// absolute resolution, bound x 2^-61, is below the rounding error of a floating-point summation of the same terms.
//
// Replaces, for Lanczos fits whose transposed side fits LDS: the second product of a las2 step (svd_las2 call sites
// /root/reference/src/dimred/pca/sparse/mod.rs:136-144, sparse_masked/mod.rs:322-330) and sum_col / sum_col_squared
// (/root/reference/src/sparse/csr.rs:259-312, 558-608) as used by fit (sparse/mod.rs:106-131, sparse_masked/mod.rs:273-311).
#include <algorithm>

#include "kernels.h"

namespace sapca {
namespace k {

namespace {

constexpr int WAVE = 64;
constexpr int SC_THREADS = 1024;

// 2^e with  total * 2^e < 2^61  (total: a bound of the magnitude of any accumulated sum); 0 for an empty / zero operator
__device__ inline double fixed_scale(double total) {
  if (!(total <= 1.7976931348623157e308)) return total - total;   // inf / nan among the values: nan, and every sum with it
  if (!(total > 0.0)) return 0.0;
  int e;
  (void)frexp(total, &e);   // total < 2^e
  // (operands of tiny magnitude: 2^(61 - e) leaves the double range below total = 2^-962; the largest finite power of two keeps
  //  every product representable -- the sums then carry fewer than 61 bits, not garbage -- and 1 / scale finite)
  return ldexp(1.0, min(61 - e, 1023));
}

__device__ inline double bits_to_double(unsigned long long b) { return __longlong_as_double((long long)b); }

// max |v| over a value array, as the bit pattern of a non-negative double (they order like unsigned integers)
template <typename T>
__global__ void __launch_bounds__(256)
absmax_kernel(const T* __restrict__ v, int64_t count, unsigned long long* __restrict__ out_bits) {
  __shared__ unsigned long long wmax[4];
  double m = 0.0;
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    const double a = fabs((double)v[i]);
    bad |= !(a <= 1.7976931348623157e308);   // inf or nan
    m = fmax(m, a);
  }
  unsigned long long b = bad ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(m);
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) b = max(b, (unsigned long long)__shfl_xor((long long)b, off));
  if ((threadIdx.x & (WAVE - 1)) == 0) wmax[threadIdx.x / WAVE] = b;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out_bits, max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3])));
}

// part[workgroup][c] = sum over the workgroup's rows r and their stored entries (r, c) of round(a_rc * y_r * scale)
template <typename T, typename I>
__global__ void __launch_bounds__(SC_THREADS)
spmvt_scatter_kernel(const int64_t* __restrict__ ptr, const I* __restrict__ idx, const T* __restrict__ val, int64_t rows, int64_t cols,
                     const double* __restrict__ y, const unsigned long long* __restrict__ amax_bits,
                     const unsigned long long* __restrict__ ymax_bits, long long* __restrict__ part) {
  extern __shared__ long long zs[];
  for (int64_t i = threadIdx.x; i < cols; i += blockDim.x) zs[i] = 0;
  const double scale = fixed_scale((double)rows * bits_to_double(*amax_bits) * bits_to_double(*ymax_bits));
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    const double yr = y[r] * scale;
    int64_t e = e0 + lane;
    for (; e + 3 * WAVE < e1; e += 4 * WAVE) {   // four loads in flight per lane
      const int c0 = __builtin_nontemporal_load(idx + e), c1 = __builtin_nontemporal_load(idx + e + WAVE);
      const int c2 = __builtin_nontemporal_load(idx + e + 2 * WAVE), c3 = __builtin_nontemporal_load(idx + e + 3 * WAVE);
      const T v0 = __builtin_nontemporal_load(val + e), v1 = __builtin_nontemporal_load(val + e + WAVE);
      const T v2 = __builtin_nontemporal_load(val + e + 2 * WAVE), v3 = __builtin_nontemporal_load(val + e + 3 * WAVE);
      atomicAdd(reinterpret_cast<unsigned long long*>(zs + c0), (unsigned long long)__double2ll_rn((double)v0 * yr));
      atomicAdd(reinterpret_cast<unsigned long long*>(zs + c1), (unsigned long long)__double2ll_rn((double)v1 * yr));
      atomicAdd(reinterpret_cast<unsigned long long*>(zs + c2), (unsigned long long)__double2ll_rn((double)v2 * yr));
      atomicAdd(reinterpret_cast<unsigned long long*>(zs + c3), (unsigned long long)__double2ll_rn((double)v3 * yr));
    }
    for (; e < e1; e += WAVE)
      atomicAdd(reinterpret_cast<unsigned long long*>(zs + (int)idx[e]), (unsigned long long)__double2ll_rn((double)val[e] * yr));
  }
  __syncthreads();
  long long* out = part + (int64_t)blockIdx.x * cols;
  for (int64_t i = threadIdx.x; i < cols; i += blockDim.x) out[i] = zs[i];
}

// z[c] = (sum over the workgroups of part[.][c]) / scale; the slot of the NEXT step's max|y| is cleared on the way.
// A workgroup takes 64 columns, its 16 waves a sixteenth of the copies each (16 independent loads in flight per thread).
__global__ void __launch_bounds__(1024)
spmvt_reduce_kernel(const long long* __restrict__ part, int nparts, int64_t rows, int64_t cols, const unsigned long long* __restrict__ amax_bits,
                    const unsigned long long* __restrict__ ymax_bits, unsigned long long* __restrict__ clear_bits, double* __restrict__ z) {
  __shared__ long long acc[16][64];
  const double scale = fixed_scale((double)rows * bits_to_double(*amax_bits) * bits_to_double(*ymax_bits));
  const double inv = scale == 0.0 ? 0.0 : 1.0 / scale;   // (a power of two: exact; nan stays nan)
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  const int per = (nparts + 15) / 16;
  long long a = 0;
  if (c < cols) {
    const int p1 = min(nparts, (grp + 1) * per);
#pragma unroll 16
    for (int p = grp * per; p < p1; ++p) a += part[(int64_t)p * cols + c];
  }
  acc[grp][lane] = a;
  __syncthreads();
  if (grp == 0 && c < cols) {
    long long t = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += acc[g][lane];
    z[c] = (double)t * inv;
  }
  if (clear_bits && blockIdx.x == 0 && threadIdx.x == 0) *clear_bits = 0ull;
}

// max |y| of a vector, into a slot that was cleared before
__global__ void __launch_bounds__(256)
vecmax_kernel(const double* __restrict__ y, int64_t len, unsigned long long* __restrict__ out_bits) {
  double m = 0.0;
  bool bad = false;   // (fmax drops a nan: a non-finite y must reach the scale, as it reaches the sums of the floating-point route)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += (int64_t)gridDim.x * blockDim.x) {
    const double a = fabs(y[i]);
    bad |= !(a <= 1.7976931348623157e308);
    m = fmax(m, a);
  }
  unsigned long long b = bad ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(m);
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) b = max(b, (unsigned long long)__shfl_xor((long long)b, off));
  if ((threadIdx.x & (WAVE - 1)) == 0) atomicMax(out_bits, b);
}

// bounds[r][p] = first entry of row r (relative to the row's start) with a column index >= p * nc_pass, p = 1 .. passes - 1:
// the statistics passes below then read only the entries of their own column range (column indices ascend along a row)
__global__ void __launch_bounds__(256)
colrange_bounds_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, int64_t rows, int nc_pass, int passes,
                       uint32_t* __restrict__ bounds) {
  const int64_t total = rows * (int64_t)(passes - 1);
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / (passes - 1);
    const int p = (int)(i % (passes - 1)) + 1;
    const int64_t e0 = ptr[r];
    int64_t lo = e0, hi = ptr[r + 1];
    const int target = p * nc_pass;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (idx[mid] < target) lo = mid + 1; else hi = mid;
    }
    bounds[i] = (uint32_t)(lo - e0);
  }
}

// Column statistics of columns [c0, c0 + nc) (pass p of `passes`): per workgroup fixed-point sums of a and a^2 and entry
// counts in LDS.  A row's entries inside the range are the run [bounds[r][p], bounds[r][p + 1]) of the row.
template <typename T>
__global__ void __launch_bounds__(SC_THREADS)
colstats_scatter_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const T* __restrict__ val, int64_t rows, int c0,
                        int nc, int p, int passes, const uint32_t* __restrict__ bounds, const unsigned long long* __restrict__ amax_bits,
                        long long* __restrict__ part_sum, long long* __restrict__ part_sq, unsigned int* __restrict__ part_cnt) {
  extern __shared__ long long cs_lds[];
  long long* ss = cs_lds;
  long long* sq = cs_lds + nc;
  unsigned int* cn = reinterpret_cast<unsigned int*>(cs_lds + 2 * (int64_t)nc);
  for (int i = threadIdx.x; i < nc; i += blockDim.x) { ss[i] = 0; sq[i] = 0; cn[i] = 0u; }
  const double amax = bits_to_double(*amax_bits);
  const double s1 = fixed_scale((double)rows * amax), s2 = fixed_scale((double)rows * amax * amax);
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e0 = ptr[r];
    const int64_t lo = e0 + (p > 0 ? (int64_t)bounds[r * (passes - 1) + (p - 1)] : 0);
    const int64_t hi = p + 1 < passes ? e0 + (int64_t)bounds[r * (passes - 1) + p] : ptr[r + 1];
    for (int64_t e = lo + lane; e < hi; e += WAVE) {
      const int c = idx[e] - c0;
      const double v = (double)val[e];
      atomicAdd(reinterpret_cast<unsigned long long*>(ss + c), (unsigned long long)__double2ll_rn(v * s1));
      atomicAdd(reinterpret_cast<unsigned long long*>(sq + c), (unsigned long long)__double2ll_rn(v * v * s2));
      atomicAdd(cn + c, 1u);
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < nc; i += blockDim.x) {
    part_sum[(int64_t)blockIdx.x * nc + i] = ss[i];
    part_sq[(int64_t)blockIdx.x * nc + i] = sq[i];
    part_cnt[(int64_t)blockIdx.x * nc + i] = cn[i];
  }
}

__global__ void __launch_bounds__(1024)
colstats_reduce_kernel(const long long* __restrict__ part_sum, const long long* __restrict__ part_sq, const unsigned int* __restrict__ part_cnt,
                       int nparts, int64_t rows, int c0, int nc, const unsigned long long* __restrict__ amax_bits, double* __restrict__ sum,
                       double* __restrict__ sumsq, double* __restrict__ cnt) {
  __shared__ long long sa[16][64], sb[16][64];
  __shared__ unsigned long long sn[16][64];
  const double amax = bits_to_double(*amax_bits);
  const double s1 = fixed_scale((double)rows * amax), s2 = fixed_scale((double)rows * amax * amax);
  const double i1 = s1 == 0.0 ? 0.0 : 1.0 / s1, i2 = s2 == 0.0 ? 0.0 : 1.0 / s2;
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int per = (nparts + 15) / 16;
  long long a = 0, b = 0;
  unsigned long long n = 0;
  if (c < nc) {
    const int p1 = min(nparts, (grp + 1) * per);
#pragma unroll 8
    for (int p = grp * per; p < p1; ++p) {
      a += part_sum[(int64_t)p * nc + c];
      b += part_sq[(int64_t)p * nc + c];
      n += part_cnt[(int64_t)p * nc + c];
    }
  }
  sa[grp][lane] = a;
  sb[grp][lane] = b;
  sn[grp][lane] = n;
  __syncthreads();
  if (grp == 0 && c < nc) {
    long long ta = 0, tb = 0;
    unsigned long long tn = 0;
#pragma unroll
    for (int g = 0; g < 16; ++g) { ta += sa[g][lane]; tb += sb[g][lane]; tn += sn[g][lane]; }
    const bool bad = !(amax <= 1.7976931348623157e308);   // inf / nan among the values: no fixed-point scale exists
    sum[c0 + c] = bad ? amax : (double)ta * i1;
    sumsq[c0 + c] = bad ? amax : (double)tb * i2;
    if (cnt) cnt[c0 + c] = (double)tn;
  }
}

constexpr int kScatterLds = 150 * 1024;
constexpr int kParts = 256;   // one workgroup per CU

}  // namespace

bool scatter_fits(int64_t cols) { return cols > 0 && cols * (int64_t)sizeof(long long) <= kScatterLds; }

template <typename T>
void absmax(const T* v, int64_t count, unsigned long long* out_bits, hipStream_t s) {
  SAPCA_HIP(hipMemsetAsync(out_bits, 0, sizeof(unsigned long long), s));
  if (count > 0)
    hipLaunchKernelGGL((absmax_kernel<T>), dim3((unsigned)std::min<int64_t>((count + 255) / 256, 4096)), dim3(256), 0, s, v, count, out_bits);
  SAPCA_HIP(hipGetLastError());
}

void vecmax(const double* y, int64_t len, unsigned long long* out_bits, hipStream_t s) {
  if (len > 0)
    hipLaunchKernelGGL(vecmax_kernel, dim3((unsigned)std::min<int64_t>((len + 255) / 256, 256)), dim3(256), 0, s, y, len, out_bits);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void spmvt_scatter(const CsrView<T>& A, const uint16_t* idx16, const double* y, const unsigned long long* amax_bits,
                   const unsigned long long* ymax_bits, unsigned long long* clear_bits, double* z, DevBuf& scratch, hipStream_t s) {
  SAPCA_CHECK(scatter_fits(A.cols), SAPCA_ERR_ARG, "spmvt_scatter: the output vector does not fit LDS");
  long long* part = scratch.as<long long>((size_t)kParts * A.cols);
  const size_t lds = (size_t)A.cols * sizeof(long long);
  static LdsAttrState attr16, attr32;   // one per instantiation of this function template
  if (idx16) {
    ensure_dynamic_lds(reinterpret_cast<const void*>(&spmvt_scatter_kernel<T, uint16_t>), kScatterLds, attr16);
    hipLaunchKernelGGL((spmvt_scatter_kernel<T, uint16_t>), dim3(kParts), dim3(SC_THREADS), lds, s, A.ptr, idx16, A.val, A.rows, A.cols, y, amax_bits,
                       ymax_bits, part);
  } else {
    ensure_dynamic_lds(reinterpret_cast<const void*>(&spmvt_scatter_kernel<T, int32_t>), kScatterLds, attr32);
    hipLaunchKernelGGL((spmvt_scatter_kernel<T, int32_t>), dim3(kParts), dim3(SC_THREADS), lds, s, A.ptr, A.idx, A.val, A.rows, A.cols, y, amax_bits,
                       ymax_bits, part);
  }
  hipLaunchKernelGGL(spmvt_reduce_kernel, dim3((unsigned)((A.cols + 63) / 64)), dim3(1024), 0, s, part, kParts, A.rows, A.cols, amax_bits, ymax_bits,
                     clear_bits, z);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void colstats_scatter(const CsrView<T>& A, const unsigned long long* amax_bits, double* sum, double* sumsq, double* cnt, DevBuf& scratch,
                      hipStream_t s) {
  if (A.cols == 0) return;
  constexpr int per_col = 2 * (int)sizeof(long long) + (int)sizeof(unsigned int);
  const int max_nc = kScatterLds / per_col;
  const int passes = (int)((A.cols + max_nc - 1) / max_nc);
  const int nc_pass = (int)((A.cols + passes - 1) / passes);
  const size_t part_bytes = (size_t)kParts * nc_pass * per_col;
  const size_t bounds_bytes = passes > 1 ? (size_t)A.rows * (passes - 1) * sizeof(uint32_t) : 0;
  char* base = static_cast<char*>(scratch.ensure(part_bytes + bounds_bytes + 64));
  long long* ps = reinterpret_cast<long long*>(base);
  long long* pq = ps + (size_t)kParts * nc_pass;
  unsigned int* pc = reinterpret_cast<unsigned int*>(pq + (size_t)kParts * nc_pass);
  uint32_t* bounds = reinterpret_cast<uint32_t*>(base + ((part_bytes + 15) & ~(size_t)15));
  if (passes > 1)
    hipLaunchKernelGGL(colrange_bounds_kernel, dim3((unsigned)std::min<int64_t>((A.rows * (passes - 1) + 255) / 256, 8192)), dim3(256), 0, s, A.ptr,
                       A.idx, A.rows, nc_pass, passes, bounds);
  static LdsAttrState attr;
  ensure_dynamic_lds(reinterpret_cast<const void*>(&colstats_scatter_kernel<T>), kScatterLds, attr);
  for (int p = 0; p < passes; ++p) {
    const int c0 = p * nc_pass, nc = (int)std::min<int64_t>(nc_pass, A.cols - c0);
    if (nc <= 0) break;
    hipLaunchKernelGGL((colstats_scatter_kernel<T>), dim3(kParts), dim3(SC_THREADS), (size_t)nc * per_col, s, A.ptr, A.idx, A.val, A.rows, c0, nc,
                       p, passes, bounds, amax_bits, ps, pq, pc);
    hipLaunchKernelGGL(colstats_reduce_kernel, dim3((unsigned)((nc + 63) / 64)), dim3(1024), 0, s, ps, pq, pc, kParts, A.rows, c0, nc, amax_bits, sum,
                       sumsq, cnt);
  }
  SAPCA_HIP(hipGetLastError());
}

template void absmax<float>(const float*, int64_t, unsigned long long*, hipStream_t);
template void absmax<double>(const double*, int64_t, unsigned long long*, hipStream_t);
template void spmvt_scatter<float>(const CsrView<float>&, const uint16_t*, const double*, const unsigned long long*, const unsigned long long*,
                                   unsigned long long*, double*, DevBuf&, hipStream_t);
template void spmvt_scatter<double>(const CsrView<double>&, const uint16_t*, const double*, const unsigned long long*, const unsigned long long*,
                                    unsigned long long*, double*, DevBuf&, hipStream_t);
template void colstats_scatter<float>(const CsrView<float>&, const unsigned long long*, double*, double*, double*, DevBuf&, hipStream_t);
template void colstats_scatter<double>(const CsrView<double>&, const unsigned long long*, double*, double*, double*, DevBuf&, hipStream_t);

}  // namespace k
}  // namespace sapca
