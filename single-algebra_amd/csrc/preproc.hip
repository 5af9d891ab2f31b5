// Upstream preprocessing and the remaining CSR statistics on a device-resident matrix:
//   Normalize<T> for CsrMatrix    /root/reference/src/sparse/csr.rs:1012-1066
//   Log1P<T> for CsrMatrix        csr.rs:1069-1078
//   sum_row / sum_row_squared     csr.rs:314-392, 610-630
//   nonzero_row / nonzero_col     csr.rs:23-134
//   min_max_row / min_max_col     csr.rs:917-1008
// The typical consumer workflow is normalize -> log1p -> PCA (src/lib.rs:28-33); with these the matrix is
// uploaded once and stays in HBM through all three.  All kernels stream the values once (HBM-bound).
#include <cfloat>

#include "kernels.h"

namespace sapca {
namespace k {

namespace {

constexpr int WAVE = 64;

inline int grid_for(int64_t work_items, int block, int cap = 8192) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// csr.rs:1019-1029: scale = sum > 0 ? target / sum : 0 (in U = f64)
__global__ void scale_factors_kernel(const double* __restrict__ sums, int64_t count, double target, double* __restrict__ scale) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) scale[i] = sums[i] > 0.0 ? target / sums[i] : 0.0;
}

// csr.rs:1032-1044 (Direction::COLUMN): val = T(U(val) * scale[col]) where scale[col] > 0
template <typename T>
__global__ void normalize_cols_kernel(const int32_t* __restrict__ idx, T* __restrict__ val, int64_t nnz,
                                      const double* __restrict__ scale) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < nnz; e += stride) {
    const double sc = scale[idx[e]];
    if (sc > 0.0) val[e] = (T)((double)val[e] * sc);
  }
}

// csr.rs:1045-1062 (Direction::ROW)
template <typename T>
__global__ void normalize_rows_kernel(const int64_t* __restrict__ ptr, T* __restrict__ val, int64_t rows,
                                      const double* __restrict__ scale) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const double sc = scale[r];
    if (!(sc > 0.0)) continue;
    const int64_t e1 = ptr[r + 1];
    for (int64_t e = ptr[r] + lane; e < e1; e += WAVE) val[e] = (T)((double)val[e] * sc);
  }
}

// csr.rs:1071-1076: val = (1 + val).ln(), in T
template <typename T>
__global__ void log1p_kernel(T* __restrict__ val, int64_t nnz) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; e < nnz; e += stride) val[e] = (T)log((T)1 + val[e]);
}

template <typename T> struct Lim;
template <> struct Lim<float> {
  __device__ static float hi() { return FLT_MAX; }
};
template <> struct Lim<double> {
  __device__ static double hi() { return DBL_MAX; }
};

// per row: sum, sum of squares (f64 accumulation), min and max over the STORED entries (rows without entries
// keep the reference's initial values Item::max_value() / Item::min_value(), csr.rs:932-933)
template <typename T>
__global__ void row_stats_kernel(const int64_t* __restrict__ ptr, const T* __restrict__ val, int64_t rows,
                                 double* __restrict__ sum, double* __restrict__ sumsq, T* __restrict__ minv,
                                 T* __restrict__ maxv) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    double a = 0, b = 0;
    T lo = Lim<T>::hi(), hi = -Lim<T>::hi();
    const int64_t e1 = ptr[r + 1];
    for (int64_t e = ptr[r] + lane; e < e1; e += WAVE) {
      const T x = val[e];
      const double v = (double)x;
      a += v;
      b += v * v;
      lo = x < lo ? x : lo;
      hi = x > hi ? x : hi;
    }
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
      a += __shfl_xor(a, off);
      b += __shfl_xor(b, off);
      const T lo2 = __shfl_xor(lo, off), hi2 = __shfl_xor(hi, off);
      lo = lo2 < lo ? lo2 : lo;
      hi = hi2 > hi ? hi2 : hi;
    }
    if (lane == 0) {
      if (sum) sum[r] = a;
      if (sumsq) sumsq[r] = b;
      if (minv) minv[r] = lo;
      if (maxv) maxv[r] = hi;
    }
  }
}

}  // namespace

template <typename T>
void normalize_csr(const CsrView<T>& A, T* values, const double* d_sums, double target, bool by_column, double* d_scale,
                   hipStream_t s) {
  const int64_t count = by_column ? A.cols : A.rows;
  if (count == 0 || A.nnz == 0) return;
  hipLaunchKernelGGL(scale_factors_kernel, dim3(grid_for(count, 256, 1 << 30)), dim3(256), 0, s, d_sums, count, target, d_scale);
  if (by_column)
    hipLaunchKernelGGL((normalize_cols_kernel<T>), dim3(grid_for(A.nnz, 256, 8192)), dim3(256), 0, s, A.idx, values, A.nnz, d_scale);
  else
    hipLaunchKernelGGL((normalize_rows_kernel<T>), dim3(grid_for(A.rows * WAVE, 256, 4096)), dim3(256), 0, s, A.ptr, values,
                       A.rows, d_scale);
  SAPCA_HIP(hipGetLastError());
}

// A 16-byte-per-lane streaming copy: what this device's HBM delivers to a kernel of this library (read + write), the
// attainable figure bench.py reports beside the 8 TB/s of the data sheet.
typedef unsigned int copy_u4 __attribute__((ext_vector_type(4)));
// (one 16-byte element per thread, one short workgroup per 4 KiB: of the launch shapes tried in tools/ubench/copy_bw.hip this is
//  the one that reaches the guide's figure -- 6.25 TB/s; grid-stride loops over 2048..16384 workgroups gave 4.5-5.2, hipMemcpy 5.2)
__global__ void __launch_bounds__(256) copy16_kernel(const copy_u4* __restrict__ src, copy_u4* __restrict__ dst, int64_t n16) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];
}

void stream_copy16(const void* src, void* dst, int64_t bytes, hipStream_t s) {
  const int64_t n16 = bytes / 16;
  if (n16 == 0) return;
  SAPCA_CHECK((n16 + 255) / 256 < ((int64_t)1 << 31), SAPCA_ERR_ARG, "stream_copy16: more than 2^31 workgroups");
  hipLaunchKernelGGL(copy16_kernel, dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, s, static_cast<const copy_u4*>(src),
                     static_cast<copy_u4*>(dst), n16);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void log1p_values(T* values, int64_t nnz, hipStream_t s) {
  if (nnz == 0) return;
  hipLaunchKernelGGL((log1p_kernel<T>), dim3(grid_for(nnz, 256, 8192)), dim3(256), 0, s, values, nnz);
  SAPCA_HIP(hipGetLastError());
}

template <typename T>
void row_stats(const CsrView<T>& A, double* sum, double* sumsq, T* minv, T* maxv, hipStream_t s) {
  if (A.rows == 0) return;
  hipLaunchKernelGGL((row_stats_kernel<T>), dim3(grid_for(A.rows * WAVE, 256, 4096)), dim3(256), 0, s, A.ptr, A.val, A.rows, sum,
                     sumsq, minv, maxv);
  SAPCA_HIP(hipGetLastError());
}

#define INSTANTIATE(T)                                                                                        \
  template void normalize_csr<T>(const CsrView<T>&, T*, const double*, double, bool, double*, hipStream_t);  \
  template void log1p_values<T>(T*, int64_t, hipStream_t);                                                    \
  template void row_stats<T>(const CsrView<T>&, double*, double*, T*, T*, hipStream_t);
INSTANTIATE(float)
INSTANTIATE(double)
#undef INSTANTIATE

}  // namespace k
}  // namespace sapca
