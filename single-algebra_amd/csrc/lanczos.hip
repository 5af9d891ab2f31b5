// R12: Lanczos SVD of the prepared operator (raw values, NO centring: quirk Q1) --
// single_svdlib::lanczos::svd_las2 call sites
//   /root/reference/src/dimred/pca/sparse/mod.rs:134-144          (iterations = max(m, n))
//   /root/reference/src/dimred/pca/sparse_masked/mod.rs:316-331   (iterations = max(2 max(m, n'), 100))
// with end interval [-1e-30, 1e30], kappa = 10e-6 and random_seed: u32.
//
// The crate's source is not available; what is restated is the published las2 algorithm
// (SVDLIBC): single-vector Lanczos on B = A^T A (or A A^T when that is the smaller side),
// eigenvalues of the tridiagonal T by implicit QL, a Ritz pair accepted when its error bound
// |beta_j s_{j,i}| <= kappa |theta_i|.  las2's selective re-orthogonalisation is an economy
// measure for a CPU; here every new vector is re-orthogonalised against the whole basis
// (classical Gram-Schmidt, twice) with two skinny GEMV kernels, which gives the same Ritz pairs
// to working precision and needs no eta/oldeta bookkeeping.  All n-sized data stays on the GPU
// in f64; at a convergence check (every step while T is small, see check_due) the host downloads
// alpha/beta (2j doubles) and runs QL on the j x j tridiagonal with the bottom eigenvector row only.
#include <algorithm>
#include <cmath>

#include <cstdlib>

#include "lanczos.h"
#include "small_svd.h"

namespace sapca {

namespace {

constexpr int WAVE = 64;

// y[r] = sum_e val_e x[col_e]  (f64 accumulate; one wave per row)
template <typename T>
__global__ void __launch_bounds__(256)
spmv_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const T* __restrict__ val, int64_t rows,
            const double* __restrict__ x, double* __restrict__ y) {
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    // four independent (value, index, gather) chains per lane: the loop is bound by memory latency, not bytes
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int64_t e = e0 + lane;
    for (; e + 3 * WAVE < e1; e += 4 * WAVE) {
      const int c0 = __builtin_nontemporal_load(idx + e), c1 = __builtin_nontemporal_load(idx + e + WAVE);
      const int c2 = __builtin_nontemporal_load(idx + e + 2 * WAVE), c3 = __builtin_nontemporal_load(idx + e + 3 * WAVE);
      const T v0 = __builtin_nontemporal_load(val + e), v1 = __builtin_nontemporal_load(val + e + WAVE);
      const T v2 = __builtin_nontemporal_load(val + e + 2 * WAVE), v3 = __builtin_nontemporal_load(val + e + 3 * WAVE);
      a0 = fma((double)v0, x[c0], a0);
      a1 = fma((double)v1, x[c1], a1);
      a2 = fma((double)v2, x[c2], a2);
      a3 = fma((double)v3, x[c3], a3);
    }
    for (; e < e1; e += WAVE) a0 = fma((double)val[e], x[idx[e]], a0);
    a0 = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) a0 += __shfl_xor(a0, off);
    if (lane == 0) y[r] = a0;
  }
}

// The same with x staged in LDS (operators with up to ~19k columns: the masked C3 matrix has 18k): the
// per-entry gather then costs an LDS read instead of an L2 sector, which is what bounds the kernel above.
// I = uint16_t: the column indices as a 2-byte copy (x fits LDS, so it has fewer than 65536 elements): 10 instead of 12
// bytes per f64 entry (6 instead of 8 for f32) on a kernel that runs at the HBM rate, once per Lanczos step.
template <typename T, typename I>
__global__ void __launch_bounds__(1024)
spmv_ldsx_kernel(const int64_t* __restrict__ ptr, const I* __restrict__ idx, const T* __restrict__ val, int64_t rows,
                 int64_t cols, const double* __restrict__ x, double* __restrict__ y, unsigned long long* __restrict__ ymax_bits) {
  extern __shared__ double xs[];
  bool ybad = false;
  double ymax = 0.0;   // (lane 0 of every wave: the largest |y| it wrote, for the fixed-point scale of the scatter product that follows)
  for (int64_t i = threadIdx.x; i < cols; i += blockDim.x) xs[i] = x[i];
  __syncthreads();
  const int lane = threadIdx.x & (WAVE - 1);
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int64_t e = e0 + lane;
    for (; e + 3 * WAVE < e1; e += 4 * WAVE) {
      const int c0 = __builtin_nontemporal_load(idx + e), c1 = __builtin_nontemporal_load(idx + e + WAVE);
      const int c2 = __builtin_nontemporal_load(idx + e + 2 * WAVE), c3 = __builtin_nontemporal_load(idx + e + 3 * WAVE);
      const T v0 = __builtin_nontemporal_load(val + e), v1 = __builtin_nontemporal_load(val + e + WAVE);
      const T v2 = __builtin_nontemporal_load(val + e + 2 * WAVE), v3 = __builtin_nontemporal_load(val + e + 3 * WAVE);
      a0 = fma((double)v0, xs[c0], a0);
      a1 = fma((double)v1, xs[c1], a1);
      a2 = fma((double)v2, xs[c2], a2);
      a3 = fma((double)v3, xs[c3], a3);
    }
    for (; e < e1; e += WAVE) a0 = fma((double)val[e], xs[idx[e]], a0);
    a0 = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) a0 += __shfl_xor(a0, off);
    if (lane == 0) y[r] = a0;
    ybad |= !(fabs(a0) <= 1.7976931348623157e308);   // (a non-finite y must reach the fixed-point scale: fmax would drop a nan)
    ymax = fmax(ymax, fabs(a0));
  }
  // non-negative doubles -- and the quiet-nan pattern above them all -- order like integers
  if (ymax_bits && lane == 0) atomicMax(ymax_bits, ybad ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(ymax));
}

// Long x (A^T y: x has one element per sample): x is cut into slices that fit LDS and a workgroup walks its
// rows (rpw per wave) slice by slice.  The column indices of a CSR row ascend, so the entries of a slice are a
// contiguous run; the run limits of all (row, slice) pairs of a wave are found up front, one binary search per
// lane, and handed out with shuffles.
constexpr int SLICE_WAVES = 16, SLICE_MAX_RPW = 8;
template <typename T>
__global__ void __launch_bounds__(SLICE_WAVES * WAVE)
spmv_sliced_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const T* __restrict__ val, int64_t rows,
                   int64_t cols, int slice, int nslices, int rpw, const double* __restrict__ x, double* __restrict__ y) {
  extern __shared__ double xs[];
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  const int64_t r0 = ((int64_t)blockIdx.x * SLICE_WAVES + wave) * rpw;
  // lane (j, t): first entry of row r0 + j whose column is >= t * slice   (t = 0..nslices; rpw * (nslices+1) <= 64)
  int64_t bnd = 0;
  {
    const int j = lane / (nslices + 1), t = lane % (nslices + 1);
    if (j < rpw && r0 + j < rows) {
      int64_t lo = ptr[r0 + j], hi = ptr[r0 + j + 1];
      const int64_t key = (int64_t)t * slice;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)idx[mid] < key) lo = mid + 1; else hi = mid;
      }
      bnd = lo;
    }
  }
  double acc[SLICE_MAX_RPW];
#pragma unroll
  for (int j = 0; j < SLICE_MAX_RPW; ++j) acc[j] = 0;
  for (int t = 0; t < nslices; ++t) {
    const int64_t c0 = (int64_t)t * slice, c1 = min(cols, c0 + (int64_t)slice);
    __syncthreads();   // the previous slice's readers are done
    for (int64_t i = threadIdx.x; i < c1 - c0; i += blockDim.x) xs[i] = x[c0 + i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < SLICE_MAX_RPW; ++j) {
      if (j < rpw) {
        const int64_t e0 = __shfl(bnd, j * (nslices + 1) + t), e1 = __shfl(bnd, j * (nslices + 1) + t + 1);
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        int64_t e = e0 + lane;
        for (; e + 3 * WAVE < e1; e += 4 * WAVE) {
          const int k0 = __builtin_nontemporal_load(idx + e), k1 = __builtin_nontemporal_load(idx + e + WAVE);
          const int k2 = __builtin_nontemporal_load(idx + e + 2 * WAVE), k3 = __builtin_nontemporal_load(idx + e + 3 * WAVE);
          const T v0 = __builtin_nontemporal_load(val + e), v1 = __builtin_nontemporal_load(val + e + WAVE);
          const T v2 = __builtin_nontemporal_load(val + e + 2 * WAVE), v3 = __builtin_nontemporal_load(val + e + 3 * WAVE);
          a0 = fma((double)v0, xs[k0 - c0], a0);
          a1 = fma((double)v1, xs[k1 - c0], a1);
          a2 = fma((double)v2, xs[k2 - c0], a2);
          a3 = fma((double)v3, xs[k3 - c0], a3);
        }
        for (; e < e1; e += WAVE) a0 = fma((double)val[e], xs[idx[e] - c0], a0);
        acc[j] += (a0 + a1) + (a2 + a3);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < SLICE_MAX_RPW; ++j) {
    double a = acc[j];
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0 && j < rpw && r0 + j < rows) y[r0 + j] = a;
  }
}

// The same product on a (slice, row group) grid: a workgroup stages ONE slice of x and walks the runs of its
// row group inside that slice, so x is read once per row group instead of once per workgroup (C3: 38 MB
// instead of 410 MB beside the 1.3 GB of entries); the per-slice partial sums go to part[slice][row] and a
// second kernel adds them in slice order.  Lane j of a wave holds the run limits of the wave's j-th row.
// REL16: the entry stream reads `rel` = column mod slice as 2-byte values (the position inside the staged slice) instead of
// the 4-byte columns; the run limits are still searched in `idx`.
template <typename T, bool REL16>
__global__ void __launch_bounds__(SLICE_WAVES * WAVE)
spmv_slice_grid_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const uint16_t* __restrict__ rel,
                       const T* __restrict__ val, int64_t rows, int64_t cols, int slice, int nslices, int rows_per_group,
                       const double* __restrict__ x, double* __restrict__ part) {
  extern __shared__ double xs[];
  const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x / WAVE;
  const int t = blockIdx.x % nslices;
  const int64_t g0 = (int64_t)(blockIdx.x / nslices) * rows_per_group;
  const int64_t c0 = (int64_t)t * slice, c1 = min(cols, c0 + (int64_t)slice);
  // rows g0 + wave, g0 + wave + 16, ...: lane j searches the limits of the j-th of them (<= 64 rows per wave)
  int64_t e0l = 0, e1l = 0;
  {
    const int64_t r = g0 + wave + (int64_t)lane * SLICE_WAVES;
    if (lane * SLICE_WAVES + wave < rows_per_group && r < rows) {
      const int64_t b = ptr[r], e = ptr[r + 1];
      int64_t lo = b, hi = e;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)idx[mid] < c0) lo = mid + 1; else hi = mid;
      }
      e0l = lo;
      hi = e;
      while (lo < hi) {
        const int64_t mid = (lo + hi) >> 1;
        if ((int64_t)idx[mid] < c1) lo = mid + 1; else hi = mid;
      }
      e1l = lo;
    }
  }
  for (int64_t i = threadIdx.x; i < c1 - c0; i += blockDim.x) xs[i] = x[c0 + i];
  __syncthreads();
  const int nmine = (rows_per_group - wave + SLICE_WAVES - 1) / SLICE_WAVES;
  for (int j = 0; j < nmine; ++j) {
    const int64_t r = g0 + wave + (int64_t)j * SLICE_WAVES;
    if (r >= rows) break;
    const int64_t e0 = __shfl(e0l, j), e1 = __shfl(e1l, j);
    double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
    int64_t e = e0 + lane;
    for (; e + 3 * WAVE < e1; e += 4 * WAVE) {
      int k0, k1, k2, k3;
      if constexpr (REL16) {
        k0 = __builtin_nontemporal_load(rel + e); k1 = __builtin_nontemporal_load(rel + e + WAVE);
        k2 = __builtin_nontemporal_load(rel + e + 2 * WAVE); k3 = __builtin_nontemporal_load(rel + e + 3 * WAVE);
      } else {
        k0 = __builtin_nontemporal_load(idx + e) - (int)c0; k1 = __builtin_nontemporal_load(idx + e + WAVE) - (int)c0;
        k2 = __builtin_nontemporal_load(idx + e + 2 * WAVE) - (int)c0; k3 = __builtin_nontemporal_load(idx + e + 3 * WAVE) - (int)c0;
      }
      const T v0 = __builtin_nontemporal_load(val + e), v1 = __builtin_nontemporal_load(val + e + WAVE);
      const T v2 = __builtin_nontemporal_load(val + e + 2 * WAVE), v3 = __builtin_nontemporal_load(val + e + 3 * WAVE);
      a0 = fma((double)v0, xs[k0], a0);
      a1 = fma((double)v1, xs[k1], a1);
      a2 = fma((double)v2, xs[k2], a2);
      a3 = fma((double)v3, xs[k3], a3);
    }
    for (; e < e1; e += WAVE) a0 = fma((double)val[e], xs[REL16 ? (int)rel[e] : idx[e] - (int)c0], a0);
    double a = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) a += __shfl_xor(a, off);
    if (lane == 0) part[(int64_t)t * rows + r] = a;
  }
}

__global__ void slice_sum_kernel(const double* __restrict__ part, int64_t rows, int nslices, double* __restrict__ y) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  double s = 0;
  for (int t = 0; t < nslices; ++t) s += part[(int64_t)t * rows + r];
  y[r] = s;
}

// h[i] (+)= sum_r V[i][r] w[r], i < nvec: one block per basis vector; four independent partial sums per
// thread keep four loads in flight (a single dependent chain made this kernel 25 us for 18k elements)
constexpr int GEMV_T_THREADS = 512;
__global__ void __launch_bounds__(GEMV_T_THREADS)
gemv_t_kernel(const double* __restrict__ V, int64_t len, int nvec, const double* __restrict__ w,
              double* __restrict__ h, int accumulate) {
  __shared__ double red[GEMV_T_THREADS / WAVE];
  const int i = blockIdx.x;
  const double* v = V + (int64_t)i * len;
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  int64_t r = threadIdx.x;
  for (; r + 3 * GEMV_T_THREADS < len; r += 4 * GEMV_T_THREADS) {
    s0 = fma(v[r], w[r], s0);
    s1 = fma(v[r + GEMV_T_THREADS], w[r + GEMV_T_THREADS], s1);
    s2 = fma(v[r + 2 * GEMV_T_THREADS], w[r + 2 * GEMV_T_THREADS], s2);
    s3 = fma(v[r + 3 * GEMV_T_THREADS], w[r + 3 * GEMV_T_THREADS], s3);
  }
  for (; r < len; r += GEMV_T_THREADS) s0 = fma(v[r], w[r], s0);
  double s = (s0 + s1) + (s2 + s3);
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
#pragma unroll
    for (int k = 0; k < GEMV_T_THREADS / WAVE; ++k) t += red[k];
    h[i] = accumulate ? h[i] + t : t;
  }
}

// w[r] -= sum_i hcur[i] V[i][r]
__global__ void __launch_bounds__(256)
gemv_n_sub_kernel(const double* __restrict__ V, int64_t len, int nvec, const double* __restrict__ hcur,
                  double* __restrict__ w) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= len) return;
  double s = 0;
  for (int i = 0; i < nvec; ++i) s = fma(hcur[i], V[(int64_t)i * len + r], s);
  w[r] -= s;
}

// The second Gram-Schmidt pass fused with what follows it: w -= V^T h2, the partial sums of ||w||^2 of this block's
// elements (for scale_kernel), and alpha[j] = h1[j] + h2[j] (the diagonal of T is the projection on v_j itself).
__global__ void __launch_bounds__(256)
gemv_n_sub_norm_kernel(const double* __restrict__ V, int64_t len, int nvec, const double* __restrict__ h1,
                       const double* __restrict__ h2, double* __restrict__ w, double* __restrict__ partial, int j,
                       double* __restrict__ alpha) {
  __shared__ double red[4];
  double acc = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < len; r += (int64_t)gridDim.x * blockDim.x) {
    double s = 0;
    for (int i = 0; i < nvec; ++i) s = fma(h2[i], V[(int64_t)i * len + r], s);
    const double x = w[r] - s;
    w[r] = x;
    acc = fma(x, x, acc);
  }
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    if (blockIdx.x == 0) alpha[j] = h1[j] + h2[j];
  }
}

// beta = ||w||; out[j] = beta; vnext = w / beta  (single block computes the norm, then all scale)
__global__ void __launch_bounds__(256)
norm2_kernel(const double* __restrict__ w, int64_t len, double* __restrict__ partial) {
  __shared__ double red[4];
  double s = 0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < len; r += (int64_t)gridDim.x * blockDim.x)
    s = fma(w[r], w[r], s);
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
  if ((threadIdx.x & (WAVE - 1)) == 0) red[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ void __launch_bounds__(256)
scale_kernel(const double* __restrict__ w, int64_t len, const double* __restrict__ partial, int nparts,
             double* __restrict__ beta_out, double* __restrict__ vnext) {
  // the partial sums of ||w||^2 (norm2_kernel or the fused second Gram-Schmidt pass), reduced in a fixed order by every wave
  const int lane = threadIdx.x & (WAVE - 1);
  double s = 0;
  for (int i = lane; i < nparts; i += WAVE) s += partial[i];
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) s += __shfl_xor(s, off);
  const double beta = sqrt(s);
  if (blockIdx.x == 0 && threadIdx.x == 0) *beta_out = beta;
  const double inv = beta > 0 ? 1.0 / beta : 0.0;
  for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < len; r += (int64_t)gridDim.x * blockDim.x)
    vnext[r] = w[r] * inv;
}

// out[r][i] = sum_j V[j][r] S[j][i]   (Ritz vectors; out row-major len x ldo as T)
template <typename T>
__global__ void __launch_bounds__(256)
combine_kernel(const double* __restrict__ V, int64_t len, int nvec, const double* __restrict__ S, int k, int ldo,
               T* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int i = blockIdx.y;
  if (r >= len) return;
  double s = 0;
  if (i < k)
    for (int j = 0; j < nvec; ++j) s = fma(V[(int64_t)j * len + r], S[(int64_t)j * k + i], s);
  out[r * ldo + i] = (T)s;
}

__global__ void column_scale_f64_kernel(double* __restrict__ v, int64_t len, double factor) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < len) v[r] *= factor;
}

template <typename T>
__global__ void store_column_kernel(const double* __restrict__ v, int64_t len, int col, int ldo, T* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < len) out[r * ldo + col] = (T)v[r];
}

template <typename T>
__global__ void load_column_kernel(const T* __restrict__ in, int64_t len, int col, int ld, double* __restrict__ v) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r < len) v[r] = (double)in[r * ld + col];
}

inline unsigned grid1(int64_t n, int block = 256, int64_t cap = 1 << 30) {
  int64_t g = (n + block - 1) / block;
  return (unsigned)std::max<int64_t>(1, std::min(g, cap));
}

// which kernel spmv_launch takes for an operator, and the slice of x its 2-byte index copy is relative to (0: absolute)
enum SpmvKind { SPMV_ROW = 0, SPMV_LDSX, SPMV_SLICE_GRID, SPMV_SLICED };
template <typename T>
SpmvKind spmv_plan(const CsrView<T>& A, bool has_scratch, int* slice_out = nullptr, int* nslices_out = nullptr) {
  static const bool no_lds = dbg_env("SAPCA_SPMV_NO_LDS") != nullptr;
  static const bool no_grid = dbg_env("SAPCA_SPMV_NO_SLICE_GRID") != nullptr;
  const size_t xbytes = (size_t)A.cols * sizeof(double);
  if (A.rows == 0 || no_lds) return SPMV_ROW;
  if (xbytes <= 150 * 1024 && A.rows >= 4096) return SPMV_LDSX;
  if (A.nnz >= 256 * A.rows && A.rows >= 1024) {
    const int max_slice = 150 * 1024 / (int)sizeof(double);
    const int nslices = (int)((A.cols + max_slice - 1) / max_slice);
    const int slice = (int)((A.cols + nslices - 1) / nslices);
    if (slice_out) *slice_out = slice;
    if (nslices_out) *nslices_out = nslices;
    if (has_scratch && !no_grid && nslices > 1) return SPMV_SLICE_GRID;
    return SPMV_SLICED;
  }
  return SPMV_ROW;
}

__global__ void narrow_idx16_kernel(const int32_t* __restrict__ idx, int64_t count, int mod, uint16_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < count; e += stride)
    out[e] = (uint16_t)(mod > 0 ? idx[e] % mod : idx[e]);
}

// the 2-byte index copy an operator's Lanczos kernel reads (nullptr: that kernel takes the 4-byte columns)
template <typename T>
const uint16_t* spmv_narrow_indices(const CsrView<T>& A, bool has_scratch, DevBuf& buf, hipStream_t s) {
  static const bool off = dbg_env("SAPCA_SPMV_IDX32") != nullptr;
  int slice = 0, nslices = 0;
  const SpmvKind kind = spmv_plan(A, has_scratch, &slice, &nslices);
  if (off || A.nnz == 0 || (kind != SPMV_LDSX && kind != SPMV_SLICE_GRID)) return nullptr;
  if (kind == SPMV_LDSX && A.cols > 65536) return nullptr;
  if (kind == SPMV_SLICE_GRID && slice > 65536) return nullptr;
  uint16_t* out = buf.as<uint16_t>((size_t)A.nnz);
  hipLaunchKernelGGL(narrow_idx16_kernel, dim3((unsigned)std::min<int64_t>((A.nnz + 255) / 256, 8192)), dim3(256), 0, s, A.idx, A.nnz,
                     kind == SPMV_SLICE_GRID ? slice : 0, out);
  SAPCA_HIP(hipGetLastError());
  return out;
}

template <typename T>
void spmv_launch(const CsrView<T>& A, const double* x, double* y, hipStream_t s, DevBuf* scratch = nullptr,
                 const uint16_t* idx16 = nullptr, unsigned long long* ymax_bits = nullptr, bool* ymax_done = nullptr) {
  if (ymax_done) *ymax_done = false;   // (only the kernel with x in LDS gathers max |y| on the way)
  if (A.rows == 0) return;
  const size_t xbytes = (size_t)A.cols * sizeof(double);
  static const bool no_lds = dbg_env("SAPCA_SPMV_NO_LDS") != nullptr;
  if (!no_lds && xbytes <= 150 * 1024 && A.rows >= 4096) {
    static LdsAttrState attr, attr16;   // one per instantiation of this function template
    if (idx16) {
      ensure_dynamic_lds(reinterpret_cast<const void*>(&spmv_ldsx_kernel<T, uint16_t>), 150 * 1024, attr16);
      hipLaunchKernelGGL((spmv_ldsx_kernel<T, uint16_t>), dim3(256), dim3(1024), xbytes, s, A.ptr, idx16, A.val, A.rows, A.cols, x, y, ymax_bits);
    } else {
      ensure_dynamic_lds(reinterpret_cast<const void*>(&spmv_ldsx_kernel<T, int32_t>), 150 * 1024, attr);
      hipLaunchKernelGGL((spmv_ldsx_kernel<T, int32_t>), dim3(256), dim3(1024), xbytes, s, A.ptr, A.idx, A.val, A.rows, A.cols, x, y, ymax_bits);
    }
    if (ymax_done) *ymax_done = ymax_bits != nullptr;
    return;
  }
  if (!no_lds && A.nnz >= 256 * A.rows && A.rows >= 1024) {   // long rows: slices of x through LDS
    const int max_slice = 150 * 1024 / (int)sizeof(double);
    const int nslices = (int)((A.cols + max_slice - 1) / max_slice);
    const int slice = (int)((A.cols + nslices - 1) / nslices);
    // (slice, row group) grid when the caller lends a buffer for the per-slice partial sums
    static const bool no_grid = dbg_env("SAPCA_SPMV_NO_SLICE_GRID") != nullptr;
    if (scratch && !no_grid && nslices > 1) {
      const int64_t groups_one_round = std::max<int64_t>(1, 256 / nslices);
      const int rows_per_group = (int)std::min<int64_t>(SLICE_WAVES * WAVE, (A.rows + groups_one_round - 1) / groups_one_round);
      const int64_t groups = (A.rows + rows_per_group - 1) / rows_per_group;
      double* part = scratch->as<double>((size_t)nslices * A.rows);
      static LdsAttrState attr3, attr3r;
      if (idx16) {
        ensure_dynamic_lds(reinterpret_cast<const void*>(&spmv_slice_grid_kernel<T, true>), 150 * 1024, attr3r);
        hipLaunchKernelGGL((spmv_slice_grid_kernel<T, true>), dim3((unsigned)(groups * nslices)), dim3(SLICE_WAVES * WAVE),
                           (size_t)slice * sizeof(double), s, A.ptr, A.idx, idx16, A.val, A.rows, A.cols, slice, nslices, rows_per_group,
                           x, part);
      } else {
        ensure_dynamic_lds(reinterpret_cast<const void*>(&spmv_slice_grid_kernel<T, false>), 150 * 1024, attr3);
        hipLaunchKernelGGL((spmv_slice_grid_kernel<T, false>), dim3((unsigned)(groups * nslices)), dim3(SLICE_WAVES * WAVE),
                           (size_t)slice * sizeof(double), s, A.ptr, A.idx, (const uint16_t*)nullptr, A.val, A.rows, A.cols, slice,
                           nslices, rows_per_group, x, part);
      }
      hipLaunchKernelGGL(slice_sum_kernel, dim3(grid1(A.rows)), dim3(256), 0, s, part, A.rows, nslices, y);
      return;
    }
    // one round of workgroups (256 CUs x 16 waves) when the wave's run-limit table fits its 64 lanes
    int rpw = (int)((A.rows + 256 * SLICE_WAVES - 1) / (256 * SLICE_WAVES));
    if (rpw < 1) rpw = 1;
    if (rpw <= SLICE_MAX_RPW && rpw * (nslices + 1) <= WAVE) {
      static LdsAttrState attr2;
      ensure_dynamic_lds(reinterpret_cast<const void*>(&spmv_sliced_kernel<T>), 150 * 1024, attr2);
      const int rows_per_block = SLICE_WAVES * rpw;
      hipLaunchKernelGGL((spmv_sliced_kernel<T>), dim3((unsigned)((A.rows + rows_per_block - 1) / rows_per_block)),
                         dim3(SLICE_WAVES * WAVE), (size_t)slice * sizeof(double), s, A.ptr, A.idx, A.val, A.rows, A.cols, slice,
                         nslices, rpw, x, y);
      return;
    }
  }
  hipLaunchKernelGGL((spmv_kernel<T>), dim3(grid1(A.rows * WAVE, 256, 8192)), dim3(256), 0, s, A.ptr, A.idx, A.val,
                     A.rows, x, y);
}

}  // namespace

namespace k {
template <typename T>
void spmv(const CsrView<T>& A, const double* x, double* y, hipStream_t s) {
  spmv_launch(A, x, y, s);
  SAPCA_HIP(hipGetLastError());
}
template void spmv<float>(const CsrView<float>&, const double*, double*, hipStream_t);
template void spmv<double>(const CsrView<double>&, const double*, double*, hipStream_t);
}  // namespace k

template <typename T>
void lanczos_fit(sapca_handle_s& h) {
  hipStream_t s = h.stream;
  const CsrView<T> A = Engine<T>::view(h.a_used), At = Engine<T>::view(h.at_used);
  const int64_t m = A.rows, n_used = A.cols;
  const int64_t m_global = (int64_t)h.m_global;
  const int k = (int)h.opt.n_components;
  // las2 iterates on the smaller Gram matrix; with rows sharded over ranks only A^T A is supported
  const bool right_side = n_used <= m_global;
  SAPCA_CHECK(right_side || !h.comm.active(), SAPCA_ERR_ARG,
              "row-sharded Lanczos needs n_features <= n_samples (iteration on A^T A)");
  const int64_t len = right_side ? n_used : m;       // Lanczos vector length
  const int64_t other = right_side ? m : n_used;     // intermediate vector length
  const bool masked = h.has_mask_maps;
  // iteration caps of the two call sites (sparse/mod.rs:135, sparse_masked/mod.rs:321)
  const int64_t ref_iters = masked ? std::max<int64_t>(2 * std::max<int64_t>(m_global, n_used), 100)
                                   : std::max<int64_t>(m_global, n_used);
  int64_t jmax = std::min<int64_t>(std::min<int64_t>(len, ref_iters), 20LL * k + 600);
  jmax = std::min<int64_t>(jmax, std::max<int64_t>(k + 2, (int64_t)(12e9 / (8.0 * (double)len))));
  SAPCA_CHECK(jmax >= k, SAPCA_ERR_SVD, "SVD computation failed: n_components exceeds the Lanczos iteration limit");
  const double kappa = 10e-6;  // sparse/mod.rs:141

  // device workspace: basis V [(jmax+1) x len], w [len], tmp [other], h1/h2 [jmax+1], alpha/beta [jmax+1], partials
  const size_t nV = (size_t)(jmax + 1) * len;
  double* base = h.lanczos_buf.as<double>(nV + len + other + 4 * (size_t)(jmax + 2) + 1024);
  double* V = base;
  double* w = V + nV;
  double* tmp = w + len;
  double* h1 = tmp + other;
  double* h2 = h1 + (jmax + 2);
  double* alpha = h2 + (jmax + 2);
  double* beta = alpha + (jmax + 2);
  double* partial = beta + (jmax + 2);
  const int nparts = 256;

  // start vector from the seed (las2 takes random_seed: u32), normalised
  k::gaussian_panel(w, len, 1, 1, h.opt.random_seed, s);
  hipLaunchKernelGGL(norm2_kernel, dim3(nparts), dim3(256), 0, s, w, len, partial);
  hipLaunchKernelGGL(scale_kernel, dim3(grid1(len, 256, 1024)), dim3(256), 0, s, w, len, partial, nparts, beta + jmax + 1, V);

  // No transposed operator (prepare() took the scatter route: the transposed side fits LDS): the second product of a step
  // scatters A's rows into per-workgroup fixed-point copies of the output (scatter.hip) -- same bytes read as a gather
  // along A^T's rows, no transposition before the first step.
  const bool scatter = h.lz_scatter;
  SAPCA_CHECK(!scatter || (right_side && At.ptr == nullptr), SAPCA_ERR_HIP, "internal: Lanczos scatter route on the wrong side");
  SAPCA_CHECK(scatter || At.ptr != nullptr || At.nnz == 0, SAPCA_ERR_HIP, "internal: Lanczos without a transposed operator");
  unsigned long long* sc = scatter ? h.lz_scalars.ptr<unsigned long long>() : nullptr;   // max |a| (prepare) ; max |y| of even / odd products
  if (scatter) SAPCA_HIP(hipMemsetAsync(sc + 1, 0, 2 * sizeof(unsigned long long), s));
  int64_t product = 0;
  // 2-byte index copies for the two products of a step (one pass over the indices each, paid back in a few steps)
  const uint16_t* first16 = spmv_narrow_indices(right_side ? A : At, false, h.idx16_a, s);
  const uint16_t* second16 = scatter ? nullptr : spmv_narrow_indices(right_side ? At : A, true, h.idx16_b, s);
  auto apply_B = [&](const double* v, double* out) {  // out = A^T A v  (or A A^T v)
    if (scatter) {
      unsigned long long* ymax = sc + 1 + (product & 1), *next = sc + 1 + ((product + 1) & 1);
      ++product;
      bool have_max = false;
      spmv_launch(A, v, tmp, s, nullptr, first16, ymax, &have_max);
      if (!have_max) k::vecmax(tmp, other, ymax, s);
      k::spmvt_scatter(A, first16, tmp, sc, ymax, next, out, h.scratch2, s);   // (the same 2-byte indices: one copy serves both products)
      if (h.comm.active()) h.comm.allreduce(out, (uint64_t)len, 1, s);
    } else if (right_side) {
      spmv_launch(A, v, tmp, s, nullptr, first16);
      spmv_launch(At, tmp, out, s, &h.scratch2, second16);
      if (h.comm.active()) h.comm.allreduce(out, (uint64_t)len, 1, s);
    } else {
      spmv_launch(At, v, tmp, s, nullptr, first16);
      spmv_launch(A, tmp, out, s, &h.scratch2, second16);
    }
  };

  std::vector<double> theta, S;
  int64_t steps = 0;
  bool converged = false;
  // How often the host looks at T: a check is one small copy, a stream sync and an O(j^2) QL pass that carries only
  // the bottom row of the eigenvector matrix (the error bounds need nothing else), so while T is small every step is
  // checked and no step is run past convergence; as T grows the checks thin out.  SAPCA_LANCZOS_CHECK=<n> fixes the
  // interval (experiments).
  static const int check_env = dbg_env("SAPCA_LANCZOS_CHECK") ? atoi(dbg_env("SAPCA_LANCZOS_CHECK")) : 0;
  auto check_due = [&](int64_t j) {
    const int every = check_env > 0 ? check_env : j <= 64 ? 1 : j <= 128 ? 2 : j <= 256 ? 4 : 8;
    return j % every == 0;
  };
  double* ab_host = (double*)h.lanczos_host.ensure((size_t)2 * (jmax + 2) * sizeof(double));
  std::vector<int> top(k);
  for (int64_t j = 0; j < jmax; ++j) {
    const double* vj = V + (size_t)j * len;
    apply_B(vj, w);
    const int nvec = (int)(j + 1);
    hipLaunchKernelGGL(gemv_t_kernel, dim3(nvec), dim3(GEMV_T_THREADS), 0, s, V, len, nvec, w, h1, 0);
    hipLaunchKernelGGL(gemv_n_sub_kernel, dim3(grid1(len)), dim3(256), 0, s, V, len, nvec, h1, w);
    hipLaunchKernelGGL(gemv_t_kernel, dim3(nvec), dim3(GEMV_T_THREADS), 0, s, V, len, nvec, w, h2, 0);
    const int nb2 = (int)std::min<int64_t>((len + 255) / 256, 1024);   // (one partial sum of ||w||^2 per block: at most the 1024 slots)
    hipLaunchKernelGGL(gemv_n_sub_norm_kernel, dim3(nb2), dim3(256), 0, s, V, len, nvec, h1, h2, w, partial, (int)j, alpha);
    hipLaunchKernelGGL(scale_kernel, dim3(grid1(len, 256, 1024)), dim3(256), 0, s, w, len, partial, nb2, beta + j,
                       V + (size_t)(j + 1) * len);
    steps = j + 1;
    const bool last = steps == jmax;
    if (steps >= k && (check_due(steps) || last)) {
      // alpha and beta are neighbours on the device: one copy of both into page-locked memory
      SAPCA_HIP(hipMemcpyAsync(ab_host, alpha, (size_t)(jmax + 2 + steps) * sizeof(double), hipMemcpyDeviceToHost, s));
      SAPCA_HIP(hipStreamSynchronize(s));
      const double* b_host = ab_host + (jmax + 2);
      std::vector<double> d(ab_host, ab_host + steps), e(steps, 0.0);
      for (int64_t i = 1; i < steps; ++i) e[i] = b_host[i - 1];
      SAPCA_CHECK(tridiag_eigh(d, e, (int)steps, S, true), SAPCA_ERR_SVD,
                  "SVD computation failed: tridiagonal QL did not converge");
      theta = d;  // ascending
      const double bj = b_host[steps - 1];
      bool ok = std::isfinite(bj);
      for (int i = 0; i < k && ok; ++i) {
        const int c = (int)steps - 1 - i;
        const double bound = std::fabs(bj * S[c]);
        ok = theta[c] > 0 && bound <= kappa * std::fabs(theta[c]);
      }
      if (ok || bj <= 1e-300 * std::fabs(theta[steps - 1])) {  // converged, or the Krylov space is exhausted
        converged = true;
        break;
      }
    }
  }
  if (!converged) {
    // the reference would index res.s[i] past a shorter result and panic (sparse/mod.rs:213-215)
    throw Error(SAPCA_ERR_SVD, "SVD computation failed: Lanczos did not converge " + std::to_string(k) +
                                   " singular triplets within " + std::to_string(steps) + " steps");
  }
  h.timings.lanczos_steps = (uint64_t)steps;
  {  // the eigenvectors of the final T (the checks kept their bottom row only)
    const double* b_host = ab_host + (jmax + 2);
    std::vector<double> d(ab_host, ab_host + steps), e(steps, 0.0);
    for (int64_t i = 1; i < steps; ++i) e[i] = b_host[i - 1];
    SAPCA_CHECK(tridiag_eigh(d, e, (int)steps, S), SAPCA_ERR_SVD, "SVD computation failed: tridiagonal QL did not converge");
    theta = d;
  }

  // Ritz vectors of the k largest eigenvalues, sigma = sqrt(theta)
  std::vector<double> Sk((size_t)steps * k);
  h.sing.assign(k, 0.0);
  for (int i = 0; i < k; ++i) {
    const int c = (int)steps - 1 - i;
    h.sing[i] = std::sqrt(std::max(theta[c], 0.0));
    for (int64_t j = 0; j < steps; ++j) Sk[(size_t)j * k + i] = S[(size_t)j * steps + c];
  }
  const int ldk = (int)round_up(k, 16);
  double* Sdev = h.small.as<double>((size_t)6 * 128 * 128 + 64 + 4 * 128 + (size_t)steps * k);
  Sdev += (size_t)6 * 128 * 128 + 64 + 4 * 128;
  SAPCA_HIP(hipMemcpyAsync(Sdev, Sk.data(), Sk.size() * sizeof(double), hipMemcpyHostToDevice, s));
  T* VtT = h.panel_w.as<T>((size_t)std::max<int64_t>(n_used, 1) * ldk);
  k::fill_zero(VtT, n_used * ldk, s);
  if (right_side) {
    hipLaunchKernelGGL((combine_kernel<T>), dim3(grid1(len), k), dim3(256), 0, s, V, len, (int)steps, Sdev, k, ldk, VtT);
  } else {
    // left vectors U from the Krylov basis, then v_i = A^T u_i / sigma_i
    T* Ut = h.panel_y.as<T>((size_t)std::max<int64_t>(m, 1) * ldk);
    hipLaunchKernelGGL((combine_kernel<T>), dim3(grid1(len), k), dim3(256), 0, s, V, len, (int)steps, Sdev, k, ldk, Ut);
    for (int i = 0; i < k; ++i) {
      hipLaunchKernelGGL((load_column_kernel<T>), dim3(grid1(m)), dim3(256), 0, s, Ut, m, i, ldk, w);
      spmv_launch(At, w, tmp, s);
      hipLaunchKernelGGL(column_scale_f64_kernel, dim3(grid1(n_used)), dim3(256), 0, s, tmp, n_used,
                         h.sing[i] > 0 ? 1.0 / h.sing[i] : 0.0);
      hipLaunchKernelGGL((store_column_kernel<T>), dim3(grid1(n_used)), dim3(256), 0, s, tmp, n_used, i, ldk, VtT);
    }
  }
  T* comps = h.components_dev.as<T>((size_t)k * std::max<int64_t>(n_used, 1));
  k::flip_transpose(VtT, n_used, ldk, k, comps, h.scratch2, s);  // R13: svd_flip(u, vt, false)
  SAPCA_HIP(hipGetLastError());
  SAPCA_HIP(hipStreamSynchronize(s));
}

template void lanczos_fit<float>(sapca_handle_s&);
template void lanczos_fit<double>(sapca_handle_s&);

}  // namespace sapca
