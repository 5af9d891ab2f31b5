// Sparse x dense-panel sweeps: Y = A X - 1 c^T   (A: CSR rows x cols, X: cols x ld panel).
//
// This one operator is every hot sweep of the path (the loops inside
// single_svdlib::randomized::randomized_svd, call sites
// /root/reference/src/dimred/pca/sparse/mod.rs:170-180, sparse_masked/mod.rs:341-351):
//   A   * Omega / Z   with the centring term  c = X^T mu  folded into the epilogue   (R8)
//   A^T * Y           by running the same kernel on the device-built CSR of A^T      (R9)
//   transform         with X = diag(cnt) V^T (quirk Q2) or V^T on mean-shifted values (Q3)
//
// Lane layout (wave64): a panel row of `ld` values is covered by LPR lanes holding VEC = 16 B
// worth of columns each (float4 / double2); the 64/LPR lane groups ("slots") take consecutive
// stored entries of the row, so one wave-instruction gathers 64/LPR panel rows of 16*LPR bytes
// each -- every panel access is a full, aligned 16-B-per-lane segment.  The slots are summed
// with wavefront shuffles at the end of a row.
#include <cstdlib>

#include "kernels.h"

namespace sapca {
namespace k {

namespace {

constexpr int WAVE = 64;

template <typename T> struct Vec;
template <> struct Vec<float> { using type = float4; static constexpr int N = 4; };
template <> struct Vec<double> { using type = double2; static constexpr int N = 2; };

template <typename T> __device__ inline void fma_vec(typename Vec<T>::type& acc, T a, const typename Vec<T>::type& x);
template <> __device__ inline void fma_vec<float>(float4& acc, float a, const float4& x) {
  acc.x = fmaf(a, x.x, acc.x); acc.y = fmaf(a, x.y, acc.y); acc.z = fmaf(a, x.z, acc.z); acc.w = fmaf(a, x.w, acc.w);
}
template <> __device__ inline void fma_vec<double>(double2& acc, double a, const double2& x) {
  acc.x = fma(a, x.x, acc.x); acc.y = fma(a, x.y, acc.y);
}
template <typename T> __device__ inline typename Vec<T>::type zero_vec();
template <> __device__ inline float4 zero_vec<float>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ inline double2 zero_vec<double>() { return make_double2(0.0, 0.0); }

template <typename T> __device__ inline T vec_get(const typename Vec<T>::type& v, int i);
template <> __device__ inline float vec_get<float>(const float4& v, int i) { return i == 0 ? v.x : i == 1 ? v.y : i == 2 ? v.z : v.w; }
template <> __device__ inline double vec_get<double>(const double2& v, int i) { return i == 0 ? v.x : v.y; }

template <typename T> __device__ inline typename Vec<T>::type shfl_xor_vec(const typename Vec<T>::type& v, int off);
template <> __device__ inline float4 shfl_xor_vec<float>(const float4& v, int off) {
  return make_float4(__shfl_xor(v.x, off), __shfl_xor(v.y, off), __shfl_xor(v.z, off), __shfl_xor(v.w, off));
}
template <> __device__ inline double2 shfl_xor_vec<double>(const double2& v, int off) {
  return make_double2(__shfl_xor(v.x, off), __shfl_xor(v.y, off));
}
template <typename T> __device__ inline void add_vec(typename Vec<T>::type& a, const typename Vec<T>::type& b);
template <> __device__ inline void add_vec<float>(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
template <> __device__ inline void add_vec<double>(double2& a, const double2& b) { a.x += b.x; a.y += b.y; }

// ---------------------------------------------------------------------------------------
// Variant 1: one wave per row, panel rows gathered straight from L2 / Infinity Cache.
// ---------------------------------------------------------------------------------------
// SHIFT: every stored value enters as (value - shift[column]) -- quirk Q3, the masked projection's centring at stored
// entries only (sparse_masked/mod.rs:488-529) -- instead of a pass that writes the shifted values out first
template <typename T, int LPR, bool SHIFT = false>
__global__ void __launch_bounds__(256)
spmm_rowgather_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const T* __restrict__ val,
                      int64_t rows, const T* __restrict__ X, int ldx, T* __restrict__ Y, int ldy, int ncols,
                      const T* __restrict__ cvec, int vec_store, const T* __restrict__ shift = nullptr) {
  using V = typename Vec<T>::type;
  constexpr int VEC = Vec<T>::N;
  constexpr int SLOTS = WAVE / LPR;
  constexpr int COVER = LPR * VEC;
  const int lane = threadIdx.x & (WAVE - 1);
  const int q = lane % LPR;
  const int slot = lane / LPR;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    for (int c0 = 0; c0 < ldx; c0 += COVER) {
      const int col = c0 + q * VEC;
      const bool live = col < ldx;
      const T* __restrict__ xq = X + (live ? col : 0);
      V acc0 = zero_vec<T>(), acc1 = zero_vec<T>();
      int64_t e = e0 + slot;
      for (; e + 3 * SLOTS < e1; e += 4 * SLOTS) {
        const int32_t ca = idx[e], cb = idx[e + SLOTS], cc = idx[e + 2 * SLOTS], cd = idx[e + 3 * SLOTS];
        T va = val[e], vb = val[e + SLOTS], vc = val[e + 2 * SLOTS], vd = val[e + 3 * SLOTS];
        if constexpr (SHIFT) { va -= shift[ca]; vb -= shift[cb]; vc -= shift[cc]; vd -= shift[cd]; }
        const V xa = *reinterpret_cast<const V*>(xq + (int64_t)ca * ldx);
        const V xb = *reinterpret_cast<const V*>(xq + (int64_t)cb * ldx);
        const V xc = *reinterpret_cast<const V*>(xq + (int64_t)cc * ldx);
        const V xd = *reinterpret_cast<const V*>(xq + (int64_t)cd * ldx);
        fma_vec<T>(acc0, va, xa);
        fma_vec<T>(acc1, vb, xb);
        fma_vec<T>(acc0, vc, xc);
        fma_vec<T>(acc1, vd, xd);
      }
      for (; e < e1; e += SLOTS) {
        const int32_t ca = idx[e];
        T va = val[e];
        if constexpr (SHIFT) va -= shift[ca];
        const V xa = *reinterpret_cast<const V*>(xq + (int64_t)ca * ldx);
        fma_vec<T>(acc0, va, xa);
      }
      add_vec<T>(acc0, acc1);
#pragma unroll
      for (int off = LPR; off < WAVE; off <<= 1) add_vec<T>(acc0, shfl_xor_vec<T>(acc0, off));
      if (slot == 0 && live) {
        T* __restrict__ y = Y + r * (int64_t)ldy + col;
        if (vec_store && col + VEC <= ncols) {
          V out = acc0;
          if (cvec) {
            const V c = *reinterpret_cast<const V*>(cvec + col);
            if constexpr (VEC == 4) { out.x -= c.x; out.y -= c.y; out.z -= c.z; out.w -= c.w; }
            else { out.x -= c.x; out.y -= c.y; }
          }
          *reinterpret_cast<V*>(y) = out;
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i)
            if (col + i < ncols) y[i] = vec_get<T>(acc0, i) - (cvec ? cvec[col + i] : (T)0);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// The same row walk for panels of 16 or 32 lanes per row, with the entry stream read once: a lane loads ONE entry of a
// chunk (index, value, and the shift of its column), coalesced, and step t of the chunk reaches the 16 lanes of a DPP
// row through row_newbcast:t -- one index/value/shift load per 16 (or 8) entries and lane instead of one per entry.
// Slot s (a 16- or 32-lane group) takes entries [16 s, 16 s + 16) of a chunk; rows shorter than a chunk and the last
// partial chunk go the per-entry way of the kernel above.
// ---------------------------------------------------------------------------------------
template <int STEP> __device__ __forceinline__ int bcast_row(int v) {
  return __builtin_amdgcn_update_dpp(0, v, 0x150 + STEP, 0xf, 0xf, false);   // row_newbcast:STEP
}
template <int STEP> __device__ __forceinline__ float bcast_row(float v) { return __int_as_float(bcast_row<STEP>(__float_as_int(v))); }
template <int STEP> __device__ __forceinline__ double bcast_row(double v) {
  const int lo = bcast_row<STEP>(__double2loint(v)), hi = bcast_row<STEP>(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

template <typename T, int STEP>
__device__ __forceinline__ void bcast_steps(int c, T v, const T* __restrict__ xq, int ldx, typename Vec<T>::type& acc0,
                                            typename Vec<T>::type& acc1) {
  using V = typename Vec<T>::type;
  if constexpr (STEP < 16) {
    const int ca = bcast_row<STEP>(c), cb = bcast_row<STEP + 1>(c);
    const T va = bcast_row<STEP>(v), vb = bcast_row<STEP + 1>(v);
    const V xa = *reinterpret_cast<const V*>(xq + (int64_t)ca * ldx);
    const V xb = *reinterpret_cast<const V*>(xq + (int64_t)cb * ldx);
    fma_vec<T>(acc0, va, xa);
    fma_vec<T>(acc1, vb, xb);
    bcast_steps<T, STEP + 2>(c, v, xq, ldx, acc0, acc1);
  }
}

template <typename T, int LPR, bool SHIFT>
__global__ void __launch_bounds__(256)
spmm_rowbcast_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ idx, const T* __restrict__ val,
                     int64_t rows, const T* __restrict__ X, int ldx, T* __restrict__ Y, int ldy, int ncols,
                     const T* __restrict__ cvec, int vec_store, const T* __restrict__ shift) {
  static_assert(LPR == 16 || LPR == 32, "a slot is one or two DPP rows");
  using V = typename Vec<T>::type;
  constexpr int VEC = Vec<T>::N;
  constexpr int SLOTS = WAVE / LPR;
  constexpr int COVER = LPR * VEC;
  constexpr int CHUNK = SLOTS * 16;
  const int lane = threadIdx.x & (WAVE - 1);
  const int q = lane % LPR;
  const int slot = lane / LPR;
  const int mine = slot * 16 + (lane & 15);      // the entry of a chunk this lane loads
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / WAVE;
  const int64_t nwaves = (int64_t)gridDim.x * blockDim.x / WAVE;
  for (int64_t r = wave; r < rows; r += nwaves) {
    const int64_t e0 = ptr[r], e1 = ptr[r + 1];
    for (int c0 = 0; c0 < ldx; c0 += COVER) {
      const int col = c0 + q * VEC;
      const bool live = col < ldx;
      const T* __restrict__ xq = X + (live ? col : 0);
      V acc0 = zero_vec<T>(), acc1 = zero_vec<T>();
      int64_t e = e0;
      for (; e + CHUNK <= e1; e += CHUNK) {
        const int c = idx[e + mine];
        T v = val[e + mine];
        if constexpr (SHIFT) v -= shift[c];
        bcast_steps<T, 0>(c, v, xq, ldx, acc0, acc1);
      }
      for (e += slot; e < e1; e += SLOTS) {
        const int32_t ca = idx[e];
        T va = val[e];
        if constexpr (SHIFT) va -= shift[ca];
        const V xa = *reinterpret_cast<const V*>(xq + (int64_t)ca * ldx);
        fma_vec<T>(acc0, va, xa);
      }
      add_vec<T>(acc0, acc1);
#pragma unroll
      for (int off = LPR; off < WAVE; off <<= 1) add_vec<T>(acc0, shfl_xor_vec<T>(acc0, off));
      if (slot == 0 && live) {
        T* __restrict__ y = Y + r * (int64_t)ldy + col;
        if (vec_store && col + VEC <= ncols) {
          V out = acc0;
          if (cvec) {
            const V cv = *reinterpret_cast<const V*>(cvec + col);
            if constexpr (VEC == 4) { out.x -= cv.x; out.y -= cv.y; out.z -= cv.z; out.w -= cv.w; }
            else { out.x -= cv.x; out.y -= cv.y; }
          }
          *reinterpret_cast<V*>(y) = out;
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i)
            if (col + i < ncols) y[i] = vec_get<T>(acc0, i) - (cvec ? cvec[col + i] : (T)0);
        }
      }
    }
  }
}

template <typename T, int LPR>
void launch_rowgather(const CsrView<T>& A, const T* X, int ldx, T* Y, int ldy, int ncols, const T* cvec, hipStream_t s,
                      const T* shift = nullptr) {
  constexpr int VEC = Vec<T>::N;
  const int vec_store = (ldy % VEC == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0) &&
                        (cvec == nullptr || (reinterpret_cast<uintptr_t>(cvec) & 15) == 0);
  int64_t blocks = (A.rows + 3) / 4;
  if (blocks > 16384) blocks = 16384;
  if (blocks < 1) blocks = 1;
  static const bool per_entry = dbg_env("SAPCA_ROWGATHER_PER_ENTRY") != nullptr;   // (experiments: the kernel without the DPP feed)
  if constexpr (LPR >= 16) {
    if (!per_entry) {
      if (shift)
        hipLaunchKernelGGL((spmm_rowbcast_kernel<T, LPR, true>), dim3((unsigned)blocks), dim3(256), 0, s, A.ptr, A.idx, A.val,
                           A.rows, X, ldx, Y, ldy, ncols, cvec, vec_store, shift);
      else
        hipLaunchKernelGGL((spmm_rowbcast_kernel<T, LPR, false>), dim3((unsigned)blocks), dim3(256), 0, s, A.ptr, A.idx, A.val,
                           A.rows, X, ldx, Y, ldy, ncols, cvec, vec_store, shift);
      return;
    }
  }
  if (shift)
    hipLaunchKernelGGL((spmm_rowgather_kernel<T, LPR, true>), dim3((unsigned)blocks), dim3(256), 0, s, A.ptr, A.idx, A.val,
                       A.rows, X, ldx, Y, ldy, ncols, cvec, vec_store, shift);
  else
    hipLaunchKernelGGL((spmm_rowgather_kernel<T, LPR>), dim3((unsigned)blocks), dim3(256), 0, s, A.ptr, A.idx, A.val,
                       A.rows, X, ldx, Y, ldy, ncols, cvec, vec_store, shift);
}

}  // namespace

template <typename T>
void spmm(const CsrView<T>& A, const TiledOp* tiled, const T* X, int ldx, T* Y, int ldy, int ncols, const T* cvec,
          int variant, DevBuf& scratch, hipStream_t s, PanelSource<T>* keep) {
  constexpr int VEC = Vec<T>::N;
  if (keep) { keep->parts = Y; keep->nsplit = 1; keep->slab_stride = 0; }
  SAPCA_CHECK(ldx % VEC == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0, SAPCA_ERR_ARG,
              "spmm: panel must be 16-byte aligned with a leading dimension that is a multiple of 16 bytes");
  SAPCA_CHECK(ncols <= ldx && ncols <= ldy, SAPCA_ERR_ARG, "spmm: ncols exceeds a leading dimension");
  if (A.rows == 0) return;
  if constexpr (sizeof(T) == 4) {
    // variant 1 forces the row kernel; otherwise the LDS-staged sweep runs whenever its format exists
    const bool geom_ok = tiled && tiled->elem == 4 && (tiled->ldp == ldx || (tiled->fmt == 1 && tiled->ldp == 64 && ldx % 64 == 0));
    if (variant != 1 && tiled && tiled->valid && geom_ok && tiled->rows == A.rows && tiled->cols == A.cols) {
      spmm_tiled(*tiled, reinterpret_cast<const float*>(X), ldx, reinterpret_cast<float*>(Y), ldy, ncols,
                 reinterpret_cast<const float*>(cvec), scratch, s, reinterpret_cast<PanelSource<float>*>(keep));
      return;
    }
  } else {
    // f64: 64-column tile geometry (512-byte rows), 128-column panels in two passes
    if (variant != 1 && tiled && tiled->valid && tiled->elem == 8 && ldx % tiled->ldp == 0 && tiled->rows == A.rows &&
        tiled->cols == A.cols) {
      spmm_tiled(*tiled, reinterpret_cast<const double*>(X), ldx, reinterpret_cast<double*>(Y), ldy, ncols,
                 reinterpret_cast<const double*>(cvec), scratch, s, reinterpret_cast<PanelSource<double>*>(keep));
      return;
    }
  }
  const int lanes_needed = (ldx + VEC - 1) / VEC;
  if (lanes_needed <= 4) launch_rowgather<T, 4>(A, X, ldx, Y, ldy, ncols, cvec, s);
  else if (lanes_needed <= 8) launch_rowgather<T, 8>(A, X, ldx, Y, ldy, ncols, cvec, s);
  else if (lanes_needed <= 16) launch_rowgather<T, 16>(A, X, ldx, Y, ldy, ncols, cvec, s);
  else launch_rowgather<T, 32>(A, X, ldx, Y, ldy, ncols, cvec, s);
  SAPCA_HIP(hipGetLastError());
}

// Y[r][j] = sum over the stored entries of row r of (value - shift[column]) X[column][j]: the row kernel with the shift folded in
template <typename T>
void spmm_rows_shifted(const CsrView<T>& A, const T* X, int ldx, T* Y, int ldy, int ncols, const T* shift, hipStream_t s) {
  constexpr int VEC = Vec<T>::N;
  SAPCA_CHECK(ldx % VEC == 0 && (reinterpret_cast<uintptr_t>(X) & 15) == 0 && ncols <= ldx && ncols <= ldy && shift != nullptr, SAPCA_ERR_ARG,
              "spmm_rows_shifted: bad panel or shift");
  if (A.rows == 0) return;
  const int lanes_needed = (ldx + VEC - 1) / VEC;
  if (lanes_needed <= 4) launch_rowgather<T, 4>(A, X, ldx, Y, ldy, ncols, (const T*)nullptr, s, shift);
  else if (lanes_needed <= 8) launch_rowgather<T, 8>(A, X, ldx, Y, ldy, ncols, (const T*)nullptr, s, shift);
  else if (lanes_needed <= 16) launch_rowgather<T, 16>(A, X, ldx, Y, ldy, ncols, (const T*)nullptr, s, shift);
  else launch_rowgather<T, 32>(A, X, ldx, Y, ldy, ncols, (const T*)nullptr, s, shift);
  SAPCA_HIP(hipGetLastError());
}
template void spmm_rows_shifted<float>(const CsrView<float>&, const float*, int, float*, int, int, const float*, hipStream_t);
template void spmm_rows_shifted<double>(const CsrView<double>&, const double*, int, double*, int, int, const double*, hipStream_t);

template void spmm<float>(const CsrView<float>&, const TiledOp*, const float*, int, float*, int, int, const float*, int, DevBuf&, hipStream_t, PanelSource<float>*);
template void spmm<double>(const CsrView<double>&, const TiledOp*, const double*, int, double*, int, int, const double*, int, DevBuf&, hipStream_t, PanelSource<double>*);

}  // namespace k
}  // namespace sapca
