// Gaussian test matrix Omega for the randomized range finder (first step of
// single_svdlib::randomized::randomized_svd; call site
// /root/reference/src/dimred/pca/sparse/mod.rs:170-180, seed = `random_seed as u64`).
// The reference draws from rand 0.9's StdRng, a stream that cannot be reproduced without the
// crate; any i.i.d. N(0,1) matrix is equivalent for the algorithm.  Here entry (r, j) is a pure
// function of (seed, r*l + j): a SplitMix64 counter hash and Box-Muller in f64 -- the same
// function as sapca.synth.gaussian_panel, so hosts can reproduce Omega exactly.
#include "kernels.h"

namespace sapca {
namespace k {

namespace {

__device__ inline uint64_t mix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

__device__ inline double hash_u01(uint64_t seed, uint64_t stream, uint64_t idx) {
  const uint64_t key = seed * 0xD1342543DE82EF95ull + stream * 0x2545F4914F6CDD1Dull + 0x1234567ull;
  uint64_t z = mix64(idx ^ key);
  z = mix64(z + key);
  return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}

template <typename T>
__global__ void gaussian_panel_kernel(T* __restrict__ out, int64_t rows, int l, int ld, uint64_t seed) {
  const int64_t total = rows * ld;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int64_t r = i / ld;
    const int j = (int)(i - r * ld);
    T v = (T)0;
    if (j < l) {
      const uint64_t idx = (uint64_t)(r * l + j);
      const double u1 = fmax(hash_u01(seed, 11, idx), 1.0 / 9007199254740992.0);
      const double u2 = hash_u01(seed, 12, idx);
      v = (T)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2));
    }
    out[i] = v;
  }
}

}  // namespace

template <typename T>
void gaussian_panel(T* out, int64_t rows, int l, int ld, uint32_t seed, hipStream_t s) {
  if (rows == 0) return;
  int64_t g = (rows * ld + 255) / 256;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL((gaussian_panel_kernel<T>), dim3((unsigned)g), dim3(256), 0, s, out, rows, l, ld, (uint64_t)seed);
  SAPCA_HIP(hipGetLastError());
}

template void gaussian_panel<float>(float*, int64_t, int, int, uint32_t, hipStream_t);
template void gaussian_panel<double>(double*, int64_t, int, int, uint32_t, hipStream_t);

}  // namespace k
}  // namespace sapca
