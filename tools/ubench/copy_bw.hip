// Streaming-copy variants (16 bytes per lane): which launch shape reaches the HBM rate the guide quotes (6.29 TB/s)?
// Build: hipcc --offload-arch=gfx950 -O3 -o bin/copy_bw copy_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k_flat(const u4* __restrict__ s, u4* __restrict__ d, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) d[i] = s[i];
}
template <int U, bool NT>
__global__ void __launch_bounds__(256) k_stride(const u4* __restrict__ s, u4* __restrict__ d, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    u4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(s + i + u * stride) : s[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) { if (NT) __builtin_nontemporal_store(v[u], d + i + u * stride); else d[i + u * stride] = v[u]; }
  }
  for (; i < n; i += stride) d[i] = s[i];
}
template <int U>
__global__ void __launch_bounds__(256) k_block(const u4* __restrict__ s, u4* __restrict__ d, int64_t n) {   // a workgroup owns U consecutive 4 KiB pieces
  int64_t i = ((int64_t)blockIdx.x * U) * 256 + threadIdx.x;
  u4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = i + u * 256 < n ? s[i + u * 256] : u4{0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < U; ++u) if (i + u * 256 < n) d[i + u * 256] = v[u];
}
#define TIME(name, ...)                                                                           \
  {                                                                                               \
    float best = 1e9;                                                                             \
    for (int r = 0; r < 6; ++r) {                                                                 \
      hipEventRecord(a); __VA_ARGS__; hipEventRecord(b); hipEventSynchronize(b);                  \
      float ms; hipEventElapsedTime(&ms, a, b); if (r && ms < best) best = ms;                    \
    }                                                                                             \
    printf("%-34s %8.1f GB/s\n", name, 2.0 * bytes / (best * 1e-3) / 1e9);                        \
  }
int main() {
  const int64_t bytes = (int64_t)1 << 30, n = bytes / 16;
  u4 *s, *d; hipMalloc(&s, bytes); hipMalloc(&d, bytes); hipMemset(s, 1, bytes); hipMemset(d, 0, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  TIME("hipMemcpyDtoD", hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0));
  TIME("flat (one 16 B per thread)", k_flat<<<(unsigned)((n + 255) / 256), 256>>>(s, d, n));
  TIME("grid-stride 2048 wg x4", k_stride<4, false><<<2048, 256>>>(s, d, n));
  TIME("grid-stride 4096 wg x4 nt", k_stride<4, true><<<4096, 256>>>(s, d, n));
  TIME("grid-stride 8192 wg x2", k_stride<2, false><<<8192, 256>>>(s, d, n));
  TIME("grid-stride 16384 wg x1", k_stride<1, false><<<16384, 256>>>(s, d, n));
  TIME("block x4 (16 KiB per workgroup)", k_block<4><<<(unsigned)((n + 1023) / 1024), 256>>>(s, d, n));
  TIME("block x8 (32 KiB per workgroup)", k_block<8><<<(unsigned)((n + 2047) / 2048), 256>>>(s, d, n));
  return 0;
}
