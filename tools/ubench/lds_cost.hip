// LDS instruction cost microbenchmark: cycles per wave-instruction per CU with 16 waves issuing.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u2 __attribute__((ext_vector_type(2)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 40960; i += 1024) reinterpret_cast<float*>(lds)[i] = (float)i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0;
  // fixed per-lane addresses (no VALU in the loop); results are consumed by an empty asm
  unsigned ad[8];
  for (int u = 0; u < 8; ++u) {
    unsigned base = wave * 4096;
    if (MODE == 0) ad[u] = (base + u * 640 + (lane >> 5) * 2048 + (lane & 31) * 8) & 0x1fff8;
    if (MODE == 1) ad[u] = (base + u * 16 + (lane >> 5) * 8) & 0x1fff8;
    if (MODE == 2) ad[u] = (base + u * 640 + lane * 4) & 0x1fffc;
    if (MODE == 3) ad[u] = (base + u * 1280 + (lane >> 4) * 2560 + (lane & 15) * 16) & 0x1fff0;
    if (MODE == 4) ad[u] = (base + u * 32 + (lane >> 4) * 8) & 0x1fff8;
    if (MODE == 5) ad[u] = (base + u * 32 + (lane >> 5) * 16) & 0x1fff0;
    if (MODE == 6) ad[u] = (base + u * 8) & 0x1fff8;
    if (MODE == 7) ad[u] = (base + u * 640 + lane * 4) & 0x1fffc;                                   // atomic add, 64 lanes contiguous
    if (MODE == 8) ad[u] = (base + u * 640 + (lane >> 5) * 2048 + (lane & 31) * 4) & 0x1fffc;       // atomic add, 2 rows x 128 B
    if (MODE == 9) ad[u] = (base + u * 640 + (lane >> 5) * 2048 + (lane & 31) * 8) & 0x1fff8;       // ds_add_f64? (pk) 2 rows x 256 B
    asm volatile("" : "+v"(ad[u]));
  }
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (MODE == 7 || MODE == 8) { atomicAdd(reinterpret_cast<float*>(lds + ad[u]), 1.0f); }
      else if (MODE == 9) { atomicAdd(reinterpret_cast<double*>(lds + ad[u]), 1.0); }
      else if (MODE == 2) { float w = *reinterpret_cast<const float*>(lds + ad[u]); asm volatile("" :: "v"(w)); }
      else if (MODE == 3 || MODE == 5) { v4f w = *reinterpret_cast<const v4f*>(lds + ad[u]); asm volatile("" :: "v"(w)); }
      else { v2f w = *reinterpret_cast<const v2f*>(lds + ad[u]); asm volatile("" :: "v"(w)); }
    }
  }
  long long t1 = clock64();
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
  if (threadIdx.x == 0) reinterpret_cast<long long*>(out + 1024 * gridDim.x)[blockIdx.x] = t1 - t0;
}

template <int MODE> void run(const char* name, float* d, int iters) {
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 163840, 0, d, iters);
  hipDeviceSynchronize();
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 163840, 0, d, iters); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  std::vector<long long> t(256); hipMemcpy(t.data(), d + 1024 * 256, 256 * 8, hipMemcpyDeviceToHost);
  double cyc = 0; for (auto x : t) cyc += x; cyc /= 256;
  double instr_per_cu = (double)iters * 8 * 16;
  printf("%-40s %.3f ms  s_memtime-ticks/instr/CU %.2f  (ns/instr/CU %.2f)\n", name, ms, cyc / instr_per_cu, ms * 1e6 / instr_per_cu);
}
int main() {
  float* d; hipMalloc(&d, 1024 * 256 * 4 + 4096);
  int iters = 20000;
  run<0>("b64 panel (2 rows x 256B)", d, iters);
  run<1>("b64 entry (2 distinct addrs)", d, iters);
  run<2>("b32 contiguous 64 lanes", d, iters);
  run<3>("b128 panel (4 rows x 256B)", d, iters);
  run<4>("b64 entry (4 distinct addrs)", d, iters);
  run<5>("b128 entry (2 distinct addrs)", d, iters);
  run<6>("b64 full broadcast", d, iters);
  run<7>("ds_add_f32 64 lanes contiguous", d, iters);
  run<8>("ds_add_f32 2 rows x 128B", d, iters);
  run<9>("ds_add_f64 2 rows x 256B", d, iters);
  return 0;
}
