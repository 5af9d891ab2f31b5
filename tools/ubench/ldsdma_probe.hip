// Probe: does global_load_lds_dwordx4 reach LDS offsets above 64 KiB through M0, and does the instruction offset move
// the LDS destination as well as the global source?
// Build: hipcc --offload-arch=gfx950 -O3 -o bin/ldsdma_probe ldsdma_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(64) k(const float* in, float* out, unsigned base, int use_off) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < 40960; i += 64) reinterpret_cast<float*>(lds)[i] = -1.f;
  __syncthreads();
  unsigned voff = threadIdx.x * 16;
  unsigned lb = (unsigned)(size_t)lds + base;
  lb = __builtin_amdgcn_readfirstlane(lb);
  if (use_off)
    asm volatile("s_mov_b32 m0, %[lb]\n s_nop 0\n global_load_lds_dwordx4 %[vo], %[p] offset:1024\n s_waitcnt vmcnt(0)\n"
                 :: [vo] "v"(voff), [p] "s"(in), [lb] "s"(lb) : "memory", "m0");
  else
    asm volatile("s_mov_b32 m0, %[lb]\n s_nop 0\n global_load_lds_dwordx4 %[vo], %[p]\n s_waitcnt vmcnt(0)\n"
                 :: [vo] "v"(voff), [p] "s"(in), [lb] "s"(lb) : "memory", "m0");
  __syncthreads();
  for (int i = threadIdx.x; i < 40960; i += 64) out[i] = reinterpret_cast<float*>(lds)[i];
}
int main() {
  float *in, *out;
  hipMalloc(&in, 1 << 16); hipMalloc(&out, 163840);
  std::vector<float> h(16384); for (int i = 0; i < 16384; ++i) h[i] = (float)i;
  hipMemcpy(in, h.data(), 65536, hipMemcpyHostToDevice);
  hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  std::vector<float> o(40960);
  for (int use_off = 0; use_off < 2; ++use_off)
    for (unsigned base : {0u, 4096u, 65536u, 81920u, 131072u, 162816u}) {
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 163840, 0, in, out, base, use_off);
      hipDeviceSynchronize();
      hipMemcpy(o.data(), out, 163840, hipMemcpyDeviceToHost);
      int first = -1, cnt = 0; float v0 = 0;
      for (int i = 0; i < 40960; ++i) if (o[i] != -1.f) { if (first < 0) { first = i; v0 = o[i]; } ++cnt; }
      printf("inst offset %d  M0 base %6u: %d dwords written, first at byte %d (value %.0f)\n", use_off ? 1024 : 0, base, cnt, first * 4, v0);
    }
  return 0;
}
