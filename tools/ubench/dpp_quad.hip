// Micro-benchmark of the DPP-fed quad sweep core (see gen_dpp_quad.py): entries come straight from global memory,
// one coalesced 512-byte load per 16 steps, and reach the lanes of their row group through DPP row_newbcast; the
// panel tile sits in LDS (80 KiB, loaded once here: no refills); accumulators are addressed through the VGPR index.
// Checks the result of wave 0 against the host and reports cycles per entry slot per CU.
// Build: python3 gen_dpp_quad.py dpp_quad_gen.h && hipcc --offload-arch=gfx950 -O3 -o bin/dpp_quad dpp_quad.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include "dpp_quad_gen.h"

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

constexpr int TILE_ROWS = 320, CHUNKS_PER_BLOCK = 30;

// VAR 0: pk_fma, 8 rows per group   1: v_fma, 8 rows   2: pk_fma, 16 rows   3: depth 3   4: groups flow across chunks
// 5: both   6: ablation, no LDS reads   7: ablation, no FMAs (6 and 7 compute nothing meaningful)
template <int VAR>
__global__ void __launch_bounds__(1024) k_dq(const float* __restrict__ P, const uint32_t* __restrict__ ent,
                                             const uint32_t* __restrict__ desc, long long wave_stride_bytes,
                                             long long desc_stride_bytes, int nblocks, float* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  for (int i = threadIdx.x; i < TILE_ROWS * 64; i += blockDim.x) reinterpret_cast<float*>(lds)[i] = P[i];
  __syncthreads();
  constexpr int RG = VAR == 2 ? 16 : 8;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * (blockDim.x >> 6) + wave;
  const int g = lane >> 4, i = lane & 15;
  unsigned lb = (unsigned)(size_t)lds + i * 16;
  unsigned eoff = i * 32 + g * 8;
  unsigned doff = lane * 4;
  unsigned long long pu = (unsigned long long)(reinterpret_cast<const char*>(ent) + (long long)gw * wave_stride_bytes);
  pu = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(pu >> 32)) << 32) |
       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)pu);
  unsigned long long du = (unsigned long long)(reinterpret_cast<const char*>(desc) + (long long)gw * desc_stride_bytes);
  du = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(du >> 32)) << 32) |
       (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)du);
  int n = __builtin_amdgcn_readfirstlane(nblocks);
  unsigned ooff = (unsigned)((gw * 4 * RG * 64 + lane) * 4);
  float tmp;
#define DQ_OPS                                                                                            \
  : [n] "+s"(n), [ooff] "+v"(ooff), [tmp] "=&v"(tmp)                                                     \
  : [lb] "v"(lb), [eoff] "v"(eoff), [doff] "v"(doff), [ptr] "s"(pu), [dptr] "s"(du), [optr] "s"(out)
  if constexpr (VAR == 0) asm volatile(DQ_ASM_PK8 DQ_OPS : DQ_CLOB_PK8);
  if constexpr (VAR == 1) asm volatile(DQ_ASM_F8 DQ_OPS : DQ_CLOB_F8);
  if constexpr (VAR == 2) asm volatile(DQ_ASM_PK16 DQ_OPS : DQ_CLOB_PK16);
  if constexpr (VAR == 3) asm volatile(DQ_ASM_PK8D3 DQ_OPS : DQ_CLOB_PK8D3);
  if constexpr (VAR == 4) asm volatile(DQ_ASM_PK8X DQ_OPS : DQ_CLOB_PK8X);
  if constexpr (VAR == 5) asm volatile(DQ_ASM_PK8XD3 DQ_OPS : DQ_CLOB_PK8XD3);
  if constexpr (VAR == 6) asm volatile(DQ_ASM_PK8_NOLDS DQ_OPS : DQ_CLOB_PK8_NOLDS);
  if constexpr (VAR == 7) asm volatile(DQ_ASM_PK8_NOFMA DQ_OPS : DQ_CLOB_PK8_NOFMA);
  if constexpr (VAR == 8) asm volatile(DQ_ASM_PK8_ON DQ_OPS : DQ_CLOB_PK8_ON);
}

static std::vector<float> hP;
static std::vector<uint32_t> hE, hD;   // wave 0's stream and descriptors
static float* dP;
static uint32_t *dE, *dD;
static float* dOut;

template <int VAR>
void run(const char* name, int threads, int nblocks, long long estride, long long dstride) {
  constexpr int RG = VAR == 2 ? 16 : 8;
  const int wpb = threads / 64;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dq<VAR>), hipFuncAttributeMaxDynamicSharedMemorySize, 81920));
  CK(hipMemset(dOut, 0, (size_t)4096 * 64 * 64 * 4));
  hipLaunchKernelGGL(k_dq<VAR>, dim3(256), dim3(threads), 81920, 0, dP, dE, dD, estride, dstride, nblocks, dOut);
  CK(hipDeviceSynchronize());
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_dq<VAR>, dim3(256), dim3(threads), 81920, 0, dP, dE, dD, estride, dstride, nblocks, dOut);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    best = ms < best ? ms : best;
  }
  // host reference for wave 0
  const int nchunks = nblocks * CHUNKS_PER_BLOCK;
  std::vector<double> ref((size_t)4 * RG * 64, 0.0);
  for (int c = 0; c < nchunks; ++c)
    for (int s = 0; s < 16; ++s) {
      const int grp = c * 8 + s / 2;                       // two steps share a descriptor byte
      const int blk = c / CHUNKS_PER_BLOCK, cb = c % CHUNKS_PER_BLOCK;
      const uint32_t dw = hD[(size_t)blk * 64 + cb * 2 + (s / 8)];
      const int j4 = (dw >> (8 * ((s / 2) & 3))) & 0xff;
      (void)grp;
      for (int g = 0; g < 4; ++g) {
        const size_t e = ((size_t)c * 64 + s * 4 + g) * 2;
        const uint32_t off = hE[e];
        float v; memcpy(&v, &hE[e + 1], 4);
        const int row = off / 256;
        for (int q = 0; q < 16; ++q)
          for (int cc = 0; cc < 4; ++cc) ref[(size_t)(j4 + cc) * 64 + g * 16 + q] += (double)v * hP[(size_t)row * 64 + q * 4 + cc];
      }
    }
  std::vector<float> o((size_t)4 * RG * 64);
  CK(hipMemcpy(o.data(), dOut, o.size() * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  double maxrel = 0;
  for (size_t k = 0; k < o.size(); ++k) {
    const double d = fabs(o[k] - ref[k]), r = d / (fabs(ref[k]) + 1e-3);
    maxrel = r > maxrel ? r : maxrel;
    if (r > 1e-4) ++bad;
  }
  const double slots = (double)nchunks * 64 * wpb * 256;
  printf("%-22s waves/SIMD %d  %.3f ms  %.1f G slots/s  %.0f GB/s entries  cycles/slot/CU @2.4GHz %.2f  mismatches %d (max rel %.2e)\n", name,
         wpb / 4, best, slots / best * 1e-6, slots * 8 / best * 1e-6, best * 1e-3 * 2.4e9 * 256 / slots, bad, maxrel);
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const int nblocks = 17;                                  // 510 chunks = 32640 slots per wave
  const long long estride = (long long)(nblocks * CHUNKS_PER_BLOCK + 4) * 512;
  const long long dstride = (long long)nblocks * 256;
  const size_t nw = 4096;
  hP.resize((size_t)TILE_ROWS * 64);
  for (size_t k = 0; k < hP.size(); ++k) hP[k] = (float)((k * 2654435761u >> 20) % 17) - 8.f;
  std::vector<uint32_t> E((size_t)nw * estride / 4), D((size_t)nw * dstride / 4);
  uint64_t s = 0x243f6a8885a308d3ull;
  auto rnd = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 33); };
  for (size_t w = 0; w < nw; ++w) {
    uint32_t* e = &E[w * estride / 4];
    for (size_t k = 0; k < (size_t)estride / 8; ++k) {
      e[2 * k] = (rnd() % TILE_ROWS) * 256;
      const float v = (float)(1 + rnd() % 3);
      memcpy(&e[2 * k + 1], &v, 4);
    }
    // descriptor bytes: 4 * (row slot), advancing every 5..9 two-step groups, cycling through 8 slots
    uint8_t* d = reinterpret_cast<uint8_t*>(&D[w * dstride / 4]);
    int j = 0, left = 5 + rnd() % 5;
    for (int blk = 0; blk < nblocks; ++blk)
      for (int gk = 0; gk < 256; ++gk) {
        d[blk * 256 + gk] = (uint8_t)(4 * j);
        if (--left == 0) { j = (j + 1) % 8; left = 5 + rnd() % 5; }
      }
  }
  hE.assign(E.begin(), E.begin() + estride / 4);
  hD.assign(D.begin(), D.begin() + dstride / 4);
  CK(hipMalloc(&dP, hP.size() * 4));
  CK(hipMalloc(&dE, E.size() * 4));
  CK(hipMalloc(&dD, D.size() * 4));
  CK(hipMalloc(&dOut, (size_t)4096 * 64 * 64 * 4 + 4096));
  CK(hipMemcpy(dP, hP.data(), hP.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dE, E.data(), E.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dD, D.data(), D.size() * 4, hipMemcpyHostToDevice));
  for (int threads : {1024}) {
    run<0>("pk_fma  rows/group 8", threads, nblocks, estride, dstride);
    run<1>("v_fma   rows/group 8", threads, nblocks, estride, dstride);
    run<2>("pk_fma  rows/group 16", threads, nblocks, estride, dstride);
    run<3>("pk_fma  depth 3", threads, nblocks, estride, dstride);
    run<4>("pk_fma  across chunks", threads, nblocks, estride, dstride);
    run<5>("pk_fma  across, depth 3", threads, nblocks, estride, dstride);
    run<6>("ablation: no LDS reads", threads, nblocks, estride, dstride);
    run<7>("ablation: no FMAs", threads, nblocks, estride, dstride);
    run<8>("index mode left on", threads, nblocks, estride, dstride);
  }
  return 0;
}
