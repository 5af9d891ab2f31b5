// Issue cost of the cross-lane moves the DPP-fed sweep could use (cycles per instruction per SIMD, 4 waves per SIMD).
// hipcc --offload-arch=gfx950 -O3 -o dpp_cost dpp_cost.hip && ./dpp_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define LOOP(NAME, BODY)                                                                                  \
  __global__ void __launch_bounds__(1024) NAME(int iters, float* out) {                                  \
    float r = threadIdx.x;                                                                                \
    asm volatile("v_mov_b32 v10, %1\n v_mov_b32 v11, %1\n v_mov_b32 v12, %1\n v_mov_b32 v13, %1\n"        \
                 "v_mov_b32 v14, %1\n v_mov_b32 v15, %1\n v_mov_b32 v16, %1\n v_mov_b32 v17, %1\n"        \
                 "v_mov_b32 v18, %1\n v_mov_b32 v19, %1\n v_mov_b32 v20, %1\n v_mov_b32 v21, %1\n"        \
                 "s_mov_b32 s40, %2\n"                                                                    \
                 "1:\n .rept 32\n" BODY " .endr\n s_sub_u32 s40, s40, 1\n s_cmp_lg_u32 s40, 0\n s_cbranch_scc1 1b\n" \
                 "v_add_f32 %0, v10, v12\n v_add_f32 %0, %0, v14\n v_add_f32 %0, %0, v16\n"               \
                 : "=v"(r) : "v"(r), "s"(iters)                                                           \
                 : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "s40", "scc"); \
    if (r == 12345.f) out[0] = r;                                                                         \
  }

// four independent instructions per repetition (128 per loop trip)
LOOP(k_mov32, "v_mov_b32 v10, v18\n v_mov_b32 v12, v19\n v_mov_b32 v14, v20\n v_mov_b32 v16, v21\n")
LOOP(k_add32, "v_add_u32 v10, v18, v19\n v_add_u32 v12, v19, v20\n v_add_u32 v14, v20, v21\n v_add_u32 v16, v21, v18\n")
LOOP(k_dpp32_newbcast, "v_mov_b32_dpp v10, v18 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v12, v19 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                       "v_mov_b32_dpp v14, v20 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v16, v21 row_newbcast:9 row_mask:0xf bank_mask:0xf\n")
LOOP(k_dppadd_newbcast, "v_add_u32_dpp v10, v18, v19 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v12, v19, v20 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                        "v_add_u32_dpp v14, v20, v21 row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp v16, v21, v18 row_newbcast:9 row_mask:0xf bank_mask:0xf\n")
LOOP(k_dpp64_newbcast, "v_mov_b64_dpp v[10:11], v[18:19] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[12:13], v[20:21] row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                       "v_mov_b64_dpp v[14:15], v[18:19] row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_mov_b64_dpp v[16:17], v[20:21] row_newbcast:9 row_mask:0xf bank_mask:0xf\n")
LOOP(k_dpp32_quadperm, "v_mov_b32_dpp v10, v18 quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v12, v19 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
                       "v_mov_b32_dpp v14, v20 quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v16, v21 quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n")
LOOP(k_dpp32_rowshr, "v_mov_b32_dpp v10, v18 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v12, v19 row_shr:2 row_mask:0xf bank_mask:0xf\n"
                     "v_mov_b32_dpp v14, v20 row_shr:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v16, v21 row_shr:4 row_mask:0xf bank_mask:0xf\n")
LOOP(k_mov64, "v_mov_b64 v[10:11], v[18:19]\n v_mov_b64 v[12:13], v[20:21]\n v_mov_b64 v[14:15], v[18:19]\n v_mov_b64 v[16:17], v[20:21]\n")
LOOP(k_pkfma, "v_pk_fma_f32 v[10:11], v[18:19], v[20:21], v[10:11]\n v_pk_fma_f32 v[12:13], v[18:19], v[20:21], v[12:13]\n"
              "v_pk_fma_f32 v[14:15], v[18:19], v[20:21], v[14:15]\n v_pk_fma_f32 v[16:17], v[18:19], v[20:21], v[16:17]\n")
LOOP(k_fmac64_dpp, "v_fmac_f64_dpp v[10:11], v[18:19], v[20:21] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[12:13], v[18:19], v[20:21] row_newbcast:5 row_mask:0xf bank_mask:0xf\n"
                   "v_fmac_f64_dpp v[14:15], v[18:19], v[20:21] row_newbcast:7 row_mask:0xf bank_mask:0xf\n v_fmac_f64_dpp v[16:17], v[18:19], v[20:21] row_newbcast:9 row_mask:0xf bank_mask:0xf\n")
LOOP(k_fma64, "v_fma_f64 v[10:11], v[18:19], v[20:21], v[10:11]\n v_fma_f64 v[12:13], v[18:19], v[20:21], v[12:13]\n"
              "v_fma_f64 v[14:15], v[18:19], v[20:21], v[14:15]\n v_fma_f64 v[16:17], v[18:19], v[20:21], v[16:17]\n")
// the sweep's mix per step: broadcast + address add + two packed FMAs, three ways
LOOP(k_mix_now, "v_add_u32_dpp v10, v18, v19 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp v11, v20 row_newbcast:3 row_mask:0xf bank_mask:0xf\n"
                "v_pk_fma_f32 v[12:13], v[18:19], v[20:21], v[12:13]\n v_pk_fma_f32 v[14:15], v[18:19], v[20:21], v[14:15]\n")
LOOP(k_mix_b64, "v_mov_b64_dpp v[10:11], v[18:19] row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_add_u32 v16, v10, v20\n"
                "v_pk_fma_f32 v[12:13], v[18:19], v[20:21], v[12:13]\n v_pk_fma_f32 v[14:15], v[18:19], v[20:21], v[14:15]\n")

int main() {
  float* out;
  CHECK(hipMalloc(&out, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const double ghz = prop.clockRate * 1e-6;
  const int cus = prop.multiProcessorCount, iters = 2000;
  struct { const char* name; void (*fn)(int, float*); } ks[] = {
      {"v_mov_b32", k_mov32}, {"v_add_u32", k_add32}, {"v_mov_b32_dpp row_newbcast", k_dpp32_newbcast},
      {"v_add_u32_dpp row_newbcast", k_dppadd_newbcast}, {"v_mov_b64_dpp row_newbcast", k_dpp64_newbcast},
      {"v_mov_b32_dpp quad_perm", k_dpp32_quadperm}, {"v_mov_b32_dpp row_shr", k_dpp32_rowshr}, {"v_mov_b64", k_mov64},
      {"v_pk_fma_f32", k_pkfma}, {"v_fmac_f64_dpp row_newbcast", k_fmac64_dpp}, {"v_fma_f64", k_fma64},
      {"mix: add_dpp + mov_dpp + 2 pk_fma", k_mix_now}, {"mix: mov_b64_dpp + add + 2 pk_fma", k_mix_b64}};
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.fn, dim3(cus), dim3(1024), 0, 0, 10, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k.fn, dim3(cus), dim3(1024), 0, 0, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double instr_per_simd = (double)iters * 128 * 4;   // 4 waves per SIMD
    printf("%-40s %.3f ms  %.2f cycles per instruction per SIMD (clock %.2f GHz)\n", k.name, ms, ms * 1e-3 * ghz * 1e9 / instr_per_simd, ghz);
  }
  return 0;
}
