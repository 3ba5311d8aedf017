// Microbenchmark: how many SALU ops fit beside an MFMA + VALU stream on gfx950?
// Each wave loops GAPS times over { 1 v_mfma_f32_32x32x16_bf16 ; NV x v_add_f32 ; NS x s_xor_b64 }.
// 512-thread workgroups (2 waves per SIMD), one workgroup per CU.  Prints cycles per gap per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NS, int NE>
__global__ __launch_bounds__(512, 2) void k(float* out, int gaps) {
  f32x16 acc;
  for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 0.001f + i;
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.01f * (threadIdx.x % 7 + i)); b[i] = (__bf16)(0.02f * (threadIdx.x % 5 + i)); }
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 0.5f;
  unsigned long long s0 = 1, s1 = 2, s2 = 3, s3 = 4;
  for (int g = 0; g < gaps; g++) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    if (NV >= 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(v1));
    if (NS >= 1) asm volatile("s_xor_b64 %0, %0, %1" : "+s"(s0) : "s"(s1));
    if (NE >= 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v5));
    if (NV >= 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v2) : "v"(v1));
    if (NS >= 2) asm volatile("s_and_b64 %0, %0, %1" : "+s"(s2) : "s"(s3));
    if (NV >= 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v3) : "v"(v1));
    if (NS >= 3) asm volatile("s_xor_b64 %0, %0, %1" : "+s"(s1) : "s"(s3));
    if (NV >= 4) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v4) : "v"(v1));
    if (NS >= 4) asm volatile("s_or_b64 %0, %0, %1" : "+s"(s3) : "s"(s0));
    if (NV >= 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(v2));
    if (NS >= 5) asm volatile("s_xor_b64 %0, %0, %1" : "+s"(s0) : "s"(s2));
    if (NS >= 6) asm volatile("s_and_b64 %0, %0, %1" : "+s"(s1) : "s"(s2));
    if (NS >= 7) asm volatile("s_xor_b64 %0, %0, %1" : "+s"(s2) : "s"(s0));
    if (NS >= 8) asm volatile("s_or_b64 %0, %0, %1" : "+s"(s3) : "s"(s1));
    __builtin_amdgcn_sched_barrier(0);
  }
  float r = v0 + v2 + v3 + v4 + v5 + (float)(s0 ^ s1 ^ s2 ^ s3);
  for (int i = 0; i < 16; i++) r += acc[i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int NV, int NS, int NE>
static void run(float* d, int gaps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NV, NS, NE>), dim3(256), dim3(512), 0, 0, d, gaps);
  hipEventRecord(e0);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<NV, NS, NE>), dim3(256), dim3(512), 0, 0, d, gaps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  // per SIMD: 2 waves x gaps MFMAs
  double ns_per_mfma = ms * 1e6 / (2.0 * gaps);
  printf("NV=%d NS=%d NE=%d: %.3f ms, %.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz, %.1f at 2.0)\n", NV, NS, NE, ms, ns_per_mfma,
         ns_per_mfma * 2.4, ns_per_mfma * 2.0);
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int gaps = 200000;
  run<0, 0, 0>(d, gaps); run<0, 0, 0>(d, gaps);
  run<3, 0, 1>(d, gaps);   // 3 adds + 1 exp = 20 cycles of VALU
  run<4, 0, 1>(d, gaps);   // 24
  run<5, 0, 1>(d, gaps);   // 28
  run<3, 2, 1>(d, gaps); run<3, 4, 1>(d, gaps); run<3, 5, 1>(d, gaps); run<3, 6, 1>(d, gaps); run<3, 8, 1>(d, gaps);
  run<4, 4, 1>(d, gaps); run<4, 8, 1>(d, gaps);
  run<0, 8, 0>(d, gaps); run<0, 4, 0>(d, gaps);
  return 0;
}
