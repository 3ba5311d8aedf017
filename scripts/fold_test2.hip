#include <hip/hip_runtime.h>
#include <stdio.h>
__device__ __forceinline__ float swap_add32(float a, float b) {
  // inline asm, not __builtin_amdgcn_permlane32_swap: hipcc (ROCm 7.2) mis-lowers `r[0] + r[1]` of the builtin's
  // two results to `r[0] + r[0]` once the SLP vectoriser packs neighbouring sums (scripts/fold_test2.hip).
  // The two v_nop are the VALU-write -> v_permlane-read wait states (cdna_hip_programming.md T21).
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}
template <int FIRST>
__device__ __forceinline__ float fold(float (&p)[64], int lane) {
  if (FIRST == 32) {
#pragma unroll
    for (int i = 0; i < 32; i++) p[i] = swap_add32(p[i], p[i + 32]);
  }
#pragma unroll
  for (int half = (FIRST == 32 ? 16 : 32); half >= 1; half >>= 1) {
    const bool hi = (lane & half) != 0;
#pragma unroll
    for (int i = 0; i < half; i++) {
      const float send = hi ? p[i] : p[i + half];
      const float keep = hi ? p[i + half] : p[i];
      p[i] = keep + __shfl_xor(send, half);
    }
  }
  return p[0];
}
__global__ void k(int* out) {
  int lane = threadIdx.x;
  float q[64], r[64], s[64];
  for (int i = 0; i < 64; i++) { q[i] = (lane == 0) ? (float)(i + 1) : 0.f; r[i] = q[i]; s[i] = (lane == 5) ? (float)(i + 1) : 0.f; }
  out[lane * 3 + 0] = (int)fold<32>(q, lane);
  out[lane * 3 + 1] = (int)fold<64>(r, lane);
  out[lane * 3 + 2] = (int)fold<32>(s, lane);
}
int main() {
  int* d; hipMalloc(&d, 64 * 12);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[192]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 7) printf("lane %2d: permlane-first %3d  shuffle-only %3d  src-lane5 %3d\n", l, h[l * 3], h[l * 3 + 1], h[l * 3 + 2]);
  return 0;
}
