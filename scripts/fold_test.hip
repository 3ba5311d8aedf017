// standalone check of the wavefront fold used by k_gru_scan (hipcc scripts/fold_test.hip -o /tmp/fold && /tmp/fold)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ float swap_add32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), false, false);
  return __builtin_bit_cast(float, r[0]) + __builtin_bit_cast(float, r[1]);
}
__device__ __forceinline__ float fold64(float (&p)[64], int lane) {
#pragma unroll
  for (int i = 0; i < 32; i++) p[i] = swap_add32(p[i], p[i + 32]);
#pragma unroll
  for (int half = 16; half >= 1; half >>= 1) {
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
__global__ void k(const float* in, float* out, int* owner) {
  int lane = threadIdx.x;
  float p[64];
  for (int i = 0; i < 64; i++) p[i] = in[lane * 64 + i];
  out[lane] = fold64(p, lane);
  float q[64];
  for (int i = 0; i < 64; i++) q[i] = (lane == 0) ? (float)(i + 1) : 0.f;
  owner[lane] = (int)fold64(q, lane) - 1;
}
int main() {
  std::vector<float> h(64 * 64);
  for (int i = 0; i < 64 * 64; i++) h[i] = (float)((i * 37) % 101) - 50.f;
  float *din, *dout; int* down;
  hipMalloc(&din, h.size() * 4); hipMalloc(&dout, 64 * 4); hipMalloc(&down, 64 * 4);
  hipMemcpy(din, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout, down);
  std::vector<float> o(64); std::vector<int> ow(64);
  hipMemcpy(o.data(), dout, 64 * 4, hipMemcpyDeviceToHost);
  hipMemcpy(ow.data(), down, 64 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; l++) {
    double ref = 0; for (int j = 0; j < 64; j++) ref += h[j * 64 + ow[l]];
    if (ref != o[l]) bad++;
    if (l < 8 || ow[l] != l) printf("lane %d owns %d total %.1f ref %.1f\n", l, ow[l], o[l], ref);
  }
  printf("bad=%d\n", bad);
  return 0;
}
