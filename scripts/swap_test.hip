#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  int lane = threadIdx.x;
  unsigned a = lane, b = 100 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane * 4 + 0] = r[0]; out[lane * 4 + 1] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  out[lane * 4 + 2] = q[0]; out[lane * 4 + 3] = q[1];
}
int main() {
  int* d; hipMalloc(&d, 64 * 16);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63})
    printf("lane %2d: swap32 -> (%3d, %3d)   swap16 -> (%3d, %3d)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
