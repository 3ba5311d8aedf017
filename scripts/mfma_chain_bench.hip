// Microbenchmark: does the serial accumulator chain of one wave limit the MFMA pipe on gfx950?
// Each wave loops over { MFMA(s) ; 4 x v_add_f32 ; 1 x v_exp_f32 ; 1 x ds_read_b128 } with
//   MODE 0: one 32x32x16 accumulator (acc = mfma(a, b, acc): every MFMA waits for the one before it)
//   MODE 1: two 32x32x16 accumulators used alternately
//   MODE 2: four 16x16x32 accumulators used round-robin (two per "gap": the same flops per gap as one 32x32x16)
// at 1 or 2 waves per SIMD (256- or 512-thread workgroups, one per CU).  Prints ns per 32,768 MFMA flops per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int FILL>
__global__ __launch_bounds__(512, 2) void k(float* out, int gaps) {
  __shared__ __attribute__((aligned(16))) unsigned char sm[16384];
  f32x16 acc0, acc1;
  f32x4 c0, c1, c2, c3;
  for (int i = 0; i < 16; i++) { acc0[i] = threadIdx.x * 0.001f + i; acc1[i] = i; }
  for (int i = 0; i < 4; i++) { c0[i] = i; c1[i] = i + 1; c2[i] = i + 2; c3[i] = i + 3; }
  bf16x8 a, b;
  for (int i = 0; i < 8; i++) { a[i] = (__bf16)(0.01f * (threadIdx.x % 7 + i)); b[i] = (__bf16)(0.02f * (threadIdx.x % 5 + i)); }
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) reinterpret_cast<float*>(sm)[i] = i;
  __syncthreads();
  float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 0.5f;
  uint4 ld = make_uint4(0, 0, 0, 0);
  const unsigned addr = (threadIdx.x & 63) * 16;
#define FILLER()                                                     \
  if (FILL) {                                                        \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v0) : "v"(v1));       \
    asm volatile("v_exp_f32 %0, %0" : "+v"(v5));                     \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v2) : "v"(v1));       \
    asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(addr));      \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v3) : "v"(v1));       \
    asm volatile("v_add_f32 %0, %0, %1" : "+v"(v4) : "v"(v1));       \
  }                                                                  \
  __builtin_amdgcn_sched_barrier(0);
  for (int g = 0; g < gaps; g += 2) {   // two gaps per trip, no branch between the accumulators
    if (MODE == 2) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c1, 0, 0, 0);
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
    }
    FILLER()
    if (MODE == 2) {
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c3, 0, 0, 0);
    } else if (MODE == 1) {
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
    } else {
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
    }
    FILLER()
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  float r = v0 + v2 + v3 + v4 + v5 + __uint_as_float(ld.x);
  for (int i = 0; i < 16; i++) r += acc0[i] + acc1[i];
  for (int i = 0; i < 4; i++) r += c0[i] + c1[i] + c2[i] + c3[i];
  out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int MODE, int FILL>
static void run(float* d, int gaps, int threads) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, FILL>), dim3(256), dim3(threads), 0, 0, d, gaps);
  hipEventRecord(e0);
  for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<MODE, FILL>), dim3(256), dim3(threads), 0, 0, d, gaps);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
  const int waves_per_simd = threads / 256;
  const double ns = ms * 1e6 / ((double)waves_per_simd * gaps);
  printf("mode %d fill %d waves/SIMD %d: %.3f ms, %.2f ns per gap per SIMD (%.1f cycles at 2.0 GHz)\n", MODE, FILL, waves_per_simd, ms, ns, ns * 2.0);
}

int main() {
  float* d; hipMalloc(&d, 256 * 512 * 4);
  const int gaps = 200000;
  run<0, 0>(d, gaps, 512);
  for (int t = 256; t <= 512; t += 256) {
    run<0, 0>(d, gaps, t); run<1, 0>(d, gaps, t); run<2, 0>(d, gaps, t);
    run<0, 1>(d, gaps, t); run<1, 1>(d, gaps, t); run<2, 1>(d, gaps, t);
  }
  return 0;
}
