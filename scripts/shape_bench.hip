// Microbenchmark for the scoring kernel's main loop WITHOUT its staging (no LDS-DMA, no barrier): which MFMA shape / epilogue mix
// delivers the most FLOP/s on RANDOM operands at the clock the chip then holds?  (MI355X_MICROARCH.md, DVFS give-back item 7.)
// One 512-thread workgroup per CU (two waves per SIMD), hidden rows (B operand) resident in 128 VGPRs, item fragments (A operand)
// read from an LDS image of random bf16 values through a two-deep register ring, one fragment feeding every row block of the wave;
// the epilogue of the previous tile's values sits in the MFMA gaps exactly as in k_score_pipe.
//   SHAPE 0: v_mfma_f32_32x32x16_bf16, two 32-row blocks per wave (production): 16 k-steps, one logit per gap and block
//   SHAPE 1: v_mfma_f32_16x16x32_bf16, four 16-row blocks per wave: 8 k-steps x 4 MFMAs, one logit per TWO gaps
//   EPI 0: none; 1: one v_add per logit; 5: fma, exp, add, cmp, addc (production); 4: fma, exp, add, alignbit (comparison folded into the MFMA)
// Each mode runs back to back for `seconds`; prints ms per launch, TFLOP/s chip-wide, ns per 32,768 flop per SIMD, and the in-kernel
// clock (s_memtime / s_memrealtime, median over workgroups).  build: hipcc --offload-arch=gfx950 -O3 scripts/shape_bench.hip -o /tmp/shape_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
static __device__ __forceinline__ f32x4 mfma16(uint4 a, uint4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
static __device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// a bf16 pair with random sign / mantissa and exponents in [2^-3, 2^0) (zero: all-zero operands)
static __device__ __forceinline__ unsigned rnd_pair(unsigned seed, int zero) {
  if (zero) return 0u;
  const unsigned h = hash32(seed);
  const unsigned lo = (h & 0x807fu) | ((124u + ((h >> 7) & 3u)) << 7);
  const unsigned hi = ((h >> 16) & 0x807fu) | ((124u + ((h >> 23) & 3u)) << 7);
  return lo | (hi << 16);
}

struct Row { float s, ep, xp, t, m2; int c; unsigned bits; };

template <int EPI>
static __device__ __forceinline__ void val(int i, float v, Row& r) {
  if (EPI == 0) return;
  if (EPI == 1) { r.s += v; asm volatile("" : "+v"(r.s)); return; }
  r.s += r.ep;
  r.ep = __builtin_amdgcn_exp2f(r.xp);
  if (EPI == 5) {
    r.xp = fmaf(v, 1.44269504f, -r.m2);
    r.c += (v > r.t) ? 1 : 0;
    asm volatile("" : "+v"(r.s), "+v"(r.c), "+v"(r.ep), "+v"(r.xp));
  } else {
    r.xp = fmaf(v, -1.44269504f, -64.0f);
    r.bits = __builtin_amdgcn_alignbit(r.bits, __float_as_uint(v), 31);
    asm volatile("" : "+v"(r.s), "+v"(r.bits), "+v"(r.ep), "+v"(r.xp));
  }
}
// half of a logit's epilogue per gap (16x16x32: one logit per two gaps)
template <int EPI>
static __device__ __forceinline__ void val_half(int i, int half, float v, Row& r) {
  if (EPI == 0) return;
  if (half == 0) {
    r.s += r.ep;
    r.ep = __builtin_amdgcn_exp2f(r.xp);
    asm volatile("" : "+v"(r.s), "+v"(r.ep));
  } else if (EPI == 5) {
    r.xp = fmaf(v, 1.44269504f, -r.m2);
    r.c += (v > r.t) ? 1 : 0;
    asm volatile("" : "+v"(r.c), "+v"(r.xp));
  } else {
    r.xp = fmaf(v, -1.44269504f, -64.0f);
    r.bits = __builtin_amdgcn_alignbit(r.bits, __float_as_uint(v), 31);
    asm volatile("" : "+v"(r.bits), "+v"(r.xp));
  }
}

constexpr int LDS_BYTES = 128 * 1024;   // 128 fragments of 1 KB: the A stream of 8 tiles (32x32: 16 per tile; 16x16: 8 per sub-tile)

template <int SHAPE, int EPI, int NOLDS = 0>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* clk, int tiles, int zero) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < LDS_BYTES / 4; i += 512) reinterpret_cast<unsigned*>(sm)[i] = rnd_pair(i * 2654435761u + blockIdx.x, zero);
  uint4 hf[128 / 4];
#pragma unroll
  for (int i = 0; i < 32; i++)
    hf[i] = make_uint4(rnd_pair(tid * 131 + i * 4 + 0x1000, zero), rnd_pair(tid * 131 + i * 4 + 0x1001, zero), rnd_pair(tid * 131 + i * 4 + 0x1002, zero),
                       rnd_pair(tid * 131 + i * 4 + 0x1003, zero));
  __syncthreads();
  Row st[4];
#pragma unroll
  for (int b = 0; b < 4; b++) { st[b].s = 0.f; st[b].ep = 0.f; st[b].xp = -100.f; st[b].t = 0.25f * b; st[b].m2 = 64.f + b; st[b].c = 0; st[b].bits = 0; }
  const unsigned char* base = sm + lane * 16;
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  uint4 ring[2];
  if (SHAPE == 0) {
    f32x16 x0, x1, y0, y1;
#pragma unroll
    for (int j = 0; j < 16; j++) { x0[j] = 0.01f * j; x1[j] = -0.01f * j; y0[j] = 0.f; y1[j] = 0.f; }
    ring[0] = *reinterpret_cast<const uint4*>(base);
    const unsigned char* bias = sm + (lane >> 5) * 16;   // the next tile's bias lands in the drained registers, four at a time
#define PHASE0(X0, X1, Y0, Y1, TI)                                                                       \
    do {                                                                                                 \
      const unsigned char* tile = base + ((TI) & 7) * 16384;                                             \
      const unsigned char* ntile = base + (((TI) + 1) & 7) * 16384;                                      \
      _Pragma("unroll") for (int s = 0; s < 16; s++) {                                                   \
        if (!NOLDS) ring[(s + 1) & 1] = *reinterpret_cast<const uint4*>(s + 1 < 16 ? tile + 1024 * (s + 1) : ntile); \
        X0 = mfma32(NOLDS ? hf[(s + 5) & 31] : ring[s & 1], hf[s], X0);                                                             \
        val<EPI>(s, Y0[s], st[0]);                                                                       \
        if ((s & 3) == 3) {                                                                              \
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 32 * (s >> 2) + ((TI) & 7) * 256);     \
          _Pragma("unroll") for (int j = 0; j < 4; j++) Y0[(s & ~3) + j] = b4[j];                        \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
        X1 = mfma32(NOLDS ? hf[(s + 21) & 31] : ring[s & 1], hf[16 + s], X1);                                                        \
        val<EPI>(s, Y1[s], st[1]);                                                                       \
        if ((s & 3) == 3) {                                                                              \
          const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 32 * (s >> 2) + ((TI) & 7) * 256);     \
          _Pragma("unroll") for (int j = 0; j < 4; j++) Y1[(s & ~3) + j] = b4[j];                        \
        }                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                               \
      }                                                                                                  \
      if (EPI == 4) { st[0].c += __builtin_popcount(st[0].bits & 0xffffu); st[1].c += __builtin_popcount(st[1].bits & 0xffffu); } \
    } while (0)
    for (int t = 0; t < tiles; t += 2) {
      PHASE0(x0, x1, y0, y1, t);
      PHASE0(y0, y1, x0, x1, t + 1);
    }
#undef PHASE0
    float r = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) r += x0[j] + x1[j] + y0[j] + y1[j];
    out[blockIdx.x * 512 + tid] = r + st[0].s + st[1].s + st[0].c + st[1].c + st[0].xp + st[1].xp;
  } else {
    f32x4 x[4], y[4];
#pragma unroll
    for (int b = 0; b < 4; b++)
#pragma unroll
      for (int j = 0; j < 4; j++) { x[b][j] = 0.01f * j; y[b][j] = 0.f; }
    ring[0] = *reinterpret_cast<const uint4*>(base);
    const unsigned char* bias = sm + (lane >> 4) * 16;
#define PHASE1(X, Y, TI)                                                                                 \
    do {                                                                                                 \
      const unsigned char* tile = base + ((TI) & 15) * 8192;                                             \
      const unsigned char* ntile = base + (((TI) + 1) & 15) * 8192;                                      \
      _Pragma("unroll") for (int s = 0; s < 8; s++) {                                                    \
        ring[(s + 1) & 1] = *reinterpret_cast<const uint4*>(s + 1 < 8 ? tile + 1024 * (s + 1) : ntile);  \
        _Pragma("unroll") for (int b = 0; b < 4; b++) {                                                  \
          X[b] = mfma16(ring[s & 1], hf[b * 8 + s], X[b]);                                               \
          /* gap g = 4 s + b of 32: logit (g >> 1) of the 16 drained ones, half (g & 1); logit l -> block l & 3, value l >> 2 */ \
          const int g = 4 * s + b, l = g >> 1;                                                           \
          val_half<EPI>(l >> 2, g & 1, Y[l & 3][l >> 2], st[l & 3]);                                     \
          if ((g & 1) && l >= 12) Y[l & 3] = *reinterpret_cast<const f32x4*>(bias + ((TI) & 15) * 64);   \
          __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                \
      }                                                                                                  \
      if (EPI == 4) {                                                                                    \
        _Pragma("unroll") for (int b = 0; b < 4; b++) st[b].c += __builtin_popcount(st[b].bits & 0xfu);  \
      }                                                                                                  \
    } while (0)
    for (int t = 0; t < tiles; t++) {   // one trip = two 16-item sub-tiles = the flops of one SHAPE-0 tile
      PHASE1(x, y, 2 * t);
      PHASE1(y, x, 2 * t + 1);
    }
#undef PHASE1
    float r = 0.f;
#pragma unroll
    for (int b = 0; b < 4; b++) {
#pragma unroll
      for (int j = 0; j < 4; j++) r += x[b][j] + y[b][j];
      r += st[b].s + st[b].c + st[b].xp;
    }
    out[blockIdx.x * 512 + tid] = r;
  }
  if (tid == 0) {
    clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int SHAPE, int EPI, int NOLDS = 0>
static void run(float* d, unsigned long long* clk, int tiles, double seconds, int zero) {
  auto kern = k<SHAPE, EPI, NOLDS>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  double ms_last = 0, total = 0;
  while (total < seconds * 1e3) {   // back-to-back launches until the clock has settled; the last launch is the one reported
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), LDS_BYTES, 0, d, clk, tiles, zero);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms_last = ms; total += ms;
  }
  std::vector<unsigned long long> h(512);
  hipMemcpy(h.data(), clk, 512 * 8, hipMemcpyDeviceToHost);
  std::vector<double> mhz;
  for (int i = 0; i < 256; i++) if (h[2 * i + 1]) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
  std::sort(mhz.begin(), mhz.end());
  const double flop = 256.0 * 8 * tiles * 32.0 * 32768.0;   // per CU: 8 waves x tiles x 32 MFMA-equivalents of 32,768 flop
  const double ns = ms_last * 1e6 / (2.0 * tiles * 32.0);   // two waves per SIMD
  printf("shape %s epi %d%s %s: %.2f ms/launch  %.0f TFLOP/s  %.2f ns per 32k flop per SIMD  in-kernel clock %.0f MHz -> %.1f cycles\n",
         SHAPE ? "16x16x32" : "32x32x16", EPI, NOLDS ? " (A operand from registers, no LDS reads)" : "", zero ? "zeros " : "random", ms_last, flop / (ms_last * 1e-3) / 1e12, ns,
         mhz.empty() ? 0.0 : mhz[mhz.size() / 2], mhz.empty() ? 0.0 : ns * mhz[mhz.size() / 2] * 1e-3);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
  float* d; hipMalloc(&d, 256 * 512 * 4);
  unsigned long long* clk; hipMalloc(&clk, 512 * 8);
  const int tiles = 20000;   // ~ 11-15 ms per launch
  for (int zero = 0; zero < 2; zero++) {
    run<0, 0>(d, clk, tiles, seconds, zero);
    run<1, 0>(d, clk, tiles, seconds, zero);
    run<0, 5>(d, clk, tiles, seconds, zero);
    run<0, 4>(d, clk, tiles, seconds, zero);
    run<1, 5>(d, clk, tiles, seconds, zero);
    run<1, 4>(d, clk, tiles, seconds, zero);
    run<0, 5, 1>(d, clk, tiles, seconds, zero);
    run<0, 1>(d, clk, tiles, seconds, zero);
    run<0, 1, 1>(d, clk, tiles, seconds, zero);
  }
  return 0;
}
