// Gate for an integer form of the parity-grade scoring product: does replacing f16 MFMAs of csrc/score_x3.hip's
// three-product k-step by v_mfma_i32_32x32x32_i8 pay on RANDOM operands at the clock the chip then holds?
// The loop is k_score_x3<256>'s without its staging (no LDS-DMA, no barrier): one 512-thread workgroup per CU (two waves per SIMD), a wave's 32
// hidden rows resident as B operands (128 VGPRs), item fragments (A operands) read from an LDS image of random values through a two-deep
// register ring, the epilogue of the previous tile's 16 values (fma, exp, add, cmp, addc: score_x3.hip's val) in the MFMA gaps, one value per k-step.
// A "k-step" = 16 channels of one 32 x 32 tile of logits; a tile = 16 k-steps (C = 256).
//   MODE 0  x3     : f16 hi.hi + hi.lo + lo.hi, three v_mfma_f32_32x32x16_f16 into ONE fp32 accumulator per k-step (production, 22 bits per factor)
//   MODE 1  hybrid : one f16 MFMA (hi.hi) + ONE v_mfma_i32_32x32x32_i8 whose K = 32 holds both cross terms of the k-step, [hh8 | hl8] . [wl8 | wh8]
//                    (8-bit fixed-point copies; the cross terms are 2^-11 of the product, so 2^-8 of them is 2^-19 of it); second accumulator (int32),
//                    +2 VALU per value (cvt, fma)
//   MODE 2  i8 x3  : 8 + 8-bit digits, H1.W1 (one i8 MFMA per TWO k-steps) + [H1 | H0] . [W0 | W1] (one per k-step): 1.5 i8 MFMAs per k-step,
//                    two int32 accumulators, +3 VALU per value (16 bits per factor, relative to the row maximum)
//   MODE 3  f16 x1 : one f16 MFMA per k-step (the 16-bit kernel's product, for scale)
// Each mode runs back to back for `seconds`; prints ms per launch, k-steps per second chip-wide (= logit products / 16,384), ns per k-step per
// SIMD, and the in-kernel clock (s_memtime / s_memrealtime, median over workgroups).
// build: hipcc --offload-arch=gfx950 -O3 scripts/i8_shape_bench.hip -o scripts/build/i8_shape_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

static __device__ __forceinline__ f32x16 mf(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
}
static __device__ __forceinline__ i32x16 mi(uint4 a, uint4 b, i32x16 c) {
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a), __builtin_bit_cast(i32x4, b), c, 0, 0, 0);
}
static __device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// an f16 pair with random sign / mantissa and exponents in [2^-3, 2^0); a dword of four random int8 (zero: all-zero operands)
static __device__ __forceinline__ unsigned rnd_h2(unsigned seed, int zero) {
  if (zero) return 0u;
  const unsigned h = hash32(seed);
  const unsigned lo = (h & 0x83ffu) | ((12u + ((h >> 10) & 3u)) << 10);
  const unsigned hi = ((h >> 16) & 0x83ffu) | ((12u + ((h >> 26) & 3u)) << 10);
  return lo | (hi << 16);
}
static __device__ __forceinline__ unsigned rnd_i8x4(unsigned seed, int zero) { return zero ? 0u : hash32(seed ^ 0x9e3779b9u); }

struct Row { float s, ep, xp, t, m2; int c; };
static __device__ __forceinline__ void val(float v, Row& r) {
  r.s += r.ep;
  r.ep = __builtin_amdgcn_exp2f(r.xp);
  r.xp = fmaf(v, 1.44269504f, -r.m2);
  r.c += (v > r.t) ? 1 : 0;
  asm volatile("" : "+v"(r.s), "+v"(r.c), "+v"(r.ep), "+v"(r.xp));
}

constexpr int LDS_BYTES = 128 * 1024;   // 128 fragments of 1 KB

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(float* out, unsigned long long* clk, int tiles, int zero) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int tid = threadIdx.x, lane = tid & 63;
  // LDS image: even KB fragments f16 pairs, odd ones int8 (modes that read both kinds take their fragments from the matching parity)
  for (int i = tid; i < LDS_BYTES / 4; i += 512)
    reinterpret_cast<unsigned*>(sm)[i] = ((i >> 8) & 1) ? rnd_i8x4(i * 2654435761u + blockIdx.x, zero) : rnd_h2(i * 2654435761u + blockIdx.x, zero);
  uint4 hf[16], hx[16];   // resident B operands: hf = f16 hi of the 32 rows; hx = f16 lo (mode 0) or int8 (modes 1, 2)
#pragma unroll
  for (int i = 0; i < 16; i++) {
    hf[i] = make_uint4(rnd_h2(tid * 131 + i * 4 + 0x1000, zero), rnd_h2(tid * 131 + i * 4 + 0x1001, zero), rnd_h2(tid * 131 + i * 4 + 0x1002, zero), rnd_h2(tid * 131 + i * 4 + 0x1003, zero));
    if (MODE == 0)
      hx[i] = make_uint4(rnd_h2(tid * 137 + i * 4 + 0x2000, zero), rnd_h2(tid * 137 + i * 4 + 0x2001, zero), rnd_h2(tid * 137 + i * 4 + 0x2002, zero), rnd_h2(tid * 137 + i * 4 + 0x2003, zero));
    else
      hx[i] = make_uint4(rnd_i8x4(tid * 137 + i * 4 + 0x2000, zero), rnd_i8x4(tid * 137 + i * 4 + 0x2001, zero), rnd_i8x4(tid * 137 + i * 4 + 0x2002, zero), rnd_i8x4(tid * 137 + i * 4 + 0x2003, zero));
  }
  __syncthreads();
  Row st;
  st.s = 0.f; st.ep = 0.f; st.xp = -100.f; st.t = 0.25f; st.m2 = 64.f; st.c = 0;
  const unsigned char* base = sm + lane * 16;
  const unsigned char* bias = sm + (lane >> 5) * 16;   // the next tile's bias lands in the drained registers, four at a time
  unsigned long long t0 = 0, r0 = 0;
  if (tid == 0) { t0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  uint4 rF[2], rX[2];
  f32x16 xf, yf;
  i32x16 xi, yi, xj, yj;
#pragma unroll
  for (int j = 0; j < 16; j++) { xf[j] = 0.01f * j; yf[j] = 0.f; xi[j] = j; yi[j] = 0; xj[j] = -j; yj[j] = 0; }
  rF[0] = *reinterpret_cast<const uint4*>(base);
  rX[0] = *reinterpret_cast<const uint4*>(base + 1024);
#define FRAG_F(t_, s_) (base + (((t_) & 3) * 32 + 2 * (s_)) * 1024)
#define FRAG_X(t_, s_) (base + (((t_) & 3) * 32 + 2 * (s_) + 1) * 1024)
  // one tile: MFMAs into X*, the 16 values of Y* drained in the gaps; Y* takes the next tile's start values behind steps 3, 7, 11, 15
#define PHASE(XF, XI, XJ, YF, YI, YJ, TI)                                                                              \
  do {                                                                                                                 \
    _Pragma("unroll") for (int s = 0; s < 16; s++) {                                                                   \
      rF[(s + 1) & 1] = *reinterpret_cast<const uint4*>(s + 1 < 16 ? FRAG_F(TI, s + 1) : FRAG_F((TI) + 1, 0));         \
      if (MODE != 3) rX[(s + 1) & 1] = *reinterpret_cast<const uint4*>(s + 1 < 16 ? FRAG_X(TI, s + 1) : FRAG_X((TI) + 1, 0)); \
      if (MODE == 0) {                                                                                                 \
        XF = mf(rF[s & 1], hf[s], XF);                                                                                 \
        val(YF[s], st);                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        XF = mf(rF[s & 1], hx[s], XF);                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        XF = mf(rX[s & 1], hf[s], XF);                                                                                 \
      } else if (MODE == 1) {                                                                                          \
        XF = mf(rF[s & 1], hf[s], XF);                                                                                 \
        val(fmaf((float)YI[s], 4.0f, YF[s]), st);                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        XI = mi(rX[s & 1], hx[s], XI);                                                                                 \
      } else if (MODE == 2) {                                                                                          \
        XI = mi(rX[s & 1], hx[s], XI);                                                                                 \
        val(fmaf((float)YJ[s], 256.0f, (float)YI[s]) * 0.125f, st);                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                             \
        if (s & 1) XJ = mi(rF[s & 1], hf[s], XJ);                                                                      \
      } else {                                                                                                         \
        XF = mf(rF[s & 1], hf[s], XF);                                                                                 \
        val(YF[s], st);                                                                                                \
      }                                                                                                                \
      if ((s & 3) == 3) {                                                                                              \
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(bias + 32 * (s >> 2) + ((TI) & 7) * 256);                     \
        _Pragma("unroll") for (int j = 0; j < 4; j++) {                                                                \
          if (MODE == 0 || MODE == 1 || MODE == 3) YF[(s & ~3) + j] = b4[j];                                           \
          if (MODE == 1 || MODE == 2) YI[(s & ~3) + j] = 0;                                                            \
          if (MODE == 2) YJ[(s & ~3) + j] = __float_as_int(b4[j]) >> 12;                                               \
        }                                                                                                              \
      }                                                                                                                \
      __builtin_amdgcn_sched_barrier(0);                                                                               \
    }                                                                                                                  \
  } while (0)
  for (int t = 0; t < tiles; t += 2) {
    PHASE(xf, xi, xj, yf, yi, yj, t);
    PHASE(yf, yi, yj, xf, xi, xj, t + 1);
  }
#undef PHASE
  float r = 0.f;
#pragma unroll
  for (int j = 0; j < 16; j++) r += xf[j] + yf[j] + (float)(xi[j] + yi[j] + xj[j] + yj[j]);
  out[blockIdx.x * 512 + tid] = r + st.s + st.c + st.xp;
  if (tid == 0) {
    clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0;
    clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
  }
}

template <int MODE>
static void run(float* d, unsigned long long* clk, int tiles, double seconds, int zero) {
  static const char* names[] = {"x3 (3 f16 MFMA / k-step)", "hybrid (1 f16 + 1 i8 MFMA / k-step)", "i8 x3 (1.5 i8 MFMA / k-step)", "x1 (1 f16 MFMA / k-step)"};
  static const double mfma_per_kstep[] = {3.0, 2.0, 1.5, 1.0};
  auto kern = k<MODE>;
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
  const double ksteps = 256.0 * 8 * tiles * 16.0;               // chip-wide k-steps per launch (8 waves per CU, 16 per tile)
  const double ns = ms_last * 1e6 / (2.0 * tiles * 16.0);       // per k-step per SIMD (two waves per SIMD)
  const double f = mhz.empty() ? 0.0 : mhz[mhz.size() / 2];
  printf("%-40s %s: %7.2f ms/launch  %6.1f G k-steps/s = %5.0f T logit-MAC/s  %6.2f ns/k-step/SIMD  clock %4.0f MHz -> %5.1f cycles/k-step = %4.1f cycles/MFMA\n",
         names[MODE], zero ? "zeros " : "random", ms_last, ksteps / (ms_last * 1e-3) / 1e9, ksteps * 16384.0 / (ms_last * 1e-3) / 1e12, ns, f, ns * f * 1e-3,
         ns * f * 1e-3 / mfma_per_kstep[MODE]);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.5;
  float* d; hipMalloc(&d, 256 * 512 * 4);
  unsigned long long* clk; hipMalloc(&clk, 512 * 8);
  const int tiles = 8000;
  for (int rep = 0; rep < 2; rep++)
    for (int zero = 0; zero < 2; zero++) {
      run<0>(d, clk, tiles, seconds, zero);
      run<1>(d, clk, tiles, seconds, zero);
      run<2>(d, clk, tiles, seconds, zero);
      run<3>(d, clk, tiles, seconds, zero);
    }
  return 0;
}
