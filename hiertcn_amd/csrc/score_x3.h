// Shared by the split-operand scoring kernels (score_x3.hip: last width 64 / 128 / 256; score_x3w.hip: 512): the IEEE-half hi + lo split of an fp32
// value, the power-of-two scales that keep both parts inside a half's range, the per-row running state of the fused epilogue.
#pragma once
#include "common.h"

struct X3Row {
  float t;        // target logit
  float m2;       // softmax reference, log2 units
  float s;        // sum of 2^(z log2e - m2)
  int c;          // #{z > t}
  float xp, ep;   // software pipeline of the main flavour: pending exponent, pending term
};

// IEEE half parts of an fp32 value (v_cvt_f16_f32: round-to-nearest-even), in both library builds
typedef _Float16 x3_h8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint16_t x3_half(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float x3_float(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
__device__ __forceinline__ f32x16 x3_mfma(uint4 a, uint4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(x3_h8, a), __builtin_bit_cast(x3_h8, b), c, 0, 0, 0);
}

// 2^k for the scale that puts a maximum magnitude mx into [2^13, 2^14): k = 14 - frexp exponent, clamped so that bias * scale stays finite
__device__ __forceinline__ int x3_scale_exp(float mx) {
  int k = 14 - __builtin_amdgcn_frexp_expf(mx);   // (frexp_exp(0) = 0; inf / NaN: whatever comes out is clamped)
  k = k < -40 ? -40 : k;
  return k > 40 ? 40 : k;
}

// eight consecutive fp32 values, scaled by the power of two sc -> the hi and lo MFMA fragments
__device__ __forceinline__ void x3_split8(const float* __restrict__ p, float sc, uint4& hi, uint4& lo) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  const float v[8] = {a.x * sc, a.y * sc, a.z * sc, a.w * sc, b.x * sc, b.y * sc, b.z * sc, b.w * sc};
  uint32_t h[4], l[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint16_t h0 = x3_half(v[2 * j]), h1 = x3_half(v[2 * j + 1]);
    h[j] = (uint32_t)h0 | ((uint32_t)h1 << 16);
    l[j] = (uint32_t)x3_half(v[2 * j] - x3_float(h0)) | ((uint32_t)x3_half(v[2 * j + 1] - x3_float(h1)) << 16);   // the subtraction is exact in fp32
  }
  hi = make_uint4(h[0], h[1], h[2], h[3]);
  lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// Paced item walk of the split-operand kernels (score_x3w.hip's file header has the why): the workgroups of an XCD class (blockIdx % 8) publish their
// position in the table (in units of 64 KB of table rows) in a slot of ModelDev::walk_pace and the one that leads a NEIGHBOUR by more than
// X3_PACE_WINDOW units sleeps, bounded.  Speed only: no result depends on it.
#define X3_PACE_WINDOW 16   // units (1 MB of table) a workgroup may lead a neighbour of its XCD by
#define X3_PACE_FAR 1024    // units beyond which another workgroup's position is not a neighbour's (another round of the grid elsewhere in the table)
#define X3_PACE_TRIES 24    // bounded wait: sleeps of ~2 us
__device__ __forceinline__ bool x3_pace_ahead(int mine, int other, int n_units) {
  int lead = mine - other;   // cyclic, signed
  lead += (lead < -(n_units >> 1)) ? n_units : 0;
  lead -= (lead > (n_units >> 1)) ? n_units : 0;
  return other >= 0 && lead > X3_PACE_WINDOW && lead <= X3_PACE_FAR;
}
