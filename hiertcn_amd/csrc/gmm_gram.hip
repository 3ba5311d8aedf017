// Gram form of the mixture losses (loss.py:247-257 on the head of model_tcn_gmm.py:39-44 / model_gru_gmm.py:50-54, fixed sigma).
//
// Both likelihood losses need, per row r and mixture m, only  |mu_m(r)|^2  and  mu_m(r)[target]  (one-hot y: |y - mu|^2 = |mu|^2 -
// 2 mu[target] + 1, csrc/gmm.hip).  The means are LINEAR in the hidden row, mu_m = W_m^T h + b_m with W_m [C, D], so
//
//     |mu_m|^2 = x^T G_m x,   x = [h; 1],   G_m = [W_m^T | b_m]^T [W_m^T | b_m]      ((C + 1) x (C + 1), symmetric, PSD)
//
// and the pass over the [D, C] mean table (2 C D flops per row and mixture: D = 20,778 outputs on XING) becomes a C x C quadratic form
// (2 C^2: 80 x fewer at C = 256) plus one gathered dot product for the target column.  G_m is formed once per hiertcn_gmm_set_weights
// in fp64 from the STORED (16-bit) table and the fp32 bias -- the numbers the table pass multiplies -- and kept as three bf16 parts
// hi + mid + lo (24 mantissa bits = fp32), so the quadratic form runs on the bf16 MFMA with fp32 accuracy: hidden rows are bf16-exact,
// every product is exact in the fp32 accumulator.  The table pass (score.hip, sum-of-squares epilogue) stays for hiertcn_gmm_head_means,
// for the IEEE-half build (Gram entries do not fit half's exponent range), for heads with 3 C >= D, and as the A/B and the check:
// HIERTCN_GMM_TABLE_PASS=1.
//
//   k_gmm_gram        G_m[i][j] = sum_d a_i(d) a_j(d), a_c = table column c (c < C) or the bias (c = C); fp64
//   k_gmm_gram_split  G_m[:C][:C] -> G3: per (m, 32-row tile) the LDS image of k_gmm_quad (row j: hi | mid | lo parts of G_m[j][0..C), pitch 6 C + 16 B), cross terms gx[m][c] = 2 G_m[c][C],
//                     constants g0[m] = G_m[C][C] = |b_m|^2
//   k_gmm_quad<C>     sq[m][r] = h_r^T G_m[:C][:C] h_r.  A workgroup = 8 waves x 32 rows; a wave keeps its rows' hidden vectors as MFMA
//                     B fragments (registers); the 32-row tiles of G3 (48 KB at C = 256) go through two LDS buffers by LDS-DMA, the next
//                     one in flight while this one is multiplied: T[j][r] = sum over the 3 C columns of the tile row (3 C / 16 MFMAs in two
//                     accumulator chains).  The accumulator tile holds rows j = 32 t + 8 q + 4 lh + p of T for the lane's hidden row;
//                     the multipliers h_r[j] are channels the lane or its partner (lane ^ 32) already holds as B fragments: four exchanges
//                     per tile, no LDS copy of the hidden rows.
//   k_gmm_prior_gram  per row (16 lanes): prior logits (as k_gmm_prior), sq += h . gx + g0, target mean = h . table_m[y] + bias_m[y]
#include <type_traits>

#include "common.h"

#ifndef HIERTCN_F16

__global__ __launch_bounds__(320) void k_gmm_gram(const uint16_t* __restrict__ muT, const float* __restrict__ mu_bias, int D, int DP, int C,
                                                  double* __restrict__ gram) {
  const int i = blockIdx.x, m = blockIdx.y;
  const int C1 = C + 1;
  const uint16_t* tab = muT + (size_t)m * DP * C;
  const float* bias = mu_bias + (size_t)m * DP;
  for (int j = threadIdx.x; j < C1; j += blockDim.x) {
    double a0 = 0.0, a1 = 0.0;
    int d = 0;
    for (; d + 1 < D; d += 2) {
      const float i0 = i < C ? bf2f(tab[(size_t)d * C + i]) : bias[d], i1 = i < C ? bf2f(tab[(size_t)(d + 1) * C + i]) : bias[d + 1];
      const float j0 = j < C ? bf2f(tab[(size_t)d * C + j]) : bias[d], j1 = j < C ? bf2f(tab[(size_t)(d + 1) * C + j]) : bias[d + 1];
      a0 = fma((double)i0, (double)j0, a0);
      a1 = fma((double)i1, (double)j1, a1);
    }
    if (d < D) {
      const float i0 = i < C ? bf2f(tab[(size_t)d * C + i]) : bias[d];
      const float j0 = j < C ? bf2f(tab[(size_t)d * C + j]) : bias[d];
      a0 = fma((double)i0, (double)j0, a0);
    }
    gram[((size_t)m * C1 + i) * C1 + j] = a0 + a1;
  }
}

__global__ void k_gmm_gram_split(const double* __restrict__ gram, int M, int C, unsigned char* __restrict__ G3, float* __restrict__ gx,
                                 float* __restrict__ g0) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int C1 = C + 1;
  if (idx >= (int64_t)M * C * C1) return;
  const int k = (int)(idx % C1);
  const int64_t mj = idx / C1;
  const int j = (int)(mj % C), m = (int)(mj / C);
  const double g = gram[((size_t)m * C1 + j) * C1 + k];
  if (k == C) {   // the bias column: cross term of row j; thread (m, 0, C) also writes the constant
    gx[(size_t)m * C + j] = (float)(2.0 * g);
    if (j == 0) g0[m] = (float)gram[((size_t)m * C1 + C) * C1 + C];
    return;
  }
  const uint16_t hi = f2bf((float)g);
  const double r1 = g - (double)bf2f(hi);
  const uint16_t mid = f2bf((float)r1);
  const double r2 = r1 - (double)bf2f(mid);
  const uint16_t lo = f2bf((float)r2);
  // tile (m, j / 32), row j % 32 at the LDS pitch (k_gmm_quad copies the tile image straight into LDS)
  const int pitch = 3 * C * 2 + 16;
  const size_t store = ((size_t)32 * pitch + 1023) / 1024 * 1024;   // QuadGeom<C>::STORE
  uint16_t* row = reinterpret_cast<uint16_t*>(G3 + ((size_t)m * (C / 32) + j / 32) * store + (size_t)(j % 32) * pitch);
  row[k] = hi;
  row[C + k] = mid;
  row[2 * C + k] = lo;
}

// compile-time loop: the body sees its index as a constant (register arrays are then never indexed dynamically, whatever the unroller decides)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

template <int C>
struct QuadGeom {
  static constexpr int KS = C / 16;                 // k-steps per part
  static constexpr int NT = C / 32;                 // 32-row tiles of G per mixture
  static constexpr int ROWB = 3 * C * 2;            // bytes of a G3 row: hi | mid | lo
  static constexpr int PITCH = ROWB + 16;           // row r starts 4 banks after row r - 1: a ds_read_b128 of 16 rows covers the 64 banks once
  static constexpr int PIECES = (32 * PITCH + 1023) / 1024;   // 1 KB LDS-DMA pieces per tile image
  static constexpr int STORE = PIECES * 1024;       // bytes of a tile image, in global memory as in LDS (copied straight)
  static constexpr int THREADS = 512;               // 8 waves x 32 hidden rows
};

template <int C>
__global__ __launch_bounds__(512, 1) void k_gmm_quad(const uint16_t* __restrict__ hidden, int64_t R, int M, const unsigned char* __restrict__ G3,
                                                     float* __restrict__ sq) {
  using Q = QuadGeom<C>;
  constexpr int KS = Q::KS, NT = Q::NT;
  static_assert(NT % 2 == 0, "the LDS buffer of a tile is its index's parity: compile-time with an even number of tiles per mixture");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // two tile images
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t row = (int64_t)blockIdx.x * 256 + wave * 32 + lr;
  const int64_t rc = row < R ? row : R - 1;

  uint4 hf[KS];
  {
    const uint16_t* hp = hidden + rc * C + 8 * lh;
    static_for<0, KS>([&](auto sc) __attribute__((always_inline)) { hf[sc.value] = *reinterpret_cast<const uint4*>(hp + 16 * sc.value); });
  }
  // LDS-DMA of a tile image (score.hip: why inline asm, and why the waits are placed by hand): wave w moves pieces w, w + 8, ...
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  const int n_it = M * NT;
  auto issue = [&](int it, int buf) __attribute__((always_inline)) {
    const unsigned char* sb = G3 + (size_t)it * Q::STORE;
    static_for<0, (Q::PIECES + 7) / 8>([&](auto ic) __attribute__((always_inline)) {
      const int p = wave + 8 * ic.value;
      if (p < Q::PIECES) {   // wave-uniform
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + buf * Q::STORE + p * 1024);
        const unsigned voff = (unsigned)(p * 1024 + lane * 16);
        const unsigned char* src = sb;   // (an asm operand alone does not capture a variable of the enclosing lambda)
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep)
                     : "v"(voff), "s"(src), "s"(dst)
                     : "memory");
      }
    });
  };

  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int m = 0; m < M; m++) {
    float qs = 0.f;
    static_for<0, NT>([&](auto tc) __attribute__((always_inline)) {
      constexpr int t = tc.value;
      constexpr int buf = t & 1;
      if (m * NT + t + 1 < n_it) issue(m * NT + t + 1, buf ^ 1);   // its previous readers passed the barrier that ended the last tile
      const unsigned char* arow = smem + buf * Q::STORE + lr * Q::PITCH + 16 * lh;   // A fragment of k-step s' of the tile row: + 32 s'
      f32x16 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 16; i++) { acc0[i] = 0.f; acc1[i] = 0.f; }
      uint4 ring[4];
      static_for<0, 3>([&](auto sc) __attribute__((always_inline)) { ring[sc.value] = *reinterpret_cast<const uint4*>(arow + 32 * sc.value); });
      static_for<0, 3 * KS / 2>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = 2 * sc.value;
        if constexpr (s + 3 < 3 * KS) ring[(s + 3) & 3] = *reinterpret_cast<const uint4*>(arow + 32 * (s + 3));
        acc0 = mfma32(ring[s & 3], hf[s % KS], acc0);
        if constexpr (s + 4 < 3 * KS) ring[(s + 4) & 3] = *reinterpret_cast<const uint4*>(arow + 32 * (s + 4));
        acc1 = mfma32(ring[(s + 1) & 3], hf[(s + 1) % KS], acc1);
        __builtin_amdgcn_sched_barrier(0);   // keep the reads three k-steps ahead: hipcc otherwise hoists them all
      });
      // multipliers of accumulator register 4 q + p: channel 32 t + 8 q + 4 lh + p = element 4 lh + p of k-step 2 t + (q >> 1) in lane half
      // (q & 1): the lane's own (x, y) / (z, w) pair when (q & 1) == lh, the partner's (lane ^ 32) otherwise
      static_for<0, 2>([&](auto uc) __attribute__((always_inline)) {
        constexpr int s = 2 * t + uc.value;
        const uint32_t oa = lh ? hf[s].z : hf[s].x, ob = lh ? hf[s].w : hf[s].y;   // what this lane uses of its own fragment
        const uint32_t sa = lh ? hf[s].x : hf[s].z, sb = lh ? hf[s].y : hf[s].w;   // what the partner needs from it
        const uint32_t pa = __shfl_xor(sa, 32), pb = __shfl_xor(sb, 32);
        static_for<0, 2>([&](auto vc) __attribute__((always_inline)) {
          constexpr int q = 2 * uc.value + vc.value;
          const bool own = vc.value == lh;
          const uint32_t a = own ? oa : pa, b = own ? ob : pb;
          const float m0 = __builtin_bit_cast(float, a << 16), m1 = __builtin_bit_cast(float, a & 0xffff0000u);
          const float m2 = __builtin_bit_cast(float, b << 16), m3 = __builtin_bit_cast(float, b & 0xffff0000u);
          qs = fmaf(acc0[4 * q] + acc1[4 * q], m0, qs);
          qs = fmaf(acc0[4 * q + 1] + acc1[4 * q + 1], m1, qs);
          qs = fmaf(acc0[4 * q + 2] + acc1[4 * q + 2], m2, qs);
          qs = fmaf(acc0[4 * q + 3] + acc1[4 * q + 3], m3, qs);
        });
      });
      asm volatile("" : "+v"(qs));
      __builtin_amdgcn_sched_barrier(0);                  // the epilogue stays in its tile (its inputs are 32 registers)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next tile have landed ...
      __syncthreads();                                    // ... and everyone's; every wave is done reading this one
    });
    qs += __shfl_xor(qs, 32);   // the two halves hold disjoint rows j of T for the same hidden row
    if (lh == 0 && row < R) sq[(size_t)m * R + row] = qs;
  }
}

// 16 lanes per row (four rows per wave): prior logits, the linear and constant terms of |mu_m|^2, the target mean (a gathered row of table m)
__global__ __launch_bounds__(256) void k_gmm_prior_gram(const uint16_t* __restrict__ hidden, const float* __restrict__ pi_w,
                                                         const float* __restrict__ pi_b, const float* __restrict__ gx, const float* __restrict__ g0,
                                                         const uint16_t* __restrict__ muT, const float* __restrict__ mu_bias,
                                                         const int32_t* __restrict__ y, int64_t R, int C, int M, int D, int DP,
                                                         float* __restrict__ pl, float* __restrict__ sq, float* __restrict__ tg) {
  const int l16 = threadIdx.x & 15;
  const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t rc = r < R ? r : R - 1;   // (every lane takes part in the shuffles)
  const int yy = y[rc];
  const bool live = yy > 0 && yy < D;
  auto dot8 = [](const uint4 hv, const float4 a, const float4 b, float acc) {
    acc = fmaf(bf2f((uint16_t)(hv.x & 0xffff)), a.x, acc); acc = fmaf(bf2f((uint16_t)(hv.x >> 16)), a.y, acc);
    acc = fmaf(bf2f((uint16_t)(hv.y & 0xffff)), a.z, acc); acc = fmaf(bf2f((uint16_t)(hv.y >> 16)), a.w, acc);
    acc = fmaf(bf2f((uint16_t)(hv.z & 0xffff)), b.x, acc); acc = fmaf(bf2f((uint16_t)(hv.z >> 16)), b.y, acc);
    acc = fmaf(bf2f((uint16_t)(hv.w & 0xffff)), b.z, acc); acc = fmaf(bf2f((uint16_t)(hv.w >> 16)), b.w, acc);
    return acc;
  };
  for (int m = 0; m < M; m++) {
    const uint16_t* trow = muT + ((size_t)m * DP + (live ? yy : 0)) * C;
    float sp = 0.f, sx = 0.f, st = 0.f;
    for (int c = 8 * l16; c < C; c += 128) {
      const uint4 hv = *reinterpret_cast<const uint4*>(hidden + rc * C + c);
      const uint4 tv = *reinterpret_cast<const uint4*>(trow + c);
      const float4* pwp = reinterpret_cast<const float4*>(pi_w + (size_t)m * C + c);
      const float4* xwp = reinterpret_cast<const float4*>(gx + (size_t)m * C + c);
      sp = dot8(hv, pwp[0], pwp[1], sp);
      sx = dot8(hv, xwp[0], xwp[1], sx);
      const float4 t0 = make_float4(bf2f((uint16_t)(tv.x & 0xffff)), bf2f((uint16_t)(tv.x >> 16)), bf2f((uint16_t)(tv.y & 0xffff)), bf2f((uint16_t)(tv.y >> 16)));
      const float4 t1 = make_float4(bf2f((uint16_t)(tv.z & 0xffff)), bf2f((uint16_t)(tv.z >> 16)), bf2f((uint16_t)(tv.w & 0xffff)), bf2f((uint16_t)(tv.w >> 16)));
      st = dot8(hv, t0, t1, st);
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) {
      sp += __shfl_xor(sp, off);
      sx += __shfl_xor(sx, off);
      st += __shfl_xor(st, off);
    }
    if (l16 == 0 && r < R) {
      pl[r * M + m] = sp + pi_b[m];
      sq[(size_t)m * R + r] += sx + g0[m];
      tg[(size_t)m * R + r] = live ? st + mu_bias[(size_t)m * DP + yy] : 0.f;
    }
  }
}

#endif  // !HIERTCN_F16

bool gmm_gram_applies(int C, int D) {
#ifdef HIERTCN_F16
  (void)C; (void)D;
  return false;   // Gram entries (sums of D products) do not fit IEEE half's exponent range
#else
  return (C == 64 || C == 128 || C == 256) && 3 * C < D;
#endif
}

size_t gmm_gram_scratch_doubles(int M, int C) { return (size_t)M * (C + 1) * (C + 1); }
size_t gmm_gram_table_bytes(int M, int C) { return (size_t)M * (C / 32) * (((size_t)32 * (3 * C * 2 + 16) + 1023) / 1024 * 1024); }

hipError_t launch_gmm_gram_build(const uint16_t* muT, const float* mu_bias, int M, int D, int DP, int C, double* gram64, unsigned char* G3, float* gx,
                                 float* g0, hipStream_t st) {
#ifdef HIERTCN_F16
  (void)muT; (void)mu_bias; (void)M; (void)D; (void)DP; (void)C; (void)gram64; (void)G3; (void)gx; (void)g0; (void)st;
  return hipErrorNotSupported;
#else
  hipLaunchKernelGGL(k_gmm_gram, dim3((unsigned)(C + 1), (unsigned)M), dim3(320), 0, st, muT, mu_bias, D, DP, C, gram64);
  const int64_t n = (int64_t)M * C * (C + 1);
  hipLaunchKernelGGL(k_gmm_gram_split, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, gram64, M, C, G3, gx, g0);
  return hipGetLastError();
#endif
}

hipError_t launch_gmm_quad(const uint16_t* hidden, int64_t R, int C, int M, const unsigned char* G3, float* sq, hipStream_t st) {
#ifdef HIERTCN_F16
  (void)hidden; (void)R; (void)C; (void)M; (void)G3; (void)sq; (void)st;
  return hipErrorNotSupported;
#else
  const dim3 grid((unsigned)((R + 255) / 256));
  hipError_t e = hipSuccess;
#define QUAD_CASE(W)                                                                                                          \
  case W:                                                                                                                     \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gmm_quad<W>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * QuadGeom<W>::STORE); \
    if (e != hipSuccess) return e;                                                                                            \
    hipLaunchKernelGGL(k_gmm_quad<W>, grid, dim3(512), 2 * QuadGeom<W>::STORE, st, hidden, R, M, G3, sq);                     \
    break;
  switch (C) {
    QUAD_CASE(64)
    QUAD_CASE(128)
    QUAD_CASE(256)
    default: return hipErrorInvalidValue;
  }
#undef QUAD_CASE
  return hipGetLastError();
#endif
}

hipError_t launch_gmm_prior_gram(const uint16_t* hidden, const float* pi_w, const float* pi_b, const float* gx, const float* g0, const uint16_t* muT,
                                 const float* mu_bias, const int32_t* y, int64_t R, int C, int M, int D, int DP, float* pl, float* sq, float* tg,
                                 hipStream_t st) {
#ifdef HIERTCN_F16
  (void)hidden; (void)pi_w; (void)pi_b; (void)gx; (void)g0; (void)muT; (void)mu_bias; (void)y; (void)R; (void)C; (void)M; (void)D; (void)DP;
  (void)pl; (void)sq; (void)tg; (void)st;
  return hipErrorNotSupported;
#else
  hipLaunchKernelGGL(k_gmm_prior_gram, dim3((unsigned)((R + 15) / 16)), dim3(256), 0, st, hidden, pi_w, pi_b, gx, g0, muT, mu_bias, y, R, C, M, D, DP,
                     pl, sq, tg);
  return hipGetLastError();
#endif
}
