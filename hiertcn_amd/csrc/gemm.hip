// The plain products of the training backward pass (model.py:139-141 differentiates customized_tcn_cell.py:109-127,
// customed_gru_cell.py:323-337 and the 'emb' projections): hand-written MFMA GEMMs, so that the library links nothing but the HIP runtime.
//
//   fp32 (GRU cells, state half of the projection): v_mfma_f32_32x32x2_f32 -- a bit-exact fp32 multiply-add chain, like the forward
//        scan's k_gemm_f32 -- in the two forms the backward needs:
//          NN   C[M][N]  = beta C + A[M][K] . B[K][N]                      (d [in | h] = d a . W)
//          TN   C[N1][N2] = beta C + sum_m A[m][N1] . B[m][N2]             (d W += d a^T . [in | h]: the reduction runs over the ROWS)
//   16-bit storage (TemporalBlock taps, Dense residual, 'emb' projection, scoring layer at widths the fused kernels do not take):
//          NT   C[M][N]  = beta C + A[M][K] . Bt[N][K]^T, fp32 out          (d a_in = d c . W^T with W kept [Cin][Cout])
//          TN   C[N1][N2] = sum_r A[r][N1] . B[r][N2], fp32 out             (d W = d c^T . a_in over all rows of the batch)
//
// One LDS-tiled kernel per element type: a 128 x 128 output tile per workgroup (4 waves x four 32 x 32 accumulators), 32-deep k chunks,
// the next chunk's global loads in flight during the MFMAs.  An operand is staged in the layout it has in memory -- [mn][k] rows
// (pitch 36 floats) or [k][mn] rows (pitch 136) -- and the fragment reads follow it (one ds_read_b128, or four ds_read_b32 down the k
// rows), so no transposed copy of anything is ever made.  The row reduction of the TN forms is cut into `splits` slices over blockIdx.z
// (a [Cout x Cin] weight gradient is 16 tiles: one slice would leave 240 CUs idle); slices write partial tiles to scratch and a second
// launch adds them in slice order -- deterministic, no atomics.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "gemm.h"

typedef float f32x16g __attribute__((ext_vector_type(16)));

#define GT 128
#define GK 32
#define GP_K 36    // pitch of a [mn][k] tile: the 16 lanes of a ds_read_b128 phase fall into 16 distinct bank groups
#define GP_M 136   // pitch of a [k][mn] tile: rows 4 apart (the two lane halves of a ds_read_b32) are 32 banks apart

struct G32 {
  const float* A; int64_t lda; const float* B; int64_t ldb;
  float* C; int64_t ldc;
  int M, N, K;          // C is M x N, the reduction has length K
  int k_per_split;      // multiple of GK
  float beta;
  float* parts;         // splits > 1: partial tiles [split][M][N], C untouched
  // k-major operands only (the TN form): reduction row k of A / B lies at (k / div) * ss + (k % div) * ld instead of k * ld when div > 0 -- the
  // rows of one GEMM gathered from `K / div` slabs that are `ss` apart (the per-slot records of a scan: train.hip)
  int a_div, b_div; int64_t a_ss, b_ss;
};

// AK: A element (i, k) at A[k * lda + i] (k-major rows) instead of A[i * lda + k]; BK likewise: B element (k, j) at B[k * ldb + j]
template <bool AK, bool BK>
__device__ __forceinline__ void gemm32_tile(const G32& g, int bx, int by, int bz, float* As, float* Bs) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = by * GT, n0 = bx * GT;
  const int wm = wave >> 1, wn = wave & 1;
  const int k_lo = bz * g.k_per_split;
  const int k_hi = min(g.K, k_lo + g.k_per_split);
  const int n_it = (k_hi - k_lo + GK - 1) / GK;

  float4 ra[4], rb[4];
  auto fetch_one = [&](const float* P, int64_t ld, int mn0, int MN, bool kmaj, int kc, int p, int div, int64_t ss) -> float4 {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kmaj) {          // tile [32 k][128 mn]: thread -> k = tid / 32 + 8 p, mn = 4 (tid % 32)
      const int k = kc + (tid >> 5) + 8 * p, mn = mn0 + (tid & 31) * 4;
      const int64_t ro = div > 0 ? (int64_t)(k / div) * ss + (int64_t)(k % div) * ld : (int64_t)k * ld;
      if (k < k_hi && mn < MN) v = *reinterpret_cast<const float4*>(P + ro + mn);
    } else {             // tile [128 mn][32 k]: thread -> mn = tid / 8 + 32 p, k = 4 (tid % 8)
      const int mn = mn0 + (tid >> 3) + 32 * p, k = kc + (tid & 7) * 4;
      if (mn < MN && k < k_hi) v = *reinterpret_cast<const float4*>(P + (int64_t)mn * ld + k);
    }
    return v;
  };
  auto fetch = [&](int it) {
    const int kc = k_lo + it * GK;
#pragma unroll
    for (int p = 0; p < 4; p++) {
      ra[p] = fetch_one(g.A, g.lda, m0, g.M, AK, kc, p, g.a_div, g.a_ss);
      rb[p] = fetch_one(g.B, g.ldb, n0, g.N, BK, kc, p, g.b_div, g.b_ss);
    }
  };
  auto stash = [&](float* T, const float4* r, bool kmaj) {
#pragma unroll
    for (int p = 0; p < 4; p++) {
      if (kmaj) *reinterpret_cast<float4*>(&T[((tid >> 5) + 8 * p) * GP_M + (tid & 31) * 4]) = r[p];
      else *reinterpret_cast<float4*>(&T[((tid >> 3) + 32 * p) * GP_K + (tid & 7) * 4]) = r[p];
    }
  };
  // fragment of tile rows [mn0, mn0 + 32) for the 8-deep step s: four values per lane, k = 8 s + 4 lh + c for MFMA c
  auto frag = [&](const float* T, bool kmaj, int mn0l, int s) -> float4 {
    if (kmaj) {
      const float* q = &T[(8 * s + 4 * lh) * GP_M + mn0l + lr];
      return make_float4(q[0], q[GP_M], q[2 * GP_M], q[3 * GP_M]);
    }
    return *reinterpret_cast<const float4*>(&T[(mn0l + lr) * GP_K + 8 * s + 4 * lh]);
  };

  f32x16g acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[i][j][q] = 0.f;

  if (n_it > 0) fetch(0);
  for (int it = 0; it < n_it; it++) {
    __syncthreads();
    stash(As, ra, AK);
    stash(Bs, rb, BK);
    __syncthreads();
    if (it + 1 < n_it) fetch(it + 1);
#pragma unroll
    for (int s = 0; s < GK / 8; s++) {
      const float4 a0 = frag(As, AK, 64 * wm, s), a1 = frag(As, AK, 64 * wm + 32, s);
      const float4 b0 = frag(Bs, BK, 64 * wn, s), b1 = frag(Bs, BK, 64 * wn + 32, s);
#define GSTEP(c)                                                                      \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b0.c, acc[0][0], 0, 0, 0);   \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b1.c, acc[0][1], 0, 0, 0);   \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b0.c, acc[1][0], 0, 0, 0);   \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b1.c, acc[1][1], 0, 0, 0);
      GSTEP(x) GSTEP(y) GSTEP(z) GSTEP(w)
#undef GSTEP
    }
  }
  float* out = g.parts ? g.parts + (int64_t)bz * g.M * g.N : g.C;
  const int64_t ldo = g.parts ? g.N : g.ldc;
  const float beta = g.parts ? 0.f : g.beta;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int n = n0 + 64 * wn + 32 * j + lr;
    if (n >= g.N) continue;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int m = m0 + 64 * wm + 32 * i + acc_row(q, lh);
        if (m >= g.M) continue;
        float v = acc[i][j][q];
        if (beta != 0.f) v += beta * out[(int64_t)m * ldo + n];
        out[(int64_t)m * ldo + n] = v;
      }
  }
}

template <bool AK, bool BK>
__global__ __launch_bounds__(256) void k_gemm32(const G32 g) {
  __shared__ float As[AK ? GK * GP_M : GT * GP_K];
  __shared__ float Bs[BK ? GK * GP_M : GT * GP_K];
  gemm32_tile<AK, BK>(g, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z, As, Bs);
}
// several independent products of one form in ONE launch (round 4: the eight weight-gradient products behind a scan at the reference's batch are 4-16
// tiles each with a reduction of ten dependent chunks -- 33 us apiece as launches of their own, about as long all together)
#define G32_GROUP_MAX 8
struct G32Group { G32 g[G32_GROUP_MAX]; int tile_off[G32_GROUP_MAX + 1]; int n; };
template <bool AK, bool BK>
__global__ __launch_bounds__(256) void k_gemm32_group(const G32Group grp) {
  __shared__ float As[AK ? GK * GP_M : GT * GP_K];
  __shared__ float Bs[BK ? GK * GP_M : GT * GP_K];
  int gi = 0;
#pragma unroll
  for (int i = 1; i < G32_GROUP_MAX; i++) gi += (i < grp.n && (int)blockIdx.x >= grp.tile_off[i]) ? 1 : 0;
  // (copy out of the kernel argument with a uniform index: one product's descriptor)
  G32 g = grp.g[0];
#pragma unroll
  for (int i = 1; i < G32_GROUP_MAX; i++) if (gi == i) g = grp.g[i];
  const int local = (int)blockIdx.x - grp.tile_off[gi];
  const int tn = (g.N + GT - 1) / GT;
  gemm32_tile<AK, BK>(g, local % tn, local / tn, (int)blockIdx.y, As, Bs);   // blockIdx.y: the slice of the reduction (g.parts set: partial tiles)
}
// C_i = beta_i C_i + sum of the P partial tiles of every product of a group (elements of all products in one index space)
__global__ void k_gemm_sum_parts_group(const G32Group grp, int P) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t base = 0;
#pragma unroll
  for (int k = 0; k < G32_GROUP_MAX; k++) {
    if (k >= grp.n) break;
    const G32& g = grp.g[k];
    const int64_t mn = (int64_t)g.M * g.N;
    if (i >= base && i < base + mn) {
      const int64_t e = i - base;
      const int m = (int)(e / g.N), n = (int)(e - (int64_t)m * g.N);
      float sum = 0.f;
      for (int p = 0; p < P; p++) sum += g.parts[(int64_t)p * mn + e];
      float* c = g.C + (int64_t)m * g.ldc + n;
      *c = g.beta != 0.f ? g.beta * *c + sum : sum;
    }
    base += mn;
  }
}

__global__ void k_gemm_sum_parts(const float* __restrict__ parts, int P, int M, int N, float* __restrict__ C, int64_t ldc, float beta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * N) return;
  const int m = (int)(i / N), n = (int)(i - (int64_t)m * N);
  float s = 0.f;
  for (int p = 0; p < P; p++) s += parts[(int64_t)p * M * N + i];
  float* c = C + (int64_t)m * ldc + n;
  *c = beta != 0.f ? beta * *c + s : s;
}

// slices of the reduction: enough workgroups for the chip (>= ~1.5 per CU) while a slice keeps >= 256 rows, bounded by the scratch
static int pick_splits(int tiles, int K, int64_t MN, size_t scratch_floats) {
  int s = 1;
  while (tiles * s < 384 && K / (2 * s) >= 256 && (size_t)(2 * s) * MN <= scratch_floats && s < 64) s *= 2;
  return s;
}

template <bool AK, bool BK>
static hipError_t run32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, float beta,
                        float* scratch, size_t scratch_floats, hipStream_t st, int a_div = 0, int64_t a_ss = 0, int b_div = 0, int64_t b_ss = 0) {
  if (M <= 0 || N <= 0) return hipSuccess;
  const int tm = (M + GT - 1) / GT, tn = (N + GT - 1) / GT;
  const int splits = scratch ? pick_splits(tm * tn, K, (int64_t)M * N, scratch_floats) : 1;
  G32 g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.beta = beta;
  g.k_per_split = (((K + splits - 1) / splits) + GK - 1) / GK * GK;
  g.parts = splits > 1 ? scratch : nullptr;
  g.a_div = a_div; g.a_ss = a_ss; g.b_div = b_div; g.b_ss = b_ss;
  hipLaunchKernelGGL((k_gemm32<AK, BK>), dim3(tn, tm, splits), dim3(256), 0, st, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || splits == 1) return e;
  const int64_t mn = (int64_t)M * N;
  hipLaunchKernelGGL(k_gemm_sum_parts, dim3((unsigned)((mn + 255) / 256)), dim3(256), 0, st, scratch, splits, M, N, C, ldc, beta);
  return hipGetLastError();
}

// NN for a handful of rows (M <= 64: the d [in | h] = d a . W products of a recurrent cell at the reference's 32 lanes).  The tiled kernel spends such a
// product on fixed costs -- a 128 x 128 tile for 32 rows, two barriers per 32-deep chunk, 3 workgroups -- 18 us per launch, 81 of them per training step.
// Here a workgroup owns 32 rows x 64 columns and takes the reduction in 128-deep chunks with every load of a chunk in flight at once.
#define SK_K 128
#define SK_PA (SK_K + 4)   // pitch of the [32 m][128 k] tile
#define SK_PB 68           // pitch of the [128 k][64 n] tile: rows 4 apart are 16 banks apart
// EPI 1 / 2 (round 4): the two products of a GRU cell's backward with the elementwise kernel that followed each folded into the epilogue (GruBwdEpi,
// gemm.h; train.hip: five launches per cell -> three) -- same accumulation, same formulas, one pass less over [B][K].
// Round 4: EIGHT waves per workgroup -- wave (k quarter wk, column half wn) runs a quarter of each chunk's MFMA chain (a wave's chain was K / 2 dependent
// 64-cycle fp32 MFMAs: 8 us at K = 512, on 8 CUs) and the four partial tiles are added in a fixed order through LDS; 512 threads also fetch a chunk in
// half the round trips.  (The sum over k is split in four: results differ from the tiled kernel's in the last bit.)
#define SK_THREADS 512
template <int EPI>
__device__ __forceinline__ void sk_body(const G32& g, const GruBwdEpi& ep, float* As, float* Bs) {
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int wave = wv & 1, wk = wv >> 1;
  const int lr = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 64;
  f32x16g acc;
#pragma unroll
  for (int q = 0; q < 16; q++) acc[q] = 0.f;
  for (int kc = 0; kc < g.K; kc += SK_K) {
    float4 ra[2], rb[4];
#pragma unroll
    for (int p = 0; p < 2; p++) {    // A tile: 32 rows x 32 float4 = 1024 float4 / 512 threads
      const int idx = p * SK_THREADS + tid, m = m0 + (idx >> 5), k = kc + (idx & 31) * 4;
      ra[p] = (m < g.M && k < g.K) ? *reinterpret_cast<const float4*>(g.A + (int64_t)m * g.lda + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {    // B tile: 128 k rows x 16 float4
      const int idx = p * SK_THREADS + tid, k = kc + (idx >> 4), n = n0 + (idx & 15) * 4;
      rb[p] = (k < g.K && n < g.N) ? *reinterpret_cast<const float4*>(g.B + (int64_t)k * g.ldb + n) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();   // the previous chunk has been consumed
#pragma unroll
    for (int p = 0; p < 2; p++) { const int idx = p * SK_THREADS + tid; *reinterpret_cast<float4*>(&As[(idx >> 5) * SK_PA + (idx & 31) * 4]) = ra[p]; }
#pragma unroll
    for (int p = 0; p < 4; p++) { const int idx = p * SK_THREADS + tid; *reinterpret_cast<float4*>(&Bs[(idx >> 4) * SK_PB + (idx & 15) * 4]) = rb[p]; }
    __syncthreads();
#pragma unroll
    for (int s4 = 0; s4 < SK_K / 32; s4++) {   // this wave's quarter of the chunk: k = 32 wk + 8 s4 + 4 lh + c for MFMA c
      const int s = wk * (SK_K / 32) + s4;
      const float4 a = *reinterpret_cast<const float4*>(&As[lr * SK_PA + 8 * s + 4 * lh]);
      const float* q = &Bs[(8 * s + 4 * lh) * SK_PB + 32 * wave + lr];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, q[0], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, q[SK_PB], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, q[2 * SK_PB], acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, q[3 * SK_PB], acc, 0, 0, 0);
    }
  }
  // the four k quarters of a column half: partial tiles through LDS, added as ((0 + 1) + 2) + 3 by the wk = 0 waves
  __syncthreads();
  float* red = Bs;
  if (wk > 0) {
#pragma unroll
    for (int q = 0; q < 16; q++) red[((wk - 1) * 16 + q) * 128 + wave * 64 + lane] = acc[q];
  }
  __syncthreads();
  if (wk > 0) return;
#pragma unroll
  for (int q = 0; q < 16; q++) {
    float v = acc[q];
#pragma unroll
    for (int k = 0; k < 3; k++) v += red[(k * 16 + q) * 128 + wave * 64 + lane];
    acc[q] = v;
  }
  const int n = n0 + 32 * wave + lr;
  if (n >= g.N) return;
  if constexpr (EPI == 1) {   // acc = d[in | r*h][m][n] (k_gru_bwd_b): the input half is kept for EPI 2, the r*h half goes into d gates (r) and d h
    if (n < ep.in) {
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int m = m0 + acc_row(q, lh);
        if (m < g.M) g.C[(int64_t)m * g.ldc + n] = acc[q];
      }
    } else {
      const int k = n - ep.in;
      float r[16], hv[16], dh[16];   // every load before the first store: the pointers may alias as far as the compiler knows, and 16 dependent round trips cost more than the kernel saved
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int m = min(m0 + acc_row(q, lh), g.M - 1);
        r[q] = ep.gates[(int64_t)m * 2 * ep.H + k];
        hv[q] = ep.h[(int64_t)m * ep.ldh + k];
        dh[q] = ep.dhpart[(int64_t)m * ep.H + k];
      }
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int m = m0 + acc_row(q, lh);
        if (m >= g.M) continue;
        ep.dag[(int64_t)m * 2 * ep.H + k] = acc[q] * hv[q] * r[q] * (1.f - r[q]);
        ep.dhpart[(int64_t)m * ep.H + k] = dh[q] + acc[q] * r[q];
      }
    }
  } else if constexpr (EPI == 2) {   // acc = d[in | h][m][n] from the gates (k_gru_bwd_c): + the candidate's input half / + d h so far
    const bool inp = n < ep.in;
    float o[16];
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int m = min(m0 + acc_row(q, lh), g.M - 1);
      o[q] = inp ? ep.dcat_c[(int64_t)m * ep.K + n] : ep.dhpart[(int64_t)m * ep.H + (n - ep.in)];
    }
    if (inp && ep.na_dac) {   // k_gru_out_grad + k_gru_bwd_a of the cell below (same slot), on the d input this lane holds
      float dc[16], ds[16], mk[16], u[16], c[16], hv[16];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int m = min(m0 + acc_row(q, lh), g.M - 1);
        dc[q] = ep.na_dcar[(int64_t)m * ep.na_ldc + n];
        ds[q] = ep.na_dsb[(int64_t)m * ep.na_ld_dsb + n];
        mk[q] = ep.na_mask[(int64_t)m * ep.na_ldm];
        u[q] = ep.na_gates[(int64_t)m * 2 * ep.H + ep.H + n];
        c[q] = ep.na_cand[(int64_t)m * ep.H + n];
        hv[q] = ep.na_h[(int64_t)m * ep.na_ldh + n];
      }
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int m = m0 + acc_row(q, lh);
        if (m >= g.M) continue;
        const float d = (dc[q] + ds[q]) * mk[q] + (o[q] + acc[q]);
        ep.na_dac[(int64_t)m * ep.H + n] = d * (1.f - u[q]) * (1.f - c[q] * c[q]);
        ep.na_dag[(int64_t)m * 2 * ep.H + ep.H + n] = d * (hv[q] - c[q]) * u[q] * (1.f - u[q]);
        ep.na_dhpart[(int64_t)m * ep.H + n] = d * u[q];
      }
      return;
    }
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int m = m0 + acc_row(q, lh);
      if (m >= g.M) continue;
      if (inp) ep.dst_in[(int64_t)m * ep.ldi + n] = o[q] + acc[q];
      else ep.dst_h[(int64_t)m * ep.ldo + (n - ep.in)] = o[q] + acc[q];
    }
  } else {
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int m = m0 + acc_row(q, lh);
      if (m >= g.M) continue;
      float v = acc[q];
      float* c = g.C + (int64_t)m * g.ldc + n;
      if (g.beta != 0.f) v += g.beta * *c;
      *c = v;
    }
  }
}

template <int EPI>
__global__ __launch_bounds__(SK_THREADS) void k_gemm32_nn_skinny(const G32 g, const GruBwdEpi ep) {
  __shared__ float As[32 * SK_PA];
  __shared__ float Bs[SK_K * SK_PB];   // (>= 4 x 32 x 64 floats: the partial tiles of the final reduction)
  sk_body<EPI>(g, ep, As, Bs);
}
// up to two independent products of one launch (blockIdx.z): the cells of one anti-diagonal of a GRU stack's backward through time (gmm_train.hip) --
// their N may differ (layer 0 of a stack on one-hot rows has no dense input half): the grid covers the wider one
struct SkPair { G32 g[2]; GruBwdEpi ep[2]; };
template <int EPI>
__global__ __launch_bounds__(SK_THREADS) void k_gemm32_nn_skinny_pair(const SkPair p) {
  __shared__ float As[32 * SK_PA];
  __shared__ float Bs[SK_K * SK_PB];
  const G32& g = p.g[blockIdx.z];
  if ((int)blockIdx.x * 64 >= g.N || (int)blockIdx.y * 32 >= g.M) return;   // (uniform per workgroup)
  sk_body<EPI>(g, p.ep[blockIdx.z], As, Bs);
}

// rows up to which the skinny kernel (32-row tiles, the reduction split over a workgroup's waves) takes an nn product instead of the tiled one, and the
// scan's backward its fused three-launch cells (train.hip).  Round 4: 64 -> 4096 -- the tiled kernel's 33 us per [128 x 512] . [512 x 256] product was
// 1.4 ms of a 3.6 ms training step at 128 lanes (2.5 ms now; 256 / 512 / 1024 / 2048 lanes: 5.3 -> 4.0, 8.5 -> 7.2, 13.6 -> 12.2, 23.8 -> 22.9 ms;
// even at 4096, the tiled form ahead from 8192).  tune.skinny_max_rows (hiertcn_set_tuning) overrides.
int gemm32_skinny_max_rows() { return tune_now().skinny_max_rows; }

hipError_t gemm32_nn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, float beta, hipStream_t st,
                     float* scratch, size_t scratch_floats) {
  if ((K & 3) || (N & 3) || (lda & 3) || (ldb & 3)) return hipErrorInvalidValue;   // 16-byte loads along k (A) and n (B)
  if (M > 0 && M <= gemm32_skinny_max_rows() && N > 0) {
    G32 g;
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.beta = beta;
    g.k_per_split = K; g.parts = nullptr; g.a_div = g.b_div = 0; g.a_ss = g.b_ss = 0;
    const GruBwdEpi ep = {};
    hipLaunchKernelGGL(k_gemm32_nn_skinny<0>, dim3((unsigned)((N + 63) / 64), (unsigned)((M + 31) / 32)), dim3(SK_THREADS), 0, st, g, ep);
    return hipGetLastError();
  }
  return run32<false, true>(A, lda, B, ldb, C, ldc, M, N, K, beta, scratch, scratch_floats, st);
}

// C-less products of a GRU cell's backward (M <= 64 rows): mode 1: [M][K] = dac[M][H] . wc[H][K] with k_gru_bwd_b's arithmetic as the epilogue (ep.dcat_c
// receives the input half); mode 2: [M][K] = dag[M][2H] . wg[2H][K] with k_gru_bwd_c's.
hipError_t gemm32_nn_gru_bwd(int mode, const float* A, int64_t lda, const float* B, int64_t ldb, int M, int N, int K, const GruBwdEpi& ep, hipStream_t st) {
  if ((K & 3) || (N & 3) || (lda & 3) || (ldb & 3) || M <= 0 || M > gemm32_skinny_max_rows() || N != ep.in + ep.H) return hipErrorInvalidValue;
  G32 g;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = const_cast<float*>(ep.dcat_c); g.ldc = ep.K; g.M = M; g.N = N; g.K = K; g.beta = 0.f;
  g.k_per_split = K; g.parts = nullptr; g.a_div = g.b_div = 0; g.a_ss = g.b_ss = 0;
  const dim3 grid((unsigned)((N + 63) / 64), (unsigned)((M + 31) / 32));
  if (mode == 1) hipLaunchKernelGGL(k_gemm32_nn_skinny<1>, grid, dim3(SK_THREADS), 0, st, g, ep);
  else hipLaunchKernelGGL(k_gemm32_nn_skinny<2>, grid, dim3(SK_THREADS), 0, st, g, ep);
  return hipGetLastError();
}

// n = 1 or 2 such products as ONE launch (GruBwdProd: the arguments of gemm32_nn_gru_bwd)
hipError_t gemm32_nn_gru_bwd_n(int mode, const GruBwdProd* pr, int n, hipStream_t st) {
  if (n < 1 || n > 2) return hipErrorInvalidValue;
  if (n == 1) return gemm32_nn_gru_bwd(mode, pr[0].A, pr[0].lda, pr[0].B, pr[0].ldb, pr[0].M, pr[0].N, pr[0].K, pr[0].ep, st);
  SkPair p;
  int maxN = 0, maxM = 0;
  for (int i = 0; i < 2; i++) {
    const GruBwdProd& q = pr[i];
    if ((q.K & 3) || (q.N & 3) || (q.lda & 3) || (q.ldb & 3) || q.M <= 0 || q.M > gemm32_skinny_max_rows() || q.N != q.ep.in + q.ep.H) return hipErrorInvalidValue;
    G32& g = p.g[i];
    g.A = q.A; g.lda = q.lda; g.B = q.B; g.ldb = q.ldb; g.C = const_cast<float*>(q.ep.dcat_c); g.ldc = q.ep.K; g.M = q.M; g.N = q.N; g.K = q.K; g.beta = 0.f;
    g.k_per_split = q.K; g.parts = nullptr; g.a_div = g.b_div = 0; g.a_ss = g.b_ss = 0;
    p.ep[i] = q.ep;
    maxN = q.N > maxN ? q.N : maxN;
    maxM = q.M > maxM ? q.M : maxM;
  }
  const dim3 grid((unsigned)((maxN + 63) / 64), (unsigned)((maxM + 31) / 32), 2);
  if (mode == 1) hipLaunchKernelGGL(k_gemm32_nn_skinny_pair<1>, grid, dim3(SK_THREADS), 0, st, p);
  else hipLaunchKernelGGL(k_gemm32_nn_skinny_pair<2>, grid, dim3(SK_THREADS), 0, st, p);
  return hipGetLastError();
}

hipError_t gemm32_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int N1, int N2, int Mred, float beta,
                     float* scratch, size_t scratch_floats, hipStream_t st) {
  if ((N1 & 3) || (N2 & 3) || (lda & 3) || (ldb & 3)) return hipErrorInvalidValue;   // 16-byte loads along the output dimensions
  return run32<true, true>(A, lda, B, ldb, C, ldc, N1, N2, Mred, beta, scratch, scratch_floats, st);
}

// n <= 8 products C_i[N1_i][N2_i] (+)= A_i^T . B_i of the TN-with-slabs form in one launch (reductions of >= 512 rows: sliced over blockIdx.y like a single
// product's, partial tiles in `scratch`, one grouped sum behind them); false where the batch form does not apply: the caller then issues them one by one
bool gemm32_tn_slabs_batch_applies(const Gemm32TnDesc* d, int n) {
  if (n < 2 || n > G32_GROUP_MAX || ABL_ENV("HIERTCN_GEMM_NO_BATCH")) return false;
  for (int i = 0; i < n; i++) {
    if ((d[i].N1 & 3) || (d[i].N2 & 3) || (d[i].lda & 3) || (d[i].ldb & 3) || (d[i].a_ss & 3) || (d[i].b_ss & 3) || d[i].a_div < 0 || d[i].b_div < 0) return false;
    if (d[i].N1 <= 0 || d[i].N2 <= 0 || d[i].Mred != d[0].Mred) return false;   // (one slicing for the whole group)
  }
  return true;
}
hipError_t gemm32_tn_slabs_batch(const Gemm32TnDesc* d, int n, float* scratch, size_t scratch_floats, hipStream_t st) {
  G32Group grp;
  memset(&grp, 0, sizeof(grp));
  grp.n = n;
  int tiles = 0;
  int64_t mn_all = 0;
  for (int i = 0; i < n; i++) {
    tiles += ((d[i].N1 + GT - 1) / GT) * ((d[i].N2 + GT - 1) / GT);
    mn_all += (int64_t)d[i].N1 * d[i].N2;
  }
  const int K = d[0].Mred;
  int splits = 1;   // pick_splits' rule over the group's tiles
  while (scratch && tiles * splits < 384 && K / (2 * splits) >= 256 && (size_t)(2 * splits) * (size_t)mn_all <= scratch_floats && splits < 64) splits *= 2;
  tiles = 0;
  int64_t off = 0;
  for (int i = 0; i < n; i++) {
    G32& g = grp.g[i];
    g.A = d[i].A; g.lda = d[i].lda; g.B = d[i].B; g.ldb = d[i].ldb; g.C = d[i].C; g.ldc = d[i].ldc; g.M = d[i].N1; g.N = d[i].N2; g.K = d[i].Mred; g.beta = d[i].beta;
    g.k_per_split = (((K + splits - 1) / splits) + GK - 1) / GK * GK;
    g.parts = splits > 1 ? scratch + off : nullptr;
    off += (int64_t)splits * d[i].N1 * d[i].N2;
    g.a_div = d[i].a_div; g.a_ss = d[i].a_ss; g.b_div = d[i].b_div; g.b_ss = d[i].b_ss;
    grp.tile_off[i] = tiles;
    tiles += ((d[i].N1 + GT - 1) / GT) * ((d[i].N2 + GT - 1) / GT);
  }
  for (int i = n; i <= G32_GROUP_MAX; i++) grp.tile_off[i] = tiles;
  hipLaunchKernelGGL((k_gemm32_group<true, true>), dim3((unsigned)tiles, (unsigned)splits), dim3(256), 0, st, grp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || splits == 1) return e;
  hipLaunchKernelGGL(k_gemm_sum_parts_group, dim3((unsigned)((mn_all + 255) / 256)), dim3(256), 0, st, grp, splits);
  return hipGetLastError();
}

hipError_t gemm32_tn_slabs(const float* A, int64_t lda, int a_div, int64_t a_ss, const float* B, int64_t ldb, int b_div, int64_t b_ss, float* C,
                           int64_t ldc, int N1, int N2, int Mred, float beta, float* scratch, size_t scratch_floats, hipStream_t st) {
  if ((N1 & 3) || (N2 & 3) || (lda & 3) || (ldb & 3) || (a_ss & 3) || (b_ss & 3) || a_div < 0 || b_div < 0) return hipErrorInvalidValue;
  return run32<true, true>(A, lda, B, ldb, C, ldc, N1, N2, Mred, beta, scratch, scratch_floats, st, a_div, a_ss, b_div, b_ss);
}

// =================================================================================================================================
// 16-bit storage operands (bf16, or IEEE half in the HIERTCN_F16 build), fp32 results, v_mfma_f32_32x32x16.
// Both kernels take a ROW SHIFT per tap: inside sequences of L rows, tap j reads row r + shift_j of the shifted operand and a zero row
// where that leaves the sequence -- the backward forms of the causal conv (customized_tcn_cell.py:46-49: o[t] = sum_j x[t - s_j] . W_j
// => d x[t] = sum_j d o[t + s_j] . W_j^T and d W_j = sum_t d o[t]^T x[t - s_j]) without materialising shifted copies.
// =================================================================================================================================
#define HK 64      // k chunk of the NT kernel (16-bit elements)
#define HP_K 144   // byte pitch of a [mn][64 k] tile: 36 words, (9 r) mod 16 distinct for the 16 lanes of a ds_read_b128 phase
#define HR 64      // row chunk of the TN kernel
#define HP_T 320   // byte pitch of a [row][128 channel] tile: 80 words, the four rows of a transposed-read block 16 banks apart

__device__ __forceinline__ uint2 lds_tr16(const unsigned char* p) {
  typedef short s16x4g __attribute__((ext_vector_type(4)));
  const s16x4g v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4g*)p);
  return __builtin_bit_cast(uint2, v);
}

struct G16nt {
  const uint16_t* A; int64_t lda;        // [M][K] rows, read at row r + shift[tap]
  const uint16_t* Bt; int64_t ldb;       // [taps][N][K]; BKM (one tap only): B [K][N], the reduction index on the rows
  float* C; int64_t ldc;
  int64_t M; int N, K;
  int taps, L; int shift[8];
  float beta;
};

// C[M][N] = beta C + sum_taps A[r + shift_tap][K] . Bt[tap][N][K]^T      (BKM: ... + A[M][K] . B[K][N], B staged as [k][n] rows and read
// through the hardware transpose read, like both operands of k_gemm16_tn)
template <bool BKM>
__global__ __launch_bounds__(256) void k_gemm16_nt(const G16nt g) {
  __shared__ __attribute__((aligned(16))) unsigned char As[GT * HP_K];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[BKM ? HK * HP_T : GT * HP_K];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.y * GT;
  const int n0 = blockIdx.x * GT;
  const int wm = wave >> 1, wn = wave & 1;
  const int ldrow = tid >> 3, ldc8 = (tid & 7) * 8;      // this thread stages rows ldrow + 32 p, elements [ldc8, ldc8 + 8) of a 64-deep chunk
  int64_t a_row[4]; int a_t[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int64_t gm = m0 + ldrow + 32 * p;
    a_row[p] = gm < g.M ? gm : -1;
    a_t[p] = (gm < g.M && g.L > 0) ? (int)(gm % g.L) : 0;
  }
  const int kchunks = (g.K + HK - 1) / HK;
  const int n_it = g.taps * kchunks;
  uint4 ra[4], rb[4];
  auto fetch = [&](int it) {
    const int j = it / kchunks, kc = (it - j * kchunks) * HK;
    const int sh = g.shift[j];
    const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; p++) {
      ra[p] = zero; rb[p] = zero;
      const bool inside = g.L > 0 ? (a_t[p] + sh >= 0 && a_t[p] + sh < g.L) : true;
      if (a_row[p] >= 0 && inside && kc + ldc8 < g.K) ra[p] = *reinterpret_cast<const uint4*>(g.A + (a_row[p] + sh) * g.lda + kc + ldc8);
      if (BKM) {   // tile [64 k][128 n]: thread -> k rows tid / 16 + 16 p, columns 8 (tid % 16)
        const int k = kc + (tid >> 4) + 16 * p, gn = n0 + (tid & 15) * 8;
        if (k < g.K && gn < g.N) rb[p] = *reinterpret_cast<const uint4*>(g.Bt + (int64_t)k * g.ldb + gn);
      } else {
        const int gn = n0 + ldrow + 32 * p;
        if (gn < g.N && kc + ldc8 < g.K) rb[p] = *reinterpret_cast<const uint4*>(g.Bt + ((int64_t)j * g.N + gn) * g.ldb + kc + ldc8);
      }
    }
  };
  const int gq = lane >> 4, w16 = lane & 15;
  const int tr_off = (8 * lh + (w16 >> 2)) * HP_T + (16 * (gq & 1) + 4 * (w16 & 3)) * 2;
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[i][j][q] = 0.f;
  fetch(0);
  for (int it = 0; it < n_it; it++) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; p++) {
      *reinterpret_cast<uint4*>(As + (ldrow + 32 * p) * HP_K + ldc8 * 2) = ra[p];
      if (BKM) *reinterpret_cast<uint4*>(Bs + ((tid >> 4) + 16 * p) * HP_T + (tid & 15) * 16) = rb[p];
      else *reinterpret_cast<uint4*>(Bs + (ldrow + 32 * p) * HP_K + ldc8 * 2) = rb[p];
    }
    __syncthreads();
    if (it + 1 < n_it) fetch(it + 1);
#pragma unroll
    for (int s = 0; s < HK / 16; s++) {
      const uint4 a0 = *reinterpret_cast<const uint4*>(As + (64 * wm + lr) * HP_K + 32 * s + 16 * lh);
      const uint4 a1 = *reinterpret_cast<const uint4*>(As + (64 * wm + 32 + lr) * HP_K + 32 * s + 16 * lh);
      uint4 b0, b1;
      if (BKM) {
        const unsigned char* pb = Bs + 16 * s * HP_T + (64 * wn) * 2 + tr_off;
        const uint2 l0 = lds_tr16(pb), h0 = lds_tr16(pb + 4 * HP_T), l1 = lds_tr16(pb + 64), h1 = lds_tr16(pb + 64 + 4 * HP_T);
        b0 = make_uint4(l0.x, l0.y, h0.x, h0.y);
        b1 = make_uint4(l1.x, l1.y, h1.x, h1.y);
      } else {
        b0 = *reinterpret_cast<const uint4*>(Bs + (64 * wn + lr) * HP_K + 32 * s + 16 * lh);
        b1 = *reinterpret_cast<const uint4*>(Bs + (64 * wn + 32 + lr) * HP_K + 32 * s + 16 * lh);
      }
      acc[0][0] = mfma32(a0, b0, acc[0][0]);
      acc[0][1] = mfma32(a0, b1, acc[0][1]);
      acc[1][0] = mfma32(a1, b0, acc[1][0]);
      acc[1][1] = mfma32(a1, b1, acc[1][1]);
    }
  }
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int n = n0 + 64 * wn + 32 * j + lr;
    if (n >= g.N) continue;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int64_t m = m0 + 64 * wm + 32 * i + acc_row(q, lh);
        if (m >= g.M) continue;
        float v = acc[i][j][q];
        float* c = g.C + m * g.ldc + n;
        if (g.beta != 0.f) v += g.beta * *c;
        *c = v;
      }
  }
}

hipError_t gemm16_nt(const uint16_t* A, int64_t lda, const uint16_t* Bt, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, int taps,
                     const int* shift, int L, float beta, hipStream_t st) {
  if (M <= 0 || N <= 0) return hipSuccess;
  if ((K & 7) || (lda & 7) || (ldb & 7) || taps < 1 || taps > 8) return hipErrorInvalidValue;
  const int64_t unit = L > 0 ? L : 1;
  const int64_t rows_per_launch = ((int64_t)65535 * GT / unit) * unit;   // gridDim.y <= 65535; cut at a sequence boundary
  for (int64_t r0 = 0; r0 < M; r0 += rows_per_launch) {
    G16nt g;
    g.A = A + r0 * lda; g.lda = lda; g.Bt = Bt; g.ldb = ldb; g.C = C + r0 * ldc; g.ldc = ldc;
    g.M = M - r0 < rows_per_launch ? M - r0 : rows_per_launch; g.N = N; g.K = K; g.taps = taps; g.L = L; g.beta = beta;
    for (int j = 0; j < 8; j++) g.shift[j] = j < taps && shift ? shift[j] : 0;
    hipLaunchKernelGGL(k_gemm16_nt<false>, dim3((unsigned)((N + GT - 1) / GT), (unsigned)((g.M + GT - 1) / GT)), dim3(256), 0, st, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

hipError_t gemm16_nn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, float beta,
                     hipStream_t st) {
  if (M <= 0 || N <= 0) return hipSuccess;
  if ((K & 7) || (N & 7) || (lda & 7) || (ldb & 7) || M > (int64_t)65535 * GT) return hipErrorInvalidValue;
  G16nt g;
  g.A = A; g.lda = lda; g.Bt = B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.taps = 1; g.L = 0; g.beta = beta;
  for (int j = 0; j < 8; j++) g.shift[j] = 0;
  hipLaunchKernelGGL(k_gemm16_nt<true>, dim3((unsigned)((N + GT - 1) / GT), (unsigned)((M + GT - 1) / GT)), dim3(256), 0, st, g);
  return hipGetLastError();
}

struct G16tn {
  const uint16_t* A; int64_t lda;        // [R][N1]
  const uint16_t* B; int64_t ldb;        // [R][N2], read at row r + shift[tap]
  int64_t R; int N1, N2;
  int taps, L; int shift[8];
  int splits; int64_t rows_per_split;    // multiple of HR
  float* parts;                          // [taps][splits][N1][N2]; null (splits == 1, taps == 1): straight into C with beta
  float* C; int64_t ldc; float beta;
};

// parts[tap][split][N1][N2] = sum over the split's rows r of A[r][N1]^T . B[r + shift_tap][N2]: both operands are staged as [row][channel]
// tiles, exactly as they lie in memory, and the MFMA fragments (8 consecutive ROWS of one channel per lane) come out of LDS through the
// hardware transpose read ds_read_b64_tr_b16 (two reads of 4 rows each); every lane takes part in every read (EXEC all ones).
__global__ __launch_bounds__(256) void k_gemm16_tn(const G16tn g) {
  __shared__ __attribute__((aligned(16))) unsigned char As[HR * HP_T];
  __shared__ __attribute__((aligned(16))) unsigned char Bs[HR * HP_T];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lh = lane >> 5;
  const int i0 = blockIdx.y * GT, j0 = blockIdx.x * GT;
  const int tap = blockIdx.z / g.splits, split = blockIdx.z - tap * g.splits;
  const int sh = g.shift[tap];
  const int wm = wave >> 1, wn = wave & 1;
  const int64_t r_lo = split * g.rows_per_split;
  const int64_t r_hi = r_lo + g.rows_per_split < g.R ? r_lo + g.rows_per_split : g.R;
  const int n_it = r_hi > r_lo ? (int)((r_hi - r_lo + HR - 1) / HR) : 0;
  const int ldrow = tid >> 4, ldc8 = (tid & 15) * 8;     // tile [64 rows][128 channels]: thread -> rows ldrow + 16 p, channels [ldc8, ldc8 + 8)
  uint4 ra[4], rb[4];
  auto fetch = [&](int it) {
    const uint4 zero = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int p = 0; p < 4; p++) {
      const int64_t r = r_lo + (int64_t)it * HR + ldrow + 16 * p;
      ra[p] = zero; rb[p] = zero;
      if (r < r_hi) {
        if (i0 + ldc8 < g.N1) ra[p] = *reinterpret_cast<const uint4*>(g.A + r * g.lda + i0 + ldc8);
        const int t = g.L > 0 ? (int)(r % g.L) : 0;
        const bool inside = g.L > 0 ? (t + sh >= 0 && t + sh < g.L) : true;
        if (inside && j0 + ldc8 < g.N2) rb[p] = *reinterpret_cast<const uint4*>(g.B + (r + sh) * g.ldb + j0 + ldc8);
      }
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[i][j][q] = 0.f;
  // transposed-read address of this lane inside a 32-channel x 16-row operand block: 16-lane group gq = lane / 16 covers rows 8 lh + 0..3
  // (second read: + 4) and channels 16 (gq & 1) + 0..15; lane 4 q + p of the group supplies row q, channels 4 p .. 4 p + 3
  const int gq = lane >> 4, w16 = lane & 15;
  const int tr_off = (8 * lh + (w16 >> 2)) * HP_T + (16 * (gq & 1) + 4 * (w16 & 3)) * 2;
  if (n_it > 0) fetch(0);
  for (int it = 0; it < n_it; it++) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; p++) {
      *reinterpret_cast<uint4*>(As + (ldrow + 16 * p) * HP_T + ldc8 * 2) = ra[p];
      *reinterpret_cast<uint4*>(Bs + (ldrow + 16 * p) * HP_T + ldc8 * 2) = rb[p];
    }
    __syncthreads();
    if (it + 1 < n_it) fetch(it + 1);
#pragma unroll
    for (int s = 0; s < HR / 16; s++) {
      uint4 a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const unsigned char* pa = As + 16 * s * HP_T + (64 * wm + 32 * i) * 2 + tr_off;
        const unsigned char* pb = Bs + 16 * s * HP_T + (64 * wn + 32 * i) * 2 + tr_off;
        const uint2 a_lo = lds_tr16(pa), a_hi = lds_tr16(pa + 4 * HP_T);
        const uint2 b_lo = lds_tr16(pb), b_hi = lds_tr16(pb + 4 * HP_T);
        a[i] = make_uint4(a_lo.x, a_lo.y, a_hi.x, a_hi.y);
        b[i] = make_uint4(b_lo.x, b_lo.y, b_hi.x, b_hi.y);
      }
      acc[0][0] = mfma32(a[0], b[0], acc[0][0]);
      acc[0][1] = mfma32(a[0], b[1], acc[0][1]);
      acc[1][0] = mfma32(a[1], b[0], acc[1][0]);
      acc[1][1] = mfma32(a[1], b[1], acc[1][1]);
    }
  }
  float* out = g.parts ? g.parts + ((int64_t)tap * g.splits + split) * g.N1 * g.N2 : g.C;
  const int64_t ldo = g.parts ? g.N2 : g.ldc;
  const float beta = g.parts ? 0.f : g.beta;
  const int lr = lane & 31;
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int n = j0 + 64 * wn + 32 * j + lr;
    if (n >= g.N2) continue;
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int m = i0 + 64 * wm + 32 * i + acc_row(q, lh);
        if (m >= g.N1) continue;
        float v = acc[i][j][q];
        float* c = out + (int64_t)m * ldo + n;
        if (beta != 0.f) v += beta * *c;
        *c = v;
      }
  }
}

// C[tap][N1][N2] (contiguous per tap, tap stride c_tap) = beta C + sum of the split partials, in split order
__global__ void k_gemm16_tn_sum(const float* __restrict__ parts, int taps, int splits, int64_t n, float* __restrict__ C, int64_t c_tap, float beta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * taps) return;
  const int tap = (int)(i / n);
  const int64_t e = i - (int64_t)tap * n;
  float s = 0.f;
  for (int p = 0; p < splits; p++) s += parts[((int64_t)tap * splits + p) * n + e];
  float* c = C + tap * c_tap + e;
  *c = beta != 0.f ? beta * *c + s : s;
}

hipError_t gemm16_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t c_tap, int N1, int N2, int64_t R, int taps,
                     const int* shift, int L, float beta, float* scratch, size_t scratch_floats, hipStream_t st) {
  if (N1 <= 0 || N2 <= 0 || taps < 1 || taps > 8) return hipErrorInvalidValue;
  if ((N1 & 7) || (N2 & 7) || (lda & 7) || (ldb & 7)) return hipErrorInvalidValue;
  const int64_t tiles = (int64_t)((N1 + GT - 1) / GT) * ((N2 + GT - 1) / GT) * taps;
  const int64_t n = (int64_t)N1 * N2;
  int splits = 1;
  while (scratch && tiles * splits < 512 && R / (2 * splits) >= 4 * HR && (size_t)(2 * splits) * taps * n <= scratch_floats && splits < 256) splits *= 2;
  G16tn g;
  g.C = C; g.ldc = N2; g.beta = beta;
  if (splits == 1 && taps == 1) {   // enough tiles already (a [N x C] table gradient): no partials, one launch
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.R = R; g.N1 = N1; g.N2 = N2; g.taps = 1; g.L = L; g.splits = 1;
    g.rows_per_split = (R + HR - 1) / HR * HR;
    g.parts = nullptr;
    for (int j = 0; j < 8; j++) g.shift[j] = j == 0 && shift ? shift[0] : 0;
    hipLaunchKernelGGL(k_gemm16_tn, dim3((unsigned)((N2 + GT - 1) / GT), (unsigned)((N1 + GT - 1) / GT), 1u), dim3(256), 0, st, g);
    return hipGetLastError();
  }
  if (!scratch || (size_t)splits * taps * n > scratch_floats) return hipErrorInvalidValue;
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.R = R; g.N1 = N1; g.N2 = N2; g.taps = taps; g.L = L; g.splits = splits;
  g.rows_per_split = ((R + splits - 1) / splits + HR - 1) / HR * HR;
  g.parts = scratch;
  for (int j = 0; j < 8; j++) g.shift[j] = j < taps && shift ? shift[j] : 0;
  hipLaunchKernelGGL(k_gemm16_tn, dim3((unsigned)((N2 + GT - 1) / GT), (unsigned)((N1 + GT - 1) / GT), (unsigned)(taps * splits)), dim3(256), 0, st, g);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_gemm16_tn_sum, dim3((unsigned)((n * taps + 255) / 256)), dim3(256), 0, st, scratch, taps, splits, n, C, c_tap, beta);
  return hipGetLastError();
}
