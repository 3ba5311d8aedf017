// High level of HierTCN: session mean-pool of target embeddings (model_hier.py:83-85) and the
// across-session GRU scan (model_hier.py:36-37,91,93; arithmetic customed_gru_cell.py:309-337,
// 1050-1073).  The scan also emits the state half of the low level's input projection
// (model_hier.py:54-55 -> model_tcn.py:35), so the TCN kernel never touches the state.
//
// Both kernels are fp32 end to end (the recurrence is the only true sequential dependency of
// the path and < 0.1 % of its FLOPs; rounding the carried state would compound over sessions).
#include <stdlib.h>
#include <string.h>

#include "common.h"

// ---------------------------------------------------------------------------------------------
// session mean-pool: one wave per session.  HBM/L2-bound row gather: each 16-byte lane load
// covers 8 bf16 channels; E/8 lanes cover a row, 64/(E/8) rows are in flight per instruction.
// ---------------------------------------------------------------------------------------------
// Small batches (round 4): the list of rows that carry a target (metrics.hip, k_compact_small: ascending row numbers with y > 0, their count, and the
// zero fill of the scoring outputs) depends on the ids alone, so it rides on this launch as ONE EXTRA WORKGROUP instead of being a launch of its own
// in front of the scoring kernel (7 us + a boundary of a 0.23 ms batch).  256 threads, four rows each per pass.
struct PoolCompact { const int32_t* y; int R; int32_t* map; int32_t* n_valid; int32_t* z0; float* z1; float* z2; };
__device__ void compact_block(const PoolCompact pc) {
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int base = 0;
  for (int r0 = 0; r0 < pc.R; r0 += 1024) {
    const int r = r0 + 4 * tid;
    int f[4], c = 0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      f[j] = (r + j < pc.R) ? (pc.y[r + j] > 0 ? 1 : 0) : 0;
      c += f[j];
      if (r + j < pc.R) {
        if (pc.z0) pc.z0[r + j] = 0;
        if (pc.z1) pc.z1[r + j] = 0.f;
        if (pc.z2) pc.z2[r + j] = 0.f;
      }
    }
    int incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int v = __shfl_up(incl, d);
      incl += (lane >= d) ? v : 0;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int off = base, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; w++) {
      off += (w < wave) ? wsum[w] : 0;
      tot += wsum[w];
    }
    int pos = off + incl - c;
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (f[j]) pc.map[pos++] = r + j;
    base += tot;
    __syncthreads();
  }
  if (tid == 0) pc.n_valid[0] = base;
}

__global__ __launch_bounds__(256) void k_session_pool(const uint16_t* __restrict__ emb, const float* __restrict__ emb_bias,
                                                      const int32_t* __restrict__ y, int64_t n_sessions, int L, int E, int N,
                                                      float* __restrict__ pooled, const PoolCompact pc) {
  if (pc.map && blockIdx.x == gridDim.x - 1) {   // (uniform per workgroup)
    compact_block(pc);
    return;
  }
  const int lane = threadIdx.x & 63;
  const int64_t sess = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sess >= n_sessions) return;
  const int lanes_per_row = E >> 3;          // 16-byte chunks per row (E multiple of 8, <= 512)
  const int rows_per_pass = 64 / lanes_per_row;
  const int sub = lane / lanes_per_row;      // which row of the pass
  const int ch = lane % lanes_per_row;       // which chunk of the row
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; j++) acc[j] = 0.f;
  int cnt = 0;
  const int32_t* ys = y + sess * L;
  for (int t0 = 0; t0 < L; t0 += rows_per_pass) {
    int t = t0 + sub;
    int id = (t < L && sub < rows_per_pass) ? ys[t] : 0;
    if (id > 0 && id < N) {   // an id >= N is a zero one-hot row: no embedding, not counted (model_hier.py:83)
      cnt++;
      uint4 v = *reinterpret_cast<const uint4*>(emb + (int64_t)id * E + ch * 8);
      uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc[2 * j] += bf2f((uint16_t)(w[j] & 0xffff));
        acc[2 * j + 1] += bf2f((uint16_t)(w[j] >> 16));
      }
    }
  }
  // combine the sub-rows: lanes with equal `ch` sit lanes_per_row apart
  for (int off = lanes_per_row; off < 64; off <<= 1) {
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] += __shfl_xor(acc[j], off);
    cnt += __shfl_xor(cnt, off);
  }
  // cnt now counts every (row, chunk-lane) pair once per chunk lane: each valid row was counted
  // by exactly one `sub` group, and the xor-reduction above summed the groups.
  if (sub == 0) {
    const float inv = 1.0f / (float)cnt;  // cnt == 0 -> inf * 0 = NaN, as the reference's 0/0 (model_hier.py:84)
    float4 o0, o1;
    const float* b = emb_bias + ch * 8;
    o0.x = acc[0] * inv + b[0]; o0.y = acc[1] * inv + b[1]; o0.z = acc[2] * inv + b[2]; o0.w = acc[3] * inv + b[3];
    o1.x = acc[4] * inv + b[4]; o1.y = acc[5] * inv + b[5]; o1.z = acc[6] * inv + b[6]; o1.w = acc[7] * inv + b[7];
    float* dst = pooled + sess * E + ch * 8;
    *reinterpret_cast<float4*>(dst) = o0;
    *reinterpret_cast<float4*>(dst + 4) = o1;
  }
}

hipError_t launch_session_pool(const ModelDev& m, const int32_t* y, int64_t n_sessions, int L, float* pooled, hipStream_t st) {
  if (n_sessions <= 0) return hipSuccess;
  unsigned blocks = (unsigned)((n_sessions + 3) / 4);
  PoolCompact pc;
  memset(&pc, 0, sizeof(pc));
  hipLaunchKernelGGL(k_session_pool, dim3(blocks), dim3(256), 0, st, m.emb, m.emb_bias, y, n_sessions, L, m.E, m.N, pooled, pc);
  return hipGetLastError();
}

// the same launch + the row compaction of launch_compact_rows (map, n_valid, zero-filled z0 / z1 / z2) as one extra workgroup; R <= 8192 rows
bool session_pool_compact_applies(int64_t R) { return R > 0 && R <= 8192 && !ABL_ENV("HIERTCN_POOL_COMPACT_OFF"); }
hipError_t launch_session_pool_compact(const ModelDev& m, const int32_t* y, int64_t n_sessions, int L, float* pooled, int32_t* map, int32_t* n_valid,
                                       int32_t* z0, float* z1, float* z2, hipStream_t st) {
  if (n_sessions <= 0) return hipSuccess;
  unsigned blocks = (unsigned)((n_sessions + 3) / 4);
  PoolCompact pc;
  pc.y = y; pc.R = (int)(n_sessions * L); pc.map = map; pc.n_valid = n_valid; pc.z0 = z0; pc.z1 = z1; pc.z2 = z2;
  hipLaunchKernelGGL(k_session_pool, dim3(blocks + 1), dim3(256), 0, st, m.emb, m.emb_bias, y, n_sessions, L, m.E, m.N, pooled, pc);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// GRU scan = a stream of small exact-fp32 MFMA GEMMs, one pair per (slot, layer):
//     gates GEMM :  sigmoid([in | h] . Wg + bg)            -> gates, r*h
//     cand  GEMM :  tanh([in | r*h] . Wc + bc), h' = u*h + (1-u)*c -> next layer input, masked next state
// over ALL lanes of the batch at once (M = B rows), so every launch fills the chip and each weight
// matrix is read from L2 once per launch instead of once per 8 lanes.  v_mfma_f32_32x32x2_f32 is
// bit-for-bit an fp32 fma chain (cdna_hip_programming.md 3, 'FP32-input MFMA'), so the carried state
// stays fp32-exact.  The state of every slot is kept ([B][S+1][HT]) and the state half of the low
// level's input projection (model_hier.py:54-55 -> model_tcn.py:35) is ONE GEMM over all slots at
// the end.  Two earlier designs (lanes-per-workgroup with VALU dot products: 7.5 ms; lanes along K
// with a wavefront shuffle fold: 11.0 ms at cfg-3) are in the git history; this one is launch-
// latency bound at ~90 launches per batch.
// ---------------------------------------------------------------------------------------------
enum { EPI_STORE = 0, EPI_GATES = 1, EPI_CAND = 2 };

struct GemmArgs {
  const float* A1; int lda1; int K1;   // activations, first K1 columns
  const float* A2; int lda2; int K2;   // activations, next K2 columns (may be null when K2 == 0)
  const float* Wt;                     // [N][K1+K2]  (output-major)
  const float* bias;                   // [N] or null
  int M, N;
  float* out; int ldo;                 // STORE: C ; GATES: gates [M][2H] ; CAND: next layer input [M][H]
  float* out2; int ldo2;               // GATES: r*h [M][H] ; CAND: masked next-state slice
  const float* h; int ldh;             // GATES / CAND: the layer's old h slice
  const float* gates; int ldg;         // CAND
  const float* mask; int ldm;          // CAND: mask[m * ldm]
  float* cand_out; int ldc;            // CAND, optional: the candidate c = tanh(.) (kept for the backward pass)
  const int32_t* seq_len; int tstep;   // CAND, optional (low-level GRU under dynamic_rnn): row m is active iff tstep < seq_len[m];
                                       //   inactive rows keep their old state, and their output row is zero
  uint16_t* hid_out; int64_t ldho;     // CAND, optional: the layer's output as a 16-bit row (scoring operand), zero when inactive
  float* hid_out32;                    // ... or as an fp32 row (the exact numerics mode, exact.hip); same row stride ldho
  int H;
  int rowmap_S;                        // STORE: >0 -> row m = b*(S+1)+s maps to out row b*S+s, s == S skipped
  const float* add_tab; int add_ld; int add_n;   // optional: + add_tab[id][n] for id = add_ids[m * add_ids_ld] in (0, add_n) -- the product of a
  const int32_t* add_ids; int add_ids_ld;        //   one-hot input row with the input half of the kernel is a row gather (model_gru_gmm on XING)
};

typedef float f32x16v __attribute__((ext_vector_type(16)));

// SPLITK (small batches: fewer tiles than CUs): the four waves of a workgroup share ONE tile and take every fourth
// 8-deep k-step each, then add their accumulators through LDS -- a 32-lane XING batch otherwise runs each of its 40
// dependent GRU GEMMs as a single 256-MFMA chain per wave (20 us per launch, 80 % of the batch's time).
struct GemmPair { GemmArgs g[2]; };

// SPLITK = 0: four tiles per 256-thread workgroup; 4: ONE tile per workgroup, wave w takes every 4th k-step (small launches).  (A 16-wave /
// 1024-thread form of the split was measured on the 32-lane XING batch: 0.49 vs 0.41 ms per batch -- the serial sum of fifteen partial
// tiles and the larger workgroups cost more than the shorter k chains save.)
template <int EPI, int SPLITK = 0>
__global__ __launch_bounds__(256) void k_gemm_f32(const GemmPair gp) {
  const GemmArgs& g = gp.g[blockIdx.y];   // up to two independent GEMMs of one shape per launch (the scan's wavefront: launch_gru_scan)
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int tiles_n = g.N >> 5;
  const int t = SPLITK ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;   // (SPLITK: all waves of the workgroup share tile blockIdx.x)
  constexpr int KSTEP = SPLITK ? SPLITK : 1;
  const int k_first = SPLITK ? wave : 0;
  const int tm = t / tiles_n, tn = t - tm * tiles_n;
  if (tm * 32 >= g.M) return;
  const int K = g.K1 + g.K2;
  int mrow = tm * 32 + lr;
  const bool mvalid = mrow < g.M;
  if (!mvalid) mrow = g.M - 1;
  // A operand = weights (rows = output columns n), B operand = activations^T (cols = lanes/users):
  // D[row = n][col = user]; per 8 k's one float4 per lane on each side, k order (j, half) -> k0 + 4*half + j.
  const float* wp = g.Wt + (int64_t)(tn * 32 + lr) * K + 4 * lh;
  const float* x1 = g.A1 + (int64_t)mrow * g.lda1 + 4 * lh;
  f32x16v acc;
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = 0.f;
  const int n1 = g.K1 >> 3;
  // four 8-deep k-steps per trip, their eight loads issued before the first MFMA (hipcc does not unroll this loop on request -- the
  // run-time trip count and stride defeat `#pragma unroll 8` -- and one k-step per trip exposes a full load latency every 4 MFMAs);
  // the MFMA order, hence every result bit, is unchanged
  auto k_run = [&](const float* wq, const float* xq, int nk) {
    int i = k_first;
    for (; i + 3 * KSTEP < nk; i += 4 * KSTEP) {
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        a[u] = *reinterpret_cast<const float4*>(wq + 8 * (i + u * KSTEP));
        b[u] = *reinterpret_cast<const float4*>(xq + 8 * (i + u * KSTEP));
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].x, b[u].x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].y, b[u].y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].z, b[u].z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u].w, b[u].w, acc, 0, 0, 0);
      }
    }
    for (; i < nk; i += KSTEP) {
      const float4 a = *reinterpret_cast<const float4*>(wq + 8 * i);
      const float4 b = *reinterpret_cast<const float4*>(xq + 8 * i);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
    }
  };
  if (n1 > 0) k_run(wp, x1, n1);
  if (g.K2 > 0) {
    const float* x2 = g.A2 + (int64_t)mrow * g.lda2 + 4 * lh;
    const float* wp2 = wp + g.K1;
    const int n2 = g.K2 >> 3;
    k_run(wp2, x2, n2);
  }
  if (SPLITK) {   // fixed-order sum of the partial tiles (deterministic)
    constexpr int NP_ = SPLITK ? SPLITK - 1 : 1;
    __shared__ float part[NP_][16][64];
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < 16; i++) part[wave - 1][i][lane] = acc[i];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      float s_ = acc[i];
#pragma unroll
      for (int w = 0; w < NP_; w++) s_ += part[w][i][lane];
      acc[i] = s_;
    }
  }
  if (!mvalid) return;
  const int m = tm * 32 + lr;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int n0 = tn * 32 + 8 * q + 4 * lh;  // 4 consecutive output columns
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = acc[4 * q + j] + (g.bias ? g.bias[n0 + j] : 0.f);
    if (g.add_tab) {
      const int id = g.add_ids[(int64_t)m * g.add_ids_ld];
      if (id > 0 && id < g.add_n) {
        const float4 a4 = *reinterpret_cast<const float4*>(g.add_tab + (int64_t)id * g.add_ld + n0);
        v[0] += a4.x; v[1] += a4.y; v[2] += a4.z; v[3] += a4.w;
      }
    }
    if (EPI == EPI_STORE) {
      int64_t orow = m;
      if (g.rowmap_S > 0) {
        const int b = m / (g.rowmap_S + 1), sl = m - b * (g.rowmap_S + 1);
        if (sl == g.rowmap_S) continue;
        orow = (int64_t)b * g.rowmap_S + sl;
      }
      *reinterpret_cast<float4*>(g.out + orow * g.ldo + n0) = make_float4(v[0], v[1], v[2], v[3]);
    } else if (EPI == EPI_GATES) {
      float s4[4];
#pragma unroll
      for (int j = 0; j < 4; j++) s4[j] = 1.0f / (1.0f + expf(-v[j]));  // customed_gru_cell.py:323
      *reinterpret_cast<float4*>(g.out + (int64_t)m * g.ldo + n0) = make_float4(s4[0], s4[1], s4[2], s4[3]);
      if (n0 < g.H) {  // r is the FIRST half (:324); r_state = r * h (:326)
        const float4 hv = *reinterpret_cast<const float4*>(g.h + (int64_t)m * g.ldh + n0);
        *reinterpret_cast<float4*>(g.out2 + (int64_t)m * g.ldo2 + n0) =
            make_float4(s4[0] * hv.x, s4[1] * hv.y, s4[2] * hv.z, s4[3] * hv.w);
      }
    } else {  // EPI_CAND: c = tanh(.), h' = u*h + (1-u)*c (:335-336); state *= mask (model_hier.py:93)
      const float4 hv = *reinterpret_cast<const float4*>(g.h + (int64_t)m * g.ldh + n0);
      const float4 uv = *reinterpret_cast<const float4*>(g.gates + (int64_t)m * g.ldg + g.H + n0);
      const float mk = g.mask ? g.mask[(int64_t)m * g.ldm] : 1.0f;
      float hn[4];
      const float c0 = tanhf(v[0]), c1 = tanhf(v[1]), c2 = tanhf(v[2]), c3 = tanhf(v[3]);
      if (g.cand_out) *reinterpret_cast<float4*>(g.cand_out + (int64_t)m * g.ldc + n0) = make_float4(c0, c1, c2, c3);
      hn[0] = uv.x * hv.x + (1.0f - uv.x) * c0;
      hn[1] = uv.y * hv.y + (1.0f - uv.y) * c1;
      hn[2] = uv.z * hv.z + (1.0f - uv.z) * c2;
      hn[3] = uv.w * hv.w + (1.0f - uv.w) * c3;
      *reinterpret_cast<float4*>(g.out + (int64_t)m * g.ldo + n0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
      if (g.seq_len) {   // [TF-sem] dynamic_rnn past a row's sequence_length: state copied through, output zero (model_gru.py:47-53)
        const bool active = g.tstep < g.seq_len[m];
        *reinterpret_cast<float4*>(g.out2 + (int64_t)m * g.ldo2 + n0) =
            active ? make_float4(hn[0], hn[1], hn[2], hn[3]) : make_float4(hv.x, hv.y, hv.z, hv.w);
        if (g.hid_out) {
          uint2 o;
          o.x = active ? pack2bf(hn[0], hn[1]) : 0u;
          o.y = active ? pack2bf(hn[2], hn[3]) : 0u;
          *reinterpret_cast<uint2*>(g.hid_out + (int64_t)m * g.ldho + n0) = o;
        }
        if (g.hid_out32)
          *reinterpret_cast<float4*>(g.hid_out32 + (int64_t)m * g.ldho + n0) =
              active ? make_float4(hn[0], hn[1], hn[2], hn[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        continue;
      }
      *reinterpret_cast<float4*>(g.out2 + (int64_t)m * g.ldo2 + n0) = make_float4(hn[0] * mk, hn[1] * mk, hn[2] * mk, hn[3] * mk);
    }
  }
}

// n = 1 or 2 GEMMs of the SAME shape (M, N) in one launch
template <int EPI>
static hipError_t launch_gemm(const GemmArgs* g, int n, hipStream_t st) {
  GemmPair gp;
  gp.g[0] = g[0];
  gp.g[1] = g[n > 1 ? 1 : 0];
  const int tiles = ((g[0].M + 31) / 32) * (g[0].N / 32);
  if (tiles * n <= tune_now().gemm_split_max_tiles)
    hipLaunchKernelGGL((k_gemm_f32<EPI, 4>), dim3(tiles, n), dim3(256), 0, st, gp);
  else
    hipLaunchKernelGGL((k_gemm_f32<EPI, 0>), dim3((tiles + 3) / 4, n), dim3(256), 0, st, gp);
  return hipGetLastError();
}
template <int EPI>
static hipError_t launch_gemm(const GemmArgs& g, hipStream_t st) { return launch_gemm<EPI>(&g, 1, st); }

__global__ void k_state_rows(const float* __restrict__ src, int64_t src_stride, float* __restrict__ dst, int64_t dst_stride,
                             int rows, int cols) {
  // dst[r][c] = src[r][c] for row strides that differ (state_in -> history slot 0, history -> state_out / state_before)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int64_t r = i / cols, c = i - r * cols;
  dst[r * dst_stride + c] = src[r * src_stride + c];
}
static hipError_t copy_rows(const float* src, int64_t ss, float* dst, int64_t ds, int rows, int cols, hipStream_t st) {
  const int64_t n = (int64_t)rows * cols;
  hipLaunchKernelGGL(k_state_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, ss, dst, ds, rows, cols);
  return hipGetLastError();
}

size_t gru_workspace_bytes(const ModelDev& m, int B, int S) {
  // history [B][S+1][HT] + per layer: gates [B][2H] + rh [B][H] + unmasked outputs [2 (slot parity)][B][H]
  return sizeof(float) * ((size_t)B * (S + 1) * m.HT + (size_t)m.nl * B * 5 * m.H);
}

size_t gru_save_floats(const ModelDev& m, int B, int S) {
  // per (slot, layer): gates [B][2H] + r*h [B][H] + candidate [B][H] + unmasked output [B][H]
  return (size_t)S * m.nl * B * 5 * m.H;
}

hipError_t launch_gru_scan(const ModelDev& m, const float* pooled, const float* mask, const float* state_in, int B, int S,
                           float* sproj, float* state_out, float* state_before, float* ws, hipStream_t st, float* save) {
  if (B <= 0) return hipSuccess;
  const int E = m.E, H = m.H, HT = m.HT;
  float* hist = ws;                                  // [B][S+1][HT]
  float* gates0 = hist + (size_t)B * (S + 1) * HT;   // [nl][B][2H]
  float* rh0 = gates0 + (size_t)m.nl * B * 2 * H;    // [nl][B][H]
  float* hcur = rh0 + (size_t)m.nl * B * H;          // [2][nl][B][H]: by slot parity -- cell (s+1, l) writes its output in the same
                                                     // launch in which cell (s, l+1) still reads the output of (s, l)
  const size_t cell = (size_t)B * 5 * H;             // one (slot, layer) record of `save`
  const int64_t hs = (int64_t)(S + 1) * HT;          // row stride of the history
  hipError_t e;
  // small batches: the whole scan as one persistent launch (gru_persist.hip); it also fills slot 0 of the history and state_out
  const int persist_groups = gru_persist_groups(m, B, S, save != nullptr);
  if (persist_groups > 0) {
    if ((e = launch_gru_scan_persist(m, pooled, mask, state_in, B, S, state_out, ws, persist_groups, st, save)) != hipSuccess) return e;
  } else if ((e = copy_rows(state_in, HT, hist, hs, B, HT, st)) != hipSuccess) return e;
  // Wavefront over (slot, layer): cell (s, l) needs (s, l-1) and (s-1, l) only, so the cells with s + l = tau are independent -- layer 1
  // of slot s runs beside layer 0 of slot s + 1.  The critical path is S + nl - 1 cells instead of S * nl, and the independent cells of
  // a step go out as ONE grouped launch per GEMM kind (two cells each): 2 (S + 1)
  // dependent launches instead of 4 S for the reference's two layers -- the scan is latency-bound at the reference's batch sizes (§4.5).
  auto cell_args = [&](int s, int l, GemmArgs& gg, GemmArgs& gc) {
    memset(&gg, 0, sizeof(gg));
    const float* in = (l == 0) ? pooled + (size_t)s * E
                      : (save ? save + ((size_t)s * m.nl + (l - 1)) * cell + (size_t)B * 4 * H : hcur + ((size_t)(s & 1) * m.nl + (l - 1)) * B * H);
    const int ldin = (l == 0) ? S * E : H;
    const int in_dim = (l == 0) ? E : H;
    const float* hold = hist + (size_t)s * HT + (size_t)l * H;
    float* gates = gates0 + (size_t)l * B * 2 * H;
    float* rh = rh0 + (size_t)l * B * H;
    float* hout = hcur + ((size_t)(s & 1) * m.nl + l) * B * H;
    if (save) {  // training: every (slot, layer) keeps its own gates / r*h / candidate / output (GruSaved in train.hip)
      float* rec = save + ((size_t)s * m.nl + l) * cell;
      gates = rec; rh = rec + (size_t)B * 2 * H; gg.cand_out = rec + (size_t)B * 3 * H; gg.ldc = H;
      hout = rec + (size_t)B * 4 * H;
    }
    gg.M = B; gg.H = H;
    gg.A1 = in; gg.lda1 = ldin; gg.K1 = in_dim;
    gg.A2 = hold; gg.lda2 = (int)hs; gg.K2 = H;
    gg.Wt = m.gru_wg[l]; gg.bias = m.gru_bg[l]; gg.N = 2 * H;
    gg.out = gates; gg.ldo = 2 * H; gg.out2 = rh; gg.ldo2 = H; gg.h = hold; gg.ldh = (int)hs;
    gc = gg;
    gc.A2 = rh; gc.lda2 = H;
    gc.Wt = m.gru_wc[l]; gc.bias = m.gru_bc[l]; gc.N = H;
    gc.out = hout; gc.ldo = H;
    gc.out2 = hist + (size_t)(s + 1) * HT + (size_t)l * H; gc.ldo2 = (int)hs;
    gc.gates = gates; gc.ldg = 2 * H;
    gc.mask = mask + s; gc.ldm = S;
  };
  for (int tau = 0; tau < S + m.nl - 1 && persist_groups == 0; tau++) {
    GemmArgs gg[HIERTCN_MAX_LAYERS], gc[HIERTCN_MAX_LAYERS];
    int n = 0;
    for (int l = 0; l < m.nl; l++) {
      const int s = tau - l;
      if (s < 0 || s >= S) continue;
      cell_args(s, l, gg[n], gc[n]);
      n++;
    }
    // two cells per launch: one (M, N) -- their reduction lengths may differ (layer 0 reads E inputs, the others H): every block reads its own arguments
    for (int i = 0; i < n; i += 2)
      if ((e = launch_gemm<EPI_GATES>(gg + i, (n - i) > 1 ? 2 : 1, st)) != hipSuccess) return e;
    for (int i = 0; i < n; i += 2)
      if ((e = launch_gemm<EPI_CAND>(gc + i, (n - i) > 1 ? 2 : 1, st)) != hipSuccess) return e;
  }
  if (sproj) {  // state half of the input projection for every slot at once: [B*(S+1), HT] x [HT, E]  (null: a caller without that projection, gmm_train.hip)
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.M = B * (S + 1); g.N = E; g.A1 = hist; g.lda1 = HT; g.K1 = HT; g.K2 = 0;
    g.Wt = m.projS; g.bias = nullptr; g.out = sproj; g.ldo = E; g.rowmap_S = S;
    if ((e = launch_gemm<EPI_STORE>(g, st)) != hipSuccess) return e;
  }
  if (persist_groups == 0 && (e = copy_rows(hist + (size_t)S * HT, hs, state_out, HT, B, HT, st)) != hipSuccess) return e;
  if (state_before) {
    for (int s = 0; s < S; s++)
      if ((e = copy_rows(hist + (size_t)s * HT, hs, state_before + (size_t)s * HT, (int64_t)S * HT, B, HT, st)) != hipSuccess) return e;
  }
  return hipSuccess;
}


// ---------------------------------------------------------------------------------------------
// HierGRU low level (model_hier.py:67-68 -> model_gru.py:16-63): the projected session rows run through a
// MultiRNNCell of GRUCells under dynamic_rnn, from a zero state, for `length` steps per session, where
// length = the NUMBER of non-zero projected rows of the session (model_gru.py:109-118, used as a prefix
// length).  Same exact-fp32 MFMA GEMM stream as the high level, M = all sessions of the batch, one
// (gates, candidate) pair per (step, layer); the top layer's outputs are written as the scoring
// kernel's 16-bit operand rows [session * L + t][H].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rows_f32_and_len(const uint16_t* __restrict__ u, int64_t n_sess, int L, int E, float* __restrict__ uf,
                                                          int32_t* __restrict__ seq_len) {
  // one wave per session: widen the projected rows to fp32 (the GEMM operand) and count the rows that are not all-zero
  const int lane = threadIdx.x & 63;
  const int64_t sess = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sess >= n_sess) return;
  int cnt = 0;
  for (int t = 0; t < L; t++) {
    const int64_t r = sess * L + t;
    int nz = 0;
    for (int c = lane; c < E; c += 64) {
      const float v = bf2f(u[r * E + c]);
      uf[r * E + c] = v;
      nz |= (v != 0.f) ? 1 : 0;
    }
    cnt += __any(nz) ? 1 : 0;
  }
  if (lane == 0) seq_len[sess] = cnt;
}

// evaluation through the session-tile / pipelined kernels (lgru_tile.hip, lgru_pipe.hip, lgru_wide.hip read the 16-bit rows themselves): only the
// lengths -- one wave per session, 16 bytes per lane and load; the fp32 copy k_rows_f32_and_len also writes is the fp32 cells' and the weight gradients' operand
__global__ __launch_bounds__(256) void k_len_u16(const uint16_t* __restrict__ u, int64_t n_sess, int L, int E, int32_t* __restrict__ seq_len) {
  const int lane = threadIdx.x & 63;
  const int64_t sess = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sess >= n_sess) return;
  const int cpr = E >> 3;                       // 16-byte chunks per row
  int cnt = 0;
  for (int t0 = 0; t0 < L; t0 += 4) {           // four rows in flight per lane
    int nz[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int t = t0 + j;
      if (t < L)
        for (int c = lane; c < cpr; c += 64) {
          const uint4 v = *reinterpret_cast<const uint4*>(u + ((sess * L + t) * (int64_t)E) + c * 8);
          // a 16-bit float is non-zero unless all bits but the sign are clear (-0 == 0: the fp32 test `v != 0.f` of k_rows_f32_and_len)
          nz[j] |= ((v.x | v.y | v.z | v.w) & 0x7fff7fffu) ? 1 : 0;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) cnt += (t0 + j < L && __any(nz[j])) ? 1 : 0;
  }
  if (lane == 0) seq_len[sess] = cnt;
}

// exact numerics mode: the projected rows arrive as fp32 (exact.hip); only the lengths are taken here
__global__ __launch_bounds__(256) void k_len_f32(const float* __restrict__ uf, int64_t n_sess, int L, int E, int32_t* __restrict__ seq_len) {
  const int lane = threadIdx.x & 63;
  const int64_t sess = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sess >= n_sess) return;
  int cnt = 0;
  for (int t = 0; t < L; t++) {
    const int64_t r = sess * L + t;
    int nz = 0;
    for (int c = lane; c < E; c += 64) nz |= (uf[r * E + c] != 0.f) ? 1 : 0;
    cnt += __any(nz) ? 1 : 0;
  }
  if (lane == 0) seq_len[sess] = cnt;
}

// split-operand cells: state hi / lo ping-pong [2][2][M][HT] + per layer r*h hi / lo and output hi / lo [nl][4][M][H]
// (step, layer) wavefront of a records-keeping GRU stack: every cell has its own record and state slot, so the cells of one anti-diagonal share nothing;
// they go out two per launch (k_gemm_f32 takes a pair of GEMMs with one (M, N); their K and input kind may differ -- each block reads its own arguments)
template <class CellArgs>
static hipError_t records_wavefront(int nl, int L, CellArgs&& cell_args, hipStream_t st) {
  hipError_t e;
  for (int tau = 0; tau < L + nl - 1; tau++) {
    GemmArgs gg[HIERTCN_MAX_LAYERS], gc[HIERTCN_MAX_LAYERS];
    int n = 0;
    for (int l = 0; l < nl; l++) {
      const int t = tau - l;
      if (t < 0 || t >= L) continue;
      cell_args(t, l, gg[n], gc[n]);
      n++;
    }
    for (int i = 0; i < n; i += 2)
      if ((e = launch_gemm<EPI_GATES>(gg + i, (n - i) > 1 ? 2 : 1, st)) != hipSuccess) return e;
    for (int i = 0; i < n; i += 2)
      if ((e = launch_gemm<EPI_CAND>(gc + i, (n - i) > 1 ? 2 : 1, st)) != hipSuccess) return e;
  }
  return hipSuccess;
}

size_t low_gru_x3_elems(const ModelDev& m, int64_t n_sess) { return (size_t)10 * n_sess * m.HT; }   // states 2 x 2, r*h 2, outputs 2 x 2 (hi, lo each)

size_t low_gru_workspace_bytes(const ModelDev& m, int64_t n_sess, int L) {
  // embedded rows + u (16-bit) + u (fp32) + lengths + two state buffers + one (gates | r*h | cand | out) record per layer + the split scratch
  const size_t R = (size_t)n_sess * L;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  return 2 * al(R * m.E * 2) + al(R * m.E * 4) + al((size_t)n_sess * 4) + 2 * al((size_t)n_sess * m.HT * 4) +
         al((size_t)m.nl * n_sess * 5 * m.H * 4) + al(low_gru_x3_elems(m, n_sess) * 2) +
         (n_sess <= 1024 ? al(low_gru_persist_floats(m, n_sess, L) * 4) : 0) +   // small batches: the persistent launch's history / exchange buffers
         al(lgru_pipe_bytes(m, n_sess, L));                                        // ... and the layer-pipelined launch's exchange slots
}

// training keeps everything the backward pass needs: L+1 states and one record per (step, layer)
size_t low_gru_save_floats(const ModelDev& m, int64_t n_sess, int L) {
  return (size_t)(L + 1) * n_sess * m.HT + (size_t)L * m.nl * n_sess * 5 * m.H;
}

hipError_t launch_low_gru_bufs(const ModelDev& m, const int32_t* x, const float* sproj, int64_t n_sess, int L, uint16_t* hidden,
                               const LowGruBufs& b, hipStream_t st) {
  if (n_sess <= 0) return hipSuccess;
  const int E = m.E, H = m.H, HT = m.HT;
  const int64_t R = n_sess * L;
  const size_t cell = (size_t)n_sess * 5 * H;   // gates [M][2H] | r*h [M][H] | cand [M][H] | out [M][H]
  hipError_t e;
  if (b.hidden32) {   // exact numerics mode: b.uf already holds the fp32 projection (exact.hip)
    hipLaunchKernelGGL(k_len_f32, dim3((unsigned)((n_sess + 3) / 4)), dim3(256), 0, st, b.uf, n_sess, L, E, b.seq_len);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  } else if (!m.hier) {
    // single-level model_gru (model.py:66-67): model_gru.py:46's 'emb' dense on one-hot rows IS the embedding lookup -- the gathered rows feed the stack
    if ((e = launch_embed_rows(m, x, R, b.u16, st)) != hipSuccess) return e;
    if (!b.keep_all && lgru_tile_applies(m, n_sess) && (E % 8) == 0)
      hipLaunchKernelGGL(k_len_u16, dim3((unsigned)((n_sess + 3) / 4)), dim3(256), 0, st, b.u16, n_sess, L, E, b.seq_len);
    else
      hipLaunchKernelGGL(k_rows_f32_and_len, dim3((unsigned)((n_sess + 3) / 4)), dim3(256), 0, st, b.u16, n_sess, L, E, b.uf, b.seq_len);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  } else {
    if ((e = launch_embed_rows(m, x, R, b.a_emb, st)) != hipSuccess) return e;
    {  // 'emb' projection of [e, state] (model_gru.py:46), rounded to the 16-bit storage type like the TCN's input
      LayerArgs a = {};
      a.in = b.a_emb; a.out = b.u16; a.W = m.projT; a.Wf = m.projF; a.seq_init = sproj;
      a.Cin = E; a.Cout = E; a.taps = 1; a.dil = 0; a.L = L; a.n_seq = n_sess; a.linear = 1;
      if ((e = launch_tcn_layer(a, st)) != hipSuccess) return e;
    }
    if (!b.keep_all && lgru_tile_applies(m, n_sess) && (E % 8) == 0)   // (evaluation on the 16-bit rows: nobody reads the fp32 copy)
      hipLaunchKernelGGL(k_len_u16, dim3((unsigned)((n_sess + 3) / 4)), dim3(256), 0, st, b.u16, n_sess, L, E, b.seq_len);
    else
      hipLaunchKernelGGL(k_rows_f32_and_len, dim3((unsigned)((n_sess + 3) / 4)), dim3(256), 0, st, b.u16, n_sess, L, E, b.uf, b.seq_len);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  if (!b.hidden32 && lgru_tile_applies(m, n_sess)) {   // small batches: one workgroup per 32 sessions through every step and layer (lgru_tile.hip); training keeps the records
    if (b.keep_all && (e = hipMemsetAsync(b.states, 0, (size_t)n_sess * HT * 4, st)) != hipSuccess) return e;   // slot 0 of the kept states: the zero state the backward pass reads at t = 0
    if (b.pipe && lgru_wide_applies(m, n_sess))   // H = 256: eight workgroups per (tile, layer), weights resident in registers (lgru_wide.hip)
      return launch_lgru_wide(m, b.u16, b.seq_len, n_sess, L, hidden, b.keep_all ? b.states : nullptr, b.keep_all ? b.rec : nullptr, b.pipe, st);
    if (b.pipe && lgru_pipe_applies(m, n_sess))   // H <= 128: one workgroup per (tile, layer), weights resident in registers (lgru_pipe.hip)
      return launch_lgru_pipe(m, b.u16, b.seq_len, n_sess, L, hidden, b.keep_all ? b.states : nullptr, b.keep_all ? b.rec : nullptr, b.pipe, st);
    return launch_lgru_tile(m, b.u16, b.seq_len, n_sess, L, hidden, b.keep_all ? b.states : nullptr, b.keep_all ? b.rec : nullptr, st);
  }
  if (b.persist && !b.keep_all && low_gru_persist_applies(m, n_sess))   // small evaluation batches (fp32 modes): the whole recurrence as ONE launch (gru_persist.hip)
    return launch_low_gru_persist(m, b.uf, b.seq_len, n_sess, L, hidden, b.hidden32, b.persist, st);
  if ((e = hipMemsetAsync(b.states, 0, (size_t)n_sess * HT * 4, st)) != hipSuccess) return e;   // zero_state (model_gru.py:52: initial_state=None)
  if (b.x3 && !b.hidden32 && gru_x3_applies(m)) {
    // split-operand cells on the 16-bit MFMA (gru_x3.hip): same records, states and outputs as the fp32 cells below
    const size_t MS = (size_t)n_sess * HT, MH = (size_t)n_sess * H;
    uint16_t* st_hi = b.x3;                 // [2][M][HT]
    uint16_t* st_lo = st_hi + 2 * MS;       // [2][M][HT]
    uint16_t* rh_hi = st_lo + 2 * MS;       // [nl][M][H]
    uint16_t* rh_lo = rh_hi + (size_t)m.nl * MH;
    uint16_t* o_hi = rh_lo + (size_t)m.nl * MH;        // [nl][2 step parities][M][H]
    uint16_t* o_lo = o_hi + 2 * (size_t)m.nl * MH;
    if ((e = hipMemsetAsync(st_hi, 0, MS * 2, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(st_lo, 0, MS * 2, st)) != hipSuccess) return e;
    // Wavefront over (step, layer): the cells (l, t = tau - l) of a diagonal are independent -- layer l at step t needs layer l - 1 at step t
    // (previous diagonal) and its own state from step t - 1 (previous diagonal) -- so a diagonal is TWO launches (gates, candidates) whatever
    // the number of layers: 2 (L + nl - 1) launches instead of 2 L nl, each with nl x the tiles (at 20,480 sessions and two layers 1280 tiles =
    // exactly five rounds of 256 CUs instead of 2.5).  The layers' output operands are double-buffered by step parity (layer l + 1 reads step
    // t's while layer l writes step t + 1's); states and records are per layer slices already.
    for (int tau = 0; tau < L + m.nl - 1; tau++) {
      GruX3Batch gb, cb;
      memset(&gb, 0, sizeof(gb));
      memset(&cb, 0, sizeof(cb));
      int n = 0;
      for (int l = 0; l < m.nl; l++) {
        const int t = tau - l;
        if (t < 0 || t >= L) continue;
        const float* s_old = b.states + (size_t)(t % b.n_states) * n_sess * HT;
        float* s_new = b.states + (size_t)((t + 1) % b.n_states) * n_sess * HT;
        const int po = t & 1, pn = (t + 1) & 1;
        float* rec = b.rec + (b.keep_all ? ((size_t)t * m.nl + l) : (size_t)l) * cell;
        GruX3Args& g = gb.g[n];
        g.M = n_sess; g.H = H; g.save_all = b.keep_all;
        if (l == 0) { g.A1h = b.u16 + (size_t)t * E; g.A1l = nullptr; g.lda1 = (int64_t)L * E; g.K1 = E; }
        else { g.A1h = o_hi + ((size_t)(l - 1) * 2 + po) * MH; g.A1l = o_lo + ((size_t)(l - 1) * 2 + po) * MH; g.lda1 = H; g.K1 = H; }
        g.A2h = st_hi + (size_t)po * MS + (size_t)l * H; g.A2l = st_lo + (size_t)po * MS + (size_t)l * H; g.lda2 = HT; g.K2 = H;
        g.Wh = m.lgru_wg_hi[l]; g.Wl = m.lgru_wg_lo[l]; g.bias = m.lgru_bg[l]; g.N = 2 * H; g.epi = X3_GATES;
        g.gates = rec; g.h = s_old + (size_t)l * H; g.ldh = HT;
        g.rh = rec + (size_t)n_sess * 2 * H; g.rh_hi = rh_hi + (size_t)l * MH; g.rh_lo = rh_lo + (size_t)l * MH;
        g.seq_len = b.seq_len; g.tstep = t;
        GruX3Args& c = cb.g[n];
        c = g;
        c.A2h = g.rh_hi; c.A2l = g.rh_lo; c.lda2 = H;
        c.Wh = m.lgru_wc_hi[l]; c.Wl = m.lgru_wc_lo[l]; c.bias = m.lgru_bc[l]; c.N = H; c.epi = X3_CAND;
        c.cand_out = rec + (size_t)n_sess * 3 * H;
        c.out = rec + (size_t)n_sess * 4 * H;
        if (l + 1 < m.nl) { c.out_hi = o_hi + ((size_t)l * 2 + po) * MH; c.out_lo = o_lo + ((size_t)l * 2 + po) * MH; }
        c.state_new = s_new + (size_t)l * H; c.state_hi = st_hi + (size_t)pn * MS + (size_t)l * H; c.state_lo = st_lo + (size_t)pn * MS + (size_t)l * H;
        c.lds_ = HT;
        if (l == m.nl - 1) { c.hid_out = hidden + (size_t)t * H; c.ldho = (int64_t)L * H; }
        n++;
      }
      gb.n = cb.n = n;
      if ((e = launch_gru_x3(gb, st)) != hipSuccess) return e;
      if ((e = launch_gru_x3(cb, st)) != hipSuccess) return e;
    }
    return hipSuccess;
  }
  auto cell_args = [&](int t, int l, GemmArgs& g, GemmArgs& gc) {
    const float* s_old = b.states + (size_t)(t % b.n_states) * n_sess * HT;
    float* s_new = b.states + (size_t)((t + 1) % b.n_states) * n_sess * HT;
    float* rec = b.rec + (b.keep_all ? ((size_t)t * m.nl + l) : (size_t)l) * cell;
    const float* below = (l == 0) ? nullptr : b.rec + (b.keep_all ? ((size_t)t * m.nl + l - 1) : (size_t)(l - 1)) * cell + (size_t)n_sess * 4 * H;
    memset(&g, 0, sizeof(g));
    const float* in = (l == 0) ? b.uf + (size_t)t * E : below;
    const int ldin = (l == 0) ? L * E : H;
    const int in_dim = (l == 0) ? E : H;
    const float* hold = s_old + (size_t)l * H;
    float* gates = rec;
    float* rh = rec + (size_t)n_sess * 2 * H;
    g.M = (int)n_sess; g.H = H;
    g.A1 = in; g.lda1 = ldin; g.K1 = in_dim;
    g.A2 = hold; g.lda2 = HT; g.K2 = H;
    g.Wt = m.lgru_wg[l]; g.bias = m.lgru_bg[l]; g.N = 2 * H;
    g.out = gates; g.ldo = 2 * H; g.out2 = rh; g.ldo2 = H; g.h = hold; g.ldh = HT;
    gc = g;
    gc.A2 = rh; gc.lda2 = H;
    gc.Wt = m.lgru_wc[l]; gc.bias = m.lgru_bc[l]; gc.N = H;
    gc.cand_out = rec + (size_t)n_sess * 3 * H; gc.ldc = H;
    gc.out = rec + (size_t)n_sess * 4 * H; gc.ldo = H;
    gc.out2 = s_new + (size_t)l * H; gc.ldo2 = HT;
    gc.gates = gates; gc.ldg = 2 * H;
    gc.seq_len = b.seq_len; gc.tstep = t;
    if (l == m.nl - 1) {
      gc.ldho = (int64_t)L * H;
      if (b.hidden32) gc.hid_out32 = b.hidden32 + (size_t)t * H; else gc.hid_out = hidden + (size_t)t * H;
    }
  };
  // every record kept (the fp32 training mode) and a state slot per step: the cells of an anti-diagonal share nothing -- the (step, layer) wavefront,
  // two cells per launch (records_wavefront); otherwise the layers share one record each and the cells go one after the other
  if (b.keep_all && b.n_states > L) return records_wavefront(m.nl, L, cell_args, st);
  for (int t = 0; t < L; t++)
    for (int l = 0; l < m.nl; l++) {
      GemmArgs g, gc;
      cell_args(t, l, g, gc);
      if ((e = launch_gemm<EPI_GATES>(g, st)) != hipSuccess) return e;
      if ((e = launch_gemm<EPI_CAND>(gc, st)) != hipSuccess) return e;
    }
  return hipSuccess;
}

hipError_t launch_low_gru(const ModelDev& m, const int32_t* x, const float* sproj, int64_t n_sess, int L, uint16_t* hidden, void* workspace,
                          hipStream_t st) {
  if (n_sess <= 0) return hipSuccess;
  const int64_t R = n_sess * L;
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  auto take = [&](size_t bytes) { unsigned char* p = w; w += (bytes + 255) & ~(size_t)255; return p; };
  LowGruBufs b;
  b.a_emb = reinterpret_cast<uint16_t*>(take((size_t)R * m.E * 2));
  b.u16 = reinterpret_cast<uint16_t*>(take((size_t)R * m.E * 2));
  b.uf = reinterpret_cast<float*>(take((size_t)R * m.E * 4));
  b.seq_len = reinterpret_cast<int32_t*>(take((size_t)n_sess * 4));
  b.states = reinterpret_cast<float*>(take(2 * (((size_t)n_sess * m.HT * 4 + 255) & ~(size_t)255)));
  b.n_states = 2;
  b.rec = reinterpret_cast<float*>(take((size_t)m.nl * n_sess * 5 * m.H * 4));
  b.keep_all = 0;
  b.hidden32 = nullptr;
  b.x3 = reinterpret_cast<uint16_t*>(take(low_gru_x3_elems(m, n_sess) * 2));
  b.persist = n_sess <= 1024 ? reinterpret_cast<float*>(take(low_gru_persist_floats(m, n_sess, L) * 4)) : nullptr;
  b.pipe = lgru_pipe_bytes(m, n_sess, L) ? take(lgru_pipe_bytes(m, n_sess, L)) : nullptr;
  return launch_low_gru_bufs(m, x, sproj, n_sess, L, hidden, b, st);
}


// ---------------------------------------------------------------------------------------------
// GRU stack on ONE-HOT input rows (model_gru_gmm.py:33-48 on XING, where x = one_hot(id) * sign(id), model.py:56-61; no 'emb' projection
// in that model): layer 0's input half of [x, h] . W is the row W[id] -- gathered in the GEMM epilogue (GemmArgs::add_tab) -- so the
// GEMMs of layer 0 run over the state half only; dynamic_rnn's sequence_length (model_gru_gmm.py:57-67: the COUNT of non-zero input rows =
// of ids > 0, used as a prefix length) as in launch_low_gru_bufs.  hidden: the top layer's outputs [session * L + t][H], storage type.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count_ids(const int32_t* __restrict__ x, int64_t n_sess, int L, int N, int32_t* __restrict__ seq_len) {
  const int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= n_sess) return;
  int c = 0;
  for (int t = 0; t < L; t++) {
    const int id = x[s * L + t];
    c += (id > 0 && id < N) ? 1 : 0;
  }
  seq_len[s] = c;
}

size_t onehot_gru_workspace_bytes(int H, int nl, int64_t n_sess) {
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  return al((size_t)n_sess * 4) + al((size_t)2 * n_sess * H * nl * 4) + al((size_t)2 * nl * n_sess * 5 * H * 4);   // (records by the step's parity: gru_rows_cells)
}
// ... on DENSE rows of width in_dim (model_hier.py:68-69: model_gru_gmm as the low level reads the [e_t, state] rows): + the rows as fp32
size_t rows_gru_workspace_bytes(int H, int nl, int in_dim, int64_t n_sess, int L) {
  return onehot_gru_workspace_bytes(H, nl, n_sess) + (((size_t)n_sess * L * in_dim * 4 + 255) & ~(size_t)255);
}

static hipError_t gru_rows_cells(const OneHotGru& w, const int32_t* x, const float* rows, int in_dim, int64_t n_sess, int L, uint16_t* hidden,
                                 int32_t* seq_len, float* states, float* recs, hipStream_t st);

hipError_t launch_rows_gru(const OneHotGru& w, const uint16_t* rows16, int in_dim, int64_t n_sess, int L, uint16_t* hidden, void* workspace, hipStream_t st) {
  if (n_sess <= 0) return hipSuccess;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  unsigned char* p = reinterpret_cast<unsigned char*>(workspace);
  int32_t* seq_len = reinterpret_cast<int32_t*>(p); p += al((size_t)n_sess * 4);
  float* states = reinterpret_cast<float*>(p);      p += al((size_t)2 * n_sess * w.H * w.nl * 4);
  float* recs = reinterpret_cast<float*>(p);        p += al((size_t)2 * w.nl * n_sess * 5 * w.H * 4);
  float* rows = reinterpret_cast<float*>(p);
  // the rows as fp32 (the GEMM operand) and dynamic_rnn's sequence_length = the count of rows that are not all-zero (model_gru_gmm.py:57-67)
  hipLaunchKernelGGL(k_rows_f32_and_len, dim3((unsigned)((n_sess + 3) / 4)), dim3(256), 0, st, rows16, n_sess, L, in_dim, rows, seq_len);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return gru_rows_cells(w, nullptr, rows, in_dim, n_sess, L, hidden, seq_len, states, recs, st);
}

hipError_t launch_onehot_gru(const OneHotGru& w, const int32_t* x, int64_t n_sess, int L, uint16_t* hidden, void* workspace, hipStream_t st) {
  if (n_sess <= 0) return hipSuccess;
  const int HT = w.H * w.nl;
  auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
  unsigned char* p = reinterpret_cast<unsigned char*>(workspace);
  int32_t* seq_len = reinterpret_cast<int32_t*>(p); p += al((size_t)n_sess * 4);
  float* states = reinterpret_cast<float*>(p);      p += al((size_t)2 * n_sess * HT * 4);
  float* recs = reinterpret_cast<float*>(p);
  hipLaunchKernelGGL(k_count_ids, dim3((unsigned)((n_sess + 255) / 256)), dim3(256), 0, st, x, n_sess, L, w.N, seq_len);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return gru_rows_cells(w, x, nullptr, 0, n_sess, L, hidden, seq_len, states, recs, st);
}

// the cells: layer 0's input half is a gathered table row (x: ids, one-hot input) or a dense fp32 row (rows, width in_dim)
static hipError_t gru_rows_cells(const OneHotGru& w, const int32_t* x, const float* rows, int in_dim, int64_t n_sess, int L, uint16_t* hidden,
                                 int32_t* seq_len, float* states, float* recs, hipStream_t st) {
  const int H = w.H, HT = w.H * w.nl;
  const size_t cell = (size_t)n_sess * 5 * H;   // gates [M][2H] | r*h [M][H] | cand [M][H] | out [M][H]
  hipError_t e;
  if ((e = hipMemsetAsync(states, 0, (size_t)n_sess * HT * 4, st)) != hipSuccess) return e;   // zero_state
  // (step, layer) wavefront, two cells per launch (records_wavefront): the layers' records alternate between two sets by the step's parity, so that cell
  // (t + 1, l - 1) does not overwrite the output row cell (t, l) is reading in the same launch; the states alternate likewise (columns of one layer each)
  auto cell_args = [&](int t, int l, GemmArgs& g, GemmArgs& gc) {
    const float* s_old = states + (size_t)(t & 1) * n_sess * HT;
    float* s_new = states + (size_t)((t + 1) & 1) * n_sess * HT;
    float* rec = recs + ((size_t)(t & 1) * w.nl + l) * cell;
    memset(&g, 0, sizeof(g));
    const float* hold = s_old + (size_t)l * H;
    float* gates = rec;
    float* rh = rec + (size_t)n_sess * 2 * H;
    g.M = (int)n_sess; g.H = H;
    if (l == 0 && rows) {
      g.A1 = rows + (size_t)t * in_dim; g.lda1 = L * in_dim; g.K1 = in_dim;
    } else if (l == 0) {
      g.A1 = nullptr; g.lda1 = 0; g.K1 = 0;
      g.add_tab = w.in_g; g.add_ld = 2 * H; g.add_n = w.N; g.add_ids = x + t; g.add_ids_ld = L;
    } else {
      g.A1 = recs + ((size_t)(t & 1) * w.nl + l - 1) * cell + (size_t)n_sess * 4 * H; g.lda1 = H; g.K1 = H;
    }
    g.A2 = hold; g.lda2 = HT; g.K2 = H;
    g.Wt = w.wg[l]; g.bias = w.bg[l]; g.N = 2 * H;
    g.out = gates; g.ldo = 2 * H; g.out2 = rh; g.ldo2 = H; g.h = hold; g.ldh = HT;
    gc = g;
    gc.A2 = rh; gc.lda2 = H;
    gc.Wt = w.wc[l]; gc.bias = w.bc[l]; gc.N = H;
    if (l == 0 && !rows) { gc.add_tab = w.in_c; gc.add_ld = H; }
    gc.cand_out = rec + (size_t)n_sess * 3 * H; gc.ldc = H;
    gc.out = rec + (size_t)n_sess * 4 * H; gc.ldo = H;
    gc.out2 = s_new + (size_t)l * H; gc.ldo2 = HT;
    gc.gates = gates; gc.ldg = 2 * H;
    gc.seq_len = seq_len; gc.tstep = t;
    if (l == w.nl - 1) { gc.hid_out = hidden + (size_t)t * H; gc.ldho = (int64_t)L * H; }
  };
  return records_wavefront(w.nl, L, cell_args, st);
}

// The same cells with EVERY record kept, fp32 output rows and the masters' fp32 kernels as they are (gmm_train.hip: the training step of model_gru_gmm):
//   states [L + 1][M][HT] (slot 0 = the zero state), recs [L][nl][gates 2H | r*h H | cand H | out H per session], hidden32 [M * L][H] rows (session, t)
hipError_t launch_onehot_gru_records(const OneHotGru& w, const int32_t* x, int64_t n_sess, int L, float* hidden32, int32_t* seq_len, float* states, float* recs,
                                     hipStream_t st) {
  if (n_sess <= 0) return hipSuccess;
  const int H = w.H, HT = w.H * w.nl;
  const size_t cell = (size_t)n_sess * 5 * H;
  hipLaunchKernelGGL(k_count_ids, dim3((unsigned)((n_sess + 255) / 256)), dim3(256), 0, st, x, n_sess, L, w.N, seq_len);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if ((e = hipMemsetAsync(states, 0, (size_t)n_sess * HT * 4, st)) != hipSuccess) return e;   // zero_state
  // wavefront over (step, layer) as in launch_gru_scan: cell (t, l) needs (t, l - 1) and (t - 1, l) only, so the cells with t + l = tau go out as ONE grouped
  // launch per GEMM kind -- 2 (L + nl - 1) dependent launches instead of 2 L nl (the chain is bound by the device's dependent-launch latency)
  auto cell_args = [&](int t, int l, GemmArgs& g, GemmArgs& gc) {
    const float* s_old = states + (size_t)t * n_sess * HT;
    float* s_new = states + (size_t)(t + 1) * n_sess * HT;
    float* rec = recs + ((size_t)t * w.nl + l) * cell;
    memset(&g, 0, sizeof(g));
    const float* hold = s_old + (size_t)l * H;
    float* gates = rec;
    float* rh = rec + (size_t)n_sess * 2 * H;
    g.M = (int)n_sess; g.H = H;
    if (l == 0) {
      g.A1 = nullptr; g.lda1 = 0; g.K1 = 0;
      g.add_tab = w.in_g; g.add_ld = 2 * H; g.add_n = w.N; g.add_ids = x + t; g.add_ids_ld = L;
    } else {
      g.A1 = recs + ((size_t)t * w.nl + l - 1) * cell + (size_t)n_sess * 4 * H; g.lda1 = H; g.K1 = H;
    }
    g.A2 = hold; g.lda2 = HT; g.K2 = H;
    g.Wt = w.wg[l]; g.bias = w.bg[l]; g.N = 2 * H;
    g.out = gates; g.ldo = 2 * H; g.out2 = rh; g.ldo2 = H; g.h = hold; g.ldh = HT;
    gc = g;
    gc.A2 = rh; gc.lda2 = H;
    gc.Wt = w.wc[l]; gc.bias = w.bc[l]; gc.N = H;
    if (l == 0) { gc.add_tab = w.in_c; gc.add_ld = H; }
    gc.cand_out = rec + (size_t)n_sess * 3 * H; gc.ldc = H;
    gc.out = rec + (size_t)n_sess * 4 * H; gc.ldo = H;
    gc.out2 = s_new + (size_t)l * H; gc.ldo2 = HT;
    gc.gates = gates; gc.ldg = 2 * H;
    gc.seq_len = seq_len; gc.tstep = t;
    if (l == w.nl - 1) { gc.hid_out32 = hidden32 + (size_t)t * H; gc.ldho = (int64_t)L * H; }
  };
  return records_wavefront(w.nl, L, cell_args, st);
}

// ... and on DENSE fp32 rows of width in_dim (model_hier.py:68-69: model_gru_gmm as the low level reads the [e_t, state] rows; gmm_train.hip: the training step
// of that model): w.wg[0] [2H][in_dim + H] / w.wc[0] [H][in_dim + H] are the full kernels; sequence_length = the count of rows that are not all-zero
__global__ __launch_bounds__(256) void k_len_rows_f32(const float* __restrict__ rows, int64_t n_sess, int L, int W, int32_t* __restrict__ seq_len) {
  const int lane = threadIdx.x & 63;
  const int64_t sess = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sess >= n_sess) return;
  int cnt = 0;
  for (int t = 0; t < L; t++) {
    const float* r = rows + (sess * L + t) * W;
    int nz = 0;
    for (int c = lane; c < W; c += 64) nz |= (r[c] != 0.f) ? 1 : 0;
    nz = __any(nz);
    cnt += nz ? 1 : 0;
  }
  if (lane == 0) seq_len[sess] = cnt;
}
hipError_t launch_rows_gru_records(const OneHotGru& w, const float* rows, int in_dim, int64_t n_sess, int L, float* hidden32, int32_t* seq_len, float* states,
                                   float* recs, hipStream_t st) {
  if (n_sess <= 0) return hipSuccess;
  const int H = w.H, HT = w.H * w.nl;
  const size_t cell = (size_t)n_sess * 5 * H;
  hipLaunchKernelGGL(k_len_rows_f32, dim3((unsigned)((n_sess + 3) / 4)), dim3(256), 0, st, rows, n_sess, L, in_dim, seq_len);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if ((e = hipMemsetAsync(states, 0, (size_t)n_sess * HT * 4, st)) != hipSuccess) return e;   // zero_state
  auto cell_args = [&](int t, int l, GemmArgs& g, GemmArgs& gc) {   // (the wavefront of launch_onehot_gru_records)
    const float* s_old = states + (size_t)t * n_sess * HT;
    float* s_new = states + (size_t)(t + 1) * n_sess * HT;
    float* rec = recs + ((size_t)t * w.nl + l) * cell;
    memset(&g, 0, sizeof(g));
    const float* hold = s_old + (size_t)l * H;
    float* gates = rec;
    float* rh = rec + (size_t)n_sess * 2 * H;
    g.M = (int)n_sess; g.H = H;
    if (l == 0) { g.A1 = rows + (size_t)t * in_dim; g.lda1 = L * in_dim; g.K1 = in_dim; }
    else { g.A1 = recs + ((size_t)t * w.nl + l - 1) * cell + (size_t)n_sess * 4 * H; g.lda1 = H; g.K1 = H; }
    g.A2 = hold; g.lda2 = HT; g.K2 = H;
    g.Wt = w.wg[l]; g.bias = w.bg[l]; g.N = 2 * H;
    g.out = gates; g.ldo = 2 * H; g.out2 = rh; g.ldo2 = H; g.h = hold; g.ldh = HT;
    gc = g;
    gc.A2 = rh; gc.lda2 = H;
    gc.Wt = w.wc[l]; gc.bias = w.bc[l]; gc.N = H;
    gc.cand_out = rec + (size_t)n_sess * 3 * H; gc.ldc = H;
    gc.out = rec + (size_t)n_sess * 4 * H; gc.ldo = H;
    gc.out2 = s_new + (size_t)l * H; gc.ldo2 = HT;
    gc.gates = gates; gc.ldg = 2 * H;
    gc.seq_len = seq_len; gc.tstep = t;
    if (l == w.nl - 1) { gc.hid_out32 = hidden32 + (size_t)t * H; gc.ldho = (int64_t)L * H; }
  };
  return records_wavefront(w.nl, L, cell_args, st);
}
