// High level of HierTCN: session mean-pool of target embeddings (model_hier.py:83-85) and the
// across-session GRU scan (model_hier.py:36-37,91,93; arithmetic customed_gru_cell.py:309-337,
// 1050-1073).  The scan also emits the state half of the low level's input projection
// (model_hier.py:54-55 -> model_tcn.py:35), so the TCN kernel never touches the state.
//
// Both kernels are fp32 end to end (the recurrence is the only true sequential dependency of
// the path and < 0.1 % of its FLOPs; rounding the carried state would compound over sessions).
#include "common.h"

// ---------------------------------------------------------------------------------------------
// session mean-pool: one wave per session.  HBM/L2-bound row gather: each 16-byte lane load
// covers 8 bf16 channels; E/8 lanes cover a row, 64/(E/8) rows are in flight per instruction.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_session_pool(const uint16_t* __restrict__ emb, const float* __restrict__ emb_bias,
                                                      const int32_t* __restrict__ y, int64_t n_sessions, int L, int E,
                                                      float* __restrict__ pooled) {
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
    if (id > 0) {
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
  hipLaunchKernelGGL(k_session_pool, dim3(blocks), dim3(256), 0, st, m.emb, m.emb_bias, y, n_sessions, L, m.E, pooled);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// GRU scan.  One workgroup owns UB lanes (users) and walks their S slots; the carried state, the
// slot input and the gate activations live in LDS; weights stream from L2 (every workgroup reads
// the same ~3.5 MB).  Thread j owns output column j of each small GEMM, so weight reads are
// coalesced along the output dimension and activations are LDS broadcasts.
// ---------------------------------------------------------------------------------------------
#define GRU_UB 8
#define GRU_THREADS 256

// acc[u] += sum_k x[u][k] * W[k][col]   for k in [0,K), x rows `xs` floats apart in LDS.
__device__ __forceinline__ void gemv_cols(float (&acc)[GRU_UB], const float* __restrict__ x, int xs, const float* __restrict__ W,
                                          int ldw, int col, int K) {
  for (int k = 0; k < K; k += 4) {
    const float w0 = W[(int64_t)(k + 0) * ldw + col];
    const float w1 = W[(int64_t)(k + 1) * ldw + col];
    const float w2 = W[(int64_t)(k + 2) * ldw + col];
    const float w3 = W[(int64_t)(k + 3) * ldw + col];
#pragma unroll
    for (int u = 0; u < GRU_UB; u++) {
      const float4 xv = *reinterpret_cast<const float4*>(x + u * xs + k);
      acc[u] = fmaf(xv.x, w0, acc[u]);
      acc[u] = fmaf(xv.y, w1, acc[u]);
      acc[u] = fmaf(xv.z, w2, acc[u]);
      acc[u] = fmaf(xv.w, w3, acc[u]);
    }
  }
}

struct GruParams {
  const float* wg[HIERTCN_MAX_LAYERS];
  const float* bg[HIERTCN_MAX_LAYERS];
  const float* wc[HIERTCN_MAX_LAYERS];
  const float* bc[HIERTCN_MAX_LAYERS];
  const float* projS;  // [HT][E]
  int E, H, nl, HT;
};

__global__ __launch_bounds__(GRU_THREADS) void k_gru_scan(GruParams p, const float* __restrict__ pooled,
                                                          const float* __restrict__ mask, const float* __restrict__ state_in,
                                                          int B, int S, float* __restrict__ sproj, float* __restrict__ state_out,
                                                          float* __restrict__ state_before) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int E = p.E, H = p.H, HT = p.HT;
  float* st = lds;                      // [UB][HT]   carried state
  float* xin = st + GRU_UB * HT;        // [UB][E]    slot input (pooled embedding)
  float* gates = xin + GRU_UB * E;      // [UB][2H]   sigmoid(r), sigmoid(u)
  float* rh = gates + GRU_UB * 2 * H;   // [UB][H]    r * h
  const int tid = threadIdx.x;
  const int b0 = blockIdx.x * GRU_UB;

  for (int i = tid; i < GRU_UB * HT; i += GRU_THREADS) {
    int u = i / HT, k = i % HT;
    st[i] = (b0 + u < B) ? state_in[(int64_t)(b0 + u) * HT + k] : 0.f;
  }
  __syncthreads();

  for (int s = 0; s < S; s++) {
    // slot input
    for (int i = tid; i < GRU_UB * E; i += GRU_THREADS) {
      int u = i / E, k = i % E;
      xin[i] = (b0 + u < B) ? pooled[((int64_t)(b0 + u) * S + s) * E + k] : 0.f;
    }
    // the state this slot's low level sees, and its projection state . Wp[E:]
    if (state_before) {
      for (int i = tid; i < GRU_UB * HT; i += GRU_THREADS) {
        int u = i / HT, k = i % HT;
        if (b0 + u < B) state_before[((int64_t)(b0 + u) * S + s) * HT + k] = st[i];
      }
    }
    for (int col = tid; col < E; col += GRU_THREADS) {
      float acc[GRU_UB];
#pragma unroll
      for (int u = 0; u < GRU_UB; u++) acc[u] = 0.f;
      gemv_cols(acc, st, HT, p.projS, E, col, HT);
#pragma unroll
      for (int u = 0; u < GRU_UB; u++)
        if (b0 + u < B) sproj[((int64_t)(b0 + u) * S + s) * E + col] = acc[u];
    }
    __syncthreads();

    for (int l = 0; l < p.nl; l++) {
      const float* in = (l == 0) ? xin : (st + (l - 1) * H);
      const int in_s = (l == 0) ? E : HT;
      const int in_dim = (l == 0) ? E : H;
      float* hl = st + l * H;  // rows HT apart
      // gates = sigmoid([in, h] . Wg + bg)      (customed_gru_cell.py:323)
      for (int col = tid; col < 2 * H; col += GRU_THREADS) {
        float acc[GRU_UB];
        const float b = p.bg[l][col];
#pragma unroll
        for (int u = 0; u < GRU_UB; u++) acc[u] = b;
        gemv_cols(acc, in, in_s, p.wg[l], 2 * H, col, in_dim);
        gemv_cols(acc, hl, HT, p.wg[l] + (int64_t)in_dim * 2 * H, 2 * H, col, H);
#pragma unroll
        for (int u = 0; u < GRU_UB; u++) gates[u * 2 * H + col] = 1.0f / (1.0f + expf(-acc[u]));
      }
      __syncthreads();
      // r_state = r * h                         (:324-326; r is the FIRST half)
      for (int i = tid; i < GRU_UB * H; i += GRU_THREADS) {
        int u = i / H, k = i % H;
        rh[i] = gates[u * 2 * H + k] * hl[u * HT + k];
      }
      __syncthreads();
      // c = tanh([in, r*h] . Wc + bc) ; h' = u*h + (1-u)*c      (:335-336)
      for (int col = tid; col < H; col += GRU_THREADS) {
        float acc[GRU_UB];
        const float b = p.bc[l][col];
#pragma unroll
        for (int u = 0; u < GRU_UB; u++) acc[u] = b;
        gemv_cols(acc, in, in_s, p.wc[l], H, col, in_dim);
        gemv_cols(acc, rh, H, p.wc[l] + (int64_t)in_dim * H, H, col, H);
#pragma unroll
        for (int u = 0; u < GRU_UB; u++) {
          const float c = tanhf(acc[u]);
          const float ug = gates[u * 2 * H + H + col];
          const float hold = hl[u * HT + col];
          acc[u] = ug * hold + (1.0f - ug) * c;
        }
        // nobody else reads st[., l*H + col] in this phase (the GEMMs above read `in` and `rh`),
        // so the new h can be committed in place; it is layer l+1's input (:1069).
#pragma unroll
        for (int u = 0; u < GRU_UB; u++) hl[u * HT + col] = acc[u];
      }
      __syncthreads();
    }
    // state *= mask[i]                          (model_hier.py:93)
    for (int i = tid; i < GRU_UB * HT; i += GRU_THREADS) {
      int u = i / HT;
      const float mk = (b0 + u < B) ? mask[(int64_t)(b0 + u) * S + s] : 0.f;
      st[i] *= mk;
    }
    __syncthreads();
  }
  for (int i = tid; i < GRU_UB * HT; i += GRU_THREADS) {
    int u = i / HT, k = i % HT;
    if (b0 + u < B) state_out[(int64_t)(b0 + u) * HT + k] = st[i];
  }
}

hipError_t launch_gru_scan(const ModelDev& m, const float* pooled, const float* mask, const float* state_in, int B, int S,
                           float* sproj, float* state_out, float* state_before, hipStream_t st) {
  if (B <= 0) return hipSuccess;
  GruParams p;
  for (int l = 0; l < HIERTCN_MAX_LAYERS; l++) {
    p.wg[l] = m.gru_wg[l]; p.bg[l] = m.gru_bg[l]; p.wc[l] = m.gru_wc[l]; p.bc[l] = m.gru_bc[l];
  }
  p.projS = m.projS; p.E = m.E; p.H = m.H; p.nl = m.nl; p.HT = m.HT;
  size_t lds = sizeof(float) * GRU_UB * (size_t)(m.HT + m.E + 2 * m.H + m.H);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gru_scan), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  unsigned blocks = (unsigned)((B + GRU_UB - 1) / GRU_UB);
  hipLaunchKernelGGL(k_gru_scan, dim3(blocks), dim3(GRU_THREADS), lds, st, p, pooled, mask, state_in, B, S, sproj, state_out,
                     state_before);
  return hipGetLastError();
}
