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
// GRU scan.  One workgroup owns UB = 8 lanes (users) and walks their S slots; the carried state,
// the concatenated cell input [in | h] and the gate activations live in LDS.  Weights are stored
// TRANSPOSED ([out][in], set_weights) and stream from L2: a wave computes a block of 8 output
// columns x 8 users with its 64 lanes spread along the REDUCTION axis, so every weight load is a
// fully coalesced 1 KiB wave access and activations are conflict-free LDS reads; the 64 partial
// sums per lane are then folded across the wavefront with v_permlane32_swap and
// five xor-shuffle steps, leaving exactly one (user, column) total on each lane, where the
// gate non-linearity is applied.
// ---------------------------------------------------------------------------------------------
#define GRU_UB 8
#define GRU_THREADS 512
#define GRU_WAVES (GRU_THREADS / 64)

__device__ __forceinline__ float swap_add32(float a, float b) {
  // inline asm, not __builtin_amdgcn_permlane32_swap: hipcc (ROCm 7.2) mis-lowers `r[0] + r[1]` of the builtin's
  // two results to `r[0] + r[0]` once the SLP vectoriser packs neighbouring sums (scripts/fold_test2.hip).
  // The two v_nop are the VALU-write -> v_permlane-read wait states (cdna_hip_programming.md T21).
  asm volatile("v_nop\n\tv_nop\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return a + b;
}

// Fold 64 per-lane values across the 64 lanes: afterwards every lane holds the wave-wide total of exactly one
// of the 64 indices.  WHICH index a lane ends up with is a fixed permutation of the lane id that depends only on
// the exchange pattern; the kernel measures it once (fold_owner) instead of assuming it.
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
__device__ __forceinline__ int fold_owner(int lane) {
  float p[64];
#pragma unroll
  for (int i = 0; i < 64; i++) p[i] = (lane == 0) ? (float)(i + 1) : 0.f;  // only lane 0 contributes: total of index i is i+1
  return (int)fold64(p, lane) - 1;
}

// out[u][c] = sum_k x[u][k] * Wt[c][k] for u < 8, c < ncols (ncols % 8 == 0), K % 4 == 0, K <= 256*NK4.
// x: LDS, rows `xs` floats apart.  epi(u, c, total) runs on the lane that owns index u*8 + (c & 7) (`own`).
template <int NK4, typename Epi>
__device__ __forceinline__ void gru_gemm(const float* x, int xs, const float* __restrict__ Wt, int K, int ncols, int wave,
                                         int lane, int own, Epi epi) {
  const int nblk = ncols >> 3;
  for (int cb = wave; cb < nblk; cb += GRU_WAVES) {
    float p[64];
#pragma unroll
    for (int i = 0; i < 64; i++) p[i] = 0.f;
#pragma unroll
    for (int i = 0; i < NK4; i++) {
      const int k = 4 * (lane + 64 * i);
      if (k < K) {
        float4 xv[GRU_UB];
#pragma unroll
        for (int u = 0; u < GRU_UB; u++) xv[u] = *reinterpret_cast<const float4*>(x + u * xs + k);
#pragma unroll
        for (int c = 0; c < 8; c++) {
          const float4 w = *reinterpret_cast<const float4*>(Wt + (int64_t)(cb * 8 + c) * K + k);
#pragma unroll
          for (int u = 0; u < GRU_UB; u++) {
            float a = p[u * 8 + c];
            a = fmaf(xv[u].x, w.x, a);
            a = fmaf(xv[u].y, w.y, a);
            a = fmaf(xv[u].z, w.z, a);
            a = fmaf(xv[u].w, w.w, a);
            p[u * 8 + c] = a;
          }
        }
      }
    }
    const float total = fold64(p, lane);
    epi(own >> 3, cb * 8 + (own & 7), total);
  }
}

struct GruParams {
  const float* wgT[HIERTCN_MAX_LAYERS];  // [2H][in+H]
  const float* bg[HIERTCN_MAX_LAYERS];
  const float* wcT[HIERTCN_MAX_LAYERS];  // [H][in+H]
  const float* bc[HIERTCN_MAX_LAYERS];
  const float* projST;  // [E][HT]
  int E, H, nl, HT, Kmax;
};

template <int NK4>
__global__ __launch_bounds__(GRU_THREADS) void k_gru_scan(GruParams p, const float* __restrict__ pooled,
                                                          const float* __restrict__ mask, const float* __restrict__ state_in,
                                                          int B, int S, float* __restrict__ sproj, float* __restrict__ state_out,
                                                          float* __restrict__ state_before) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int E = p.E, H = p.H, HT = p.HT, Kmax = p.Kmax;
  float* st = lds;                      // [UB][HT]    carried state
  float* xcat = st + GRU_UB * HT;       // [UB][Kmax]  [in | h] then [in | r*h]
  float* gates = xcat + GRU_UB * Kmax;  // [UB][2H]    sigmoid(r), sigmoid(u)
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b0 = blockIdx.x * GRU_UB;
  const int own = fold_owner(lane);  // which (user, column-in-block) total this lane receives from fold64

  for (int i = tid; i < GRU_UB * HT; i += GRU_THREADS) {
    int u = i / HT, k = i % HT;
    st[i] = (b0 + u < B) ? state_in[(int64_t)(b0 + u) * HT + k] : 0.f;
  }
  __syncthreads();

  for (int s = 0; s < S; s++) {
    // the state this slot's low level sees, and its projection state . Wp[E:]  (model_hier.py:54-55 -> model_tcn.py:35)
    if (state_before) {
      for (int i = tid; i < GRU_UB * HT; i += GRU_THREADS) {
        int u = i / HT, k = i % HT;
        if (b0 + u < B) state_before[((int64_t)(b0 + u) * S + s) * HT + k] = st[i];
      }
    }
    gru_gemm<NK4>(st, HT, p.projST, HT, E, wave, lane, own, [&](int u, int c, float v) {
      if (b0 + u < B) sproj[((int64_t)(b0 + u) * S + s) * E + c] = v;
    });

    for (int l = 0; l < p.nl; l++) {
      const int in_dim = (l == 0) ? E : H;
      const int K = in_dim + H;
      float* hl = st + l * H;  // rows HT apart
      // xcat = [in | h_l]
      for (int i = tid; i < GRU_UB * K; i += GRU_THREADS) {
        const int u = i / K, k = i % K;
        float v;
        if (k < in_dim)
          v = (l == 0) ? ((b0 + u < B) ? pooled[((int64_t)(b0 + u) * S + s) * E + k] : 0.f) : st[u * HT + (l - 1) * H + k];
        else
          v = hl[u * HT + (k - in_dim)];
        xcat[u * Kmax + k] = v;
      }
      __syncthreads();
      // gates = sigmoid([in, h] . Wg + bg)      (customed_gru_cell.py:323)
      gru_gemm<NK4>(xcat, Kmax, p.wgT[l], K, 2 * H, wave, lane, own, [&](int u, int c, float v) {
        gates[u * 2 * H + c] = 1.0f / (1.0f + expf(-(v + p.bg[l][c])));
      });
      __syncthreads();
      // r_state = r * h                         (:324-326; r is the FIRST half), written over the h half of xcat
      for (int i = tid; i < GRU_UB * H; i += GRU_THREADS) {
        const int u = i / H, k = i % H;
        xcat[u * Kmax + in_dim + k] = gates[u * 2 * H + k] * hl[u * HT + k];
      }
      __syncthreads();
      // c = tanh([in, r*h] . Wc + bc) ; h' = u*h + (1-u)*c      (:335-336).  Only the owning lane touches
      // st[u][l*H + c] in this phase (the GEMM reads xcat), so h' is committed in place; it is layer l+1's input.
      gru_gemm<NK4>(xcat, Kmax, p.wcT[l], K, H, wave, lane, own, [&](int u, int c, float v) {
        const float cand = tanhf(v + p.bc[l][c]);
        const float ug = gates[u * 2 * H + H + c];
        const float hold = hl[u * HT + c];
        hl[u * HT + c] = ug * hold + (1.0f - ug) * cand;
      });
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
    p.wgT[l] = m.gru_wg[l]; p.bg[l] = m.gru_bg[l]; p.wcT[l] = m.gru_wc[l]; p.bc[l] = m.gru_bc[l];
  }
  p.projST = m.projS; p.E = m.E; p.H = m.H; p.nl = m.nl; p.HT = m.HT;
  int kmax = (m.E > m.H ? m.E : m.H) + m.H;
  if (m.HT > kmax) kmax = m.HT;
  p.Kmax = kmax;
  const int nk4 = (kmax / 4 + 63) / 64;
  size_t lds = sizeof(float) * GRU_UB * (size_t)(m.HT + kmax + 2 * m.H);
  unsigned blocks = (unsigned)((B + GRU_UB - 1) / GRU_UB);
#define GRU_LAUNCH(NK)                                                                                                         \
  do {                                                                                                                         \
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gru_scan<NK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
    if (e != hipSuccess) return e;                                                                                             \
    hipLaunchKernelGGL(k_gru_scan<NK>, dim3(blocks), dim3(GRU_THREADS), lds, st, p, pooled, mask, state_in, B, S, sproj, state_out, \
                       state_before);                                                                                          \
  } while (0)
  switch (nk4) {
    case 1: GRU_LAUNCH(1); break;
    case 2: GRU_LAUNCH(2); break;
    case 3: GRU_LAUNCH(3); break;
    case 4: GRU_LAUNCH(4); break;
    default: return hipErrorInvalidValue;
  }
#undef GRU_LAUNCH
  return hipGetLastError();
}
