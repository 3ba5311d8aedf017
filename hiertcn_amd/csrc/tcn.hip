// Low level of HierTCN, fused: item-embedding gather (model.py:59-61 + model_hier.py:50), input
// projection (model_hier.py:54-55 -> model_tcn.py:35) and the dilated causal conv stack
// (customized_tcn_cell.py:46-49,109-127,147-161; conv arithmetic
// customized_convolution_layer.py:171-198) for G sessions per workgroup, entirely in LDS.
//
// LDS image: per session a slab of (PAD + 32) rows x C bf16 channels; the first PAD rows are the
// zero left-padding of the causal conv (customized_tcn_cell.py:47-48) and are never written, the
// next 32 rows are the time steps (L <= 32).  Each layer is computed in place: MFMA accumulators
// hold the layer's output while every wave still reads the layer's input, then one barrier, then
// the bf16 write-back.  16-byte chunks of a row are XOR-swizzled by the row index so that the
// ds_read_b128 of 32 consecutive rows is bank-conflict free (cdna_hip_programming.md T2).
//
// MFMA orientation: A = weights [cout][cin] (internal layout, transposed at set_weights),
// B = activations^T, D[row = cout][col = time step].  A lane then owns ONE time step and four
// runs of 4 consecutive output channels -> 8-byte packed bf16 stores and residual loads.
#include <stdlib.h>
#include <string.h>

#include "common.h"

#define TCN_G 4          // sessions per workgroup
#define TCN_THREADS 512  // 8 waves

template <int C, int GSO = 0>   // GSO: sessions per workgroup of the small-grid instances (0: the default below)
struct TcnGeom {
  static constexpr int GS = GSO ? GSO : (C == 512) ? 2 : TCN_G;   // sessions per workgroup: the LDS slabs of four 512-channel sessions would take all 160 KB
  static constexpr int NCB = C / 32;        // output-channel blocks
  static constexpr int CBW = NCB > 8 ? NCB / 8 : 1;   // channel blocks per wave (C = 512: two)
  static constexpr int WPC = NCB >= 8 ? 1 : 8 / NCB;  // waves sharing one channel block
  static constexpr int RPW = GS / WPC;      // sessions (row blocks) per wave
  static constexpr int KS = C / 16;         // MFMA k-steps
  static constexpr int CH = C / 8;          // 16-byte chunks per row
  static constexpr int SWZ = (CH < 16 ? CH : 16) - 1;
  static_assert(NCB >= 1 && NCB <= 16 && (NCB <= 8 ? (8 % NCB) == 0 : (NCB % 8) == 0) && (GS % WPC) == 0, "unsupported channel count");
};

struct TcnParams {
  const uint16_t* emb;                         // [N][C] bf16, row 0 zero
  const uint16_t* projT;                       // [C][C] bf16 (cout-major) or null (single level: no projection)
  const uint16_t* convT[HIERTCN_MAX_BLOCKS];   // [K][C][C]
  const uint16_t* projF;                       // fragment-major copies ([channel block][tap][k-step][lane][8], tcn_layers.hip), FRAG kernels
  const uint16_t* convF[HIERTCN_MAX_BLOCKS];
  const float* conv_bias[HIERTCN_MAX_BLOCKS];  // [C]
  int n_blocks, K, pad;
  int N;                                       // item_num: ids outside [0, N) embed like the null id
  // training (optional, all [n_sessions * L][C]): what the backward pass needs -- the embedded rows, every layer's output
  // (save_a[0] = projection output or, single level, unused; save_a[k + 1] = output of block k) and relu(conv + b) of every block
  uint16_t* save_emb;
  uint16_t* save_a[HIERTCN_MAX_BLOCKS + 1];
  uint16_t* save_c[HIERTCN_MAX_BLOCKS];
};

template <int C>
__device__ __forceinline__ int lds_off(int row, int chunk) {
  // byte offset of 16-byte chunk `chunk` of slab row `row`
  return row * (2 * C) + ((chunk ^ (row & TcnGeom<C>::SWZ)) << 4);
}

// DEEP: all of a tap's weight fragments in flight at once (156 VGPRs at C = 256: one workgroup per CU) -- for grids that do not fill the
// chip anyway (the reference's 32-lane batches: 80 workgroups), where the kernel is bound by the L2 round trips of its weight loads;
// otherwise four fragments at a time and two workgroups per CU (cfg-3: 3.1 ms per 81,920 sessions against 5.3 ms with DEEP)
// FRAG (round 3; one channel block per wave): the wave's weight stream of a layer -- taps x k-steps fragments -- is read from the
// fragment-major copy (one contiguous, fully coalesced 1 KB per fragment instead of 32 rows x 32 bytes) in batches of eight through two
// register sets, the loads of batch g + 1 issued in one run before the MFMAs of batch g (the recipe of k_tcn_layer_lds); the accumulation
// order per output is unchanged, so the three low-level paths stay bitwise identical.
template <int C, bool DEEP = false, bool FRAG = false, int GSO = 0>
__global__ __launch_bounds__(TCN_THREADS) void k_tcn_fused(TcnParams p, const int32_t* __restrict__ x,
                                                           const float* __restrict__ sproj, int64_t n_sessions, int L,
                                                           uint16_t* __restrict__ hidden) {
  using G = TcnGeom<C, GSO>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int ROWS = p.pad + 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t sess0 = (int64_t)blockIdx.x * G::GS;

  // ---- phase 0: zero halo rows, gather embedding rows (16 B per lane) -------------------------
  for (int i = tid; i < G::GS * ROWS * G::CH; i += TCN_THREADS) {
    const int row = i / G::CH, ch = i % G::CH;
    const int g = row / ROWS, r = row % ROWS;
    uint4 v = make_uint4(0, 0, 0, 0);
    const int t = r - p.pad;
    const int64_t sess = sess0 + g;
    if (t >= 0 && t < L && sess < n_sessions) {
      int id = x[sess * L + t];
      id = (unsigned)id < (unsigned)p.N ? id : 0;   // out of range = null id: tf.one_hot gives a zero row (model.py:59)
      v = *reinterpret_cast<const uint4*>(p.emb + (int64_t)id * C + ch * 8);  // row 0 is stored as zeros
      if (p.save_emb) *reinterpret_cast<uint4*>(p.save_emb + (sess * L + t) * C + ch * 8) = v;
    }
    *reinterpret_cast<uint4*>(smem + lds_off<C>(row, ch)) = v;
  }
  __syncthreads();

  const int cb0 = (wave / G::WPC) * G::CBW;      // first output-channel block of this wave (C = 512: two blocks per wave)
  const int g0 = (wave % G::WPC) * G::RPW;      // first session (row block) of this wave

  // One layer = (taps x k-steps) MFMAs per row block; layer < 0 is the input projection.
  const int n_layers = p.n_blocks + (p.projT ? 1 : 0);
  for (int li = 0; li < n_layers; li++) {
    const bool is_proj = (p.projT != nullptr) && (li == 0);
    const int blk = li - (p.projT ? 1 : 0);
    const int dil = is_proj ? 0 : (1 << blk);
    const int taps = is_proj ? 1 : p.K;
    const uint16_t* W = is_proj ? p.projT : p.convT[blk];

    f32x16 acc[G::CBW][G::RPW];
#pragma unroll
    for (int cbi = 0; cbi < G::CBW; cbi++)
#pragma unroll
    for (int r = 0; r < G::RPW; r++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        f32x4 init = {0.f, 0.f, 0.f, 0.f};
        const int c0 = (cb0 + cbi) * 32 + 8 * q + 4 * lh;  // 4 consecutive output channels of this register quad
        if (is_proj) {
          const int64_t sess = sess0 + g0 + r;
          if (sproj != nullptr && sess < n_sessions) init = *reinterpret_cast<const f32x4*>(sproj + sess * C + c0);
        } else {
          init = *reinterpret_cast<const f32x4*>(p.conv_bias[blk] + c0);
        }
#pragma unroll
        for (int j = 0; j < 4; j++) acc[cbi][r][4 * q + j] = init[j];
      }
    }

    if constexpr (FRAG) {
      static_assert(G::CBW == 1, "the fragment-major stream is written for one channel block per wave");
      const int n_steps = taps * G::KS, n_full = n_steps & ~7;
      const uint16_t* wp = (is_proj ? p.projF : p.convF[blk]) + (int64_t)cb0 * n_steps * 512 + lane * 8;
      int s = 0;
      int shift = is_proj ? 0 : (p.K - 1) * dil;              // tap 0 reaches furthest back: o[t] += x[t - (K-1-j)*dil] . W[j]
      auto step = [&](const uint4 av) {
#pragma unroll
        for (int r = 0; r < G::RPW; r++) {
          const int row = (g0 + r) * ROWS + p.pad + lr - shift;
          const uint4 b = *reinterpret_cast<const uint4*>(smem + lds_off<C>(row, 2 * s + lh));
          acc[0][r] = mfma32(av, b, acc[0][r]);
        }
        s++;
        const bool wrap = s == G::KS;
        s = wrap ? 0 : s;
        shift -= wrap ? dil : 0;
      };
      auto load8 = [&](uint4* w, int g0s) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
          const int idx = g0s + u < n_steps ? g0s + u : n_steps - 1;
          w[u] = *reinterpret_cast<const uint4*>(wp + (int64_t)idx * 512);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      auto comp8 = [&](const uint4* w) {
#pragma unroll
        for (int u = 0; u < 8; u++) step(w[u]);
      };
      auto tail = [&](const uint4* w) {
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (n_full + u < n_steps) step(w[u]);
      };
      uint4 wa[8], wb[8];
      load8(wa, 0);
      int i0 = 0;
      for (; i0 + 16 <= n_full; i0 += 16) {
        load8(wb, i0 + 8);
        comp8(wa);
        load8(wa, i0 + 16);
        comp8(wb);
      }
      if (i0 < n_full) {
        load8(wb, i0 + 8);
        comp8(wa);
        tail(wb);
      } else {
        tail(wa);
      }
    } else
    for (int j = 0; j < taps; j++) {
      const int shift = is_proj ? 0 : (p.K - 1 - j) * dil;  // o[t] += x[t - (K-1-j)*dil] . W[j]
      const uint16_t* Wj = W + (int64_t)j * C * C + (int64_t)(cb0 * 32 + lr) * C + 8 * lh;   // (+ 32 C per further channel block of the wave)
      if constexpr (DEEP) {
        static_assert(G::CBW == 1, "the deep prefetch is written for one channel block per wave");
        uint4 af[G::KS];
#pragma unroll
        for (int s = 0; s < G::KS; s++) af[s] = *reinterpret_cast<const uint4*>(Wj + 16 * s);
#pragma unroll
        for (int s = 0; s < G::KS; s++) {
#pragma unroll
          for (int r = 0; r < G::RPW; r++) {
            const int row = (g0 + r) * ROWS + p.pad + lr - shift;
            const uint4 b = *reinterpret_cast<const uint4*>(smem + lds_off<C>(row, 2 * s + lh));
            acc[0][r] = mfma32(af[s], b, acc[0][r]);
          }
        }
      } else {
#pragma unroll 4
        for (int s = 0; s < G::KS; s++) {
          uint4 a[G::CBW];
#pragma unroll
          for (int cbi = 0; cbi < G::CBW; cbi++) a[cbi] = *reinterpret_cast<const uint4*>(Wj + (int64_t)cbi * 32 * C + 16 * s);
#pragma unroll
          for (int r = 0; r < G::RPW; r++) {
            const int row = (g0 + r) * ROWS + p.pad + lr - shift;
            const uint4 b = *reinterpret_cast<const uint4*>(smem + lds_off<C>(row, 2 * s + lh));   // one LDS read feeds every channel block of the wave
#pragma unroll
            for (int cbi = 0; cbi < G::CBW; cbi++) acc[cbi][r] = mfma32(a[cbi], b, acc[cbi][r]);
          }
        }
      }
    }

    // epilogue in registers: projection -> identity; block -> relu(relu(conv+b) + x)
    uint2 outv[G::CBW][G::RPW][4];
#pragma unroll
    for (int cbi = 0; cbi < G::CBW; cbi++)
#pragma unroll
    for (int r = 0; r < G::RPW; r++) {
      const int cb = cb0 + cbi;
      const int row = (g0 + r) * ROWS + p.pad + lr;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = acc[cbi][r][4 * q + j];
        if (!is_proj) {
          if (p.save_c[blk]) {   // training: relu(conv + b) before the residual add (its sign pattern gates the backward pass)
            const int64_t sess = sess0 + g0 + r;
            if (lr < L && sess < n_sessions) {
              uint2 co;
              co.x = pack2bf(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
              co.y = pack2bf(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
              *reinterpret_cast<uint2*>(p.save_c[blk] + (sess * L + lr) * C + cb * 32 + 8 * q + 4 * lh) = co;
            }
          }
          const uint2 res = *reinterpret_cast<const uint2*>(smem + lds_off<C>(row, cb * 4 + q) + 8 * lh);
          const float r0 = bf2f((uint16_t)(res.x & 0xffff)), r1 = bf2f((uint16_t)(res.x >> 16));
          const float r2 = bf2f((uint16_t)(res.y & 0xffff)), r3 = bf2f((uint16_t)(res.y >> 16));
          v[0] = fmaxf(fmaxf(v[0], 0.f) + r0, 0.f);
          v[1] = fmaxf(fmaxf(v[1], 0.f) + r1, 0.f);
          v[2] = fmaxf(fmaxf(v[2], 0.f) + r2, 0.f);
          v[3] = fmaxf(fmaxf(v[3], 0.f) + r3, 0.f);
        }
        outv[cbi][r][q].x = pack2bf(v[0], v[1]);
        outv[cbi][r][q].y = pack2bf(v[2], v[3]);
      }
    }
    __syncthreads();  // every wave has finished reading this layer's input
#pragma unroll
    for (int cbi = 0; cbi < G::CBW; cbi++)
#pragma unroll
    for (int r = 0; r < G::RPW; r++) {
      const int row = (g0 + r) * ROWS + p.pad + lr;
#pragma unroll
      for (int q = 0; q < 4; q++)
        *reinterpret_cast<uint2*>(smem + lds_off<C>(row, (cb0 + cbi) * 4 + q) + 8 * lh) = outv[cbi][r][q];
    }
    __syncthreads();
    uint16_t* sv = p.save_a[is_proj ? 0 : blk + 1];
    if (sv) {   // training: this layer's output, 16 B per lane (the next layer's first barrier also covers these reads)
      for (int i = tid; i < G::GS * 32 * G::CH; i += TCN_THREADS) {
        const int ch = i % G::CH, rt = i / G::CH;
        const int g = rt >> 5, t = rt & 31;
        const int64_t sess = sess0 + g;
        if (t < L && sess < n_sessions)
          *reinterpret_cast<uint4*>(sv + (sess * L + t) * C + ch * 8) = *reinterpret_cast<const uint4*>(smem + lds_off<C>(g * ROWS + p.pad + t, ch));
      }
    }
  }

  // ---- write the valid rows out, 16 B per lane, rows compacted to [session][t < L] -----------
  for (int i = tid; i < G::GS * 32 * G::CH; i += TCN_THREADS) {
    const int ch = i % G::CH, rt = i / G::CH;
    const int g = rt >> 5, t = rt & 31;
    const int64_t sess = sess0 + g;
    if (t < L && sess < n_sessions) {
      const uint4 v = *reinterpret_cast<const uint4*>(smem + lds_off<C>(g * ROWS + p.pad + t, ch));
      *reinterpret_cast<uint4*>(hidden + (sess * L + t) * C + ch * 8) = v;
    }
  }
}

template <int C>
static hipError_t launch_tcn_c(const TcnParams& p, const int32_t* x, const float* sproj, int64_t n_sessions, int L,
                               uint16_t* hidden, hipStream_t st, int64_t gs2_max) {
  constexpr int GS = TcnGeom<C>::GS;
  size_t lds = (size_t)GS * (p.pad + 32) * 2 * C;
  unsigned blocks = (unsigned)((n_sessions + GS - 1) / GS);
  const bool frag_ok = C <= 256 && p.convF[0] && (!p.projT || p.projF) && !ABL_ENV("HIERTCN_TCN_FUSED_V1");
  if constexpr (C == 128 || C == 256) {
    // Small grids (round 4; the reference's 32-lane batch = 320 sessions = 80 workgroups of four): every workgroup streams ALL the weights
    // (1.4 MB at C = 256) through its CU whatever it computes, so two sessions per workgroup put twice the CUs on the same stream:
    // 56 -> 44 us with the fragment-major stream instead of DEEP's row reads, -> 27 us with two sessions per workgroup (one: 37 us, 320
    // workgroups are a round and a quarter).  Same accumulation order per output: bitwise identical to the other instances.
    // (gs2_max: tune.tcn_gs2_max_sessions, default 512)
    if (frag_ok && n_sessions <= gs2_max) {
      const size_t l2 = (size_t)2 * (p.pad + 32) * 2 * C;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tcn_fused<C, false, true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((k_tcn_fused<C, false, true, 2>), dim3((unsigned)((n_sessions + 1) / 2)), dim3(TCN_THREADS), l2, st, p, x, sproj, n_sessions, L, hidden);
      return hipGetLastError();
    }
  }
  if (C <= 256 && blocks <= 384 && !p.save_emb && !frag_ok) {   // no fragment-major tables: all of a tap's row reads in flight (DEEP)
    constexpr bool DEEP_OK = C <= 256;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tcn_fused<C, DEEP_OK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_tcn_fused<C, DEEP_OK>), dim3(blocks), dim3(TCN_THREADS), lds, st, p, x, sproj, n_sessions, L, hidden);
    return hipGetLastError();
  }
  if constexpr (C <= 256) {
    if (frag_ok) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tcn_fused<C, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL((k_tcn_fused<C, false, true>), dim3(blocks), dim3(TCN_THREADS), lds, st, p, x, sproj, n_sessions, L, hidden);
      return hipGetLastError();
    }
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tcn_fused<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_tcn_fused<C, false>), dim3(blocks), dim3(TCN_THREADS), lds, st, p, x, sproj, n_sessions, L, hidden);
  return hipGetLastError();
}

#include <stdlib.h>
#include <string.h>
static bool use_fused(const ModelDev& m, int L) {
  if (m.tune.tcn_layered == 1) return false;   // (tests: both paths must agree bitwise)
  const int halo = (m.K - 1) * (1 << (m.n_blocks - 1));
  const size_t lds = (size_t)(m.E == 512 ? 2 : TCN_G) * (((halo + 7) & ~7) + 32) * 2 * m.E;   // TcnGeom<C>::GS sessions per workgroup
  return m.fused_tcn && L <= 32 && lds <= 160 * 1024;
}

size_t tcn_workspace_bytes(const ModelDev& m, int64_t n_sessions, int L) {
  if (m.low_gru) return low_gru_workspace_bytes(m, n_sessions, L);
  return use_fused(m, L) ? 0 : tcn_layers_workspace_bytes(m, n_sessions, L);
}

bool tcn_fused_applies(const ModelDev& m, int L) { return !m.low_gru && use_fused(m, L); }

hipError_t launch_tcn_forward(const ModelDev& m, const int32_t* x, const float* sproj, int64_t n_sessions, int L,
                              uint16_t* hidden, void* workspace, hipStream_t st, const TcnSaves* sv) {
  if (n_sessions <= 0) return hipSuccess;
  if (m.low_gru) return launch_low_gru(m, x, sproj, n_sessions, L, hidden, workspace, st);
  if (!use_fused(m, L)) return sv ? hipErrorInvalidValue : launch_tcn_layers(m, x, sproj, n_sessions, L, hidden, workspace, st);
  TcnParams p;
  memset(&p, 0, sizeof(p));
  if (sv) {
    p.save_emb = sv->emb;
    for (int i = 0; i <= HIERTCN_MAX_BLOCKS; i++) p.save_a[i] = sv->a[i];
    for (int i = 0; i < HIERTCN_MAX_BLOCKS; i++) p.save_c[i] = sv->c[i];
  }
  p.emb = m.emb;
  p.projT = m.hier ? m.projT : nullptr;
  for (int i = 0; i < HIERTCN_MAX_BLOCKS; i++) { p.convT[i] = m.convT[i]; p.conv_bias[i] = m.conv_bias[i]; p.convF[i] = m.convF[i]; }
  p.projF = m.hier ? m.projF : nullptr;
  p.n_blocks = m.n_blocks; p.K = m.K; p.N = m.N;
  int halo = (m.K - 1) * (1 << (m.n_blocks - 1));
  p.pad = (halo + 7) & ~7;
  switch (m.E) {
    case 64: return launch_tcn_c<64>(p, x, sproj, n_sessions, L, hidden, st, m.tune.tcn_gs2_max_sessions);
    case 128: return launch_tcn_c<128>(p, x, sproj, n_sessions, L, hidden, st, m.tune.tcn_gs2_max_sessions);
    case 256: return launch_tcn_c<256>(p, x, sproj, n_sessions, L, hidden, st, m.tune.tcn_gs2_max_sessions);
    case 512: return launch_tcn_c<512>(p, x, sproj, n_sessions, L, hidden, st, m.tune.tcn_gs2_max_sessions);
    default: return hipErrorInvalidValue;
  }
}
