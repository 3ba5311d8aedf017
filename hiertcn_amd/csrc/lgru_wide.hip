// HierGRU low level of small batches at H = 256 (round 4; model_gru.py:47-55, customed_gru_cell.py:309-337): register-resident weights as in
// lgru_pipe.hip, with a layer's units AND its reduction split so that they fit.
//
// At H = 256 a layer's weights are 6 H^2 x 4 bytes (hi + lo) = 1.5 MB: lgru_tile.hip streams them from the L2 for every cell (34 us per cell,
// 10 workgroups of 40 sequential cells per 320-session batch = 0.85 ms), and one CU's register file holds a quarter of a megabyte.  Here a
// (32-session tile, layer) is EIGHT workgroups: workgroup p owns unit block p (32 units: its r, z and candidate tiles, 196 KB of fragments), and its
// four waves split the REDUCTION -- wave w keeps k-steps [8 w, 8 w + 8) of the three tiles, 192 registers, loaded once -- so a cell is 72
// MFMAs per wave, a reduction of the four partial tiles through LDS, and an epilogue on four units per lane.  What crosses workgroups per cell
// (inside the launch, drained-`sc1` hand-offs as in lgru_pipe.hip, one arrival counter per (layer, tile, step, phase)):
//   phase 1: r * h of the own 32 units (split 16-bit rows) -> the layer's 8 workgroups (the candidate's recurrent half needs all 256);
//   phase 2: the new state of the own 32 units -> the layer's 8 workgroups (the next step's recurrent half) and the 8 of the layer above (its input).
// Layers run pipelined (layer l + 1 of step t beside layer l of step t + 1).  At most 256 workgroups, all resident (tiles x layers x 8 <= CUs);
// every wait is bounded and a time-out sets the handle's persistent-launch word.  Same split-operand arithmetic as the other HierGRU kernels
// (a.w = a_hi w_hi + a_lo w_hi + a_hi w_lo, fp32 accumulation); the sum over k is formed as four partial sums, so results equal lgru_tile.hip's up
// to fp32 summation order (a storage-type ulp in a handful of outputs).
#include <stdlib.h>
#include <string.h>

#include "common.h"

#define LW_THREADS 256
#define LW_PARTS 8
#define LW_SPIN_MAX (1u << 22)
// Batches of more tiles than one launch can keep resident run as several launches over consecutive tile ranges (stream order: each has the chip
// to itself) -- at H = 256 up to three (measured, training step of 32 x 20 x 30 / 64 / 128 / 200 lanes x 10 x 20: 4.0 / 2.6 / 5.4 / 7.5 ms against
// 5.7 / 3.8 / 6.1 / 7.5 ms with the session-tile kernels of lgru_tile.hip, which keep every CU busy with a tile of its own); at H = 128 one (two
// launches of the backward cells at 1,280 sessions: 3.0 against 2.9 ms).
static int lw_max_launches(const ModelDev& m, int parts) { return parts == 8 ? m.tune.lgru_wide_max_launches : 1; }
static int64_t lw_tiles_per_launch(const ModelDev& m, int parts) {
  const int64_t cus = m.n_cu > 0 ? m.n_cu : 256;
  return cus / ((int64_t)m.nl * parts);
}

struct LgruWideArgs {
  const uint16_t* u16;                       // [M][L][E] projected [e, state] rows, storage type
  const int32_t* seq_len;                    // [M]
  const uint16_t* wgF_hi[HIERTCN_MAX_LAYERS]; const uint16_t* wgF_lo[HIERTCN_MAX_LAYERS];   // [2H/32][K/16][64][8]
  const uint16_t* wcF_hi[HIERTCN_MAX_LAYERS]; const uint16_t* wcF_lo[HIERTCN_MAX_LAYERS];   // [H/32][K/16][64][8]
  const float* bg[HIERTCN_MAX_LAYERS]; const float* bc[HIERTCN_MAX_LAYERS];
  uint16_t* hidden;                          // [M * L][H]: the top layer's outputs, zero past the length
  float* states;                             // training: [(L + 1)][M][HT]; null in evaluation
  float* rec;                                // training: per (t, l): gates [M][2H] | r*h [M][H] | cand [M][H] | out [M][H]
  uint16_t* hx;                              // [nl][L][2 (hi, lo)][M][H]: the state rows of (layer, step)
  uint16_t* rx;                              // [nl][L][2][M][H]: r * h rows
  int* cnt;                                  // [nl][tiles][L][2]: workgroups that have published phase 1 / phase 2 of (layer, tile, step)
  unsigned long long* err;
  int M, L, HT, nl;
  int tile0, tiles;                          // this launch's first tile; tiles of the whole batch (the counters' stride)
};

__device__ __forceinline__ float lw_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float lw_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
__device__ __forceinline__ void lw_split(float v, uint16_t& hi, uint16_t& lo) {
  hi = f2bf(v);
  lo = f2bf(v - bf2f(hi));
}

__global__ __launch_bounds__(LW_THREADS, 1) void k_lgru_wide(const LgruWideArgs a) {
  constexpr int H = 256, HB = 8, KS = 32, KSI = 16, KW = KS / 4;   // k-steps per wave and tile
  constexpr int PITCH = 2 * H + 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_dead;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nl = a.nl, L = a.L, M = a.M, HT = a.HT;
  const int tile = a.tile0 + (int)blockIdx.x, p = (int)blockIdx.y, l = (int)blockIdx.z, tiles = a.tiles;   // (a launch covers tiles [tile0, tile0 + gridDim.x))
  const int64_t sess0 = (int64_t)tile * 32;
  const int64_t sess = sess0 + lr;
  const bool live = sess < M;
  const int64_t sc = live ? sess : (int64_t)M - 1;
  const int len = a.seq_len[sc];
  const bool first = l == 0, last = l == nl - 1;

  unsigned char* XH = smem;                        // [32][PITCH] the layer's input rows of step t, hi parts (layer 0: the projected rows, no lo part)
  unsigned char* XL = XH + 32 * PITCH;
  unsigned char* SH = XL + 32 * PITCH;             // the layer's state before step t
  unsigned char* SL = SH + 32 * PITCH;
  unsigned char* RHh = SL + 32 * PITCH;            // r * h of step t
  unsigned char* RHl = RHh + 32 * PITCH;
  float* RED = reinterpret_cast<float*>(RHl + 32 * PITCH);   // [2 tiles][4 waves][16][64] partial accumulators
  for (int i = tid; i < 2 * 32 * PITCH / 16; i += LW_THREADS) reinterpret_cast<uint4*>(SH)[i] = make_uint4(0, 0, 0, 0);   // zero_state
  if (tid == 0) s_dead = 0;

  // ---- this wave's share of the layer's weights, once: tiles r, z, candidate of unit block p, k-steps [KW wave, KW wave + KW)
  uint4 Wh[3][KW], Wl[3][KW];
#pragma unroll
  for (int sel = 0; sel < 3; sel++) {
    const size_t T = (size_t)((sel == 1 ? HB : 0) + p);
    const uint16_t* th = (sel < 2 ? a.wgF_hi[l] : a.wcF_hi[l]) + ((T * KS + (size_t)KW * wave) * 64 + lane) * 8;
    const uint16_t* tl = (sel < 2 ? a.wgF_lo[l] : a.wcF_lo[l]) + ((T * KS + (size_t)KW * wave) * 64 + lane) * 8;
#pragma unroll
    for (int i = 0; i < KW; i++) {
      Wh[sel][i] = *reinterpret_cast<const uint4*>(th + (size_t)i * 512);
      Wl[sel][i] = *reinterpret_cast<const uint4*>(tl + (size_t)i * 512);
    }
  }
#pragma unroll
  for (int sel = 0; sel < 3; sel++)
#pragma unroll
    for (int i = 0; i < KW; i++) {   // opaque: keep them in registers (lgru_pipe.hip: hipcc would re-load read-only fragments at every use)
      asm volatile("" : "+v"(Wh[sel][i].x), "+v"(Wh[sel][i].y), "+v"(Wh[sel][i].z), "+v"(Wh[sel][i].w));
      asm volatile("" : "+v"(Wl[sel][i].x), "+v"(Wl[sel][i].y), "+v"(Wl[sel][i].z), "+v"(Wl[sel][i].w));
    }
  const int unit0 = 32 * p + 8 * wave + 4 * lh;    // this lane's four units (of its session) behind the reduction: accumulator registers 4 wave .. 4 wave + 3
  float bias_r[4], bias_z[4], bias_c[4];
#pragma unroll
  for (int j = 0; j < 4; j++) { bias_r[j] = a.bg[l][unit0 + j]; bias_z[j] = a.bg[l][H + unit0 + j]; bias_c[j] = a.bc[l][unit0 + j]; }
  float hreg[4] = {0.f, 0.f, 0.f, 0.f};

  const size_t slot = (size_t)2 * M * H;           // 16-bit elements of one (layer, step) slot: hi rows, then lo rows
  auto cnt_of = [&](int ll, int t, int ph) { return a.cnt + (((size_t)ll * tiles + tile) * L + t) * 2 + ph; };
  auto wait_for = [&](int* c) {   // one lane, bounded; the caller's barrier follows
    unsigned spins = 0;
    while (!s_dead && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < LW_PARTS) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > LW_SPIN_MAX) {
        __hip_atomic_store(a.err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_dead = 1;
      }
    }
  };
  // 32 rows x 256 units of a slot (hi and lo parts) -> two LDS buffers; agent-scope 8-byte loads the compiler can see
  auto gather = [&](const uint16_t* src, unsigned char* dh, unsigned char* dl) {
    unsigned long long vh[8], vl[8];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = tid + k * LW_THREADS;          // chunk of 16 bytes: row = c / 32, chunk = c % 32
      const int row = c >> 5, ch = c & 31;
      const int64_t s = (sess0 + row < M) ? sess0 + row : (int64_t)M - 1;
      const unsigned long long* ph = reinterpret_cast<const unsigned long long*>(src + (size_t)s * H + ch * 8);
      const unsigned long long* pl = reinterpret_cast<const unsigned long long*>(src + (size_t)M * H + (size_t)s * H + ch * 8);
      vh[2 * k] = __hip_atomic_load(ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      vh[2 * k + 1] = __hip_atomic_load(ph + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      vl[2 * k] = __hip_atomic_load(pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      vl[2 * k + 1] = __hip_atomic_load(pl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int c = tid + k * LW_THREADS;
      const int row = c >> 5, ch = c & 31;
      *reinterpret_cast<unsigned long long*>(dh + row * PITCH + ch * 16) = vh[2 * k];
      *reinterpret_cast<unsigned long long*>(dh + row * PITCH + ch * 16 + 8) = vh[2 * k + 1];
      *reinterpret_cast<unsigned long long*>(dl + row * PITCH + ch * 16) = vl[2 * k];
      *reinterpret_cast<unsigned long long*>(dl + row * PITCH + ch * 16 + 8) = vl[2 * k + 1];
    }
  };
  // this lane's four values (units unit0 .. + 3 of its session), split, to a slot: 8 bytes hi + 8 bytes lo, write-through
  auto publish = [&](uint16_t* dst, const float* v4) {
    uint16_t hi[4], lo[4];
#pragma unroll
    for (int j = 0; j < 4; j++) lw_split(v4[j], hi[j], lo[j]);
    if (live) {
      const unsigned long long ph = (unsigned long long)hi[0] | ((unsigned long long)hi[1] << 16) | ((unsigned long long)hi[2] << 32) | ((unsigned long long)hi[3] << 48);
      const unsigned long long pl = (unsigned long long)lo[0] | ((unsigned long long)lo[1] << 16) | ((unsigned long long)lo[2] << 32) | ((unsigned long long)lo[3] << 48);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + (size_t)sess * H + unit0), ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(dst + (size_t)M * H + (size_t)sess * H + unit0), pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto arrive = [&](int* c) {   // every wave's stores have completed, then ONE add for the workgroup
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(c, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  // this wave's KW k-steps of one tile: k-steps [KW wave, ..) lie in the input half for waves 0, 1 and in the second half (state / r * h) for waves 2, 3
  const bool in_half = wave < 2;
  const int ks0 = (KW * wave) % KSI;
  auto partial = [&](int sel, const unsigned char* h2, const unsigned char* l2) -> f32x16 {
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = 0.f;
    const unsigned char* bh = in_half ? XH : h2;
    const unsigned char* bl = in_half ? XL : l2;
    const bool has_lo = !(in_half && first);
#pragma unroll
    for (int i = 0; i < KW; i++) {
      const int ko = (ks0 + i) * 32 + lh * 16;
      const uint4 qh = *reinterpret_cast<const uint4*>(bh + lr * PITCH + ko);
      acc = mfma32(Wh[sel][i], qh, acc);
      acc = mfma32(Wl[sel][i], qh, acc);
      if (has_lo) {
        const uint4 ql = *reinterpret_cast<const uint4*>(bl + lr * PITCH + ko);
        acc = mfma32(Wh[sel][i], ql, acc);
      }
    }
    return acc;
  };
  auto red_put = [&](int slot_i, const f32x16& acc) {
#pragma unroll
    for (int i = 0; i < 16; i++) RED[((slot_i * 4 + wave) * 16 + i) * 64 + lane] = acc[i];
  };
  auto red_get = [&](int slot_i, const float* bias, float* out4) {   // registers 4 wave .. 4 wave + 3, the four partial tiles in wave order
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float v = bias[j];
#pragma unroll
      for (int w = 0; w < 4; w++) v += RED[((slot_i * 4 + w) * 16 + 4 * wave + j) * 64 + lane];
      out4[j] = v;
    }
  };
  __syncthreads();

  const size_t cellf = (size_t)M * 5 * H;
  for (int t = 0; t < L; t++) {
    const bool active = t < len;
    float* rec = a.rec ? a.rec + ((size_t)t * nl + l) * cellf : nullptr;
    // ---------------- inputs of the cell: the layer's input rows of step t, the layer's state after step t - 1
    if (tid == 0) {
      if (!first) wait_for(cnt_of(l - 1, t, 1));
      if (t > 0) wait_for(cnt_of(l, t - 1, 1));
    }
    __syncthreads();
    if (first) {
#pragma unroll
      for (int k = 0; k < 4; k++) {   // projected rows: 32 x 32 chunks of 16 bytes, plain loads
        const int c = tid + k * LW_THREADS;
        const int row = c >> 5, ch = c & 31;
        const int64_t s = (sess0 + row < M) ? sess0 + row : (int64_t)M - 1;
        *reinterpret_cast<uint4*>(XH + row * PITCH + ch * 16) = *reinterpret_cast<const uint4*>(a.u16 + ((size_t)s * L + t) * H + ch * 8);
      }
    } else {
      gather(a.hx + ((size_t)(l - 1) * L + t) * slot, XH, XL);
    }
    if (t > 0) gather(a.hx + ((size_t)l * L + (t - 1)) * slot, SH, SL);
    __syncthreads();
    // ---------------- gates: partial r and z tiles, reduction, epilogue on this lane's four units
    {
      const f32x16 pr = partial(0, SH, SL);
      red_put(0, pr);
      const f32x16 pz = partial(1, SH, SL);
      red_put(1, pz);
    }
    __syncthreads();
    float r4[4], z4[4], rh4[4];
    red_get(0, bias_r, r4);
    red_get(1, bias_z, z4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      r4[j] = lw_sigmoid(r4[j]);
      z4[j] = lw_sigmoid(z4[j]);
      rh4[j] = r4[j] * hreg[j];
    }
    publish(a.rx + ((size_t)l * L + t) * slot, rh4);
    if (rec && live) {
      *reinterpret_cast<float4*>(rec + sess * 2 * H + unit0) = make_float4(r4[0], r4[1], r4[2], r4[3]);
      *reinterpret_cast<float4*>(rec + sess * 2 * H + H + unit0) = make_float4(z4[0], z4[1], z4[2], z4[3]);
      *reinterpret_cast<float4*>(rec + (size_t)M * 2 * H + sess * H + unit0) = make_float4(rh4[0], rh4[1], rh4[2], rh4[3]);
    }
    arrive(cnt_of(l, t, 0));
    // ---------------- candidate: r * h of all 256 units, partial tile, reduction, blend
    if (tid == 0) wait_for(cnt_of(l, t, 0));
    __syncthreads();
    gather(a.rx + ((size_t)l * L + t) * slot, RHh, RHl);
    __syncthreads();
    {
      const f32x16 pc = partial(2, RHh, RHl);
      red_put(0, pc);
    }
    __syncthreads();
    float c4[4], hn[4], st4[4];
    red_get(0, bias_c, c4);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      c4[j] = lw_tanh(c4[j]);
      hn[j] = z4[j] * hreg[j] + (1.0f - z4[j]) * c4[j];
      st4[j] = active ? hn[j] : hreg[j];   // dynamic_rnn past the session's length: state through, output zero (model_gru.py:47-53)
      hreg[j] = st4[j];
    }
    publish(a.hx + ((size_t)l * L + t) * slot, st4);
    if (live) {
      if (rec) {
        *reinterpret_cast<float4*>(rec + (size_t)M * 3 * H + sess * H + unit0) = make_float4(c4[0], c4[1], c4[2], c4[3]);
        *reinterpret_cast<float4*>(rec + (size_t)M * 4 * H + sess * H + unit0) = make_float4(hn[0], hn[1], hn[2], hn[3]);
        *reinterpret_cast<float4*>(a.states + ((size_t)(t + 1) * M + sess) * HT + (size_t)l * H + unit0) = make_float4(st4[0], st4[1], st4[2], st4[3]);
      }
      if (last) {
        uint2 o;
        o.x = active ? pack2bf(hn[0], hn[1]) : 0u;
        o.y = active ? pack2bf(hn[2], hn[3]) : 0u;
        *reinterpret_cast<uint2*>(a.hidden + (sess * L + t) * H + unit0) = o;
      }
    }
    arrive(cnt_of(l, t, 1));   // (its barrier also closes the cell: RED, X, S, RH are free for the next one)
  }
}

static size_t lw_lds_bytes() { return 6 * 32 * (2 * 256 + 16) + 2 * 4 * 16 * 64 * sizeof(float); }

bool lgru_wide_applies(const ModelDev& m, int64_t n_sess) {
  if (m.tune.lgru_wide == 0) return false;         // never (A/B runs, tests)
  if (m.tune.lgru_persistent == 0) return false;   // set by the shared-GPU guard (distributed.py): no launch that waits on other workgroups
  if (!lgru_tile_applies(m, n_sess) || m.H != 256 || m.gru_bar == nullptr) return false;
  const int64_t tiles = (n_sess + 31) / 32, per = lw_tiles_per_launch(m, LW_PARTS);   // every workgroup of a launch resident at once
  return per >= 1 && tiles <= lw_max_launches(m, LW_PARTS) * per;
}

// bytes of the exchange slots + counters (0: no such launch for this model; sizes past the launch's range count as the largest one in it)
size_t lgru_wide_bytes(const ModelDev& m, int64_t n_sess, int L) {
  if (n_sess <= 0 || m.H != 256 || m.E != m.H || m.nl > 2) return 0;
  const int64_t cap = 32 * lw_max_launches(m, LW_PARTS) * lw_tiles_per_launch(m, LW_PARTS);   // a workspace sized for a larger batch still serves the smaller ones
  if (n_sess > cap) n_sess = cap;
  const size_t tiles = (size_t)(n_sess + 31) / 32;
  const size_t x = (size_t)m.nl * L * 2 * (size_t)n_sess * m.H * 2;
  return 2 * ((x + 255) & ~(size_t)255) + ((tiles * m.nl * L * 2 * 4 + 255) & ~(size_t)255);
}

hipError_t launch_lgru_wide(const ModelDev& m, const uint16_t* u16, const int32_t* seq_len, int64_t n_sess, int L, uint16_t* hidden, float* states, float* rec,
                            void* scratch, hipStream_t st) {
  LgruWideArgs a;
  memset(&a, 0, sizeof(a));
  a.u16 = u16; a.seq_len = seq_len; a.hidden = hidden; a.states = states; a.rec = rec;
  for (int l = 0; l < m.nl; l++) {
    a.wgF_hi[l] = m.lgru_wgF_hi[l]; a.wgF_lo[l] = m.lgru_wgF_lo[l]; a.wcF_hi[l] = m.lgru_wcF_hi[l]; a.wcF_lo[l] = m.lgru_wcF_lo[l];
    a.bg[l] = m.lgru_bg[l]; a.bc[l] = m.lgru_bc[l];
  }
  a.M = (int)n_sess; a.L = L; a.HT = m.HT; a.nl = m.nl;
  const unsigned tiles = (unsigned)((n_sess + 31) / 32);
  const size_t xbytes = (((size_t)m.nl * L * 2 * (size_t)n_sess * m.H * 2) + 255) & ~(size_t)255;
  unsigned char* s = reinterpret_cast<unsigned char*>(scratch);
  a.hx = reinterpret_cast<uint16_t*>(s);
  a.rx = reinterpret_cast<uint16_t*>(s + xbytes);
  a.cnt = reinterpret_cast<int*>(s + 2 * xbytes);
  a.err = reinterpret_cast<unsigned long long*>(m.gru_bar) + 8 * 16;   // (gru_persist.hip: the word behind the 8 group counters)
  hipError_t e;
  if ((e = hipMemsetAsync(a.cnt, 0, (size_t)tiles * m.nl * L * 2 * 4, st)) != hipSuccess) return e;
  const size_t lds = lw_lds_bytes();
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lgru_wide), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
  const unsigned per = (unsigned)lw_tiles_per_launch(m, LW_PARTS);
  a.tiles = (int)tiles;
  for (unsigned t0 = 0; t0 < tiles; t0 += per) {
    a.tile0 = (int)t0;
    hipLaunchKernelGGL(k_lgru_wide, dim3(tiles - t0 < per ? tiles - t0 : per, LW_PARTS, (unsigned)m.nl), dim3(LW_THREADS), lds, st, a);
  }
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward through time of the same cells (model_gru.py:47-55 differentiated by model.py:139-141) in the same decomposition, H = 128 and 256:
// a (32-session tile, layer) is HB workgroups; workgroup p owns units [32 p, 32 p + 32) of the layer -- as elements of a cell's output (everything
// elementwise is local) and as the OUTPUT COLUMNS 32 p .. and in + 32 p .. of the two backward products
//     [d x_c | d(r h)] = da_c . Wc      (reduction over the layer's H units)
//     [d x_g | d h_g]  = [da_r | da_z] . Wg      (reduction over 2 H)
// whose transposed weight fragments (two tiles each: 24 HB registers per lane with the reduction split over the four waves) stay in registers
// for all L steps.  Per cell two all-to-all hand-offs inside the launch -- da_c, then [da_r | da_z], both through their slots of dac_all / dag_all
// (fp32, what the weight-gradient products behind the loop read anyway; the readers split them into 16-bit hi + lo rows on the way into LDS) -- and
// one point-to-point hand-off: d x of block p goes to workgroup p of the layer BELOW (with E = H its input units are that workgroup's own units),
// so the layers run pipelined top down.  Formulas and arithmetic of k_lgru_tile_bwd (lgru_tile.hip); the sums over the reduction are four partial sums.
struct LgruWideBwdArgs {
  const float* dH;                           // [M * L][H]: d loss / d (top layer output)
  const int32_t* seq_len;
  const float* states;                       // [(L + 1)][M][HT]
  const float* rec;                          // per (t, l): gates [M][2H] | r*h | cand | out
  const uint16_t* wcT_hi[HIERTCN_MAX_LAYERS]; const uint16_t* wcT_lo[HIERTCN_MAX_LAYERS];   // (Wc)^T fragment-major: [K/32][H/16][64][8]
  const uint16_t* wgT_hi[HIERTCN_MAX_LAYERS]; const uint16_t* wgT_lo[HIERTCN_MAX_LAYERS];   // (Wg)^T: [K/32][2H/16][64][8]
  float* du;                                 // [M * L][E]
  float* dac_all;                            // [(l L + t)][M][H]
  float* dag_all;                            // [(l L + t)][M][2H]
  float* dxb;                                // [nl][L][M][H]: d x of (layer, step), for the layer below
  int* cnt;                                  // [nl][tiles][L][2]: arrivals of phase 1 (da_c) / phase 2 ([da_r | da_z])
  int* flagx;                                // [nl][tiles][L][HB]: d x of (layer, tile, step, part) published
  unsigned long long* err;
  int M, L, HT, nl;
  int tile0, tiles;                          // this launch's first tile; tiles of the whole batch (the counters' stride)
};

template <int HB>   // H / 32 in {4, 8} = workgroups per (tile, layer)
__global__ __launch_bounds__(LW_THREADS, 1) void k_lgru_wide_bwd(const LgruWideBwdArgs a) {
  constexpr int H = 32 * HB;
  constexpr int KSC = 2 * HB, KSG = 4 * HB;        // k-steps of the two products (reduction over H resp. 2H)
  constexpr int KWC = KSC / 4, KWG = KSG / 4;      // ... per wave
  constexpr int PC = 2 * H + 16, PG = 4 * H + 16;  // LDS pitches of the split da_c resp. [da_r | da_z] rows
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_dead;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nl = a.nl, L = a.L, M = a.M, HT = a.HT;
  const int tile = a.tile0 + (int)blockIdx.x, p = (int)blockIdx.y, tiles = a.tiles;
  const int l = nl - 1 - (int)blockIdx.z;          // (the top layer's workgroups first in dispatch order: they are the producers)
  const int64_t sess0 = (int64_t)tile * 32;
  const int64_t sess = sess0 + lr;
  const bool live = sess < M;
  const int64_t sc = live ? sess : (int64_t)M - 1;
  const int len = a.seq_len[sc];
  const bool top = l == nl - 1;
  unsigned char* DCh = smem;
  unsigned char* DCl = DCh + 32 * PC;
  unsigned char* DGh = DCl + 32 * PC;
  unsigned char* DGl = DGh + 32 * PG;
  float* RED = reinterpret_cast<float*>(DGl + 32 * PG);   // [2 tiles][4 waves][16][64]
  if (tid == 0) s_dead = 0;

  // ---- this wave's share of the transposed weights, once: output tiles (input part, state part) of block p, its quarter of each reduction
  uint4 Ch[2][KWC], Cl[2][KWC], Gh[2][KWG], Gl[2][KWG];
#pragma unroll
  for (int part = 0; part < 2; part++) {
    const size_t T = (size_t)(part * HB + p);
#pragma unroll
    for (int i = 0; i < KWC; i++) {
      const size_t off = ((T * KSC + (size_t)KWC * wave + i) * 64 + lane) * 8;
      Ch[part][i] = *reinterpret_cast<const uint4*>(a.wcT_hi[l] + off);
      Cl[part][i] = *reinterpret_cast<const uint4*>(a.wcT_lo[l] + off);
    }
#pragma unroll
    for (int i = 0; i < KWG; i++) {
      const size_t off = ((T * KSG + (size_t)KWG * wave + i) * 64 + lane) * 8;
      Gh[part][i] = *reinterpret_cast<const uint4*>(a.wgT_hi[l] + off);
      Gl[part][i] = *reinterpret_cast<const uint4*>(a.wgT_lo[l] + off);
    }
  }
#pragma unroll
  for (int part = 0; part < 2; part++) {   // opaque: keep them in registers
#pragma unroll
    for (int i = 0; i < KWC; i++) {
      asm volatile("" : "+v"(Ch[part][i].x), "+v"(Ch[part][i].y), "+v"(Ch[part][i].z), "+v"(Ch[part][i].w));
      asm volatile("" : "+v"(Cl[part][i].x), "+v"(Cl[part][i].y), "+v"(Cl[part][i].z), "+v"(Cl[part][i].w));
    }
#pragma unroll
    for (int i = 0; i < KWG; i++) {
      asm volatile("" : "+v"(Gh[part][i].x), "+v"(Gh[part][i].y), "+v"(Gh[part][i].z), "+v"(Gh[part][i].w));
      asm volatile("" : "+v"(Gl[part][i].x), "+v"(Gl[part][i].y), "+v"(Gl[part][i].z), "+v"(Gl[part][i].w));
    }
  }
  const int unit0 = 32 * p + 8 * wave + 4 * lh;    // this lane's four units of its session behind a reduction
  float dcar[4] = {0.f, 0.f, 0.f, 0.f};            // the low level's final state is dropped (model_gru.py:47)

  auto cnt_of = [&](int ll, int t, int ph) { return a.cnt + (((size_t)ll * tiles + tile) * L + t) * 2 + ph; };
  auto flag_of = [&](int ll, int t, int pp) { return a.flagx + (((size_t)ll * tiles + tile) * L + t) * HB + pp; };
  auto wait_ge = [&](int* c, int n) {
    unsigned spins = 0;
    while (!s_dead && __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < n) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > LW_SPIN_MAX) {
        __hip_atomic_store(a.err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_dead = 1;
      }
    }
  };
  auto st2 = [&](float* ptr, float x, float y) {   // two floats, write-through
    const unsigned long long v = (unsigned long long)__float_as_uint(x) | ((unsigned long long)__float_as_uint(y) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(ptr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto ld2 = [&](const float* ptr, float& x, float& y) {
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(ptr), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    x = __uint_as_float((unsigned)(v & 0xffffffffull));
    y = __uint_as_float((unsigned)(v >> 32));
  };
  // 32 rows x W floats of an fp32 slot -> split 16-bit rows in LDS (hi, lo), 8 agent-scope loads of 8 bytes in flight per thread
  auto gather_split = [&](const float* src, int W, unsigned char* dh, unsigned char* dl, int pitch) {
    const int per_row = W >> 1, total = 32 * per_row;
    for (int c0 = 0; c0 < total; c0 += 8 * LW_THREADS) {
      unsigned long long v[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int c = min(c0 + k * LW_THREADS + tid, total - 1);
        const int row = c / per_row, col = c - row * per_row;
        const int64_t s = (sess0 + row < M) ? sess0 + row : (int64_t)M - 1;
        v[k] = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(src + (size_t)s * W + 2 * col), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int c = c0 + k * LW_THREADS + tid;
        if (c < total) {
          const int row = c / per_row, col = c - row * per_row;
          uint16_t h0, l0, h1, l1;
          lw_split(__uint_as_float((unsigned)(v[k] & 0xffffffffull)), h0, l0);
          lw_split(__uint_as_float((unsigned)(v[k] >> 32)), h1, l1);
          *reinterpret_cast<uint32_t*>(dh + row * pitch + col * 4) = (uint32_t)h0 | ((uint32_t)h1 << 16);
          *reinterpret_cast<uint32_t*>(dl + row * pitch + col * 4) = (uint32_t)l0 | ((uint32_t)l1 << 16);
        }
      }
    }
  };
  auto red_put = [&](int slot_i, const f32x16& acc) {
#pragma unroll
    for (int i = 0; i < 16; i++) RED[((slot_i * 4 + wave) * 16 + i) * 64 + lane] = acc[i];
  };
  auto red_get = [&](int slot_i, float* out4) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; w++) v += RED[((slot_i * 4 + w) * 16 + 4 * wave + j) * 64 + lane];
      out4[j] = v;
    }
  };
  __syncthreads();

  const size_t cellf = (size_t)M * 5 * H;
  for (int t = L - 1; t >= 0; t--) {
    const bool active = t < len;
    const float* rec = a.rec + ((size_t)t * nl + l) * cellf;
    const float* hp = a.states + ((size_t)t * M + sc) * HT + (size_t)l * H;
    float* dacs = a.dac_all + ((size_t)l * L + t) * M * H;
    float* dags = a.dag_all + ((size_t)l * L + t) * M * 2 * H;
    // ---------------- the gradient of the cell's output: d loss / d output (top layer) or the d x of the layer above (same part, same step)
    float dout[4];
    if (top) {
      const float4 o4 = *reinterpret_cast<const float4*>(a.dH + (sc * L + t) * H + unit0);
      dout[0] = o4.x; dout[1] = o4.y; dout[2] = o4.z; dout[3] = o4.w;
    } else {
      if (tid == 0) wait_ge(flag_of(l + 1, t, p), 1);
      __syncthreads();
      const float* src = a.dxb + (((size_t)(l + 1) * L + t) * M + sc) * H + unit0;
      ld2(src, dout[0], dout[1]);
      ld2(src + 2, dout[2], dout[3]);
    }
    // ---------------- P1 (local): through the blend and the tanh
    float dhp[4], dagz[4], hv[4];
    {
      const float4 z4 = *reinterpret_cast<const float4*>(rec + sc * 2 * H + H + unit0);
      const float4 c4 = *reinterpret_cast<const float4*>(rec + (size_t)M * 3 * H + sc * H + unit0);
      const float4 h4 = *reinterpret_cast<const float4*>(hp + unit0);
      const float zv[4] = {z4.x, z4.y, z4.z, z4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w};
      hv[0] = h4.x; hv[1] = h4.y; hv[2] = h4.z; hv[3] = h4.w;
      float dac[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        const float d = active ? dcar[j] + dout[j] : 0.f;
        dac[j] = d * (1.f - zv[j]) * (1.f - cv[j] * cv[j]);
        dhp[j] = d * zv[j];
        dagz[j] = d * (hv[j] - cv[j]) * zv[j] * (1.f - zv[j]);
      }
      if (live) {
        st2(dacs + (size_t)sess * H + unit0, dac[0], dac[1]);
        st2(dacs + (size_t)sess * H + unit0 + 2, dac[2], dac[3]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt_of(l, t, 0), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_ge(cnt_of(l, t, 0), HB);
    }
    __syncthreads();
    gather_split(dacs, H, DCh, DCl, PC);
    __syncthreads();
    // ---------------- P2: [d x_c | d(r h)] = da_c . Wc (own output columns), through the reset gate
    {
#pragma unroll
      for (int part = 0; part < 2; part++) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
        for (int i = 0; i < KWC; i++) {
          const int ko = (KWC * wave + i) * 32 + lh * 16;
          const uint4 qh = *reinterpret_cast<const uint4*>(DCh + lr * PC + ko);
          const uint4 ql = *reinterpret_cast<const uint4*>(DCl + lr * PC + ko);
          acc = mfma32(Ch[part][i], qh, acc);
          acc = mfma32(Cl[part][i], qh, acc);
          acc = mfma32(Ch[part][i], ql, acc);
        }
        red_put(part, acc);
      }
    }
    __syncthreads();
    float dxc[4], drh[4];
    red_get(0, dxc);
    red_get(1, drh);
    {
      const float4 r4 = *reinterpret_cast<const float4*>(rec + sc * 2 * H + unit0);
      const float rv[4] = {r4.x, r4.y, r4.z, r4.w};
      float dagr[4];
#pragma unroll
      for (int j = 0; j < 4; j++) {
        dagr[j] = drh[j] * hv[j] * rv[j] * (1.f - rv[j]);
        dhp[j] += drh[j] * rv[j];
      }
      if (live) {
        st2(dags + (size_t)sess * 2 * H + unit0, dagr[0], dagr[1]);
        st2(dags + (size_t)sess * 2 * H + unit0 + 2, dagr[2], dagr[3]);
        st2(dags + (size_t)sess * 2 * H + H + unit0, dagz[0], dagz[1]);
        st2(dags + (size_t)sess * 2 * H + H + unit0 + 2, dagz[2], dagz[3]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // (also: every wave has read RED)
    if (tid == 0) {
      __hip_atomic_fetch_add(cnt_of(l, t, 1), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      wait_ge(cnt_of(l, t, 1), HB);
    }
    __syncthreads();
    gather_split(dags, 2 * H, DGh, DGl, PG);
    __syncthreads();
    // ---------------- P3: [d x_g | d h_g] = [da_r | da_z] . Wg (own output columns)
    {
#pragma unroll
      for (int part = 0; part < 2; part++) {
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 0.f;
#pragma unroll
        for (int i = 0; i < KWG; i++) {
          const int ko = (KWG * wave + i) * 32 + lh * 16;
          const uint4 qh = *reinterpret_cast<const uint4*>(DGh + lr * PG + ko);
          const uint4 ql = *reinterpret_cast<const uint4*>(DGl + lr * PG + ko);
          acc = mfma32(Gh[part][i], qh, acc);
          acc = mfma32(Gl[part][i], qh, acc);
          acc = mfma32(Gh[part][i], ql, acc);
        }
        red_put(part, acc);
      }
    }
    __syncthreads();
    float dxg[4], dhg[4], dx[4];
    red_get(0, dxg);
    red_get(1, dhg);
#pragma unroll
    for (int j = 0; j < 4; j++) {
      dx[j] = dxc[j] + dxg[j];                       // zero for a finished session (its da rows are zero)
      dcar[j] = active ? dhp[j] + dhg[j] : dcar[j];
    }
    if (l == 0) {
      if (live) *reinterpret_cast<float4*>(a.du + (sess * L + t) * H + unit0) = make_float4(dx[0], dx[1], dx[2], dx[3]);
      __syncthreads();   // (every wave has read RED before the next cell's products overwrite it)
    } else {
      if (live) {
        float* dst = a.dxb + (((size_t)l * L + t) * M + sess) * H + unit0;
        st2(dst, dx[0], dx[1]);
        st2(dst + 2, dx[2], dx[3]);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) __hip_atomic_store(flag_of(l, t, p), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <int HB>
static size_t lwb_lds_bytes() { return 2 * 32 * (2 * 32 * HB + 16) + 2 * 32 * (4 * 32 * HB + 16) + 2 * 4 * 16 * 64 * sizeof(float); }

bool lgru_wide_bwd_applies(const ModelDev& m, int64_t n_sess) {
  if (m.tune.lgru_wide_bwd == 0 || m.tune.lgru_persistent == 0) return false;
  if (!lgru_tile_applies(m, n_sess) || !(m.H == 128 || m.H == 256) || m.gru_bar == nullptr || !m.lgru_wgTF_hi[0]) return false;
  const int64_t tiles = (n_sess + 31) / 32, per = lw_tiles_per_launch(m, m.H / 32);
  return per >= 1 && tiles <= lw_max_launches(m, m.H / 32) * per;
}

// bytes of the d x slots + counters + flags (0: no such launch for this model; sizes past its range count as the largest one in it)
size_t lgru_wide_bwd_bytes(const ModelDev& m, int64_t n_sess, int L) {
  if (n_sess <= 0 || !(m.H == 128 || m.H == 256) || m.E != m.H || m.nl > 2) return 0;
  const int64_t cap = 32 * lw_max_launches(m, m.H / 32) * lw_tiles_per_launch(m, m.H / 32);
  if (n_sess > cap) n_sess = cap;
  const size_t tiles = (size_t)(n_sess + 31) / 32;
  const size_t x = (size_t)m.nl * L * (size_t)n_sess * m.H * 4;
  return ((x + 255) & ~(size_t)255) + ((tiles * m.nl * L * (2 + (size_t)m.H / 32) * 4 + 255) & ~(size_t)255);
}

hipError_t launch_lgru_wide_bwd(const ModelDev& m, const float* dH, const int32_t* seq_len, const float* states, const float* rec, int64_t n_sess, int L, float* du,
                                float* dac_all, float* dag_all, void* scratch, hipStream_t st) {
  LgruWideBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.dH = dH; a.seq_len = seq_len; a.states = states; a.rec = rec; a.du = du; a.dac_all = dac_all; a.dag_all = dag_all;
  for (int l = 0; l < m.nl; l++) {
    a.wcT_hi[l] = m.lgru_wcTF_hi[l]; a.wcT_lo[l] = m.lgru_wcTF_lo[l]; a.wgT_hi[l] = m.lgru_wgTF_hi[l]; a.wgT_lo[l] = m.lgru_wgTF_lo[l];
  }
  a.M = (int)n_sess; a.L = L; a.HT = m.HT; a.nl = m.nl;
  const unsigned tiles = (unsigned)((n_sess + 31) / 32);
  const int HB = m.H / 32;
  const size_t xbytes = (((size_t)m.nl * L * (size_t)n_sess * m.H * 4) + 255) & ~(size_t)255;
  unsigned char* s = reinterpret_cast<unsigned char*>(scratch);
  a.dxb = reinterpret_cast<float*>(s);
  a.cnt = reinterpret_cast<int*>(s + xbytes);
  a.flagx = a.cnt + (size_t)tiles * m.nl * L * 2;
  a.err = reinterpret_cast<unsigned long long*>(m.gru_bar) + 8 * 16;
  hipError_t e;
  if ((e = hipMemsetAsync(a.cnt, 0, (size_t)tiles * m.nl * L * (2 + HB) * 4, st)) != hipSuccess) return e;
  const unsigned per = (unsigned)lw_tiles_per_launch(m, HB);
  a.tiles = (int)tiles;
  if (HB == 8) {
    const size_t lds = lwb_lds_bytes<8>();
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lgru_wide_bwd<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    for (unsigned t0 = 0; t0 < tiles; t0 += per) {
      a.tile0 = (int)t0;
      hipLaunchKernelGGL(k_lgru_wide_bwd<8>, dim3(tiles - t0 < per ? tiles - t0 : per, 8, (unsigned)m.nl), dim3(LW_THREADS), lds, st, a);
    }
  } else {
    const size_t lds = lwb_lds_bytes<4>();
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_lgru_wide_bwd<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess) return e;
    for (unsigned t0 = 0; t0 < tiles; t0 += per) {
      a.tile0 = (int)t0;
      hipLaunchKernelGGL(k_lgru_wide_bwd<4>, dim3(tiles - t0 < per ? tiles - t0 : per, 4, (unsigned)m.nl), dim3(LW_THREADS), lds, st, a);
    }
  }
  return hipGetLastError();
}
