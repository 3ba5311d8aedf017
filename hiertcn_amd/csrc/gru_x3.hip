// HierGRU low level (model_hier.py:67-68 -> model_gru.py:47-55; cell arithmetic customed_gru_cell.py:309-337) on the 16-bit MFMA with SPLIT
// operands: every fp32 value v that enters a product is carried as v_hi + v_lo (v_hi = v rounded to the 16-bit storage type, v_lo = the
// rounded remainder) and a . w is formed as a_hi w_hi + a_lo w_hi + a_hi w_lo in fp32 accumulators -- 16 mantissa bits per factor, relative
// error ~2^-16 per product, three bf16 MFMAs (2.5 PFLOP/s dense) instead of one fp32 MFMA (157 TFLOP/s).
//
// Why here and not in the across-session scan: the low level runs dynamic_rnn from a ZERO state for at most L = 30 steps per session and its
// outputs are rounded to the storage type for the scoring layer anyway (2^-9 relative), so a 2^-16 perturbation of the recurrence never
// compounds past that rounding; the scan carries its state across sessions AND batches without bound and stays on exact fp32 chains
// (pool_gru.hip, gru_persist.hip).  At the bench shape this cell is M = 20,480 sessions x K = 512: the fp32 kernels spent 18.6 of the
// evaluation step's 91 ms here at 45-56 TFLOP/s.  The exact numerics mode (exact.hip) and HIERTCN_LGRU_FP32=1 keep the fp32 cells.
//
// One kernel, two epilogues (the fp32 kernel's, pool_gru.hip): D[row = output unit n][col = session m], so a lane holds four consecutive
// units of one session and every epilogue access is a 16-byte row access.  Workgroup = 128 units x 128 sessions (4 waves x four 32 x 32
// accumulators); the reduction runs over [in | h] in 64- or 32-deep chunks; W_hi, W_lo, A_hi, A_lo chunks arrive in LDS by LDS-DMA (two buffers,
// source-side swizzle) with the next chunk in flight during the MFMAs.  The layer-0 input (the projected rows, already 16-bit) has no low part: two products.
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

#define XT 128
#define XK_MIN 32   // reduction widths are multiples of this (the kernel's chunk depth is 64 or 32: k_gru_x3<XK>)

// sigmoid / tanh on the hardware exp2 and reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each) instead of libm's expf / tanhf + a division: at 64
// activations per thread the libm forms were as long as the cell's whole MFMA loop.  Both saturate correctly (exp2 -> 0 or inf, rcp(inf) = 0).
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float fast_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }

__device__ __forceinline__ void split2(float v, uint16_t& hi, uint16_t& lo) {
  hi = f2bf(v);
  lo = f2bf(v - bf2f(hi));
}

// XK = reduction columns per chunk: 64 (128-byte tile rows, 128 KB of LDS: one workgroup per CU, half as many barriers -- launches too small to fill
// the chip) or 32 (64-byte rows, 66 KB: two workgroups per CU, whose prologues and epilogues overlap each other's products)
template <int XK>
__global__ __launch_bounds__(256) void k_gru_x3(const GruX3Batch bt) {
  const GruX3Args& g = bt.g[blockIdx.z];   // the cells of one wavefront diagonal are independent products: one launch
  // Operand path: LDS-DMA (global_load_lds_dwordx4, 1 KB = 8 rows x 128 B per wave instruction) into two 64 KB chunk buffers -- the register-
  // staged path it replaces (global -> VGPR -> ds_write_b128, two register sets) left the waves waiting half of their cycles at 11 % MFMA busy
  // (scripts/profile_gru_x3.sh).  A tile row is 128 bytes = eight 16-byte chunks; DMA writes rows back to back (no padding possible), so the
  // chunk order is swizzled on the SOURCE side: LDS position (row r, chunk c) holds the row's chunk c ^ ((r >> 1) & 7), which puts the 16 rows
  // of a ds_read_b128 phase (same logical chunk) on 16 different 16-byte bank groups.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 x [Wh | Wl | Ah | Al] x 16 KB; behind the main loop: the fp32 result tile
  constexpr int ROWB = XK * 2, CPR = ROWB / 16, RPP = 1024 / ROWB, RW = 256 / ROWB, PPW = XT / RPP / 4;   // row bytes, chunks per row, rows per piece, rows per bank sweep, pieces per wave
  constexpr int RWS = RW == 2 ? 1 : 2;   // log2(RW)
  constexpr int TILE_B = XT * ROWB, CHUNK_B = 4 * TILE_B;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  // Tile order: the N / 128 unit tiles of one 128-session tile read the SAME activation rows, and a workgroup's XCD is its linear id mod 8 (observed
  // round-robin placement: a speed hint, never correctness) -- so consecutive slots of one XCD take the unit tiles of one session tile, and its rows
  // come from HBM into that XCD's L2 once instead of once per unit tile (the grid's y extent is padded to a multiple of 8 session tiles).
  const int NT = gridDim.x, MT = (int)((g.M + XT - 1) / XT);
  const int lin = blockIdx.x + NT * blockIdx.y;
  const int xcd = lin & 7, slot = lin >> 3;
  const int m_tile = (slot / NT) * 8 + xcd, n_tile = slot % NT;
  if (m_tile >= MT) return;
  const int n0 = n_tile * XT;
  const int64_t m0 = (int64_t)m_tile * XT;
  const int wn = wave >> 1, wm = wave & 1;
  const int K = g.K1 + g.K2;
  const int n_it = K / XK;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  auto dma = [&](const void* gsrc, unsigned lds_dst) {   // (score.hip: why inline asm, and why the waits are placed by hand)
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
  };
  // wave w moves pieces 4 w .. 4 w + 3 (rows 32 w .. 32 w + 31) of each of the four tiles; a lane's row inside a piece and its source chunk
  const int prow = lane / CPR;
  auto issue = [&](int it, int buf) {
    const int kc = it * XK;
    const bool seg2 = kc >= g.K1;
    const uint16_t* ah = seg2 ? g.A2h : g.A1h;
    const uint16_t* al = seg2 ? g.A2l : g.A1l;
    const int64_t lda = seg2 ? g.lda2 : g.lda1;
    const int ka = seg2 ? kc - g.K1 : kc;
#pragma unroll
    for (int pp = 0; pp < PPW; pp++) {
      const int row = 32 * wave + RPP * pp + prow;
      const int cs = ((lane & (CPR - 1)) ^ ((row >> RWS) & (CPR - 1))) * 8;  // source chunk (elements) for this LDS position
      const int64_t n = (n0 + row < g.N) ? n0 + row : g.N - 1;               // (rows past the end re-read the last one: never stored)
      const int64_t m = (m0 + row < g.M) ? m0 + row : g.M - 1;
      const unsigned dst = lds_base + buf * CHUNK_B + (PPW * wave + pp) * 1024;
      dma(g.Wh + n * K + kc + cs, dst);
      dma(g.Wl + n * K + kc + cs, dst + TILE_B);
      dma(ah + m * lda + ka + cs, dst + 2 * TILE_B);
      if (al) dma(al + m * lda + ka + cs, dst + 3 * TILE_B);                 // (wave-uniform)
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[i][j][q] = 0.f;
  // fragment addresses inside a tile: row * ROWB + ((2 s + lh) ^ swizzle(row)) * 16
  int wrow[2], arow[2], wsw[2], asw[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int rw = 64 * wn + 32 * i + lr, ra = 64 * wm + 32 * i + lr;
    wrow[i] = rw * ROWB; wsw[i] = (rw >> RWS) & (CPR - 1);
    arow[i] = ra * ROWB; asw[i] = (ra >> RWS) & (CPR - 1);
  }
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int it = 0; it < n_it; it++) {
    const int buf = it & 1;
    if (it + 1 < n_it) issue(it + 1, buf ^ 1);   // its readers passed the barrier that ended the previous chunk
    const unsigned char* Wh = smem + buf * CHUNK_B;
    const unsigned char* Wl = Wh + TILE_B;
    const unsigned char* Ah = Wh + 2 * TILE_B;
    const unsigned char* Al = Wh + 3 * TILE_B;
    const bool has_lo = (it * XK >= g.K1 ? g.A2l : g.A1l) != nullptr;   // the layer-0 input rows are 16-bit already: two products
#pragma unroll
    for (int s = 0; s < XK / 16; s++) {
      uint4 wh[2], wl[2], ah[2], al[2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
        const int wo = wrow[i] + (((2 * s + lh) ^ wsw[i]) << 4), ao = arow[i] + (((2 * s + lh) ^ asw[i]) << 4);
        wh[i] = *reinterpret_cast<const uint4*>(Wh + wo);
        wl[i] = *reinterpret_cast<const uint4*>(Wl + wo);
        ah[i] = *reinterpret_cast<const uint4*>(Ah + ao);
        if (has_lo) al[i] = *reinterpret_cast<const uint4*>(Al + ao);
      }
      // smallest terms first -- (a_lo w_hi + a_hi w_lo) + a_hi w_hi inside each fp32 accumulator -- and product-major, so that the four tiles'
      // MFMAs of one product sit between the dependent MFMAs of a tile
      if (has_lo) {
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 2; j++) acc[i][j] = mfma32(wh[i], al[j], acc[i][j]);
      }
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mfma32(wl[i], ah[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = mfma32(wh[i], ah[j], acc[i][j]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the next chunk have landed ...
    __syncthreads();                                    // ... and everyone's; every wave is done reading this one
  }
  // Epilogues.  D[row = output unit n][col = session m]: a lane holds four consecutive units of one session, and 32 lanes hold 32 different
  // sessions -- written from the registers, every instruction touches 32 rows of the global arrays with 32 bytes each (a quarter of every
  // 128-byte line it opens; the four q of a tile come in four separate instructions), and the cell was bound by that: ~190 MB of useful
  // epilogue traffic per gates + candidate pair moved at ~1 TB/s.  So the result tile goes through LDS (the operand tiles are dead): written as
  // it lies in the accumulators, read back ROW-major -- a wave covers two sessions x 128 units, 512 contiguous bytes per row and array.
  // Evaluation (save_all == 0) writes only what the next product reads: u, the split r*h, the split state and output; the training forward
  // also keeps r, r*h, the candidate and the output in fp32 for the backward pass.
  constexpr int TP = XT + 4;   // floats per row of the result tile: rows 4 banks apart for the 16-byte writes of 16 lanes
  // (the launch sizes the dynamic LDS as max(two chunk buffers, this tile))
  float* T = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int q = 0; q < 4; q++)
        *reinterpret_cast<float4*>(T + (64 * wm + 32 * j + lr) * TP + 64 * wn + 32 * i + 8 * q + 4 * lh) =
            make_float4(acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]);
  __syncthreads();
  const int H = g.H;
  auto pack4 = [](const uint16_t* v) { return make_uint2(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16)); };
  for (int pass = 0; pass < XT * XT / 4 / 256; pass++) {
    const int idx = pass * 256 + tid;
    const int ml = idx >> 5, n = n0 + 4 * (idx & 31);
    const int64_t m = m0 + ml;
    if (m >= g.M || n >= g.N) continue;
    const bool active = g.seq_len ? g.tstep < g.seq_len[m] : true;
    const float4 t4 = *reinterpret_cast<const float4*>(T + ml * TP + 4 * (idx & 31));
    const float4 b4 = *reinterpret_cast<const float4*>(g.bias + n);
    const float v[4] = {t4.x + b4.x, t4.y + b4.y, t4.z + b4.z, t4.w + b4.w};
    uint16_t hi[4], lo[4];
    if (g.epi == X3_GATES) {   // customed_gru_cell.py:323-326: value = sigmoid(.), r first, u second; r_state = r * h
      float s4[4];
#pragma unroll
      for (int c = 0; c < 4; c++) s4[c] = fast_sigmoid(v[c]);
      if (n >= H || g.save_all) *reinterpret_cast<float4*>(g.gates + m * 2 * H + n) = make_float4(s4[0], s4[1], s4[2], s4[3]);
      if (n < H) {
        const float4 hv = *reinterpret_cast<const float4*>(g.h + m * g.ldh + n);
        const float r4[4] = {s4[0] * hv.x, s4[1] * hv.y, s4[2] * hv.z, s4[3] * hv.w};
        if (g.save_all) *reinterpret_cast<float4*>(g.rh + m * H + n) = make_float4(r4[0], r4[1], r4[2], r4[3]);
#pragma unroll
        for (int c = 0; c < 4; c++) split2(r4[c], hi[c], lo[c]);
        *reinterpret_cast<uint2*>(g.rh_hi + m * H + n) = pack4(hi);
        *reinterpret_cast<uint2*>(g.rh_lo + m * H + n) = pack4(lo);
      }
    } else {                   // :335-336: c = tanh(.), h' = u h + (1 - u) c; dynamic_rnn past the row's length: state through, output 0
      const float4 hv = *reinterpret_cast<const float4*>(g.h + m * g.ldh + n);
      const float4 uv = *reinterpret_cast<const float4*>(g.gates + m * 2 * H + H + n);
      const float c4[4] = {fast_tanh(v[0]), fast_tanh(v[1]), fast_tanh(v[2]), fast_tanh(v[3])};
      float hn[4];
      hn[0] = uv.x * hv.x + (1.0f - uv.x) * c4[0];
      hn[1] = uv.y * hv.y + (1.0f - uv.y) * c4[1];
      hn[2] = uv.z * hv.z + (1.0f - uv.z) * c4[2];
      hn[3] = uv.w * hv.w + (1.0f - uv.w) * c4[3];
      if (g.save_all) {
        *reinterpret_cast<float4*>(g.cand_out + m * H + n) = make_float4(c4[0], c4[1], c4[2], c4[3]);
        *reinterpret_cast<float4*>(g.out + m * H + n) = make_float4(hn[0], hn[1], hn[2], hn[3]);
      }
      const float st4[4] = {active ? hn[0] : hv.x, active ? hn[1] : hv.y, active ? hn[2] : hv.z, active ? hn[3] : hv.w};
      *reinterpret_cast<float4*>(g.state_new + m * g.lds_ + n) = make_float4(st4[0], st4[1], st4[2], st4[3]);
#pragma unroll
      for (int c = 0; c < 4; c++) split2(st4[c], hi[c], lo[c]);
      *reinterpret_cast<uint2*>(g.state_hi + m * g.lds_ + n) = pack4(hi);
      *reinterpret_cast<uint2*>(g.state_lo + m * g.lds_ + n) = pack4(lo);
      if (g.out_hi) {          // the next layer's input operand
#pragma unroll
        for (int c = 0; c < 4; c++) split2(hn[c], hi[c], lo[c]);
        *reinterpret_cast<uint2*>(g.out_hi + m * H + n) = pack4(hi);
        *reinterpret_cast<uint2*>(g.out_lo + m * H + n) = pack4(lo);
      }
      if (g.hid_out) {         // top layer: the scoring operand row, zero past the length
        uint2 o;
        o.x = active ? pack2bf(hn[0], hn[1]) : 0u;
        o.y = active ? pack2bf(hn[2], hn[3]) : 0u;
        *reinterpret_cast<uint2*>(g.hid_out + m * g.ldho + n) = o;
      }
    }
  }
}

template <int XK>
static hipError_t launch_gru_x3_k(const GruX3Batch& bt, hipStream_t st) {
  const GruX3Args& g = bt.g[0];
  size_t lds = 2 * 4 * XT * XK * 2;   // two chunk buffers of four tiles ...
  if (lds < (size_t)XT * (XT + 4) * 4) lds = (size_t)XT * (XT + 4) * 4;   // ... or the fp32 result tile of the epilogue
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_gru_x3<XK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const unsigned mt = (unsigned)((g.M + XT - 1) / XT);
  hipLaunchKernelGGL(k_gru_x3<XK>, dim3((unsigned)((g.N + XT - 1) / XT), (mt + 7) / 8 * 8, (unsigned)bt.n), dim3(256), lds, st, bt);
  return hipGetLastError();
}

hipError_t launch_gru_x3(const GruX3Batch& bt, hipStream_t st) {
  if (bt.n <= 0 || bt.g[0].M <= 0) return hipSuccess;
  const GruX3Args& g = bt.g[0];
  bool k64 = true;
  for (int i = 0; i < bt.n; i++) {   // one grid for the batch: same shape, same epilogue kind
    const GruX3Args& q = bt.g[i];
    if ((q.K1 % XK_MIN) || (q.K2 % XK_MIN) || (q.N & 31) || q.N != g.N || q.M != g.M || q.epi != g.epi) return hipErrorInvalidValue;
    k64 = k64 && (q.K1 % 64) == 0 && (q.K2 % 64) == 0;
  }
  const int64_t tiles = (int64_t)((g.N + XT - 1) / XT) * ((g.M + XT - 1) / XT) * bt.n;
  if (k64 && tiles <= 256) return launch_gru_x3_k<64>(bt, st);   // fewer workgroups than CUs: nothing to overlap with, half the barriers
  return launch_gru_x3_k<32>(bt, st);
}

// v [rows][cols] fp32 -> hi, lo (16-bit storage type): v ~ hi + lo
__global__ void k_split_f32(const float* __restrict__ v, int64_t n, uint16_t* __restrict__ hi, uint16_t* __restrict__ lo) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float4 x = *reinterpret_cast<const float4*>(v + i);
  uint16_t h[4], l[4];
  split2(x.x, h[0], l[0]); split2(x.y, h[1], l[1]); split2(x.z, h[2], l[2]); split2(x.w, h[3], l[3]);
  *reinterpret_cast<uint2*>(hi + i) = make_uint2(h[0] | ((uint32_t)h[1] << 16), h[2] | ((uint32_t)h[3] << 16));
  *reinterpret_cast<uint2*>(lo + i) = make_uint2(l[0] | ((uint32_t)l[1] << 16), l[2] | ((uint32_t)l[3] << 16));
}

hipError_t launch_split_f32(const float* v, int64_t n, uint16_t* hi, uint16_t* lo, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  if (n & 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_split_f32, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, v, n, hi, lo);
  return hipGetLastError();
}

bool gru_x3_applies(const ModelDev& m) {
  if (!m.low_gru || !m.lgru_split || m.tune.lgru_fp32_cells == 1) return false;
  return (m.E % XK_MIN) == 0 && (m.H % XK_MIN) == 0;
}

// Every derived copy of the low level's weights in ONE launch (round 4: after an optimizer step the split, its fragment-major copies and the
// fragment-major copies of the transposes were 10 launches per layer -- a fifth of a 32-lane training step's launches): blockIdx.y picks the
// matrix; thread i takes the i-th 8-element fragment chunk of the fragment-major order (every element exactly once: it also writes the
// row-major split there) and, where asked, the i-th chunk of the transpose's fragment-major order (8 strided reads).
struct LgruRefreshDesc { const float* W; int N, K; uint16_t *hi, *lo, *Fhi, *Flo, *TFhi, *TFlo; };
struct LgruRefreshGroup { LgruRefreshDesc d[2 * HIERTCN_MAX_LAYERS]; };
__global__ void k_lgru_refresh(const LgruRefreshGroup grp) {
  const LgruRefreshDesc& d = grp.d[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int N = d.N, K = d.K;
  if (i >= (int64_t)N * K / 8) return;
  const int lane = (int)(i & 63);
  const int64_t f = i >> 6;
  auto pack8 = [](const uint16_t* v) { return make_uint4(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16), v[6] | ((uint32_t)v[7] << 16)); };
  {
    const int ks = K >> 4;
    const int s = (int)(f % ks), cb = (int)(f / ks);
    const int64_t off = (int64_t)(cb * 32 + (lane & 31)) * K + 16 * s + 8 * (lane >> 5);
    const float4 x0 = *reinterpret_cast<const float4*>(d.W + off), x1 = *reinterpret_cast<const float4*>(d.W + off + 4);
    const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
    uint16_t h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; j++) split2(x[j], h[j], l[j]);
    const uint4 ph = pack8(h), pl = pack8(l);
    *reinterpret_cast<uint4*>(d.hi + off) = ph;
    *reinterpret_cast<uint4*>(d.lo + off) = pl;
    *reinterpret_cast<uint4*>(d.Fhi + i * 8) = ph;
    *reinterpret_cast<uint4*>(d.Flo + i * 8) = pl;
  }
  if (d.TFhi) {
    const int ns = N >> 4;
    const int s = (int)(f % ns), kb = (int)(f / ns);
    uint16_t h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; j++) split2(d.W[(int64_t)(16 * s + 8 * (lane >> 5) + j) * K + 32 * kb + (lane & 31)], h[j], l[j]);
    *reinterpret_cast<uint4*>(d.TFhi + i * 8) = pack8(h);
    *reinterpret_cast<uint4*>(d.TFlo + i * 8) = pack8(l);
  }
}

hipError_t refresh_lgru_split(const ModelDev& m, hipStream_t st) {
  if (!m.low_gru || !m.lgru_split) return hipSuccess;
  hipError_t e;
  {
    bool all = !ABL_ENV("HIERTCN_LGRU_REFRESH_UNFUSED");
    LgruRefreshGroup grp;
    memset(&grp, 0, sizeof(grp));
    int64_t most = 0;
    for (int l = 0; l < m.nl && all; l++) {
      const int K = ((l == 0) ? m.E : m.H) + m.H;
      all = m.lgru_wgF_hi[l] && m.lgru_wcF_hi[l] && (K % 32) == 0 && (m.H % 32) == 0 && (!m.lgru_wgTF_hi[l] == !m.lgru_wcTF_hi[l]);
      grp.d[2 * l] = LgruRefreshDesc{m.lgru_wg[l], 2 * m.H, K, m.lgru_wg_hi[l], m.lgru_wg_lo[l], m.lgru_wgF_hi[l], m.lgru_wgF_lo[l], m.lgru_wgTF_hi[l], m.lgru_wgTF_lo[l]};
      grp.d[2 * l + 1] = LgruRefreshDesc{m.lgru_wc[l], m.H, K, m.lgru_wc_hi[l], m.lgru_wc_lo[l], m.lgru_wcF_hi[l], m.lgru_wcF_lo[l], m.lgru_wcTF_hi[l], m.lgru_wcTF_lo[l]};
      most = most > (int64_t)2 * m.H * K / 8 ? most : (int64_t)2 * m.H * K / 8;
    }
    if (all && m.nl > 0) {
      hipLaunchKernelGGL(k_lgru_refresh, dim3((unsigned)((most + 255) / 256), (unsigned)(2 * m.nl)), dim3(256), 0, st, grp);
      return hipGetLastError();
    }
  }
  for (int l = 0; l < m.nl; l++) {
    const int in = (l == 0) ? m.E : m.H;
    const int64_t ng = (int64_t)(in + m.H) * 2 * m.H, nc = (int64_t)(in + m.H) * m.H;
    if ((e = launch_split_f32(m.lgru_wg[l], ng, m.lgru_wg_hi[l], m.lgru_wg_lo[l], st)) != hipSuccess) return e;
    if ((e = launch_split_f32(m.lgru_wc[l], nc, m.lgru_wc_hi[l], m.lgru_wc_lo[l], st)) != hipSuccess) return e;
    if (m.lgru_wgF_hi[l] && ((in + m.H) % 16) == 0 && (m.H % 32) == 0) {   // fragment-major copies for the session-tile kernel (lgru_tile.hip)
      if ((e = launch_frag_major(m.lgru_wg_hi[l], m.lgru_wgF_hi[l], 1, 2 * m.H, in + m.H, st)) != hipSuccess) return e;
      if ((e = launch_frag_major(m.lgru_wg_lo[l], m.lgru_wgF_lo[l], 1, 2 * m.H, in + m.H, st)) != hipSuccess) return e;
      if ((e = launch_frag_major(m.lgru_wc_hi[l], m.lgru_wcF_hi[l], 1, m.H, in + m.H, st)) != hipSuccess) return e;
      if ((e = launch_frag_major(m.lgru_wc_lo[l], m.lgru_wcF_lo[l], 1, m.H, in + m.H, st)) != hipSuccess) return e;
      if (m.lgru_wgTF_hi[l] && ((in + m.H) % 32) == 0) {   // ... and of the transposes (its backward pass)
        if ((e = launch_frag_major_T(m.lgru_wg_hi[l], m.lgru_wgTF_hi[l], 2 * m.H, in + m.H, st)) != hipSuccess) return e;
        if ((e = launch_frag_major_T(m.lgru_wg_lo[l], m.lgru_wgTF_lo[l], 2 * m.H, in + m.H, st)) != hipSuccess) return e;
        if ((e = launch_frag_major_T(m.lgru_wc_hi[l], m.lgru_wcTF_hi[l], m.H, in + m.H, st)) != hipSuccess) return e;
        if ((e = launch_frag_major_T(m.lgru_wc_lo[l], m.lgru_wcTF_lo[l], m.H, in + m.H, st)) != hipSuccess) return e;
      }
    }
  }
  return hipSuccess;
}
