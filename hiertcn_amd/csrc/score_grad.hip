// Training backward of the scoring layer, first half: G = d loss / d logits as bf16, never for the whole batch at once.
//   g[r][n] = w_r * (softmax(z_r)[n] - [n == y_r]),  z = h . Wo + bo (model_tcn.py:41), loss.py:21, weights w_r from the
//   lane means of model.py:111-117.
// ITEM-stationary restatement of the forward kernel (score.hip, k_score_pipe) with the two operands swapped: a wave keeps
// the Wo rows of 64 items in registers (MFMA B operand) for the whole launch and the workgroup streams the hidden rows of
// a ROW chunk through the same three-stage LDS ring (asm LDS-DMA, XOR-swizzled, one barrier per 64-row stage).  D[row =
// step][col = item] puts one ITEM on each lane, so
//   * the column sums of G (d bo) are lane-local adds carried in a register across the whole chunk -- no second pass over G;
//   * the accumulator starts from (ln w_r - lse_r), a per-ROW scalar staged beside the rows exactly like the forward
//     kernel's per-item bias, and the item's own bias rides in the exponent's fma: g = 2^((acc) * log2(e) + bo * log2(e)),
//     three VALU ops per logit (fma, exp, add) software-pipelined into the MFMA gaps;
//   * lane n of item block 0 holds item 2n and lane n of block 1 item 2n+1, so one v_cvt_pk_bf16_f32 makes a dword of two
//     neighbouring items and every global_store_dword of a half-wave is one full 128-byte line of a G row.
// The one-hot term touches one element per row: k_dlogits_target rewrites G[r][y_r] (and d bo[y_r]) afterwards from the
// target logit the forward pass saved -- bitwise the value the row's own MFMA chain would give (score.hip file header).
#include <stdlib.h>

#include "score_geom.h"

// v2 of the kernel body (v1: two accumulator tiles, XOR-swizzled image, 114 ms per cfg-3 step): the four-tile seamless shape of the
// forward kernel (score.hip, SC_PHASE3).  One A fragment (32 hidden rows x 16 k) read from LDS feeds the MFMAs of BOTH item blocks,
// so the two halves of every packed dword (items 2n, 2n+1) come out of the same phase; the next row block's (ln w - lse) scalars are
// read straight into the drained accumulator registers and the fragment ring runs on across the tile boundary.  C = 256: padded LDS
// image (fragment k-steps at immediate offsets); C = 64 / 128: XOR-swizzled image.
template <int C>
__global__ __launch_bounds__(SC_THREADS, 2) void k_dlogits(const uint16_t* __restrict__ hrows, const float* __restrict__ rowbias, int n_stages,
                                                           const uint16_t* __restrict__ outT, const float* __restrict__ out_bias, int N, int NP,
                                                           uint16_t* __restrict__ gout, int64_t ldg, float* __restrict__ dbias) {
  using G = ScGeom<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 3 stages of 64 rows + 3 x 64 row scalars
  // gridDim.y > 1 splits the ROW stages of the chunk: the item dimension alone gives N / 512 workgroups (586 at N = 300k: 2.3
  // rounds of 256 CUs, the last one 29 % full); workgroup (x, p) makes the G rows of stages [p * per, (p + 1) * per) for its items
  // and adds its column sums to d bo atomically.
  if (gridDim.y > 1) {
    const int per = (n_stages + (int)gridDim.y - 1) / (int)gridDim.y;
    const int s_lo = (int)blockIdx.y * per;
    const int s_hi = (s_lo + per < n_stages) ? s_lo + per : n_stages;
    if (s_hi <= s_lo) return;
    hrows += (int64_t)s_lo * SC_STAGE * C;
    rowbias += (int64_t)s_lo * SC_STAGE;
    gout += (int64_t)s_lo * SC_STAGE * ldg;
    n_stages = s_hi - s_lo;
  }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int item_base = (blockIdx.x * 8 + wave) * 64;

  uint4 hf[2][G::KS];   // stationary operand: Wo rows of this lane's two items
  float b2[2];          // bo * log2(e) of the two items (-inf for padding items: their g is 0)
#pragma unroll
  for (int b = 0; b < 2; b++) {
    const int it = item_base + 2 * lr + b;
    const int itc = it < NP ? it : NP - 1;
    const uint16_t* wp = outT + (int64_t)itc * C + 8 * lh;
#pragma unroll
    for (int s = 0; s < G::KS; s++) hf[b][s] = *reinterpret_cast<const uint4*>(wp + 16 * s);
    b2[b] = it < NP ? out_bias[itc] * LOG2E_F : -INFINITY;
  }

  constexpr int PIECES = G::STAGE_BYTES / 1024;
  constexpr int PPW = PIECES / 8;
  constexpr bool PADL = C == 256;
  constexpr int PITCH = PADL ? 1024 + 32 : 1024;
  constexpr int STAGE_LDS = PIECES * PITCH;
  constexpr int TILE_LDS = STAGE_LDS / 2;
  constexpr int BIAS_OFF = 3 * STAGE_LDS;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  unsigned dma_voff[PPW];   // byte offset of this lane's 16 B of piece i inside a 64-row stage (source-side swizzle applied), computed once
#pragma unroll
  for (int i = 0; i < PPW; i++) {
    const int off = (wave * PPW + i) * 1024 + lane * 16;
    const int row = off / (2 * C), slot = (off % (2 * C)) >> 4;
    const int sw = PADL ? (row & 1) : (row & G::SWZ);
    dma_voff[i] = (unsigned)(row * (2 * C) + ((slot ^ sw) << 4));
  }
  auto stage_issue = [&](int stg, int buf) {   // LDS-DMA by hand (score.hip: why asm); SGPR stage base + 32-bit lane offset
    const uint16_t* sbase = hrows + (int64_t)stg * SC_STAGE * C;
#pragma unroll
    for (int i = 0; i < PPW; i++) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_LDS + (wave * PPW + i) * PITCH);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(dma_voff[i]), "s"(sbase), "s"(dst)
                   : "memory");
    }
    if (wave == 0 && lane < SC_STAGE / 4) {
      const float* src = rowbias + (int64_t)stg * SC_STAGE + lane * 4;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + BIAS_OFF + buf * SC_STAGE * 4);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src), "s"(dst)
                   : "memory");
    }
  };
  const float* sbias = reinterpret_cast<const float*>(smem + BIAS_OFF);
  const int a_k0 = PADL ? (lr >> 1) * PITCH + (lr & 1) * 512 + ((lh ^ (lr & 1)) << 4) : lr * (2 * C) + ((lh ^ (lr & G::SWZ)) << 4);
#define DL_FRAG(s_) (PADL ? (a_k0 + ((s_) << 5)) : (a_k0 ^ ((s_) << 5)))
  auto bias_init = [&](f32x16& a0, int buf, int rb) {   // accumulator row r starts from ln w_r - lse_r
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + buf * SC_STAGE + rb * 32 + 8 * q + 4 * lh);
#pragma unroll
      for (int j = 0; j < 4; j++) a0[4 * q + j] = b4[j];
    }
  };

  float cs0 = 0.f, cs1 = 0.f;   // column sums of this lane's two items over the rows of its half-wave
  float xp0 = 0.f, xp1 = 0.f, e0 = 0.f, e1 = 0.f;   // pipeline: pending exponents / terms of the current value pair
  const unsigned voff = (unsigned)(((int64_t)4 * lh * ldg + item_base + 2 * lr) * 2);   // bytes, < 2^32 (ldg < 2^27)
  // value k of a tile sits in accumulator register k = row (k & 3) + 8 * (k >> 2) + 4 * lh of the 32-row block;
  // (uniform row pointer in an SGPR pair) + (per-lane 32-bit byte offset): no VALU address arithmetic per store
  auto put = [&](uint16_t* gtile, int k, uint32_t dw) {
    const uint16_t* rowp = gtile + (int64_t)((k & 3) + 8 * (k >> 2)) * ldg;
    asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(voff), "v"(dw), "s"(rowp));   // nt: G is written once, 17 GB per chunk
  };
  auto pack2 = [&](float lo, float hi) -> uint32_t {   // one v_cvt_pk_bf16_f32
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  };
  // value pair i over the two MFMA gaps of k-step i: gap A consumes pair i-1 (column sums, pack, store) and starts pair i (fma),
  // gap B runs pair i through v_exp
  auto gap_a = [&](int i, float v0, float v1, uint16_t* gtile) {
    if (i >= 1) {
      cs0 += e0;
      cs1 += e1;
      put(gtile, i - 1, pack2(e0, e1));
    }
    xp0 = fmaf(v0, LOG2E_F, b2[0]);
    xp1 = fmaf(v1, LOG2E_F, b2[1]);
    asm volatile("" : "+v"(xp0), "+v"(xp1), "+v"(cs0), "+v"(cs1));
  };
  auto gap_b = [&]() {
    e0 = __builtin_amdgcn_exp2f(xp0);
    e1 = __builtin_amdgcn_exp2f(xp1);
    asm volatile("" : "+v"(e0), "+v"(e1));
  };
  auto post = [&](uint16_t* gtile) {
    cs0 += e0;
    cs1 += e1;
    put(gtile, 15, pack2(e0, e1));
  };

  // One phase: the 32 MFMAs of (row block rb of stage buffer buf) x (both item blocks) into (x0, x1); in the gaps the previous row
  // block's tiles (y0, y1; rows from srow) become G, and (y0, y1) are refilled with the next row block's row scalars.
#define DL_PHASE(x0, x1, y0, y1, buf_, rb_, MF, nbuf_, nrb_, srow)                           \
  do {                                                                                       \
    uint16_t* gtile_ = gout + (int64_t)(srow) * ldg;                                         \
    const unsigned char* tile_ = smem + (buf_) * STAGE_LDS + (rb_) * TILE_LDS;               \
    const unsigned char* ntile_ = smem + (nbuf_) * STAGE_LDS + (nrb_) * TILE_LDS;            \
    const float* nbias_ = sbias + (nbuf_) * SC_STAGE + (nrb_) * 32 + 4 * lh;                 \
    _Pragma("unroll") for (int s = 0; s < 16; s++) {                                         \
      if (MF && s < G::KS) {                                                                 \
        ring_[(s + 1) & 1] = *reinterpret_cast<const uint4*>((s + 1 < G::KS) ? tile_ + DL_FRAG(s + 1) : ntile_ + DL_FRAG(0)); \
        x0 = mfma32(ring_[s & 1], hf[0][s], x0);                                             \
      }                                                                                      \
      gap_a(s, y0[s], y1[s], gtile_);                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      if (MF && s < G::KS) x1 = mfma32(ring_[s & 1], hf[1][s], x1);                          \
      gap_b();                                                                               \
      if ((s & 3) == 3) {   /* values 4q..4q+3 of both tiles are dead: next row block's scalars */ \
        const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_ + 8 * (s >> 2));           \
        _Pragma("unroll") for (int j = 0; j < 4; j++) { y0[(s & ~3) + j] = b4_[j]; y1[(s & ~3) + j] = b4_[j]; } \
      }                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                     \
    }                                                                                        \
    post(gtile_);                                                                            \
  } while (0)

  f32x16 xa0, xa1, xb0, xb1;
  uint4 ring_[2];
  if (n_stages > 0) stage_issue(0, 0);
  if (n_stages > 1) stage_issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (n_stages > 0) {
    bias_init(xa0, 0, 0);
    bias_init(xa1, 0, 0);
    bias_init(xb0, 0, 1);
    bias_init(xb1, 0, 1);
#pragma unroll
    for (int s = 0; s < G::KS; s++) {
      const uint4 a = *reinterpret_cast<const uint4*>(smem + DL_FRAG(s));
      xa0 = mfma32(a, hf[0][s], xa0);
      xa1 = mfma32(a, hf[1][s], xa1);
    }
    ring_[0] = *reinterpret_cast<const uint4*>(smem + TILE_LDS + DL_FRAG(0));   // first fragment of (stage 0, rows 32..63)
  }
  int buf = 0;
  for (int stg = 0; stg + 1 < n_stages; stg++) {   // the last stage is peeled: "is there a next tile" stays a compile-time flag
    const int buf1 = (buf == 2) ? 0 : buf + 1;
    const int buf2 = (buf1 == 2) ? 0 : buf1 + 1;
    const int r0 = stg * SC_STAGE;
    if (stg + 2 < n_stages) stage_issue(stg + 2, buf2);
    DL_PHASE(xb0, xb1, xa0, xa1, buf, 1, true, buf1, 0, r0);          // MFMA (stg, rows1)   || G of (stg, rows0);  xa <- scalars (stg+1, rows0)
    DL_PHASE(xa0, xa1, xb0, xb1, buf1, 0, true, buf1, 1, r0 + 32);    // MFMA (stg+1, rows0) || G of (stg, rows1);  xb <- scalars (stg+1, rows1)
    // vmcnt counts in issue order on gfx9-family parts (loads and stores share the counter): the LDS-DMA of stage stg+2 was issued
    // before this iteration's 32 stores, so "at most 32 outstanding" means it has landed while the stores may still be in flight
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    buf = buf1;
  }
  if (n_stages > 0) {
    const int r0 = (n_stages - 1) * SC_STAGE;
    DL_PHASE(xb0, xb1, xa0, xa1, buf, 1, true, buf, 1, r0);           // (the "next tile" reads land in dead registers)
    DL_PHASE(xa0, xa1, xb0, xb1, buf, 1, false, buf, 1, r0 + 32);     // G of the last row block only
  }
#undef DL_PHASE
#undef DL_FRAG

  {
    const float t0 = cs0 + __shfl_xor(cs0, 32), t1 = cs1 + __shfl_xor(cs1, 32);
    const int it = item_base + 2 * lr;
    if (gridDim.y > 1) {
      if (lh == 0 && it < N) atomicAdd(dbias + it, t0);
      if (lh == 0 && it + 1 < N) atomicAdd(dbias + it + 1, t1);
    } else {
      if (lh == 0 && it < N) dbias[it] += t0;   // one owner per item and launch; launches of a step are stream-ordered
      if (lh == 0 && it + 1 < N) dbias[it + 1] += t1;
    }
  }
}

// rowbias[p] = ln w_p - lse_p for the scored rows, -inf for the padding rows up to the next stage boundary (their g is 0)
__global__ void k_dlogits_rowbias(const float* __restrict__ w, const float* __restrict__ lse, int64_t Rv, int64_t Rpad, float* __restrict__ rb) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Rpad) return;
  rb[p] = (p < Rv && w[p] > 0.f) ? logf(w[p]) - lse[p] : -INFINITY;
}

// the one-hot term: G[p - r0][y_p] = bf16(w_p * (softmax_p[y_p] - 1)) and d bo[y_p] -= w_p (k_dlogits added w_p * softmax_p[y_p])
__global__ void k_dlogits_target(const int32_t* __restrict__ yc, const float* __restrict__ wc, const float* __restrict__ lc,
                                 const float* __restrict__ tgt, const int32_t* __restrict__ map, int64_t r0, int64_t r1, int N,
                                 uint16_t* __restrict__ gout, int64_t ldg, float* __restrict__ dbias) {
  const int64_t p = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= r1) return;
  const int yv = yc[p];
  if (yv <= 0 || yv >= N) return;
  const float w = wc[p];
  const float pr = __expf(tgt[map[p]] - lc[p]);
  gout[(p - r0) * ldg + yv] = f2bf(w * (pr - 1.f));
  atomicAdd(dbias + yv, -w);
}

template <int C>
static hipError_t launch_dlogits_c(const ModelDev& m, const uint16_t* hrows, const float* rowbias, int n_stages, uint16_t* gout, int64_t ldg,
                                   float* dbias, hipStream_t st) {
  using G = ScGeom<C>;
  const size_t pitch = C == 256 ? 1024 + 32 : 1024;
  const size_t lds = 3 * (G::STAGE_BYTES / 1024) * pitch + 3 * SC_STAGE * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dlogits<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const unsigned blocks = (unsigned)((m.NP + 511) / 512);
  // row parts: the smallest p <= 8 (each part at least 32 stages) that fills the last round of 256 workgroups to >= 93 %
  unsigned parts = 1;
  if (!ABL_ENV("HIERTCN_DLOGITS_NOSPLIT")) {
    double best = 0.0;
    for (unsigned p = 1; p <= 8 && (p == 1 || n_stages / (int)p >= 32); p++) {
      const double eff = ((double)blocks * p / 256.0) / (double)((blocks * p + 255) / 256);
      if (eff > best + 1e-9) { best = eff; parts = p; }
      if (eff >= 0.93) break;
    }
  }
  hipLaunchKernelGGL(k_dlogits<C>, dim3(blocks, parts), dim3(SC_THREADS), lds, st, hrows, rowbias, n_stages, m.outT, m.out_bias, m.N, m.NP, gout, ldg,
                     dbias);
  return hipGetLastError();
}

bool dlogits_supported(const ModelDev& m) { return m.Clast == 64 || m.Clast == 128 || m.Clast == 256; }
int64_t dlogits_ld(const ModelDev& m) { return ((int64_t)m.NP + 511) & ~(int64_t)511; }

hipError_t launch_dlogits_rowbias(const float* wc, const float* lc, int64_t Rv, int64_t Rpad, float* rb, hipStream_t st) {
  if (Rpad <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_dlogits_rowbias, dim3((unsigned)((Rpad + 255) / 256)), dim3(256), 0, st, wc, lc, Rv, Rpad, rb);
  return hipGetLastError();
}

// G[rows_pad][ldg] for the compact rows [r0, r1) (rows_pad = their count rounded up to 64; hrows / rowbias start at row r0 and are
// readable up to the padded end); d bo accumulated into dbias
hipError_t launch_dlogits_rows(const ModelDev& m, const uint16_t* hc, const float* rb, const int32_t* yc, const float* wc, const float* lc,
                               const float* tgt, const int32_t* map, int64_t r0, int64_t r1, uint16_t* gout, float* dbias, hipStream_t st) {
  if (r1 <= r0) return hipSuccess;
  const int n_stages = (int)((r1 - r0 + 63) / 64);
  const int64_t ldg = dlogits_ld(m);
  hipError_t e;
  switch (m.Clast) {
    case 64: e = launch_dlogits_c<64>(m, hc + r0 * 64, rb + r0, n_stages, gout, ldg, dbias, st); break;
    case 128: e = launch_dlogits_c<128>(m, hc + r0 * 128, rb + r0, n_stages, gout, ldg, dbias, st); break;
    case 256: e = launch_dlogits_c<256>(m, hc + r0 * 256, rb + r0, n_stages, gout, ldg, dbias, st); break;
    default: return hipErrorInvalidValue;
  }
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_dlogits_target, dim3((unsigned)((r1 - r0 + 255) / 256)), dim3(256), 0, st, yc, wc, lc, tgt, map, r0, r1, m.N, gout, ldg,
                     dbias);
  return hipGetLastError();
}
