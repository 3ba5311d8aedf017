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
#include "score_geom.h"

template <int C>
__global__ __launch_bounds__(SC_THREADS, 2) void k_dlogits(const uint16_t* __restrict__ hrows, const float* __restrict__ rowbias, int n_stages,
                                                           const uint16_t* __restrict__ outT, const float* __restrict__ out_bias, int N, int NP,
                                                           uint16_t* __restrict__ gout, int64_t ldg, float* __restrict__ dbias) {
  using G = ScGeom<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 3 stages of 64 rows + 3 x 64 row scalars
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
  constexpr int BIAS_OFF = 3 * G::STAGE_BYTES;
  constexpr int RING = 3;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  auto glds16 = [&](const void* gsrc, unsigned lds_dst) {   // LDS-DMA by hand (score.hip: why asm)
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
  };
  auto stage_issue = [&](int stg, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; i++) {
      const int piece = wave * PPW + i;
      const int off = piece * 1024 + lane * 16;
      const int row = off / (2 * C), slot = (off % (2 * C)) >> 4;
      const uint16_t* src = hrows + ((int64_t)stg * SC_STAGE + row) * C + ((slot ^ (row & G::SWZ)) << 3);
      glds16(src, lds_base + buf * G::STAGE_BYTES + piece * 1024);
    }
    if (wave == 0 && lane < SC_STAGE / 4)
      glds16(rowbias + (int64_t)stg * SC_STAGE + lane * 4, lds_base + BIAS_OFF + buf * SC_STAGE * 4);
  };
  const float* sbias = reinterpret_cast<const float*>(smem + BIAS_OFF);
  const int a_k0 = lr * (2 * C) + ((lh ^ (lr & G::SWZ)) << 4);
  auto bias_init = [&](f32x16& a0, int buf, int ib) {   // accumulator row r starts from ln w_r - lse_r
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + buf * SC_STAGE + ib * 32 + 8 * q + 4 * lh);
#pragma unroll
      for (int j = 0; j < 4; j++) a0[4 * q + j] = b4[j];
    }
  };

  float cs[2] = {0.f, 0.f};   // column sums of this lane's two items over the rows of its half-wave
  float xp = 0.f;             // pipeline: pending exponent
  float g0[16];               // block-0 values of the current row block, waiting for their block-1 neighbours
  float e1[2] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 16; i++) g0[i] = 0.f;
  const unsigned voff = (unsigned)(((int64_t)4 * lh * ldg + item_base + 2 * lr) * 2);   // bytes, < 2^32 (ldg < 2^27)

  // value k of a tile sits in accumulator register k = row (k & 3) + 8 * (k >> 2) + 4 * lh of the 32-row block
  // (uniform row pointer in an SGPR pair) + (per-lane 32-bit byte offset): no VALU address arithmetic per store
  auto put = [&](uint16_t* gtile, int k, uint32_t dw) {
    const uint16_t* rowp = gtile + (int64_t)((k & 3) + 8 * (k >> 2)) * ldg;
    asm volatile("global_store_dword %0, %1, %2" : : "v"(voff), "v"(dw), "s"(rowp));
  };
  auto pack2 = [&](float lo, float hi) -> uint32_t {   // one v_cvt_pk_bf16_f32
    typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
  };
  // gap i of a tile: value i-2 is consumed (column sum; block 1: packed with its block-0 neighbour and stored),
  // value i-1 goes through v_exp, value i through the fma
  auto val = [&](int i, const f32x16& v, int cb, uint16_t* gtile) {
    if (cb == 0) {
      if (i >= 2) cs[0] += g0[i - 2];
      if (i >= 1) g0[i - 1] = __builtin_amdgcn_exp2f(xp);
      xp = fmaf(v[i], LOG2E_F, b2[0]);
      asm volatile("" : "+v"(xp), "+v"(cs[0]));
    } else {
      if (i >= 2) {
        cs[1] += e1[i & 1];
        put(gtile, i - 2, pack2(g0[i - 2], e1[i & 1]));
      }
      if (i >= 1) e1[(i - 1) & 1] = __builtin_amdgcn_exp2f(xp);
      xp = fmaf(v[i], LOG2E_F, b2[1]);
      asm volatile("" : "+v"(xp), "+v"(cs[1]), "+v"(e1[0]), "+v"(e1[1]));
    }
  };
  auto post = [&](int cb, uint16_t* gtile) {
    if (cb == 0) {
      cs[0] += g0[14];
      g0[15] = __builtin_amdgcn_exp2f(xp);
      cs[0] += g0[15];
    } else {
      cs[1] += e1[0];
      put(gtile, 14, pack2(g0[14], e1[0]));
      e1[1] = __builtin_amdgcn_exp2f(xp);
      cs[1] += e1[1];
      put(gtile, 15, pack2(g0[15], e1[1]));
    }
  };

  // One pipeline step: the MFMAs of (row block ib of stage buffer buf) x (item block CBX) into x, with the epilogue of the
  // previous tile y (item block CBY, rows starting at srow) in the gaps.
#define DL_PHASE(x, CBX, y, CBY, buf_, ib_, do_mfma, srow)                                   \
  do {                                                                                       \
    uint16_t* gtile_ = gout + (int64_t)(srow) * ldg;                                         \
    if (do_mfma) {                                                                           \
      const unsigned char* tile_ = smem + (buf_) * G::STAGE_BYTES + (ib_) * 32 * (2 * C);    \
      uint4 af_[RING];                                                                       \
      _Pragma("unroll") for (int s = 0; s < RING - 1; s++)                                   \
        af_[s] = *reinterpret_cast<const uint4*>(tile_ + (a_k0 ^ (s << 5)));                 \
      bias_init(x, buf_, ib_);                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      _Pragma("unroll") for (int s = 0; s < G::KS; s++) {                                    \
        if (s + RING - 1 < G::KS)                                                            \
          af_[(s + RING - 1) % RING] = *reinterpret_cast<const uint4*>(tile_ + (a_k0 ^ ((s + RING - 1) << 5))); \
        x = mfma32(af_[s % RING], hf[CBX][s], x);                                            \
        val(s & 15, y, CBY, gtile_);                                                         \
        __builtin_amdgcn_sched_barrier(0);                                                   \
      }                                                                                      \
      _Pragma("unroll") for (int s = G::KS; s < 16; s++) val(s, y, CBY, gtile_);             \
    } else {                                                                                 \
      _Pragma("unroll") for (int s = 0; s < 16; s++) val(s, y, CBY, gtile_);                 \
    }                                                                                        \
    post(CBY, gtile_);                                                                       \
  } while (0)

  f32x16 accA, accB;
  if (n_stages > 0) stage_issue(0, 0);
  if (n_stages > 1) stage_issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (n_stages > 0) {
    bias_init(accA, 0, 0);
#pragma unroll
    for (int s = 0; s < G::KS; s++) {
      const uint4 a = *reinterpret_cast<const uint4*>(smem + (a_k0 ^ (s << 5)));
      accA = mfma32(a, hf[0][s], accA);
    }
  }
  int buf = 0;
  for (int stg = 0; stg < n_stages; stg++) {
    const int buf1 = (buf == 2) ? 0 : buf + 1;
    const int buf2 = (buf1 == 2) ? 0 : buf1 + 1;
    const int r0 = stg * SC_STAGE;
    if (stg + 2 < n_stages) stage_issue(stg + 2, buf2);
    DL_PHASE(accB, 1, accA, 0, buf, 0, true, r0);                         // MFMA (stg, rows0, items1)   || epilogue (stg, rows0, items0)
    DL_PHASE(accA, 0, accB, 1, buf, 1, true, r0);                         // MFMA (stg, rows1, items0)   || epilogue (stg, rows0, items1) + stores
    DL_PHASE(accB, 1, accA, 0, buf, 1, true, r0 + 32);                    // MFMA (stg, rows1, items1)   || epilogue (stg, rows1, items0)
    DL_PHASE(accA, 0, accB, 1, buf1, 0, stg + 1 < n_stages, r0 + 32);     // MFMA (stg+1, rows0, items0) || epilogue (stg, rows1, items1) + stores
    // vmcnt counts in issue order on gfx9-family parts (loads and stores share the counter): the LDS-DMA of stage stg+2 was issued
    // before this iteration's 32 stores, so "at most 32 outstanding" means it has landed while the stores may still be in flight
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    buf = buf1;
  }
#undef DL_PHASE

#pragma unroll
  for (int b = 0; b < 2; b++) {
    const float tot = cs[b] + __shfl_xor(cs[b], 32);
    const int it = item_base + 2 * lr + b;
    if (lh == 0 && it < N) dbias[it] += tot;   // one owner per item and launch; launches of a step are stream-ordered
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
  const size_t lds = 3 * G::STAGE_BYTES + 3 * SC_STAGE * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_dlogits<C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const unsigned blocks = (unsigned)((m.NP + 511) / 512);
  hipLaunchKernelGGL(k_dlogits<C>, dim3(blocks), dim3(SC_THREADS), lds, st, hrows, rowbias, n_stages, m.outT, m.out_bias, m.N, m.NP, gout, ldg,
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
