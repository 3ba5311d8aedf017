// Ranked top-k lists (loss.py:120, tf.nn.top_k order) beside rank / cross entropy (loss.py:21,179) at the headline width C = 256, on the
// FOUR-TILE shape of the loss + rank kernel: one LDS item fragment feeds the MFMAs of both 32-step blocks of a wave.
//
// The four-tile kernel of score.hip (k_score_pipe<256, 14, 2, 3, pad>) runs two waves per SIMD and has 241 of its 256 registers in use: there
// is no room for two per-lane lists (2 x (2 KT + 5) registers), which is why the list flavour there keeps ONE step block per wave
// (k_score_pipe<256, 15, 1, 3, pad, KT>: one LDS fragment read per MFMA, 7.4 VALU per MFMA, matrix pipe 46 % busy, +32 % over loss + rank).
// This kernel takes the other way out: ONE wave per SIMD (256-thread workgroups, one per CU) with the whole 512-register file -- 128 registers
// of resident step rows, four accumulator tiles, two lists, a three-deep fragment ring -- and the single wave hides its own epilogue in the
// 24 free issue cycles of each 32-cycle MFMA (MI355X_MICROARCH.md, one wave per SIMD: issue costs summing to <= 24 cycles per gap hide).
// Same arithmetic as every other 16-bit scoring kernel: bias-initialised accumulators, k-steps in order, the target logit from the same MFMA
// sequence on a gathered table row -- its logits, ranks and lists are bit-identical to theirs (tests/test_gpu_parity.py).
#include "common.h"
#include "score_geom.h"
#include "score_lists.h"

#define T2_THREADS 256
#define T2_C 256

struct T2Row {
  float t, m2, s;
  int c;
  float xp, ep;
};

template <int KT>
__global__ __launch_bounds__(T2_THREADS, 1) void k_score_topk2(const uint16_t* __restrict__ hidden, const int32_t* __restrict__ y, int64_t R,
                                                               const uint16_t* __restrict__ outT, const float* __restrict__ out_bias, int N, int NP,
                                                               int32_t* __restrict__ o_rank, float* __restrict__ o_ce, float* __restrict__ o_tgt,
                                                               const int32_t* __restrict__ row_map, const int32_t* __restrict__ n_rows_dev,
                                                               int32_t* __restrict__ o_topi, float* __restrict__ o_topv, int topk) {
  constexpr int C = T2_C;
  using G = ScGeom<C>;
  constexpr int KS = G::KS;                          // 16 k-steps
  constexpr int PIECES = G::STAGE_BYTES / 1024;      // 32 one-KB LDS-DMA pieces per 64-item stage
  constexpr int PPW = PIECES / 4;                    // 8 per wave
  constexpr int PITCH = 1024 + 32;                   // two 512-byte rows per piece, padded (score.hip, PAD layout at C = 256)
  constexpr int STAGE_LDS = PIECES * PITCH;
  constexpr int TILE_LDS = STAGE_LDS / 2;
  constexpr int NSTG = 3;
  constexpr int BIAS_OFF = NSTG * STAGE_LDS;
  const int64_t Rv = n_rows_dev ? (int64_t)n_rows_dev[0] : R;
  if ((int64_t)blockIdx.x * 256 >= Rv) return;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + wave) * 64;

  uint4 hf[2][KS];
  int myy[2];
  int64_t mysrc[2];
  T2Row st[2];
  TopList<KT> top[2];
#pragma unroll
  for (int cb = 0; cb < 2; cb++) {
    const int64_t r = row0 + cb * 32 + lr;
    const int64_t rp = (r < Rv) ? r : (Rv - 1);
    const int64_t rc = row_map ? (int64_t)row_map[rp] : rp;
    mysrc[cb] = rc;
    const int yy = (r >= Rv) ? 0 : y[rc];
    myy[cb] = (yy > 0 && yy < N) ? yy : 0;
    const uint16_t* hp = hidden + rc * C + 8 * lh;
#pragma unroll
    for (int s = 0; s < KS; s++) hf[cb][s] = *reinterpret_cast<const uint4*>(hp + 16 * s);
    top_init<KT>(top[cb]);
  }
#pragma unroll
  for (int cb = 0; cb < 2; cb++) {   // target logits: the same MFMA sequence as the main loop on the gathered rows of the targets, diagonal kept
    f32x16 acc;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) acc[reg] = out_bias[__shfl(myy[cb], acc_row(reg, lh))];
    const uint16_t* ap = outT + (int64_t)myy[cb] * C + 8 * lh;
#pragma unroll
    for (int s = 0; s < KS; s++) acc = mfma32(*reinterpret_cast<const uint4*>(ap + 16 * s), hf[cb][s], acc);
    const int dreg = (lr & 3) + 4 * (lr >> 3);
    float d = acc[0];
#pragma unroll
    for (int reg = 1; reg < 16; reg++) d = (dreg == reg) ? acc[reg] : d;
    const float other = __shfl_xor(d, 32);
    const float t = (((lr >> 2) & 1) == lh) ? d : other;
    st[cb].t = t;
    st[cb].m2 = t * LOG2E_F + 64.0f;   // fixed reference (score.hip): a sum past fp32 is left as ce = +inf for the fix-up launch
    st[cb].s = 0.f; st[cb].c = 0; st[cb].xp = 0.f; st[cb].ep = 0.f;
  }

  const int n_stages = NP / SC_STAGE;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  unsigned dma_voff[PPW];
#pragma unroll
  for (int i = 0; i < PPW; i++) {
    const int off = (wave * PPW + i) * 1024 + lane * 16;
    const int row = off / (2 * C), slot = (off % (2 * C)) >> 4;
    dma_voff[i] = (unsigned)(row * (2 * C) + ((slot ^ (row & 1)) << 4));
  }
  auto stage_issue = [&](int stg, int buf) {   // (the walk starts at item 0 in every workgroup: equal values must arrive in index order)
    const uint16_t* sbase = outT + (int64_t)stg * SC_STAGE * C;
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
      const float* gsrc = out_bias + (int64_t)stg * SC_STAGE + lane * 4;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + BIAS_OFF + buf * SC_STAGE * 4);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(gsrc), "s"(dst)
                   : "memory");
    }
  };
  const float* sbias = reinterpret_cast<const float*>(smem + BIAS_OFF);
  const int a_k0 = (lr >> 1) * PITCH + (lr & 1) * 512 + ((lh ^ (lr & 1)) << 4);
#define T2_FRAG(s_) (a_k0 + ((s_) << 5))
#define T2_LDS(p_) (*reinterpret_cast<const uint4*>(p_))
  auto bias_init = [&](f32x16& a0, int buf, int ib) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + buf * SC_STAGE + ib * 32 + 8 * q + 4 * lh);
#pragma unroll
      for (int j = 0; j < 4; j++) a0[4 * q + j] = b4[j];
    }
  };
  auto tile_ptr = [&](int buf, int ib) { return smem + buf * STAGE_LDS + ib * TILE_LDS; };
  // one value of the epilogue, software-pipelined over three k-steps (score.hip, val5)
  auto val = [&](int i, float v, T2Row& rs) {
    const bool gt = v > rs.t;
    if (i >= 2) rs.s += rs.ep;
    if (i >= 1) rs.ep = __builtin_amdgcn_exp2f(rs.xp);
    rs.xp = fmaf(v, LOG2E_F, -rs.m2);
    rs.c += gt ? 1 : 0;
    asm volatile("" : "+v"(rs.s), "+v"(rs.c), "+v"(rs.ep), "+v"(rs.xp));
  };
  auto post = [&](T2Row& rs) {
    rs.s += rs.ep;
    rs.s += __builtin_amdgcn_exp2f(rs.xp);
  };

  // A phase: the 32 MFMAs of (one 32-item block) x (both step blocks) into x0 / x1, each LDS fragment feeding two MFMAs, with the 32 values of
  // the previous item block (y0 / y1) drained in their gaps.  Seamless tile boundary as in score.hip's four-tile kernel: a drained value is
  // dead, so y takes the NEXT tile's bias four registers at a time and is the next phase's accumulator; the fragment ring (read two k-steps =
  // four MFMAs ahead: a single wave has no partner to cover the LDS latency) runs on into the next tile.  Lists: each tile's maximum is folded
  // over k-steps 0..3 and one wave vote in step 3 decides whether any lane has a candidate.
  uint4 ring[4];
  float tk0 = -INFINITY, tk1 = -INFINITY;
#define T2_STEP(x0, x1, y0, y1, s)                                                                              \
  if (MF_) {                                                                                                   \
    ring[((s) + 2) & 3] = T2_LDS((((s) + 2 < KS) ? tile_ : ntile_) + T2_FRAG(((s) + 2) % KS));                  \
    x0 = mfma32(ring[(s) & 3], hf[0][s], x0);                                                                  \
  }                                                                                                            \
  val(s, y0[s], st[0]);
#define T2_STEP_B(x0, x1, y0, y1, s)                                                                            \
  __builtin_amdgcn_sched_barrier(0);                                                                           \
  if (MF_) x1 = mfma32(ring[(s) & 3], hf[1][s], x1);                                                           \
  val(s, y1[s], st[1]);
#define T2_PHASE(x0, x1, y0, y1, tile__, ntile__, nbias__, MF, dit_)                                             \
  do {                                                                                                         \
    constexpr bool MF_ = MF;                                                                                   \
    const unsigned char* tile_ = tile__;                                                                       \
    const unsigned char* ntile_ = ntile__;                                                                     \
    const float* nbias_ = nbias__ + 4 * lh;                                                                    \
    (void)tile_; (void)ntile_;                                                                                 \
    T2_STEP(x0, x1, y0, y1, 0) tk0 = fmaxf(fmaxf(fmaxf(y0[0], y0[1]), y0[2]), fmaxf(y0[3], y0[4]));            \
    T2_STEP_B(x0, x1, y0, y1, 0) tk1 = fmaxf(fmaxf(fmaxf(y1[0], y1[1]), y1[2]), fmaxf(y1[3], y1[4]));          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    T2_STEP(x0, x1, y0, y1, 1) tk0 = fmaxf(fmaxf(fmaxf(tk0, y0[5]), y0[6]), fmaxf(y0[7], y0[8]));              \
    T2_STEP_B(x0, x1, y0, y1, 1) tk1 = fmaxf(fmaxf(fmaxf(tk1, y1[5]), y1[6]), fmaxf(y1[7], y1[8]));            \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    T2_STEP(x0, x1, y0, y1, 2) tk0 = fmaxf(fmaxf(fmaxf(tk0, y0[9]), y0[10]), fmaxf(y0[11], y0[12]));           \
    T2_STEP_B(x0, x1, y0, y1, 2) tk1 = fmaxf(fmaxf(fmaxf(tk1, y1[9]), y1[10]), fmaxf(y1[11], y1[12]));         \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    T2_STEP(x0, x1, y0, y1, 3)                                                                                 \
    tk0 = fmaxf(fmaxf(fmaxf(tk0, y0[13]), y0[14]), y0[15]);                                                    \
    if (__any(tk0 > top[0].v[KT - 1])) top_tile_slow<KT>(top[0], y0, dit_);                                    \
    {                                                                                                          \
      const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_);                                               \
      _Pragma("unroll") for (int j = 0; j < 4; j++) y0[j] = b4_[j];                                            \
    }                                                                                                          \
    T2_STEP_B(x0, x1, y0, y1, 3)                                                                               \
    tk1 = fmaxf(fmaxf(fmaxf(tk1, y1[13]), y1[14]), y1[15]);                                                    \
    if (__any(tk1 > top[1].v[KT - 1])) top_tile_slow<KT>(top[1], y1, dit_);                                    \
    {                                                                                                          \
      const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_);                                               \
      _Pragma("unroll") for (int j = 0; j < 4; j++) y1[j] = b4_[j];                                            \
    }                                                                                                          \
    __builtin_amdgcn_sched_barrier(0);                                                                         \
    _Pragma("unroll") for (int s = 4; s < 16; s++) {                                                           \
      T2_STEP(x0, x1, y0, y1, s)                                                                               \
      if ((s & 3) == 3) {                                                                                      \
        const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_ + 8 * (s >> 2));                             \
        _Pragma("unroll") for (int j = 0; j < 4; j++) y0[(s & ~3) + j] = b4_[j];                               \
      }                                                                                                        \
      T2_STEP_B(x0, x1, y0, y1, s)                                                                             \
      if ((s & 3) == 3) {                                                                                      \
        const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_ + 8 * (s >> 2));                             \
        _Pragma("unroll") for (int j = 0; j < 4; j++) y1[(s & ~3) + j] = b4_[j];                               \
      }                                                                                                        \
      __builtin_amdgcn_sched_barrier(0);                                                                       \
    }                                                                                                          \
    post(st[0]);                                                                                               \
    post(st[1]);                                                                                               \
  } while (0)

  f32x16 xa0, xa1, xb0, xb1;
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
    for (int s = 0; s < KS; s++) {
      const uint4 a = T2_LDS(smem + T2_FRAG(s));
      xa0 = mfma32(a, hf[0][s], xa0);
      xa1 = mfma32(a, hf[1][s], xa1);
    }
    ring[0] = T2_LDS(tile_ptr(0, 1) + T2_FRAG(0));   // first fragments of (stage 0, item block 1)
    ring[1] = T2_LDS(tile_ptr(0, 1) + T2_FRAG(1));
  }
  int buf = 0;
  for (int stg = 0; stg + 1 < n_stages; stg++) {   // the last stage is peeled: no run-time "is there a next tile" inside the phases
    const int buf1 = (buf == 2) ? 0 : buf + 1;
    const int buf2 = (buf1 == 2) ? 0 : buf1 + 1;
    if (stg + 2 < n_stages) stage_issue(stg + 2, buf2);   // buf2 held stage stg - 1: its last readers finished before the barrier that ended iteration stg - 1
    T2_PHASE(xb0, xb1, xa0, xa1, tile_ptr(buf, 1), tile_ptr(buf1, 0), sbias + buf1 * SC_STAGE, true, stg * SC_STAGE + 4 * lh);
    T2_PHASE(xa0, xa1, xb0, xb1, tile_ptr(buf1, 0), tile_ptr(buf1, 1), sbias + buf1 * SC_STAGE + 32, true, stg * SC_STAGE + 32 + 4 * lh);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    buf = buf1;
  }
  if (n_stages > 0) {
    T2_PHASE(xb0, xb1, xa0, xa1, tile_ptr(buf, 1), tile_ptr(buf, 1), sbias + buf * SC_STAGE + 32, true, (n_stages - 1) * SC_STAGE + 4 * lh);
    T2_PHASE(xa0, xa1, xb0, xb1, tile_ptr(buf, 1), tile_ptr(buf, 1), sbias + buf * SC_STAGE + 32, false, (n_stages - 1) * SC_STAGE + 32 + 4 * lh);
  }
#undef T2_PHASE
#undef T2_STEP
#undef T2_STEP_B
#undef T2_FRAG
#undef T2_LDS

  float* lv = reinterpret_cast<float*>(smem);                       // [256 lanes][KT]
  int* li = reinterpret_cast<int*>(smem + T2_THREADS * KT * 4);     // [256 lanes][KT]
#pragma unroll
  for (int cb = 0; cb < 2; cb++) {
    const int64_t myr = row0 + cb * 32 + lr;
    const float l_own = st[cb].m2 + log2f(st[cb].s);
    const float l_o = __shfl_xor(l_own, 32);
    const int c_o = __shfl_xor(st[cb].c, 32);
    const float s_other = __shfl_xor(st[cb].s, 32);
    const float M2 = fmaxf(l_own, l_o);
    const float S = __builtin_amdgcn_exp2f(l_own - M2) + __builtin_amdgcn_exp2f(l_o - M2);
    const bool valid = myy[cb] > 0;
    if (lh == 0 && myr < Rv) {
      float lse = (M2 + log2f(S)) * LN2_F;
      if ((__float_as_uint(st[cb].s) & 0x7f800000u) == 0x7f800000u || (__float_as_uint(s_other) & 0x7f800000u) == 0x7f800000u) lse = INFINITY;
      if (o_rank) o_rank[mysrc[cb]] = valid ? st[cb].c + c_o : 0;
      if (o_ce) o_ce[mysrc[cb]] = valid ? (lse - st[cb].t) : 0.f;
      if (o_tgt) o_tgt[mysrc[cb]] = valid ? st[cb].t : 0.f;
    }
    // the two half-waves' sorted lists (value descending, then index ascending) merged through LDS, which the item stream no longer needs
    top_flush<KT>(top[cb]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KT; k++) { lv[tid * KT + k] = top[cb].v[k]; li[tid * KT + k] = top[cb].i[k]; }
    __syncthreads();
    if (lh == 0 && myr < Rv) {
      const float* av = lv + tid * KT; const int* ai = li + tid * KT;
      const float* bv = lv + (tid + 32) * KT; const int* bi = li + (tid + 32) * KT;
      int pa = 0, pb = 0;
      for (int k = 0; k < topk; k++) {
        // masked rows are all-zero logit rows (model.py:105): every value ties at 0 -> indices 0..k-1
        const float va = pa < KT ? av[pa] : -INFINITY, vb = pb < KT ? bv[pb] : -INFINITY;
        const int ia = pa < KT ? ai[pa] : 0x7fffffff, ib = pb < KT ? bi[pb] : 0x7fffffff;
        const bool ta = (va > vb) || (va == vb && ia < ib);
        o_topi[mysrc[cb] * topk + k] = valid ? (ta ? ia : ib) : k;
        o_topv[mysrc[cb] * topk + k] = valid ? (ta ? va : vb) : 0.f;
        pa += ta ? 1 : 0;
        pb += ta ? 0 : 1;
      }
    }
  }
}

template <int KT>
static hipError_t launch_topk2_k(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out, hipStream_t st,
                                 const int32_t* row_map, const int32_t* n_rows_dev) {
  using G = ScGeom<T2_C>;
  size_t lds = 3 * (size_t)(G::STAGE_BYTES / 1024) * (1024 + 32) + 3 * SC_STAGE * sizeof(float);
  const size_t merge = (size_t)T2_THREADS * KT * 8;
  if (merge > lds) lds = merge;
  auto kern = k_score_topk2<KT>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)((R + 255) / 256)), dim3(T2_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP, out->rank, out->ce,
                     out->target, row_map, n_rows_dev, out->topk_idx, out->topk_val, (int)out->topk);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return out->ce ? launch_score_fixup_only(m, hidden, y, R, out, st, row_map, n_rows_dev) : hipSuccess;
}

// C_last == 256 only (the caller checks); out->topk in 1..16, out->rank and out->ce given
hipError_t launch_score_topk2(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out, hipStream_t st,
                              const int32_t* row_map, const int32_t* n_rows_dev) {
  if (R <= 0) return hipSuccess;
  if (out->topk <= 8) return launch_topk2_k<8>(m, hidden, y, R, out, st, row_map, n_rows_dev);
  if (out->topk <= 10) return launch_topk2_k<10>(m, hidden, y, R, out, st, row_map, n_rows_dev);
  return launch_topk2_k<16>(m, hidden, y, R, out, st, row_map, n_rows_dev);
}
