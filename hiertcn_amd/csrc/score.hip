// Full-catalogue scoring with a fused epilogue: model_tcn.py:41 (logits = h . Wo + bo),
// model.py:105 (row mask), loss.py:21 (cross entropy), loss.py:179 (rank = #{n : z_n > z_target}).
// The [rows, N] logits are never written (hiertcn_score_logits instantiates the same main loop
// with a store instead of the epilogue, for parity tests only).
//
// Decomposition: a workgroup of 8 waves owns 512 rows (time steps); each wave keeps the bf16
// activations of its 64 rows in registers as the MFMA B operand for the whole kernel and streams
// the item table Wo^T [N][C] (internal layout: item-major, 2C bytes per item) through a
// double-buffered LDS stage of 64 items shared by the 8 waves.  MFMA orientation is "swapped":
// A = item rows, B = steps^T, so D[row = item][col = step] puts ONE step on each lane and the
// per-step reductions (count-greater, running max, sum of exp) are lane-local VALU work with no
// cross-lane traffic until the final merge of the two half-waves.
//
// The target logit z[y] is produced by the SAME MFMA sequence (bias-initialised accumulator,
// k-steps in the same order) on a gathered item row, so `z_n > z_target` is exactly false for
// n == y, as in the reference where both come from one tensor (loss.py:179).
#include "common.h"

#include "score_geom.h"
#include "score_lists.h"

// Catalogue parts for a small batch: enough (row tile, part) workgroups to cover the 256 CUs about once, at least 16
// stages (1024 items) per part, at most 32 parts.  1 = no split (large batches fill the chip with row tiles alone).
int score_split_parts(int64_t R, int NP, int rows_per_wg) {
  const int64_t tiles = (R + rows_per_wg - 1) / rows_per_wg;
  const int64_t by_len = (NP / SC_STAGE) / 16;
  if (tiles < 128) {
    int64_t parts = 256 / tiles;   // floor: tiles x parts workgroups must fit ONE round of 256 (13 tiles x 20 parts = 260 took two)
    if (parts > by_len) parts = by_len;
    if (parts > 32) parts = 32;
    return parts < 1 ? 1 : (int)parts;
  }
  // Mid-size batches: a few rounds of 256 workgroups, the last one partly empty (600 row tiles = 2.34 rounds: the chip idles for
  // 22 % of the launch).  Splitting the catalogue in p parts makes p x more, p x shorter workgroups: take the smallest p <= 4 whose
  // last round is at least 93 % full, else the best one.
  if (tiles >= 2048 || ABL_ENV("HIERTCN_NO_MIDSPLIT")) return 1;
  int best = 1;
  double best_eff = 0.0;
  for (int p = 1; p <= 4 && p <= by_len; p++) {
    const double rounds = (double)(tiles * p) / 256.0;
    const double eff = rounds / (double)((tiles * p + 255) / 256);
    if (eff >= 0.93) return p;
    if (eff > best_eff) { best_eff = eff; best = p; }
  }
  return best;
}

static bool score_dyn_applies(int64_t R, int NP, int rows_per_wg);
int score_split_scratch_parts(int64_t R, int NP, int rows_per_wg) {
  const int parts = score_split_parts(R, NP, rows_per_wg);
  if (rows_per_wg == 512 && score_dyn_applies(R, NP, 1024)) return parts > SC_DYN_MAX_PARTS ? parts : SC_DYN_MAX_PARTS;   // (the two-block kernel's device-side split)
  return parts > 1 ? (parts > SC_DYN_MAX_PARTS ? parts : SC_DYN_MAX_PARTS) : 1;
}
// small batch = the device-side split applies: fewer than 128 row tiles at the host's bound (score_split_parts' first branch) and a split at all.
// The kernels pass TWICE their rows per workgroup: the 256-workgroup grid deals one (row tile, part) to each workgroup, so up to 256 live tiles fit --
// one-block workgroups (256 rows) up to 65,024 rows, two-block ones (512 rows) up to 130,048.  The bound is the host's PADDED row count; the live
// count (ragged sessions: about half of it) is only known on the device, which is why a static split there misjudges the grid (512 lanes x 10 x 20:
// 200 padded tiles, 105 live ones, one part = 105 workgroups of the whole catalogue on 256 CUs: 0.73 ms of a 1.46 ms batch).
static bool score_dyn_applies(int64_t R, int NP, int rows_per_wg) {
  static const bool off = ABL_ENV("HIERTCN_SCORE_DYN_OFF");
  return !off && (R + rows_per_wg - 1) / rows_per_wg < 128 && score_split_parts(R, NP, rows_per_wg) > 1;
}

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

// diagnostic cycle stamp (separate STAMP build only; cdna_hip_programming.md 7 'In-kernel stamps')
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define SC_NSTAMP 8

struct RowState {
  float t;    // target logit
  float m2;   // reference point of the running softmax sum, in log2 units (>= max logit * log2(e) - SC_DEFER)
  float s;    // sum of 2^(z*log2(e) - m2)
  int cnt;    // #{z > t}
};

#define SC_DEFER 8.0f  // rescale only when a logit exceeds the reference by 2^8 (T13): exact in fp32, rarely taken

// MODE: 0 = full (production); 1 = rank only; 2 = no epilogue (accumulators kept alive).  Modes 1/2 exist for
// the ablation runs recorded in profiles/ (HIERTCN_SCORE_VARIANT); they produce wrong loss values by design.
template <int MODE>
__device__ __forceinline__ void epilogue(const f32x16& v, RowState& st) {
  if (MODE == 2) {
#pragma unroll
    for (int i = 0; i < 16; i++) asm volatile("" ::"v"(v[i]));
    return;
  }
  if (MODE == 1) {
#pragma unroll
    for (int i = 0; i < 16; i++) st.cnt += (v[i] > st.t) ? 1 : 0;
    return;
  }
  if (MODE == 3) {   // mixture head (gmm.hip): st.s = sum of squares of the row's outputs (|mu_m|^2 of loss.py:247-257's Gaussians)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 4) {
      s0 = fmaf(v[i], v[i], s0);
      s1 = fmaf(v[i + 1], v[i + 1], s1);
      s2 = fmaf(v[i + 2], v[i + 2], s2);
      s3 = fmaf(v[i + 3], v[i + 3], s3);
    }
    st.s += (s0 + s1) + (s2 + s3);
    return;
  }
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = fmaf(v[i], LOG2E_F, -st.m2);  // fma results are canonical: no v_max canonicalisation below
  float xm = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
  xm = fmaxf(xm, fmaxf(fmaxf(fmaxf(x[8], x[9]), fmaxf(x[10], x[11])), fmaxf(fmaxf(x[12], x[13]), fmaxf(x[14], x[15]))));
  if (__any(xm > SC_DEFER)) {  // wave-uniform, rare once the reference has settled
    const float d = fmaxf(xm, 0.f);
    st.s *= __builtin_amdgcn_exp2f(-d);
    st.m2 += d;
#pragma unroll
    for (int i = 0; i < 16; i++) x[i] -= d;
  }
  // four independent partial sums / counters: a single accumulation chain is latency-bound (one wave
  // issues a dependent VALU op only every ~6.6 cycles instead of 4)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
#pragma unroll
  for (int i = 0; i < 16; i += 4) {
    s0 += __builtin_amdgcn_exp2f(x[i]);
    s1 += __builtin_amdgcn_exp2f(x[i + 1]);
    s2 += __builtin_amdgcn_exp2f(x[i + 2]);
    s3 += __builtin_amdgcn_exp2f(x[i + 3]);
    c0 += (v[i] > st.t) ? 1 : 0;
    c1 += (v[i + 1] > st.t) ? 1 : 0;
    c2 += (v[i + 2] > st.t) ? 1 : 0;
    c3 += (v[i + 3] > st.t) ? 1 : 0;
  }
  st.s += (s0 + s1) + (s2 + s3);
  st.cnt += (c0 + c1) + (c2 + c3);
}

// NB = step blocks of 32 rows per wave: 2 everywhere except the top-k build, whose per-lane lists would not fit
// the 256-VGPR budget of a 512-thread workgroup beside two resident step blocks.
template <int C, bool LOGITS, int MODE, bool STAMP = false, int KT = 0, int NB = 2>
__global__ __launch_bounds__(SC_THREADS, 2) void k_score(const uint16_t* __restrict__ hidden, const int32_t* __restrict__ y,
                                                         int64_t R, const uint16_t* __restrict__ outT,
                                                         const float* __restrict__ out_bias, int N, int NP,
                                                         int32_t* __restrict__ o_rank, float* __restrict__ o_ce,
                                                         float* __restrict__ o_tgt, float* __restrict__ logits, int stagger,
                                                         unsigned long long* __restrict__ stamps = nullptr,
                                                         int32_t* __restrict__ o_topi = nullptr, float* __restrict__ o_topv = nullptr,
                                                         int topk = 0, const float* __restrict__ g_lse = nullptr,
                                                         const float* __restrict__ g_w = nullptr, int stg_begin = 0, int stg_end = -1,
                                                         const int32_t* __restrict__ row_map = nullptr,
                                                         const int32_t* __restrict__ n_rows_dev = nullptr) {
  // row_map / n_rows_dev (fused and top-k flavours): compact position p stands for row row_map[p], as in k_score_pipe.
  const int64_t Rin = R;
  if (n_rows_dev) R = (int64_t)n_rows_dev[0];
  if ((int64_t)blockIdx.x * (256 * NB) >= R && Rin > 0) return;
  // LOGITS && MODE == 9: gradient build (training).  Instead of fp32 logits the kernel writes, for the stages
  // [stg_begin, stg_end) only, G[r][n] = w_r * (exp(z_rn - lse_r) - [n == y_r]) as bf16 into `logits` viewed as
  // uint16_t [R][(stg_end - stg_begin) * 64]: d loss / d z of model.py:111-117 + loss.py:21, never held for all N at once.
  constexpr bool GRAD = LOGITS && MODE == 9;
  using G = ScGeom<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 stages + 2 x 64 bias floats
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t row0 = ((int64_t)blockIdx.x * 8 + wave) * (32 * NB);
  // Waves w and w+4 share a SIMD.  Waves 4-7 run half a tile behind waves 0-3 (they defer the epilogue of a
  // stage's second item block past the barrier), so that on every SIMD one wave's MFMA burst runs beside the
  // other's VALU epilogue instead of both bursts colliding after each barrier (MI355X_MICROARCH.md, two waves
  // per SIMD, item 9).
  const bool late = !LOGITS && KT == 0 && stagger && wave >= 4;
  constexpr int KTA = KT > 0 ? KT : 1;
  TopList<KTA> top[NB];
  if (KT > 0) {
#pragma unroll
    for (int cbk = 0; cbk < NB; cbk++) top_init<KTA>(top[cbk]);
  }

  // ---- this wave's 64 steps as the B operand, resident for the whole kernel -------------------
  uint4 hf[NB][G::KS];
  int64_t myrow[NB];
  int myy[NB];
#pragma unroll
  for (int cbk = 0; cbk < NB; cbk++) {
    int64_t r = row0 + cbk * 32 + lr;
    const int64_t rp = (r < R) ? r : (R - 1);
    const int64_t rc = row_map ? (int64_t)row_map[rp] : rp;
    myrow[cbk] = (r < R) ? rc : Rin;   // output / input row; Rin = "no row" (fails every `< R` test below, R <= Rin)
    int yy = ((LOGITS && !GRAD) || r >= R) ? 0 : y[rc];
    myy[cbk] = (yy > 0 && yy < N) ? yy : 0;
    const uint16_t* hp = hidden + rc * C + 8 * lh;
#pragma unroll
    for (int s = 0; s < G::KS; s++) hf[cbk][s] = *reinterpret_cast<const uint4*>(hp + 16 * s);
  }
  float my_lse[NB], my_w[NB];
#pragma unroll
  for (int cbk = 0; cbk < NB; cbk++) {
    const bool live = GRAD && myrow[cbk] < Rin && myy[cbk] > 0;
    my_lse[cbk] = live ? g_lse[myrow[cbk]] : 0.f;
    my_w[cbk] = live ? g_w[myrow[cbk]] : 0.f;
  }

  RowState st[NB];
  if (!LOGITS) {
    // ---- target logits: D[i][j] = bo[y_i] + Wo^T[y_i] . h_j, keep the diagonal ----------------
#pragma unroll
    for (int cbk = 0; cbk < NB; cbk++) {
      f32x16 acc;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) {
        const int yi = __shfl(myy[cbk], acc_row(reg, lh));  // y of step i = tile row i (lanes 0..31 hold it)
        acc[reg] = out_bias[yi];
      }
      const uint16_t* ap = outT + (int64_t)myy[cbk] * C + 8 * lh;  // lane (r,h) feeds A row r = item y_r
#pragma unroll
      for (int s = 0; s < G::KS; s++) {
        const uint4 a = *reinterpret_cast<const uint4*>(ap + 16 * s);
        acc = mfma32(a, hf[cbk][s], acc);
      }
      // diagonal element (i == j == lr) sits in half (lr>>2)&1, register (lr&3) + 4*(lr>>3)
      const int dreg = (lr & 3) + 4 * (lr >> 3);
      float d = acc[0];
#pragma unroll
      for (int reg = 1; reg < 16; reg++) d = (dreg == reg) ? acc[reg] : d;
      const float other = __shfl_xor(d, 32);
      const float t = (((lr >> 2) & 1) == lh) ? d : other;
      st[cbk].t = t;
      st[cbk].m2 = t * LOG2E_F;
      st[cbk].s = 0.f;
      st[cbk].cnt = 0;
    }
  }

  // ---- stream the item table: LDS-DMA (global_load_lds, 16 B per lane, 1 KiB per wave instruction) -------
  // The LDS image is lane-linear per instruction, so the bank swizzle is applied to the SOURCE address
  // (cdna_hip_programming.md rule 21): LDS slot (row, c) holds global chunk c ^ (row & SWZ) of that row.
  const int n_stages = NP / SC_STAGE;
  constexpr int PIECES = G::STAGE_BYTES / 1024;  // wave instructions per stage
  constexpr int PPW = PIECES / 8;                // per wave
  static_assert(PIECES % 8 == 0, "stage must split over 8 waves");
  auto stage_issue = [&](int stg, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; i++) {
      const int piece = wave * PPW + i;
      const int off = piece * 1024 + lane * 16;          // byte offset inside the stage image
      const int row = off / (2 * C), slot = (off % (2 * C)) >> 4;
      const uint16_t* src = outT + ((int64_t)stg * SC_STAGE + row) * C + ((slot ^ (row & G::SWZ)) << 3);
      __builtin_amdgcn_global_load_lds((gbl_void*)src, (lds_void*)(smem + buf * G::STAGE_BYTES + piece * 1024), 16, 0, 0);
    }
    if (wave == 0 && lane < SC_STAGE / 4)
      __builtin_amdgcn_global_load_lds((gbl_void*)(out_bias + (int64_t)stg * SC_STAGE + lane * 4),
                                       (lds_void*)(smem + 2 * G::STAGE_BYTES + buf * SC_STAGE * 4), 16, 0, 0);
  };
  const float* sbias = reinterpret_cast<const float*>(smem + 2 * G::STAGE_BYTES);

  f32x16 acc0, acc1;
  auto mfma_tile = [&](int buf, int ib) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + buf * SC_STAGE + ib * 32 + 8 * q + 4 * lh);
#pragma unroll
      for (int j = 0; j < 4; j++) {
        acc0[4 * q + j] = b4[j];
        if (NB == 2) acc1[4 * q + j] = b4[j];
      }
    }
    const unsigned char* arow = smem + buf * G::STAGE_BYTES + (ib * 32 + lr) * (2 * C);
#pragma unroll
    for (int s = 0; s < G::KS; s++) {
      const uint4 a = *reinterpret_cast<const uint4*>(arow + (((2 * s + lh) ^ (lr & G::SWZ)) << 4));
      acc0 = mfma32(a, hf[0][s], acc0);
      if (NB == 2) acc1 = mfma32(a, hf[NB - 1][s], acc1);
    }
  };
  auto store_logits = [&](int stg, int ib) {
    const int item0 = stg * SC_STAGE + ib * 32 + 4 * lh;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int it = item0 + 8 * q;
#pragma unroll
      for (int j = 0; j < 4; j++) {
        if (it + j < N) {
          if (myrow[0] < Rin) logits[myrow[0] * N + it + j] = acc0[4 * q + j];
          if (NB == 2 && myrow[NB - 1] < Rin) logits[myrow[NB - 1] * N + it + j] = acc1[4 * q + j];
        }
      }
    }
  };

  auto store_grad = [&](int stg, int ib) {
    uint16_t* gout = reinterpret_cast<uint16_t*>(logits);
    const int64_t NC = (int64_t)(stg_end - stg_begin) * SC_STAGE;
    const int item0 = stg * SC_STAGE + ib * 32 + 4 * lh;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int it = item0 + 8 * q;
#pragma unroll
      for (int cbk = 0; cbk < NB; cbk++) {
        if (myrow[cbk] >= Rin) continue;
        const f32x16& a = (cbk == 0) ? acc0 : acc1;
        float g[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float p = __expf(a[4 * q + j] - my_lse[cbk]);      // softmax probability (pad items: exp(-inf) = 0)
          g[j] = my_w[cbk] != 0.f ? my_w[cbk] * (p - ((it + j == myy[cbk]) ? 1.f : 0.f)) : 0.f;
        }
        uint2 o;
        o.x = pack2bf(g[0], g[1]);
        o.y = pack2bf(g[2], g[3]);
        *reinterpret_cast<uint2*>(gout + myrow[cbk] * NC + (it - stg_begin * SC_STAGE)) = o;
      }
    }
  };
  const int stg_last = (GRAD && stg_end >= 0) ? stg_end : n_stages;
  const int stg_first = GRAD ? stg_begin : 0;

  stage_issue(stg_first, 0);
  __syncthreads();  // (hipcc drains vmcnt(0) here because an LDS-DMA is in flight)

  unsigned long long tsum[SC_NSTAMP];
#pragma unroll
  for (int i = 0; i < SC_NSTAMP; i++) tsum[i] = 0;
#define STAMP_AT(i)                         \
  if (STAMP) {                              \
    const unsigned long long tn_ = stamp(); \
    tsum[i] += tn_ - tprev;                 \
    tprev = tn_;                            \
  }
  unsigned long long tprev = STAMP ? stamp() : 0;
  for (int stg = stg_first; stg < stg_last; stg++) {
    const int buf = (stg - stg_first) & 1;
    // buf^1 was last read (by MFMA operand / bias reads) before the barrier that ended stage stg-1
    if (stg + 1 < stg_last) stage_issue(stg + 1, buf ^ 1);
    STAMP_AT(0)
    if (late && stg > 0) {  // deferred epilogue of the previous stage's second item block
      epilogue<MODE>(acc0, st[0]);
      if (NB == 2) epilogue<MODE>(acc1, st[NB - 1]);
    }
    STAMP_AT(1)
    mfma_tile(buf, 0);
    STAMP_AT(2)
    if (GRAD) {
      store_grad(stg, 0);
    } else if (LOGITS) {
      store_logits(stg, 0);
    } else {
      epilogue<MODE>(acc0, st[0]);
      if (NB == 2) epilogue<MODE>(acc1, st[NB - 1]);
      if (KT > 0) {
        top_tile<KTA>(top[0], acc0, stg * SC_STAGE + 4 * lh);
        if (NB == 2) top_tile<KTA>(top[NB - 1], acc1, stg * SC_STAGE + 4 * lh);
      }
    }
    STAMP_AT(3)
    mfma_tile(buf, 1);
    STAMP_AT(4)
    if (GRAD) {
      store_grad(stg, 1);
    } else if (LOGITS) {
      store_logits(stg, 1);
    } else if (!late) {
      epilogue<MODE>(acc0, st[0]);
      if (NB == 2) epilogue<MODE>(acc1, st[NB - 1]);
      if (KT > 0) {
        top_tile<KTA>(top[0], acc0, stg * SC_STAGE + 32 + 4 * lh);
        if (NB == 2) top_tile<KTA>(top[NB - 1], acc1, stg * SC_STAGE + 32 + 4 * lh);
      }
    }
    STAMP_AT(5)
    if (STAMP) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP_AT(6)
    __syncthreads();
    STAMP_AT(7)
  }
  if (STAMP && lane == 0) {
#pragma unroll
    for (int i = 0; i < SC_NSTAMP; i++) stamps[((int64_t)blockIdx.x * 8 + wave) * SC_NSTAMP + i] = tsum[i];
  }
  if (late) {
    epilogue<MODE>(acc0, st[0]);
    if (NB == 2) epilogue<MODE>(acc1, st[NB - 1]);
  }

  if (!LOGITS) {
#pragma unroll
    for (int cbk = 0; cbk < NB; cbk++) {
      // merge the two half-waves (same step, disjoint item subsets)
      const float m_o = __shfl_xor(st[cbk].m2, 32);
      const float s_o = __shfl_xor(st[cbk].s, 32);
      const int c_o = __shfl_xor(st[cbk].cnt, 32);
      const float M2 = fmaxf(st[cbk].m2, m_o);
      const float S = st[cbk].s * __builtin_amdgcn_exp2f(st[cbk].m2 - M2) + s_o * __builtin_amdgcn_exp2f(m_o - M2);
      const int cnt = st[cbk].cnt + c_o;
      if (MODE == 3) {   // sum of squares over ALL outputs of the row (null-id rows too: their y is the zero vector, not "no row")
        if (lh == 0 && myrow[cbk] < Rin) {
          if (o_ce) o_ce[myrow[cbk]] = st[cbk].s + s_o;
          if (o_tgt) o_tgt[myrow[cbk]] = myy[cbk] > 0 ? st[cbk].t : 0.f;
        }
        continue;
      }
      if (lh == 0 && myrow[cbk] < Rin) {
        const bool valid = myy[cbk] > 0;
        const float lse = (M2 + log2f(S)) * LN2_F;
        if (o_rank) o_rank[myrow[cbk]] = valid ? cnt : 0;
        if (o_ce) o_ce[myrow[cbk]] = valid ? (lse - st[cbk].t) : 0.f;
        if (o_tgt) o_tgt[myrow[cbk]] = valid ? st[cbk].t : 0.f;
      }
    }
    if (KT > 0) {
      // The two half-waves hold disjoint item subsets of the same step: merge the two sorted lists
      // (value descending, then index ascending) in LDS, which the item stream no longer needs.
      float* lv = reinterpret_cast<float*>(smem);                       // [512 lanes][KT]
      int* li = reinterpret_cast<int*>(smem + SC_THREADS * KTA * 4);    // [512 lanes][KT]
#pragma unroll
      for (int cbk = 0; cbk < NB; cbk++) {
        top_flush<KTA>(top[cbk]);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < KTA; k++) { lv[tid * KTA + k] = top[cbk].v[k]; li[tid * KTA + k] = top[cbk].i[k]; }
        __syncthreads();
        if (lh == 0 && myrow[cbk] < Rin) {
          const bool valid = myy[cbk] > 0;
          const float* av = lv + tid * KTA; const int* ai = li + tid * KTA;
          const float* bv = lv + (tid + 32) * KTA; const int* bi = li + (tid + 32) * KTA;
          int pa = 0, pb = 0;
          for (int k = 0; k < topk; k++) {
            // masked rows are all-zero logit rows (model.py:105): every value ties at 0 -> indices 0..k-1
            const float va = pa < KTA ? av[pa] : -INFINITY, vb = pb < KTA ? bv[pb] : -INFINITY;
            const int ia = pa < KTA ? ai[pa] : 0x7fffffff, ib = pb < KTA ? bi[pb] : 0x7fffffff;
            const bool ta = (va > vb) || (va == vb && ia < ib);
            o_topi[myrow[cbk] * topk + k] = valid ? (ta ? ia : ib) : k;
            o_topv[myrow[cbk] * topk + k] = valid ? (ta ? va : vb) : 0.f;
            pa += ta ? 1 : 0;
            pb += ta ? 0 : 1;
          }
        }
      }
    }
  }
}

// =============================================================================================
// Production kernel: software-pipelined.  Inside ONE wave the MFMAs of item tile t+1 are issued
// between the VALU epilogue instructions of tile t (two accumulator sets), because on gfx950 a
// wave's own VALU work hides in the ~24 free issue cycles of each 32-cycle MFMA, while two
// co-resident waves that alternate "all MFMA" and "all VALU" phases mostly time-slice the SIMD
// (measured: profiles/r1_score_ablation.md).  Three LDS stages of 64 items; the LDS-DMA for stage
// s+2 is issued one full iteration before its first use; one barrier per stage.
// =============================================================================================
struct RowState2 {
  float t;     // target logit
  float m2;    // softmax reference in log2 units
  float s;     // sum of 2^(z*log2(e) - m2)
  int c;       // count of z > t
  float xp, ep, ssave;  // PMODE 5: software-pipeline registers (pending exponent, pending term) and the tile checkpoint of s
  f32x2 s2, xp2, ep2;   // PMODE 7/8: the same pipeline on PAIRS of logits (v_pk_fma_f32 / v_pk_add_f32: two values per issue slot)
  unsigned bits;        // PMODE 7: sign bits of (t - z) of the current tile, one per logit (popcount at the end of the tile)
};

// PMODE (HIERTCN_PIPE_VARIANT): 0 full, reference checked before every tile (v3); 1 count only; 2 no epilogue;
// 3 softmax sum only; 4 full without the reference check (1-4: ablation builds, wrong outputs by design);
// 5 full, epilogue software-pipelined across values (the fma / exp / add of one logit sit in three consecutive
// MFMA gaps, so no VALU op waits on the one before it) and the reference moved only when the running sum
// overflows (checked once per tile; the tile is then redone from its checkpoint with a new reference).
// 7: as 5 on pairs of logits with the packed fp32 ops (fma and sum cost half an issue slot per logit) and the count taken
// from the sign of (t - z) shifted into a bit register (v_pk_add + v_alignbit, popcount once per tile) instead of
// v_cmp + v_addc; 8: packed sum, compare/add-carry count.
// NB = 32-step blocks per wave, NSTG = LDS stages: <2, 3> for C <= 256; <1, 2> for C = 512, where 64 rows of resident B operand
// would take all 256 VGPRs and three 64 KB stages do not fit the 160 KB of LDS.
// PAD (C >= 256): the LDS image of a stage is padded per 1 KB LDS-DMA piece instead of XOR-swizzled per row, so that the A fragment of
// k-step s sits at (lane base + 32 s) -- an immediate offset -- instead of (lane base ^ 32 s): one address register instead of sixteen.
// C = 256: a piece holds two rows; pitch 1024 + 32 and chunk' = chunk ^ (row & 1) put row r of a 16-lane read group in bank group
// r mod 16.  C = 512: one row per piece, pitch 1024 + 16.
// FIXUP: launched behind a DUAL kernel (which has no room for the overflow redo): a row tile is re-done, with the per-tile redo,
// only if the fast kernel left a non-finite ce in it (running sum past fp32); every other workgroup exits after one vote.
// KT > 0 (NB = 1 shape only): the ranked top-KT list of every row beside rank / ce (loss.py:120) -- per lane a sorted list, each drained
// tile tested against the list's last entry with one wave-uniform branch behind its epilogue, while the next tile's MFMAs run; the walk
// must start at item 0 (walk_pos null) so that equal values arrive in index order, which is tf.nn.top_k's order.
template <int C, int PMODE = 0, int NB = 2, int NSTG = 3, bool PAD = false, bool FIXUP = false, int KT = 0>
__global__ __launch_bounds__(SC_THREADS, 2) void k_score_pipe(const uint16_t* __restrict__ hidden, const int32_t* __restrict__ y,
                                                              int64_t R, const uint16_t* __restrict__ outT,
                                                              const float* __restrict__ out_bias, int N, int NP,
                                                              int32_t* __restrict__ o_rank, float* __restrict__ o_ce,
                                                              float* __restrict__ o_tgt, const int32_t* __restrict__ row_map,
                                                              const int32_t* __restrict__ n_rows_dev, int part_stages,
                                                              float* __restrict__ partials, int* __restrict__ walk_pos,
                                                              int32_t* __restrict__ o_topi, float* __restrict__ o_topv, int topk) {
  static_assert(KT == 0 || NB == 1, "the per-lane lists fit beside ONE resident step block");
  // Small batches: gridDim.y > 1 splits the CATALOGUE -- workgroup (x, p) walks stages [p*part_stages, (p+1)*part_stages)
  // for row tile x and writes (count, log2-sum-exp2) per row into `partials` [gridDim.y][2][R]; k_score_merge adds them
  // up.  A 6,400-row XING batch (32 lanes) otherwise occupies 13 of the 256 CUs for a whole pass over the table.
  // Ragged batches: `row_map` (optional) lists the rows that have a target (y > 0) in ascending order and
  // n_rows_dev[0] is their count (both produced on the device, hiertcn_forward_eval); compact position p then
  // stands for row row_map[p], workgroups past the count exit at once, and padded steps cost no MFMA at all.
  // Every row's arithmetic is independent of its position, so results are bit-identical to the dense launch.
  const int64_t Rv = n_rows_dev ? (int64_t)n_rows_dev[0] : R;
  static_assert((NB == 2 && NSTG == 3) || (NB == 1 && NSTG == 2) || (NB == 1 && (PMODE == 15 || PMODE == 22) && (NSTG == 3 || (NSTG == 2 && C == 512))), "the pipeline shapes written out below");
  int bx = (int)blockIdx.x, by = (int)blockIdx.y, part_st = part_stages;
  if (part_stages < 0) {   // the split decided here, from the live row count (score_geom.h): -part_stages = the parts the scratch holds
    const ScDynSplit d = score_dyn_split(Rv, 256 * NB, NP / SC_STAGE, (int)gridDim.x, -part_stages);
    by = bx / d.tiles;
    bx -= by * d.tiles;
    if (by >= d.parts) return;
    part_st = d.part_stages;
  }
  if ((int64_t)bx * (256 * NB) >= Rv) return;
  using G = ScGeom<C>;
  constexpr bool DUAL = PMODE == 12 || PMODE == 13 || PMODE == 14 || PMODE == 16 || PMODE == 21;   // one A fragment feeds both step blocks (four accumulator tiles in flight); 13: ring of 3
  constexpr bool SUMSQ = PMODE == 21;   // mixture head (gmm.hip): the four-tile seamless kernel with a one-fma epilogue -- sum of squares of the row's outputs
  constexpr bool SIGMA = PMODE == 22;   // mixture head with a learned scale (gmm.hip): tiles come in PAIRS -- the 32 means of a stage, then the 32 log-scales of the same
                                        // outputs -- and the row accumulates sum_d mu_d^2 exp(-2 s_d): the seamless one-block skeleton with a two-phase epilogue
  constexpr bool SEAM1 = PMODE == 15 || PMODE == 22;   // ONE step block per wave (room for per-lane top-k lists) with the seamless tile boundary of PMODE 14: see SC_PHASE1
  constexpr bool SEAMLESS = PMODE == 14 || PMODE == 16 || PMODE == 21;   // 16: timing experiment, no stage barrier (racy by design)   // DUAL + no tile-boundary bubble: the next tile's bias and first fragment are fetched during this one
  if constexpr (FIXUP) {
    const int64_t r = (int64_t)bx * (256 * NB) + threadIdx.x;
    int bad = 0;
    if ((int)threadIdx.x < 256 * NB && r < Rv) {
      const int64_t src = row_map ? (int64_t)row_map[r] : r;
      bad = (__float_as_uint(o_ce[src]) & 0x7f800000u) == 0x7f800000u ? 1 : 0;   // inf or NaN, whatever the fast-math flags assume
    }
    if (!__syncthreads_or(bad)) return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // NSTG stages + NSTG x 64 bias floats
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t row0 = ((int64_t)bx * 8 + wave) * (32 * NB);

  uint4 hf[NB][G::KS];
  int myy[NB];
  int64_t mysrc[NB];
#pragma unroll
  for (int cbk = 0; cbk < NB; cbk++) {
    int64_t r = row0 + cbk * 32 + lr;
    const int64_t rp = (r < Rv) ? r : (Rv - 1);
    const int64_t rc = row_map ? (int64_t)row_map[rp] : rp;
    mysrc[cbk] = rc;
    int yy = (r >= Rv) ? 0 : y[rc];
    myy[cbk] = (yy > 0 && yy < N) ? yy : 0;
    const uint16_t* hp = hidden + rc * C + 8 * lh;
#pragma unroll
    for (int s = 0; s < G::KS; s++) hf[cbk][s] = *reinterpret_cast<const uint4*>(hp + 16 * s);
  }

  RowState2 st[NB];
#pragma unroll
  for (int cbk = 0; cbk < NB; cbk++) {  // target logits: same MFMA sequence as the main loop (see file header)
    f32x16 acc;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int yi = __shfl(myy[cbk], acc_row(reg, lh));
      acc[reg] = out_bias[yi];
    }
    const uint16_t* ap = outT + (int64_t)myy[cbk] * C + 8 * lh;
#pragma unroll
    for (int s = 0; s < G::KS; s++) {
      const uint4 a = *reinterpret_cast<const uint4*>(ap + 16 * s);
      acc = mfma32(a, hf[cbk][s], acc);
    }
    const int dreg = (lr & 3) + 4 * (lr >> 3);
    float d = acc[0];
#pragma unroll
    for (int reg = 1; reg < 16; reg++) d = (dreg == reg) ? acc[reg] : d;
    const float other = __shfl_xor(d, 32);
    const float t = (((lr >> 2) & 1) == lh) ? d : other;
    st[cbk].t = t;
    // Reference of the softmax sum, log2 units: the target logit plus 64.  Terms are 2^(z log2e - m2), so the running sum overflows
    // fp32 only when some logit sits more than (128 + 64) ln 2 = 133 nats above the target (88 without the headroom) -- the
    // overflow path (per-tile redo / the fix-up launch) is for pathological rows -- and a term underflows only 43 nats BELOW the
    // target, where it is < 2^-62 of the target's own term.  The shift is an exact integer; lse = (m2 + log2 s) ln 2 as before.
    st[cbk].m2 = t * LOG2E_F + 64.0f;
    st[cbk].s = 0.f;
    st[cbk].c = 0;
    st[cbk].xp = st[cbk].ep = st[cbk].ssave = 0.f;
    st[cbk].s2 = st[cbk].xp2 = st[cbk].ep2 = f32x2{0.f, 0.f};
    st[cbk].bits = 0u;
    if (PMODE == 7 || PMODE == 9) st[cbk].t = t + 0.0f;   // -0 -> +0: then (t - z) is negative exactly when z > t (no -0 result possible)
  }

  const int n_all = NP / SC_STAGE;
  const int stg_lo = partials ? by * part_st : 0;
  const int stg_hi0 = partials ? stg_lo + part_st : n_all;
  const int n_stages = ((stg_hi0 < n_all) ? stg_hi0 : n_all) - stg_lo;   // stages of THIS workgroup (may be <= 0 for a tail part)
  constexpr int PIECES = G::STAGE_BYTES / 1024;
  constexpr int PPW = PIECES / 8;
  constexpr bool PADL = PAD && C >= 256;
  constexpr int PITCH = PADL ? 1024 + (C == 256 ? 32 : 16) : 1024;   // LDS bytes per 1 KB piece
  constexpr int STAGE_LDS = PIECES * PITCH;
  constexpr int TILE_LDS = STAGE_LDS / 2;                             // one 32-item block
  constexpr int BIAS_OFF = NSTG * STAGE_LDS;
  // LDS-DMA issued through inline asm: hipcc then neither drains it (vmcnt(0)) in front of the next ds_read -- it
  // cannot prove that the DMA and the read touch different stages -- nor waits for it at the barrier; the
  // wait is placed by hand, one full iteration after the issue (cdna_hip_programming.md 5.7 'LDS-DMA recipe').
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  auto glds16 = [&](const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
  };
  // walk_pos: a workgroup may start anywhere in the table and wrap around (count and softmax sum do not depend on the
  // item order beyond fp32 rounding).  Workgroups dealt to one XCD (blockIdx % 8: observed round-robin placement, a
  // speed hint, never correctness) start where their already-running neighbours are, so that late starters re-use the
  // neighbours' lines in that XCD's 4 MiB L2 instead of pulling the 150 MB table through the fabric again: measured
  // 161 -> 32 GB per launch at the L2's memory-side port, L2 hit rate 76 -> 96 % (profiles/r1_score_ablation.md).
  // Price: the low bits of `ce` depend on the start offset (ranks never do); HIERTCN_FIXED_WALK=1 pins it to 0.
  // ONE read per workgroup, shared through LDS: the position word moves while the workgroup starts, and eight waves that each read
  // it for themselves can disagree -- they would then fill one LDS stage with pieces of two different item stages (items counted
  // twice or not at all: ranks off by a few; found by scripts/soak_score.py at full size, not by the small race screen).
  int s0 = 0;
  if (walk_pos && !partials) {
    int* s0_sh = reinterpret_cast<int*>(smem);   // (the stage buffers are not in use yet)
    if (tid == 0) *s0_sh = __hip_atomic_load(walk_pos + (blockIdx.x & 7) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    s0 = __builtin_amdgcn_readfirstlane(*s0_sh);
    __syncthreads();                             // everyone has read it before the first LDS-DMA may overwrite the word
    s0 = (s0 >= 0 && s0 < n_stages) ? s0 : 0;
  }
  unsigned dma_voff[PPW];   // byte offset of this lane's 16 B of piece i inside a stage of outT (source-side swizzle applied)
#pragma unroll
  for (int i = 0; i < PPW; i++) {
    const int off = (wave * PPW + i) * 1024 + lane * 16;
    const int row = off / (2 * C), slot = (off % (2 * C)) >> 4;
    const int sw = PADL ? (C == 256 ? (row & 1) : 0) : (row & G::SWZ);
    dma_voff[i] = (unsigned)(row * (2 * C) + ((slot ^ sw) << 4));
  }
  auto stage_issue = [&](int stg_rel, int buf) {
    int stg = s0 + stg_rel;
    stg -= (stg >= n_stages) ? n_stages : 0;
    if (walk_pos && !partials && wave == 0 && lane == 0 && (stg_rel & 7) == 0)
      __hip_atomic_store(walk_pos + (blockIdx.x & 7) * 32, stg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    stg += stg_lo;
    {
      // SGPR stage base + per-piece 32-bit lane offsets (dma_voff, computed once): no 64-bit per-lane pointers, no address VALU per stage
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
      if (wave == 0 && lane < SC_STAGE / 4)
        glds16(out_bias + (int64_t)stg * SC_STAGE + lane * 4, lds_base + BIAS_OFF + buf * SC_STAGE * 4);
    }
  };
  const float* sbias = reinterpret_cast<const float*>(smem + BIAS_OFF);

  // per-tile pre-pass of the epilogue: running-reference check on the raw logits (rare rescale, T13)
  constexpr bool NOLDS = PMODE == 10 || PMODE == 11;   // timing experiments (wrong results): one A fragment per tile instead of one per MFMA
  constexpr bool PIPED = PMODE >= 5 && PMODE != 10;    // 10: no epilogue either; 11: production epilogue
  constexpr bool PACKED = PMODE >= 7 && PMODE <= 9;
  constexpr int GPV = G::KS > 16 ? G::KS / 16 : 1;   // MFMA gaps per epilogue value
  constexpr int RING = PMODE == 6 ? 4 : (PMODE == 12 ? 2 : 3);  // LDS fragment ring: a ds_read is issued RING-1 MFMAs ahead of its use
  auto pre = [&](const f32x16& v, RowState2& rs) {
    if (PMODE == 1 || PMODE == 2 || PMODE == 4 || PMODE == 10) return;
    if (PACKED) {
      rs.ssave = rs.s2.x + rs.s2.y;
      asm volatile("" : "+v"(rs.ssave));
      return;
    }
    if (PIPED) {
      rs.ssave = rs.s;
      asm volatile("" : "+v"(rs.ssave));
      return;
    }
    float tm = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
    tm = fmaxf(tm, fmaxf(fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])), fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]))));
    const float xm = fmaf(tm, LOG2E_F, -rs.m2);
    if (__any(xm > SC_DEFER)) {
      const float d = fmaxf(xm, 0.f);
      const float sc = __builtin_amdgcn_exp2f(-d);
      rs.s *= sc;
      rs.m2 += d;
    }
  };
  // one value of the epilogue (placed between two MFMAs)
  // The empty asm statements are ordering anchors: hipcc otherwise hoists these (pure) VALU ops out from between
  // the MFMAs, where they are free, into one block of their own, where they are not.
  // PMODE 5: value i of a tile.  s += e(i-2); e(i-1) = 2^x(i-1); x(i) = v*log2(e) - m2; c += v > t.
  auto val5 = [&](int i, float v, RowState2& rs) {
    if (SUMSQ) { rs.s = fmaf(v, v, rs.s); asm volatile("" : "+v"(rs.s)); return; }
    const bool gt = v > rs.t;
    if (i >= 2) rs.s += rs.ep;
    if (i >= 1) rs.ep = __builtin_amdgcn_exp2f(rs.xp);
    rs.xp = fmaf(v, LOG2E_F, -rs.m2);
    rs.c += gt ? 1 : 0;
    asm volatile("" : "+v"(rs.s), "+v"(rs.c), "+v"(rs.ep), "+v"(rs.xp));
  };
  // PMODE 7/8: gap i of a tile handles pair j = i/2.  Even gap: s2 += e2(j-2); e(j-1).lo = 2^x(j-1).lo; d2 = t - v2(j).
  // Odd gap: e(j-1).hi = 2^x(j-1).hi; x2(j) = v2(j)*log2(e) - m2.  Every gap: one sign bit into rs.bits.
  // PMODE 9 = 7 with the packed ops written in asm: hipcc (SIPreEmitPeephole) splits v_pk_*_f32 that sit behind an MFMA back
  // into two scalar ops on gfx950, so 7 as compiled is only partly packed.
  auto pk_add = [&](f32x2 a, f32x2 b) { f32x2 r; asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; };
  auto val7 = [&](int i, const f32x16& v, RowState2& rs, f32x2& d2) {
    const int j = i >> 1;
    const f32x2 v2 = {v[2 * j], v[2 * j + 1]};
    if (PMODE == 9) {
      const f32x2 tm = {rs.t, rs.m2};
      const f32x2 l2 = {LOG2E_F, LOG2E_F};
      if ((i & 1) == 0) {
        if (j >= 2) rs.s2 = pk_add(rs.s2, rs.ep2);
        if (j >= 1) rs.ep2.x = __builtin_amdgcn_exp2f(rs.xp2.x);
        asm volatile("v_pk_add_f32 %0, %1, %2 op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d2) : "v"(tm), "v"(v2));
        rs.bits = (i == 0) ? (__float_as_uint(d2.x) >> 31) : __builtin_amdgcn_alignbit(rs.bits, __float_as_uint(d2.x), 31);
      } else {
        if (j >= 1) rs.ep2.y = __builtin_amdgcn_exp2f(rs.xp2.y);
        asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1] op_sel_hi:[1,0,1] neg_lo:[0,0,1] neg_hi:[0,0,1]"
                     : "=v"(rs.xp2) : "v"(v2), "s"(l2), "v"(tm));
        rs.bits = __builtin_amdgcn_alignbit(rs.bits, __float_as_uint(d2.y), 31);
      }
      asm volatile("" : "+v"(rs.s2), "+v"(rs.c), "+v"(rs.ep2), "+v"(rs.xp2), "+v"(rs.bits), "+v"(d2));
      return;
    }
    if ((i & 1) == 0) {
      if (j >= 2) rs.s2 += rs.ep2;
      if (j >= 1) rs.ep2.x = __builtin_amdgcn_exp2f(rs.xp2.x);
      if (PMODE == 7) {
        d2 = f32x2{rs.t, rs.t} - v2;
        rs.bits = (i == 0) ? (__float_as_uint(d2.x) >> 31) : __builtin_amdgcn_alignbit(rs.bits, __float_as_uint(d2.x), 31);
      } else {
        rs.c += (v2.x > rs.t) ? 1 : 0;
      }
    } else {
      if (j >= 1) rs.ep2.y = __builtin_amdgcn_exp2f(rs.xp2.y);
      rs.xp2 = __builtin_elementwise_fma(v2, f32x2{LOG2E_F, LOG2E_F}, f32x2{-rs.m2, -rs.m2});
      if (PMODE == 7) rs.bits = __builtin_amdgcn_alignbit(rs.bits, __float_as_uint(d2.y), 31);
      else rs.c += (v2.y > rs.t) ? 1 : 0;
    }
    asm volatile("" : "+v"(rs.s2), "+v"(rs.c), "+v"(rs.ep2), "+v"(rs.xp2), "+v"(rs.bits), "+v"(d2));
  };
  auto post7 = [&](const f32x16& v, RowState2& rs) {
    rs.s2 += rs.ep2;
    rs.ep2.x = __builtin_amdgcn_exp2f(rs.xp2.x);
    rs.ep2.y = __builtin_amdgcn_exp2f(rs.xp2.y);
    rs.s2 += rs.ep2;
    if (PMODE == 7 || PMODE == 9) rs.c += __builtin_popcount(rs.bits);
    const float tot = rs.s2.x + rs.s2.y;
    if (__any(!(tot < 1.0e38f))) {
      float tm = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
      tm = fmaxf(tm, fmaxf(fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])), fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]))));
      const bool bad = !(tot < 1.0e38f);
      float d = fmaxf(ceilf(fmaf(tm, LOG2E_F, -rs.m2)), 0.f);
      d = fmaxf(d, (float)__builtin_amdgcn_frexp_expf(rs.ssave));
      d = bad ? d : 0.f;
      rs.m2 += d;
      float acc = rs.ssave * __builtin_amdgcn_exp2f(-d);
#pragma unroll
      for (int i = 0; i < 16; i++) acc += __builtin_amdgcn_exp2f(fmaf(v[i], LOG2E_F, -rs.m2));
      rs.s2 = f32x2{acc, 0.f};
    }
  };
  // PMODE 5: end of a tile: drain the two pending values, then the overflow check (wave-uniform branch, rare).
  auto post5 = [&](const f32x16& v, RowState2& rs) {
    if (SUMSQ || SIGMA) return;
    rs.s += rs.ep;
    rs.s += __builtin_amdgcn_exp2f(rs.xp);
    if (!DUAL && !SEAM1 && __any(!(rs.s < 1.0e38f))) {
      // redo this tile from the checkpoint with a reference that bounds every term and the carried sum by 1
      float tm = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
      tm = fmaxf(tm, fmaxf(fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])), fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]))));
      const bool bad = !(rs.s < 1.0e38f);
      float d = fmaxf(ceilf(fmaf(tm, LOG2E_F, -rs.m2)), 0.f);
      d = fmaxf(d, (float)__builtin_amdgcn_frexp_expf(rs.ssave));   // ssave * 2^-d <= 1 (frexp_exp(0) = 0)
      d = bad ? d : 0.f;                                             // lanes that did not overflow keep their reference
      rs.m2 += d;
      float acc = rs.ssave * __builtin_amdgcn_exp2f(-d);             // exact: d is an integer
#pragma unroll
      for (int i = 0; i < 16; i++) acc += __builtin_amdgcn_exp2f(fmaf(v[i], LOG2E_F, -rs.m2));
      rs.s = acc;
    }
  };
  auto one = [&](float v, RowState2& rs) {
    asm volatile("" : "+v"(v));
    if (PMODE == 2 || PMODE == 10) return;
    if (PMODE != 1) rs.s += __builtin_amdgcn_exp2f(fmaf(v, LOG2E_F, -rs.m2));
    if (PMODE != 3) rs.c += (v > rs.t) ? 1 : 0;
    asm volatile("" : "+v"(rs.s), "+v"(rs.c));
  };
  // LDS byte offset of this lane's A fragment for k-step 0 of item row (ib*32 + lr); k-step s is `^ (s << 5)`:
  // ((2s + lh) ^ sw) << 4 == ((lh ^ sw) << 4) ^ (s << 5), so one VGPR and one v_xor replace 16 address registers
  const int a_k0 = PADL ? (C == 256 ? (lr >> 1) * PITCH + (lr & 1) * 512 + ((lh ^ (lr & 1)) << 4) : lr * PITCH + (lh << 4))
                        : lr * (2 * C) + ((lh ^ (lr & G::SWZ)) << 4);
#define SC_FRAG(s_) (PADL ? (a_k0 + ((s_) << 5)) : (a_k0 ^ ((s_) << 5)))
  auto bias_init = [&](f32x16& a0, int buf, int ib) {
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + buf * SC_STAGE + ib * 32 + 8 * q + 4 * lh);
#pragma unroll
      for (int j = 0; j < 4; j++) a0[4 * q + j] = b4[j];
    }
  };
  // One pipeline step: the 16 MFMAs of (item block ib of stage buffer buf) x (step block CBX) into x, with the
  // epilogue of the previous tile y (step block CBY) issued between them, one value per MFMA.
  // Both step blocks re-read the item fragments from LDS (1 ds_read_b128 per MFMA = 128 B/clk/CU, half the
  // LDS peak) -- the price of needing only two accumulator tiles.
#define SC_PHASE(x, CBX, y, CBY, buf_, ib_, do_mfma)                                         \
  do {                                                                                       \
    f32x2 d2_ = {0.f, 0.f};                                                                  \
    int constexpr_i_ = 0;                                                                    \
    pre(y, st[CBY]);                                                                         \
    if (do_mfma) {                                                                           \
      const unsigned char* tile_ = smem + (buf_) * STAGE_LDS + (ib_) * TILE_LDS;             \
      uint4 af_[RING];                                                                       \
      _Pragma("unroll") for (int s = 0; s < RING - 1; s++)                                   \
        af_[s] = *reinterpret_cast<const uint4*>(tile_ + SC_FRAG(s));                        \
      bias_init(x, buf_, ib_);                                                               \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      _Pragma("unroll") for (int s = 0; s < G::KS; s++) {                                    \
        if (s + RING - 1 < G::KS && !NOLDS)                                                  \
          af_[(s + RING - 1) % RING] = *reinterpret_cast<const uint4*>(tile_ + SC_FRAG(s + RING - 1)); \
        x = mfma32(af_[NOLDS ? 0 : s % RING], hf[CBX][s], x);                                \
        if (s % GPV == 0) {   /* KS > 16 (C = 512): one value per GPV gaps */                \
          constexpr_i_ = s / GPV;                                                            \
          if (PACKED) val7(constexpr_i_, y, st[CBY], d2_); else if (PIPED) val5(constexpr_i_, y[constexpr_i_], st[CBY]); else one(y[constexpr_i_], st[CBY]); \
        }                                                                                    \
        __builtin_amdgcn_sched_barrier(0); /* keep exactly this order: hipcc otherwise clusters the MFMAs */ \
      }                                                                                      \
      _Pragma("unroll") for (int s = G::KS; s < 16; s++) { if (PACKED) val7(s, y, st[CBY], d2_); else if (PIPED) val5(s, y[s], st[CBY]); else one(y[s], st[CBY]); } \
    } else {                                                                                 \
      _Pragma("unroll") for (int s = 0; s < 16; s++) { if (PACKED) val7(s, y, st[CBY], d2_); else if (PIPED) val5(s, y[s], st[CBY]); else one(y[s], st[CBY]); } \
    }                                                                                        \
    if (PACKED) post7(y, st[CBY]); else if (PIPED) post5(y, st[CBY]);                        \
  } while (0)

  // DUAL: the 32 MFMAs of (item block ib) x (both step blocks) share each A fragment (0.5 ds_read_b128 per MFMA); the epilogue of the
  // previous item block's two tiles runs in the 32 gaps, the two rows' chains alternating.
#define SC_PHASE2(x0, x1, y0, y1, buf_, ib_, do_mfma)                                        \
  do {                                                                                       \
    pre(y0, st[0]);                                                                          \
    pre(y1, st[1]);                                                                          \
    if (do_mfma) {                                                                           \
      const unsigned char* tile_ = smem + (buf_) * STAGE_LDS + (ib_) * TILE_LDS;             \
      uint4 af_[RING];                                                                       \
      _Pragma("unroll") for (int s = 0; s < RING - 1; s++)                                   \
        af_[s] = *reinterpret_cast<const uint4*>(tile_ + SC_FRAG(s));                        \
      bias_init(x0, buf_, ib_);                                                              \
      bias_init(x1, buf_, ib_);                                                              \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      _Pragma("unroll") for (int s = 0; s < G::KS; s++) {                                    \
        if (s + RING - 1 < G::KS)                                                            \
          af_[(s + RING - 1) % RING] = *reinterpret_cast<const uint4*>(tile_ + SC_FRAG(s + RING - 1)); \
        x0 = mfma32(af_[s % RING], hf[0][s], x0);                                            \
        val5(s, y0[s], st[0]);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        x1 = mfma32(af_[s % RING], hf[1][s], x1);                                            \
        val5(s, y1[s], st[1]);                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                   \
      }                                                                                      \
      _Pragma("unroll") for (int s = G::KS; s < 16; s++) { val5(s, y0[s], st[0]); val5(s, y1[s], st[1]); } \
    } else {                                                                                 \
      _Pragma("unroll") for (int s = 0; s < 16; s++) { val5(s, y0[s], st[0]); val5(s, y1[s], st[1]); } \
    }                                                                                        \
    post5(y0, st[0]);                                                                        \
    post5(y1, st[1]);                                                                        \
  } while (0)

  // SEAMLESS: as SC_PHASE2, but nothing waits at a tile boundary.  The accumulators of the NEXT phase are the ones this phase's
  // epilogue is draining (y0, y1), and a value is dead once its gap has passed (no per-tile redo in DUAL), so the next tile's bias
  // is read straight into registers 4q..4q+3 right after gap 4q+3; the A-fragment ring runs on across the boundary (the read issued
  // at the last k-step is the next tile's first fragment).
#define SC_PHASE3(x0, x1, y0, y1, buf_, ib_, MF, nbuf_, nib_)                                 \
  do {                                                                                       \
    const unsigned char* tile_ = smem + (buf_) * STAGE_LDS + (ib_) * TILE_LDS;               \
    const unsigned char* ntile_ = smem + (nbuf_) * STAGE_LDS + (nib_) * TILE_LDS;            \
    const float* nbias_ = sbias + (nbuf_) * SC_STAGE + (nib_) * 32 + 4 * lh;                 \
    _Pragma("unroll") for (int s = 0; s < 16; s++) {                                         \
      if (MF && s < G::KS) {   /* compile-time; C < 256: fewer k-steps than values, the rest of the epilogue trails */ \
        ring_[(s + 1) & 1] = *reinterpret_cast<const uint4*>((s + 1 < G::KS) ? tile_ + SC_FRAG(s + 1) : ntile_ + SC_FRAG(0)); \
        x0 = mfma32(ring_[s & 1], hf[0][s], x0);                                             \
      }                                                                                      \
      val5(s, y0[s], st[0]);                                                                 \
      if ((s & 3) == 3) {                                                                    \
        const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_ + 8 * (s >> 2));           \
        _Pragma("unroll") for (int j = 0; j < 4; j++) y0[(s & ~3) + j] = b4_[j];             \
      }                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                     \
      if (MF && s < G::KS) x1 = mfma32(ring_[s & 1], hf[1][s], x1);                          \
      val5(s, y1[s], st[1]);                                                                 \
      if ((s & 3) == 3) {                                                                    \
        const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_ + 8 * (s >> 2));           \
        _Pragma("unroll") for (int j = 0; j < 4; j++) y1[(s & ~3) + j] = b4_[j];             \
      }                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                     \
    }                                                                                        \
    post5(y0, st[0]);                                                                        \
    post5(y1, st[1]);                                                                        \
  } while (0)

  constexpr int KTA = KT > 0 ? KT : 1;
  TopList<KTA> top;
  if (KT > 0) top_init<KTA>(top);
  // SEAM1 (PMODE 15, NB = 1, three LDS stages): the ranked top-k flavour on the seamless skeleton.  As SC_PHASE3 with one step block: the
  // accumulator tile being drained takes the NEXT tile's bias four registers at a time (behind gaps 3, 7, 11, 15) and the A-fragment
  // ring (four deep here: with one MFMA per fragment a read must be issued three MFMAs ahead to cover the LDS latency) runs on across
  // the tile boundary, so no wave waits for LDS at a tile start -- the two-stage flavour it replaces loses a third of its MFMA time
  // there (MFMA pipe busy 43 %).  The tile's maximum is folded over gaps 0..3 (two v_max per gap); in gap 3 -- the last one before the
  // bias starts to overwrite the drained values -- one wave vote decides whether any lane has a candidate for its list (top_tile's
  // slow path: per-lane sorted lists in registers, candidates parked two at a time).  The softmax sum has no per-tile redo here: a sum
  // past fp32 is handed to the fix-up launch, as in PMODE 14.
  float tk_m = -INFINITY;
  float mu2_[16];   // SIGMA: squares of the stage's means, kept from the first tile's drain to the second's
  // half 0: a tile of means -> their squares; half 1: the log-scales of the same outputs -> sum += mu^2 exp(-2 s), exponent clamped below fp32 overflow
  auto sig_val = [&](int half, float& m2, float v, RowState2& rs) {
    if (half == 0) m2 = v * v;
    else rs.s = fmaf(m2, __builtin_amdgcn_exp2f(fminf(v * (-2.0f * LOG2E_F), 126.0f)), rs.s);
    asm volatile("" : "+v"(rs.s), "+v"(m2));
  };
#define SC_GAP1(x, y, s)                                                                     \
      if (MF_ && (s) < G::KS) {                                                              \
        ring4_[((s) + 3) & 3] = *reinterpret_cast<const uint4*>(((s) + 3 < G::KS) ? tile_ + SC_FRAG((s) + 3) : ntile_ + SC_FRAG((s) + 3 - G::KS)); \
        x = mfma32(ring4_[(s) & 3], hf[0][s], x);                                            \
      }                                                                                      \
      if (SIGMA) sig_val(HALF_, mu2_[s], y[s], st[0]); else val5(s, y[s], st[0]);
#define SC_PHASE1(x, y, buf_, ib_, MF, nbuf_, nib_, dit_, half_)                               \
  do {                                                                                       \
    constexpr bool MF_ = MF;                                                                 \
    constexpr int HALF_ = half_;                                                             \
    const unsigned char* tile_ = smem + (buf_) * STAGE_LDS + (ib_) * TILE_LDS;               \
    const unsigned char* ntile_ = smem + (nbuf_) * STAGE_LDS + (nib_) * TILE_LDS;            \
    const float* nbias_ = sbias + (nbuf_) * SC_STAGE + (nib_) * 32 + 4 * lh;                 \
    /* gaps 0..3, the tile's maximum folded in beside them (the slow path is kept OUT of the unrolled loops) */ \
    SC_GAP1(x, y, 0) if (KT > 0) tk_m = fmaxf(fmaxf(fmaxf(y[0], y[1]), y[2]), fmaxf(y[3], y[4]));      __builtin_amdgcn_sched_barrier(0); \
    SC_GAP1(x, y, 1) if (KT > 0) tk_m = fmaxf(fmaxf(fmaxf(tk_m, y[5]), y[6]), fmaxf(y[7], y[8]));      __builtin_amdgcn_sched_barrier(0); \
    SC_GAP1(x, y, 2) if (KT > 0) tk_m = fmaxf(fmaxf(fmaxf(tk_m, y[9]), y[10]), fmaxf(y[11], y[12]));   __builtin_amdgcn_sched_barrier(0); \
    SC_GAP1(x, y, 3)                                                                         \
    if (KT > 0) {                                                                            \
      tk_m = fmaxf(fmaxf(fmaxf(tk_m, y[13]), y[14]), y[15]);                                 \
      if (__any(tk_m > top.v[KTA - 1])) top_tile_slow<KTA>(top, y, dit_);                    \
    }                                                                                        \
    {                                                                                        \
      const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_);                             \
      _Pragma("unroll") for (int j = 0; j < 4; j++) y[j] = b4_[j];                           \
    }                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    _Pragma("unroll") for (int s = 4; s < 16; s++) {                                         \
      SC_GAP1(x, y, s)                                                                       \
      if ((s & 3) == 3) {                                                                    \
        const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_ + 8 * (s >> 2));           \
        _Pragma("unroll") for (int j = 0; j < 4; j++) y[(s & ~3) + j] = b4_[j];              \
      }                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                     \
    }                                                                                        \
    post5(y, st[0]);                                                                         \
  } while (0)

  // C = 512 (32 k-steps per tile, a value drained every second gap): the same seamless phase.  With lists (KT > 0) the tile's maximum is folded in
  // the odd gaps 1, 3, 5 and the wave votes in gap 7 -- the last one before the next tile's bias starts to overwrite the drained values.
#define SC_GAPW(x, y, s)                                                                     \
      if (MF_) {                                                                             \
        ring4_[((s) + 3) & 3] = *reinterpret_cast<const uint4*>(((s) + 3 < 32) ? tile_ + SC_FRAG((s) + 3) : ntile_ + SC_FRAG((s) + 3 - 32)); \
        x = mfma32(ring4_[(s) & 3], hf[0][s], x);                                            \
      }                                                                                      \
      if (((s) & 1) == 0) val5((s) >> 1, y[(s) >> 1], st[0]);
#define SC_PHASE1W(x, y, buf_, ib_, MF, nbuf_, nib_, dit_)                                    \
  do {                                                                                       \
    constexpr bool MF_ = MF;                                                                 \
    const unsigned char* tile_ = smem + (buf_) * STAGE_LDS + (ib_) * TILE_LDS;               \
    const unsigned char* ntile_ = smem + (nbuf_) * STAGE_LDS + (nib_) * TILE_LDS;            \
    const float* nbias_ = sbias + (nbuf_) * SC_STAGE + (nib_) * 32 + 4 * lh;                 \
    /* gaps 0..7 written out: the list test (its slow path is a call-sized body) must not sit inside an unrolled loop, or hipcc gives up the unroll \
       and indexes the 128 registers of the B operand dynamically -- through scratch */      \
    SC_GAPW(x, y, 0) __builtin_amdgcn_sched_barrier(0);                                      \
    SC_GAPW(x, y, 1) if (KT > 0) tk_m = fmaxf(fmaxf(fmaxf(y[0], y[1]), y[2]), fmaxf(y[3], y[4]));      __builtin_amdgcn_sched_barrier(0); \
    SC_GAPW(x, y, 2) __builtin_amdgcn_sched_barrier(0);                                      \
    SC_GAPW(x, y, 3) if (KT > 0) tk_m = fmaxf(fmaxf(fmaxf(tk_m, y[5]), y[6]), fmaxf(y[7], y[8]));      __builtin_amdgcn_sched_barrier(0); \
    SC_GAPW(x, y, 4) __builtin_amdgcn_sched_barrier(0);                                      \
    SC_GAPW(x, y, 5) if (KT > 0) tk_m = fmaxf(fmaxf(fmaxf(tk_m, y[9]), y[10]), fmaxf(y[11], y[12]));   __builtin_amdgcn_sched_barrier(0); \
    SC_GAPW(x, y, 6) __builtin_amdgcn_sched_barrier(0);                                      \
    SC_GAPW(x, y, 7)                                                                         \
    if (KT > 0) {                                                                            \
      tk_m = fmaxf(fmaxf(fmaxf(tk_m, y[13]), y[14]), y[15]);                                 \
      if (__any(tk_m > top.v[KTA - 1])) top_tile_slow<KTA>(top, y, dit_);                    \
    }                                                                                        \
    {                                                                                        \
      const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_);                             \
      _Pragma("unroll") for (int j = 0; j < 4; j++) y[j] = b4_[j];                           \
    }                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                       \
    _Pragma("unroll") for (int s = 8; s < 32; s++) {                                         \
      SC_GAPW(x, y, s)                                                                       \
      if ((s & 7) == 7) {                                                                    \
        const f32x4 b4_ = *reinterpret_cast<const f32x4*>(nbias_ + 8 * (s >> 3));           \
        _Pragma("unroll") for (int j = 0; j < 4; j++) y[4 * (s >> 3) + j] = b4_[j];          \
      }                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                     \
    }                                                                                        \
    post5(y, st[0]);                                                                         \
  } while (0)

#ifdef HIERTCN_SCORE_SETPRIO   // experiment (MI355X_MICROARCH.md, two waves per SIMD, item 4): static priority for the second-dispatched half
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
  f32x16 accA, accB;
  if constexpr (SEAM1 && NSTG == 2) {
    // two LDS stages (three 64 KB stages do not fit at C = 512): the freed buffer is refilled between the two phases of an iteration, and
    // a second wait + barrier behind the second phase makes sure the stage has landed before the next iteration's first phase reads its
    // first fragments across the tile boundary
    static_assert(C == 512, "the two-stage seamless loop is written for C = 512");
    uint4 ring4_[4];
    if (n_stages > 0) stage_issue(0, 0);
    if (n_stages > 1) stage_issue(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (n_stages > 0) {
      bias_init(accA, 0, 0);
      bias_init(accB, 0, 1);
#pragma unroll
      for (int s = 0; s < G::KS; s++) {
        const uint4 a = *reinterpret_cast<const uint4*>(smem + SC_FRAG(s));
        accA = mfma32(a, hf[0][s], accA);
      }
#pragma unroll
      for (int s = 0; s < 3; s++) ring4_[s] = *reinterpret_cast<const uint4*>(smem + TILE_LDS + SC_FRAG(s));
    }
    for (int stg = 0; stg + 1 < n_stages; stg++) {
      const int buf = stg & 1;
      SC_PHASE1W(accB, accA, buf, 1, true, buf ^ 1, 0, (stg_lo + stg) * SC_STAGE + 4 * lh);        // MFMA (stg, ib1)   || epilogue (stg, ib0); accA <- bias (stg+1, ib0)
      __syncthreads();                                          // every wave is done with stage stg: its buffer takes stage stg + 2
      if (stg + 2 < n_stages) stage_issue(stg + 2, buf);
      SC_PHASE1W(accA, accB, buf ^ 1, 0, true, buf ^ 1, 1, (stg_lo + stg) * SC_STAGE + 32 + 4 * lh);    // MFMA (stg+1, ib0) || epilogue (stg, ib1); accB <- bias (stg+1, ib1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                                          // stage stg + 2 has landed (the next first phase reads it at its tail)
    }
    if (n_stages > 0) {
      const int buf = (n_stages - 1) & 1;
      SC_PHASE1W(accB, accA, buf, 1, true, buf, 1, (stg_lo + n_stages - 1) * SC_STAGE + 4 * lh);            // (the "next tile" reads land in dead registers)
      SC_PHASE1W(accA, accB, buf, 1, false, buf, 1, (stg_lo + n_stages - 1) * SC_STAGE + 32 + 4 * lh);           // epilogue of the last tile only
    }
  } else if constexpr (SEAM1) {
    static_assert(G::KS >= 4, "the four-deep fragment ring needs at least four k-steps");
    uint4 ring4_[4];
    if (n_stages > 0) stage_issue(0, 0);
    if (n_stages > 1) stage_issue(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (n_stages > 0) {
      bias_init(accA, 0, 0);
      bias_init(accB, 0, 1);
#pragma unroll
      for (int s = 0; s < G::KS; s++) {
        const uint4 a = *reinterpret_cast<const uint4*>(smem + SC_FRAG(s));
        accA = mfma32(a, hf[0][s], accA);
      }
#pragma unroll
      for (int s = 0; s < 3; s++) ring4_[s] = *reinterpret_cast<const uint4*>(smem + TILE_LDS + SC_FRAG(s));   // first fragments of (stage 0, ib1)
    }
    int buf = 0;
    for (int stg = 0; stg + 1 < n_stages; stg++) {   // the last stage is peeled: no run-time "is there a next tile" inside the phases
      const int buf1 = (buf == 2) ? 0 : buf + 1;
      const int buf2 = (buf1 == 2) ? 0 : buf1 + 1;
      if (stg + 2 < n_stages) stage_issue(stg + 2, buf2);
      SC_PHASE1(accB, accA, buf, 1, true, buf1, 0, (stg_lo + stg) * SC_STAGE + 4 * lh, 0);        // MFMA (stg, ib1)   || epilogue (stg, ib0); accA <- bias (stg+1, ib0)
      SC_PHASE1(accA, accB, buf1, 0, true, buf1, 1, (stg_lo + stg) * SC_STAGE + 32 + 4 * lh, 1);  // MFMA (stg+1, ib0) || epilogue (stg, ib1); accB <- bias (stg+1, ib1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      buf = buf1;
    }
    if (n_stages > 0) {
      SC_PHASE1(accB, accA, buf, 1, true, buf, 1, (stg_lo + n_stages - 1) * SC_STAGE + 4 * lh, 0);         // (the "next tile" reads land in dead registers)
      SC_PHASE1(accA, accB, buf, 1, false, buf, 1, (stg_lo + n_stages - 1) * SC_STAGE + 32 + 4 * lh, 1);   // epilogue of the last tile only
    }
  } else if constexpr (NB == 2 && SEAMLESS && C <= 256) {
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
        const uint4 a = *reinterpret_cast<const uint4*>(smem + SC_FRAG(s));
        xa0 = mfma32(a, hf[0][s], xa0);
        xa1 = mfma32(a, hf[1][s], xa1);
      }
      ring_[0] = *reinterpret_cast<const uint4*>(smem + TILE_LDS + SC_FRAG(0));   // first fragment of (stage 0, ib1)
    }
    int buf = 0;
    for (int stg = 0; stg + 1 < n_stages; stg++) {   // the last stage is peeled: no run-time "is there a next tile" inside the phases
      const int buf1 = (buf == 2) ? 0 : buf + 1;
      const int buf2 = (buf1 == 2) ? 0 : buf1 + 1;
      if (stg + 2 < n_stages) stage_issue(stg + 2, buf2);
      SC_PHASE3(xb0, xb1, xa0, xa1, buf, 1, true, buf1, 0);       // MFMA (stg, ib1)   || epilogue (stg, ib0); xa <- bias (stg+1, ib0)
      SC_PHASE3(xa0, xa1, xb0, xb1, buf1, 0, true, buf1, 1);      // MFMA (stg+1, ib0) || epilogue (stg, ib1); xb <- bias (stg+1, ib1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (PMODE != 16) __syncthreads();
      buf = buf1;
    }
    if (n_stages > 0) {
      SC_PHASE3(xb0, xb1, xa0, xa1, buf, 1, true, buf, 1);        // (the "next tile" reads land in dead registers)
      SC_PHASE3(xa0, xa1, xb0, xb1, buf, 1, false, buf, 1);       // epilogue of the last tile only
    }
  } else if constexpr (NB == 2 && DUAL) {
    f32x16 xa0, xa1, xb0, xb1;
    if (n_stages > 0) stage_issue(0, 0);
    if (n_stages > 1) stage_issue(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (n_stages > 0) {
      bias_init(xa0, 0, 0);
      bias_init(xa1, 0, 0);
#pragma unroll
      for (int s = 0; s < G::KS; s++) {
        const uint4 a = *reinterpret_cast<const uint4*>(smem + SC_FRAG(s));
        xa0 = mfma32(a, hf[0][s], xa0);
        xa1 = mfma32(a, hf[1][s], xa1);
      }
    }
    int buf = 0;
    for (int stg = 0; stg < n_stages; stg++) {
      const int buf1 = (buf == 2) ? 0 : buf + 1;
      const int buf2 = (buf1 == 2) ? 0 : buf1 + 1;
      if (stg + 2 < n_stages) stage_issue(stg + 2, buf2);
      SC_PHASE2(xb0, xb1, xa0, xa1, buf, 1, true);                    // MFMA (stg, ib1)   || epilogue (stg, ib0)
      SC_PHASE2(xa0, xa1, xb0, xb1, buf1, 0, stg + 1 < n_stages);     // MFMA (stg+1, ib0) || epilogue (stg, ib1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      buf = buf1;
    }
  } else if constexpr (NB == 2) {
    f32x16 accA, accB;
    if (n_stages > 0) stage_issue(0, 0);
    if (n_stages > 1) stage_issue(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (n_stages > 0) {  // tile (stage 0, ib 0, step block 0) has nothing to overlap with
      bias_init(accA, 0, 0);
  #pragma unroll
      for (int s = 0; s < G::KS; s++) {
        const uint4 a = *reinterpret_cast<const uint4*>(smem + SC_FRAG(s));
        accA = mfma32(a, hf[0][s], accA);
      }
    }
    int buf = 0;  // stg % 3
    for (int stg = 0; stg < n_stages; stg++) {
      const int buf1 = (buf == 2) ? 0 : buf + 1;
      const int buf2 = (buf1 == 2) ? 0 : buf1 + 1;
      // buf2 held stage stg-1: its last readers finished before the barrier that ended iteration stg-1
      if (stg + 2 < n_stages) stage_issue(stg + 2, buf2);
      SC_PHASE(accB, 1, accA, 0, buf, 0, true);                   // MFMA (stg, ib0, cb1)   || epilogue (stg, ib0, cb0)
      SC_PHASE(accA, 0, accB, 1, buf, 1, true);                   // MFMA (stg, ib1, cb0)   || epilogue (stg, ib0, cb1)
      SC_PHASE(accB, 1, accA, 0, buf, 1, true);                   // MFMA (stg, ib1, cb1)   || epilogue (stg, ib1, cb0)
      SC_PHASE(accA, 0, accB, 1, buf1, 0, stg + 1 < n_stages);    // MFMA (stg+1, ib0, cb0) || epilogue (stg, ib1, cb1)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of stage stg+2 (issued a whole iteration ago)
      __syncthreads();                                   // ... and everyone else's; buf is free again
      buf = buf1;
    }
  } else {
    // One step block per wave, two LDS stages.  Stage stg+1 is needed by the second phase of iteration stg, so the barrier sits
    // between the two phases: by then every wave has finished reading stage stg (the first phase was its last reader) and has
    // waited for its own pieces of stage stg+1; the buffer of stage stg is refilled with stage stg+2 right behind the barrier,
    // a whole iteration (64 MFMAs) before its first use.
    if (n_stages > 0) stage_issue(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (n_stages > 1) stage_issue(1, 1);
    if (n_stages > 0) {
      bias_init(accA, 0, 0);
#pragma unroll
      for (int s = 0; s < G::KS; s++) {
        const uint4 a = *reinterpret_cast<const uint4*>(smem + SC_FRAG(s));
        accA = mfma32(a, hf[0][s], accA);
      }
    }
    for (int stg = 0; stg < n_stages; stg++) {
      const int buf = stg & 1;
      SC_PHASE(accB, 0, accA, 0, buf, 1, true);                        // MFMA (stg, ib1)   || epilogue (stg, ib0)
      if (KT > 0) top_tile<KTA>(top, accA, (stg_lo + stg) * SC_STAGE + 4 * lh);        // (walk_pos is null: stage stg IS item stage stg)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (stg + 2 < n_stages) stage_issue(stg + 2, buf);
      SC_PHASE(accA, 0, accB, 0, buf ^ 1, 0, stg + 1 < n_stages);      // MFMA (stg+1, ib0) || epilogue (stg, ib1)
      if (KT > 0) top_tile<KTA>(top, accB, (stg_lo + stg) * SC_STAGE + 32 + 4 * lh);
    }
  }
#undef SC_PHASE
#undef SC_PHASE2
#undef SC_PHASE3
#undef SC_PHASE1
#undef SC_PHASE1W
#undef SC_GAPW
#undef SC_GAP1
#undef SC_FRAG

#pragma unroll
  for (int cbk = 0; cbk < NB; cbk++) {
    const float s_own = PACKED ? st[cbk].s2.x + st[cbk].s2.y : st[cbk].s;
    const int c_own = st[cbk].c;
    const int64_t myr = row0 + cbk * 32 + lr;
    // each half's log2-sum-exp2 first (s may be as large as 1e38 in PMODE 5), then the two-term merge
    const float l_own = st[cbk].m2 + log2f(s_own);
    const float l_o = __shfl_xor(l_own, 32);
    const int c_o = __shfl_xor(c_own, 32);
    const float s_other = __shfl_xor(s_own, 32);
    const float M2 = fmaxf(l_own, l_o);
    const float S = __builtin_amdgcn_exp2f(l_own - M2) + __builtin_amdgcn_exp2f(l_o - M2);
    const int cnt = c_own + c_o;
    if (SUMSQ || SIGMA) {   // sum of squares (SIGMA: of mu^2 exp(-2 s)) over ALL outputs of the row (null-id rows too: their y is the zero vector, not "no row")
      if (lh == 0 && myr < Rv) {
        if (o_ce) o_ce[mysrc[cbk]] = s_own + s_other;
        if (o_tgt) o_tgt[mysrc[cbk]] = myy[cbk] > 0 ? st[cbk].t : 0.f;
      }
      continue;
    }
    if (partials) {
      if (lh == 0 && myr < Rv) {
        float* pp = partials + (int64_t)by * 2 * R;
        pp[myr] = __int_as_float(cnt);
        pp[R + myr] = (n_stages > 0) ? M2 + log2f(S) : -INFINITY;   // log2 of this part's sum of 2^(z log2 e)
        if (by == 0 && o_tgt) o_tgt[mysrc[cbk]] = (myy[cbk] > 0) ? st[cbk].t : 0.f;
      }
      continue;
    }
    if (lh == 0 && myr < Rv) {
      const bool valid = myy[cbk] > 0;
      float lse = (M2 + log2f(S)) * LN2_F;
      // DUAL has no overflow redo: a running sum past fp32 is handed to the fix-up launch as an all-ones exponent in ce
      if ((DUAL || SEAM1) && ((__float_as_uint(s_own) & 0x7f800000u) == 0x7f800000u || (__float_as_uint(s_other) & 0x7f800000u) == 0x7f800000u))
        lse = INFINITY;
      const int64_t dst = mysrc[cbk];
      if (o_rank) o_rank[dst] = valid ? cnt : 0;
      if (o_ce) o_ce[dst] = valid ? (lse - st[cbk].t) : 0.f;
      if (o_tgt) o_tgt[dst] = valid ? st[cbk].t : 0.f;
    }
  }
  if constexpr (KT > 0) {
    // the two half-waves hold disjoint item subsets of the same step: merge the two sorted lists (value descending, then index
    // ascending) through LDS, which the item stream no longer needs
    float* lv = reinterpret_cast<float*>(smem);                       // [512 lanes][KT]
    int* li = reinterpret_cast<int*>(smem + SC_THREADS * KTA * 4);    // [512 lanes][KT]
    top_flush<KTA>(top);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KTA; k++) { lv[tid * KTA + k] = top.v[k]; li[tid * KTA + k] = top.i[k]; }
    __syncthreads();
    if (lh == 0 && row0 + lr < Rv) {
      const bool valid = myy[0] > 0;
      const int64_t dst = mysrc[0];
      const float* av = lv + tid * KTA; const int* ai = li + tid * KTA;
      const float* bv = lv + (tid + 32) * KTA; const int* bi = li + (tid + 32) * KTA;
      int pa = 0, pb = 0;
      for (int k = 0; k < topk; k++) {
        // masked rows are all-zero logit rows (model.py:105): every value ties at 0 -> indices 0..k-1
        const float va = pa < KTA ? av[pa] : -INFINITY, vb = pb < KTA ? bv[pb] : -INFINITY;
        const int ia = pa < KTA ? ai[pa] : 0x7fffffff, ib = pb < KTA ? bi[pb] : 0x7fffffff;
        const bool ta = (va > vb) || (va == vb && ia < ib);
        o_topi[dst * topk + k] = valid ? (ta ? ia : ib) : k;
        o_topv[dst * topk + k] = valid ? (ta ? va : vb) : 0.f;
        pa += ta ? 1 : 0;
        pb += ta ? 0 : 1;
      }
    }
  }
}

// adds up the catalogue parts of k_score_pipe (small-batch split): rank = sum of counts, lse = log-sum-exp over parts
// parts < 0: the device-side split (score_geom.h) of a launch of `dyn_grid` workgroups of `rows_per_wg` rows over n_all stages, -parts = its cap
__global__ void k_score_merge(const float* __restrict__ partials, int parts, int64_t R, const int32_t* __restrict__ row_map,
                              const int32_t* __restrict__ n_rows_dev, const int32_t* __restrict__ y, const float* __restrict__ tgt,
                              int32_t* __restrict__ o_rank, float* __restrict__ o_ce, int dyn_grid, int rows_per_wg, int n_all) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t Rv = n_rows_dev ? (int64_t)n_rows_dev[0] : R;
  if (p >= Rv) return;
  if (parts < 0) parts = score_dyn_split(Rv, rows_per_wg, n_all, dyn_grid, -parts).parts;
  const int64_t row = row_map ? (int64_t)row_map[p] : p;
  int cnt = 0;
  float M = -INFINITY;
  for (int k = 0; k < parts; k++) {
    cnt += __float_as_int(partials[(int64_t)k * 2 * R + p]);
    M = fmaxf(M, partials[(int64_t)k * 2 * R + R + p]);
  }
  float S = 0.f;
  for (int k = 0; k < parts; k++) S += __builtin_amdgcn_exp2f(partials[(int64_t)k * 2 * R + R + p] - M);
  const bool valid = y[row] > 0;
  o_rank[row] = valid ? cnt : 0;
  o_ce[row] = valid ? (M + log2f(S)) * LN2_F - tgt[row] : 0.f;
}

hipError_t launch_score_merge(const float* partials, int parts, int64_t R, const int32_t* row_map, const int32_t* n_rows_dev, const int32_t* y,
                               const float* tgt, int32_t* o_rank, float* o_ce, hipStream_t st, int dyn_grid, int rows_per_wg, int n_all) {
  hipLaunchKernelGGL(k_score_merge, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, partials, parts, R, row_map, n_rows_dev, y, tgt, o_rank, o_ce, dyn_grid,
                     rows_per_wg, n_all);
  return hipGetLastError();
}

template <int C, int NB, int NSTG, bool PAD>
static hipError_t launch_score_fixup(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                                     hipStream_t st, const int32_t* row_map, const int32_t* n_rows_dev, int* walk) {
  using G = ScGeom<C>;
  const size_t pitch = (PAD && C >= 256) ? 1024 + (C == 256 ? 32 : 16) : 1024;
  const size_t lds = NSTG * (G::STAGE_BYTES / 1024) * pitch + NSTG * SC_STAGE * sizeof(float);
  auto kern = k_score_pipe<C, 5, NB, NSTG, PAD, true>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)((R + 256 * NB - 1) / (256 * NB))), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N,
                     m.NP, out->rank, out->ce, out->target, row_map, n_rows_dev, 0, (float*)nullptr, walk, (int32_t*)nullptr, (float*)nullptr, 0);
  return hipGetLastError();
}

template <int C, int PMODE = 0, int NB = 2, int NSTG = 3, bool PAD = false>
static hipError_t launch_score_pipe(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R,
                                    const hiertcn_score_out_t* out, hipStream_t st, const int32_t* row_map = nullptr,
                                    const int32_t* n_rows_dev = nullptr, float* split_scratch = nullptr) {
  using G = ScGeom<C>;
  const size_t pitch = (PAD && C >= 256) ? 1024 + (C == 256 ? 32 : 16) : 1024;
  const size_t lds = NSTG * (G::STAGE_BYTES / 1024) * pitch + NSTG * SC_STAGE * sizeof(float);
  const unsigned blocks = (unsigned)((R + 256 * NB - 1) / (256 * NB));
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_score_pipe<C, PMODE, NB, NSTG, PAD, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  int* walk = m.walk_pos;   // owned by the handle (per device): hiertcn_create; null with HIERTCN_FIXED_WALK=1
  constexpr int FNB = (NB == 1 && NSTG == 3) ? 2 : NB;   // the fix-up flavour exists in the two-block shape at three stages (as launch_score_topk uses it)
  const int parts = score_split_parts(R, m.NP);
  if ((PMODE == 5 || PMODE >= 7) && split_scratch && out->rank && out->ce && out->target && score_dyn_applies(R, m.NP, 512 * NB)) {
    // small batch: SC_DYN_GRID workgroups, (row tile, catalogue part) decided on the device from the live row count (score_geom.h)
    const int n_all = m.NP / SC_STAGE;
    hipLaunchKernelGGL((k_score_pipe<C, PMODE, NB, NSTG, PAD, false>), dim3(SC_DYN_GRID), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP,
                       out->rank, out->ce, out->target, row_map, n_rows_dev, -SC_DYN_MAX_PARTS, split_scratch, (int*)nullptr, (int32_t*)nullptr, (float*)nullptr, 0);
    hipLaunchKernelGGL(k_score_merge, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, split_scratch, -SC_DYN_MAX_PARTS, R, row_map, n_rows_dev, y,
                       out->target, out->rank, out->ce, SC_DYN_GRID, 256 * NB, n_all);
    if constexpr (PMODE == 12 || PMODE == 14 || PMODE == 15) return launch_score_fixup<C, FNB, NSTG, PAD>(m, hidden, y, R, out, st, row_map, n_rows_dev, walk);
    return hipGetLastError();
  }
  if ((PMODE == 5 || PMODE >= 7) && split_scratch && parts > 1 && out->rank && out->ce && out->target) {
    const int n_all = m.NP / SC_STAGE;
    const int part_stages = (n_all + parts - 1) / parts;
    hipLaunchKernelGGL((k_score_pipe<C, PMODE, NB, NSTG, PAD, false>), dim3(blocks, parts), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP,
                       out->rank, out->ce, out->target, row_map, n_rows_dev, part_stages, split_scratch, (int*)nullptr, (int32_t*)nullptr, (float*)nullptr, 0);
    hipLaunchKernelGGL(k_score_merge, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, split_scratch, parts, R, row_map, n_rows_dev, y,
                       out->target, out->rank, out->ce, 0, 0, 0);
    if constexpr (PMODE == 12 || PMODE == 14 || PMODE == 15) return launch_score_fixup<C, FNB, NSTG, PAD>(m, hidden, y, R, out, st, row_map, n_rows_dev, walk);
    return hipGetLastError();
  }
  hipLaunchKernelGGL((k_score_pipe<C, PMODE, NB, NSTG, PAD, false>), dim3(blocks), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP,
                     out->rank, out->ce, out->target, row_map, n_rows_dev, 0, (float*)nullptr, walk, (int32_t*)nullptr, (float*)nullptr, 0);
  if constexpr (PMODE == 12 || PMODE == 14 || PMODE == 15) {
    if (out->ce) return launch_score_fixup<C, FNB, NSTG, PAD>(m, hidden, y, R, out, st, row_map, n_rows_dev, walk);
  }
  return hipGetLastError();
}

#include <stdio.h>
#include <stdlib.h>

#include <vector>

template <int C, bool LOGITS, int MODE>
static hipError_t launch_score_v(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R,
                                 const hiertcn_score_out_t* out, float* logits, int stagger, hipStream_t st) {
  using G = ScGeom<C>;
  const size_t lds = 2 * G::STAGE_BYTES + 2 * SC_STAGE * sizeof(float);
  const unsigned blocks = (unsigned)((R + 511) / 512);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_score<C, LOGITS, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_score<C, LOGITS, MODE>), dim3(blocks), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP,
                     out ? out->rank : nullptr, out ? out->ce : nullptr, out ? out->target : nullptr, logits, stagger);
  return hipGetLastError();
}

template <int C, int KT>
static hipError_t launch_score_topk(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R,
                                    const hiertcn_score_out_t* out, hipStream_t st, const int32_t* row_map = nullptr,
                                    const int32_t* n_rows_dev = nullptr) {
  using G = ScGeom<C>;
  if constexpr (C >= 64 && C <= 256) {
    // the seamless one-step-block kernel (three LDS stages, four-deep fragment ring) + the fix-up launch for a softmax sum past fp32;
    // (ablation build only -- HIERTCN_TOPK_V2=1: the two-stage flavour it replaces, HIERTCN_TOPK_V1=1: the un-pipelined kernel)
#if HIERTCN_ABLATION
    if (!getenv("HIERTCN_TOPK_V1") && !getenv("HIERTCN_TOPK_V2"))
#endif
    {
      constexpr bool PADI = C >= 256;
      const size_t pitch = PADI ? 1024 + (C == 256 ? 32 : 16) : 1024;
      size_t lds = 3 * (G::STAGE_BYTES / 1024) * pitch + 3 * SC_STAGE * sizeof(float);
      const size_t merge = (size_t)SC_THREADS * KT * 8;
      if (merge > lds) lds = merge;
      auto kern = k_score_pipe<C, 15, 1, 3, PADI, false, KT>;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3((unsigned)((R + 255) / 256)), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP, out->rank,
                         out->ce, out->target, row_map, n_rows_dev, 0, (float*)nullptr, (int*)nullptr, out->topk_idx, out->topk_val, (int)out->topk);
      if ((e = hipGetLastError()) != hipSuccess) return e;
      return out->ce ? launch_score_fixup<C, 2, 3, PADI>(m, hidden, y, R, out, st, row_map, n_rows_dev, (int*)nullptr) : hipSuccess;
    }
  }
  if constexpr (C == 512) {
    // the two-stage seamless kernel of the loss + rank flavour (k_score_pipe<512, 15, 1, 2, pad>) with the list test in its gaps + the fix-up launch;
    // (ablation build only -- HIERTCN_TOPK512_V1=1: the un-pipelined kernel it replaces, launch_score_512: 175 ms per 307,200 x 300,000 launch against 72)
#if HIERTCN_ABLATION
    if (!getenv("HIERTCN_TOPK512_V1"))
#endif
    {
      const size_t pitch = 1024 + 16;
      size_t lds = 2 * (G::STAGE_BYTES / 1024) * pitch + 2 * SC_STAGE * sizeof(float);
      const size_t merge = (size_t)SC_THREADS * KT * 8;
      if (merge > lds) lds = merge;
      auto kern = k_score_pipe<512, 15, 1, 2, true, false, KT>;
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(kern, dim3((unsigned)((R + 255) / 256)), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP, out->rank,
                         out->ce, out->target, row_map, n_rows_dev, 0, (float*)nullptr, (int*)nullptr, out->topk_idx, out->topk_val, (int)out->topk);
      if ((e = hipGetLastError()) != hipSuccess) return e;
      return out->ce ? launch_score_fixup<512, 1, 2, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, (int*)nullptr) : hipSuccess;
    }
#if HIERTCN_ABLATION
    return hipErrorNotSupported;   // (the caller falls back to launch_score_512)
#endif
  } else {
#if !HIERTCN_ABLATION
  return hipErrorInvalidValue;   // (not reached: every width returned above)
#else
  if (!getenv("HIERTCN_TOPK_V1")) {   // the pipelined skeleton (epilogue in the MFMA gaps, one step block per wave, two LDS stages); V1: A/B
    constexpr bool PADI = C >= 256;
    const size_t pitch = PADI ? 1024 + (C == 256 ? 32 : 16) : 1024;
    size_t lds = 2 * (G::STAGE_BYTES / 1024) * pitch + 2 * SC_STAGE * sizeof(float);
    const size_t merge = (size_t)SC_THREADS * KT * 8;
    if (merge > lds) lds = merge;
    auto kern = k_score_pipe<C, 5, 1, 2, PADI, false, KT>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((R + 255) / 256)), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP, out->rank,
                       out->ce, out->target, row_map, n_rows_dev, 0, (float*)nullptr, (int*)nullptr, out->topk_idx, out->topk_val, (int)out->topk);
    return hipGetLastError();
  }
  size_t lds = 2 * G::STAGE_BYTES + 2 * SC_STAGE * sizeof(float);
  const size_t merge = (size_t)SC_THREADS * KT * 8;
  if (merge > lds) lds = merge;
  const unsigned blocks = (unsigned)((R + 255) / 256);  // NB = 1: 8 waves x 32 rows
  auto kern = k_score<C, false, 0, false, KT, 1>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP, out->rank, out->ce,
                     out->target, (float*)nullptr, 0, (unsigned long long*)nullptr, out->topk_idx, out->topk_val, (int)out->topk,
                     (const float*)nullptr, (const float*)nullptr, 0, -1, row_map, n_rows_dev);
  return hipGetLastError();
#endif
  }
}

template <int C>
static hipError_t launch_score_c(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R,
                                 const hiertcn_score_out_t* out, float* logits, hipStream_t st, const int32_t* row_map,
                                 const int32_t* n_rows_dev, float* split_scratch) {
  if (logits) return launch_score_v<C, true, 0>(m, hidden, y, R, nullptr, logits, 0, st);
  if (out && out->topk > 0)  // loss.py:120 ranked indices beside rank / ce: the list kernel (slower epilogue)
    return out->topk <= 8    ? launch_score_topk<C, 8>(m, hidden, y, R, out, st, row_map, n_rows_dev)
           : out->topk <= 10 ? launch_score_topk<C, 10>(m, hidden, y, R, out, st, row_map, n_rows_dev)   // the reference's recall@10 lists
                             : launch_score_topk<C, 16>(m, hidden, y, R, out, st, row_map, n_rows_dev);
#ifdef HIERTCN_ABLATION
  // Ablation / timing flavours (profiles/r1_score_ablation.md; several give wrong results by design).  Only in libraries built with
  // -DHIERTCN_ABLATION=1 (HIERTCN_ABLATION=1 python -m hiertcn_amd.build): the production library holds production kernels only.
  if (getenv("HIERTCN_SCORE_STAMPS")) {  // diagnostic build: per-phase cycle sums of the first workgroups, printed to stderr
    using G = ScGeom<C>;
    const int stagger_d = getenv("HIERTCN_SCORE_VARIANT") && getenv("HIERTCN_SCORE_VARIANT")[1] == '0' ? 0 : 1;
    const size_t lds = 2 * G::STAGE_BYTES + 2 * SC_STAGE * sizeof(float);
    const unsigned blocks = (unsigned)((R + 511) / 512);
    unsigned long long* d = nullptr;
    const size_t n = (size_t)blocks * 8 * SC_NSTAMP;
    if (hipMalloc(&d, n * 8) != hipSuccess) return hipErrorOutOfMemory;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_score<C, false, 0, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((k_score<C, false, 0, true>), dim3(blocks), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N,
                       m.NP, out->rank, out->ce, out->target, (float*)nullptr, stagger_d, d);
    (void)hipStreamSynchronize(st);
    std::vector<unsigned long long> hbuf(n);
    (void)hipMemcpy(hbuf.data(), d, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    const char* names[SC_NSTAMP] = {"issue", "defer-epi", "mfma0", "epi0", "mfma1", "epi1", "vmcnt", "barrier"};
    const double ns = (double)(m.NP / SC_STAGE);
    for (unsigned b = 0; b < blocks && b < 600; b += 257) {
      for (int w = 0; w < 8; w++) {
        fprintf(stderr, "[stamps] wg %u wave %d:", b, w);
        double tot = 0;
        for (int i = 0; i < SC_NSTAMP; i++) {
          const double v = hbuf[((size_t)b * 8 + w) * SC_NSTAMP + i] / ns;
          tot += v;
          fprintf(stderr, " %s=%.0f", names[i], v);
        }
        fprintf(stderr, " total=%.0f cyc/stage\n", tot);
      }
    }
    return hipGetLastError();
  }
  // ablation switch for profiling runs only: "<mode><stagger>", e.g. "20" = no epilogue, no stagger
  static const char* var = getenv("HIERTCN_SCORE_VARIANT");
  static const char* pvar = getenv("HIERTCN_PIPE_VARIANT");
  if (pvar && pvar[0] == '5') return launch_score_pipe<C, 5>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);   // previous production, for A/B runs
  if (pvar && pvar[0] == '0') return launch_score_pipe<C, 0>(m, hidden, y, R, out, st);  // v3, for A/B runs
  if (pvar && pvar[0] == '6') return launch_score_pipe<C, 6>(m, hidden, y, R, out, st);
  if (pvar && pvar[0] == '7') return launch_score_pipe<C, 7>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  if (pvar && pvar[0] == '8') return launch_score_pipe<C, 8>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  if (pvar && pvar[0] == '9') return launch_score_pipe<C, 9>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  if (C == 256 && pvar && pvar[0] == 'c') return launch_score_pipe<256, 12, 2, 3, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  if (C == 256 && pvar && pvar[0] == 'd') return launch_score_pipe<256, 13, 2, 3, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  if (C == 256 && pvar && pvar[0] == 'h') return launch_score_pipe<256, 16, 2, 3, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  if (C == 256 && pvar && pvar[0] == 'f') return launch_score_pipe<256, 14, 2, 3, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  if (C == 256 && pvar && pvar[0] == 'e') return launch_score_pipe<256, 5, 2, 3, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  if (C == 256 && pvar && pvar[0] == 'a') return launch_score_pipe<256, 10>(m, hidden, y, R, out, st);
  if (C == 256 && pvar && pvar[0] == 'b') return launch_score_pipe<256, 11>(m, hidden, y, R, out, st);
  if (C == 256 && pvar && pvar[0]) {
    switch (pvar[0]) {
      case '1': return launch_score_pipe<256, 1>(m, hidden, y, R, out, st);
      case '2': return launch_score_pipe<256, 2>(m, hidden, y, R, out, st);
      case '3': return launch_score_pipe<256, 3>(m, hidden, y, R, out, st);
      case '4': return launch_score_pipe<256, 4>(m, hidden, y, R, out, st);
      default: break;
    }
  }
  if (var && var[0]) {
    int mode = 0, stagger = 1;
    mode = var[0] - '0';
    if (var[1]) stagger = var[1] - '0';
    if (mode == 1) return launch_score_v<C, false, 1>(m, hidden, y, R, out, nullptr, stagger, st);
    if (mode == 2) return launch_score_v<C, false, 2>(m, hidden, y, R, out, nullptr, stagger, st);
    return launch_score_v<C, false, 0>(m, hidden, y, R, out, nullptr, stagger, st);
  }
#endif
  // production
  if constexpr (C == 256) {  // four accumulator tiles, padded LDS image, overflow handled by the fix-up launch
    // small batches: ONE step block per wave (256-row tiles, the list flavour's skeleton without lists): twice the row tiles for the device-side
    // split to deal out, half the MFMA chain per stage (a 32-lane XING batch: ~14 live tiles x 18 parts on 252 CUs instead of ~7 x 19 on 133)
    static const bool nb1 = !ABL_ENV("HIERTCN_SCORE_SMALL_NB2");
    if (nb1 && split_scratch && out->rank && out->ce && out->target && score_dyn_applies(R, m.NP, 512))
      return launch_score_pipe<256, 15, 1, 3, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
    return launch_score_pipe<256, 14, 2, 3, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  }
  else {  // C = 64 / 128: the same four-tile seamless shape on the XOR-swizzled image (registers are not the problem here)
    static const bool nb1 = !ABL_ENV("HIERTCN_SCORE_SMALL_NB2");
    if (nb1 && split_scratch && out->rank && out->ce && out->target && score_dyn_applies(R, m.NP, 512))
      return launch_score_pipe<C, 15, 1, 3>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
    return launch_score_pipe<C, 14>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  }
}

template <bool LOGITS, int MODE, int KT>
static hipError_t launch_score_512(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                                   float* logits, const float* lse, const float* w, int stg_begin, int stg_end, hipStream_t st,
                                   const int32_t* row_map = nullptr, const int32_t* n_rows_dev = nullptr);

#if HIERTCN_ABLATION   // the legacy training backward's d-logits chunks (train.hip, HIERTCN_TRAIN_LEGACY=1)
template <int C>
static hipError_t launch_score_grad_c(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const float* lse,
                                      const float* w, int stg_begin, int stg_end, uint16_t* gout, hipStream_t st) {
  using G = ScGeom<C>;
  const size_t lds = 2 * G::STAGE_BYTES + 2 * SC_STAGE * sizeof(float);
  const unsigned blocks = (unsigned)((R + 511) / 512);
  auto kern = k_score<C, true, 9>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP, (int32_t*)nullptr,
                     (float*)nullptr, (float*)nullptr, reinterpret_cast<float*>(gout), 0, (unsigned long long*)nullptr, (int32_t*)nullptr,
                     (float*)nullptr, 0, lse, w, stg_begin, stg_end, (const int32_t*)nullptr, (const int32_t*)nullptr);
  return hipGetLastError();
}

// d loss / d logits for the items [stg_begin*64, stg_end*64) as bf16 [R][(stg_end-stg_begin)*64] (training backward)
hipError_t launch_score_grad_chunk(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const float* lse, const float* w,
                                   int stg_begin, int stg_end, uint16_t* gout, hipStream_t st) {
  if (R <= 0 || stg_end <= stg_begin) return hipSuccess;
  switch (m.Clast) {
    case 64: return launch_score_grad_c<64>(m, hidden, y, R, lse, w, stg_begin, stg_end, gout, st);
    case 128: return launch_score_grad_c<128>(m, hidden, y, R, lse, w, stg_begin, stg_end, gout, st);
    case 256: return launch_score_grad_c<256>(m, hidden, y, R, lse, w, stg_begin, stg_end, gout, st);
    case 512:
      return launch_score_512<true, 9, 0>(m, hidden, y, R, nullptr, reinterpret_cast<float*>(gout), lse, w, stg_begin, stg_end, st);
    default: return hipErrorInvalidValue;
  }
}

#endif

// d = 512 (BASELINE.json configs[3]: the scoring-GEMM stress shape): 64 rows x 512 channels of resident B operand
// would take 256 VGPRs, so every flavour (fused, top-k, logits, gradient chunk) runs the two-stage kernel with ONE
// 32-step block per wave (128 VGPRs of B operand, 2 x 64 KB LDS stages, 32 MFMAs per 16 logits of epilogue).
template <bool LOGITS, int MODE, int KT>
static hipError_t launch_score_512(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                                   float* logits, const float* lse, const float* w, int stg_begin, int stg_end, hipStream_t st,
                                   const int32_t* row_map, const int32_t* n_rows_dev) {
  using G = ScGeom<512>;
  size_t lds = 2 * G::STAGE_BYTES + 2 * SC_STAGE * sizeof(float);
  const unsigned blocks = (unsigned)((R + 255) / 256);
  auto kern = k_score<512, LOGITS, MODE, false, KT, 1>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP, out ? out->rank : nullptr,
                     out ? out->ce : nullptr, out ? out->target : nullptr, logits, 0, (unsigned long long*)nullptr,
                     out ? out->topk_idx : nullptr, out ? out->topk_val : nullptr, out ? (int)out->topk : 0, lse, w, stg_begin, stg_end,
                     row_map, n_rows_dev);
  return hipGetLastError();
}

hipError_t launch_score_fixup_only(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                                   hipStream_t st, const int32_t* row_map, const int32_t* n_rows_dev) {
  if (R <= 0) return hipSuccess;
  switch (m.Clast) {
    case 64: return launch_score_fixup<64, 2, 3, false>(m, hidden, y, R, out, st, row_map, n_rows_dev, m.walk_pos);
    case 128: return launch_score_fixup<128, 2, 3, false>(m, hidden, y, R, out, st, row_map, n_rows_dev, m.walk_pos);
    case 256: return launch_score_fixup<256, 2, 3, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, m.walk_pos);
    case 512: return launch_score_fixup<512, 1, 2, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, m.walk_pos);
    default: return hipErrorInvalidValue;
  }
}

// Mixture head (gmm.hip): for the table (m.outT [NP][C], m.out_bias [NP], pad rows / biases ZERO) -> o_sq[r] = sum_n z[r][n]^2,
// o_tgt[r] = z[r][y[r]] (0 for the null id); the same streaming kernel with a sum-of-squares epilogue.
hipError_t launch_score_sumsq(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, float* o_sq, float* o_tgt, hipStream_t st) {
  if (R <= 0) return hipSuccess;
  hiertcn_score_out_t out;
  out.rank = nullptr; out.ce = o_sq; out.target = o_tgt; out.topk_idx = nullptr; out.topk_val = nullptr; out.topk = 0;
  if (!ABL_ENV("HIERTCN_SUMSQ_V1")) {   // the four-tile seamless skeleton with the one-fma epilogue (ablation build, HIERTCN_SUMSQ_V1: the un-pipelined kernel)
    switch (m.Clast) {
      case 64: return launch_score_pipe<64, 21>(m, hidden, y, R, &out, st);
      case 128: return launch_score_pipe<128, 21>(m, hidden, y, R, &out, st);
      case 256: return launch_score_pipe<256, 21, 2, 3, true>(m, hidden, y, R, &out, st);
      default: break;
    }
  }
  switch (m.Clast) {
#if HIERTCN_ABLATION
    case 64: return launch_score_v<64, false, 3>(m, hidden, y, R, &out, nullptr, 0, st);
    case 128: return launch_score_v<128, false, 3>(m, hidden, y, R, &out, nullptr, 0, st);
    case 256: return launch_score_v<256, false, 3>(m, hidden, y, R, &out, nullptr, 0, st);
#endif
    case 512: return launch_score_512<false, 3, 0>(m, hidden, y, R, &out, nullptr, nullptr, nullptr, 0, -1, st);
    default: return hipErrorInvalidValue;
  }
}

// Mixture head with a learned scale (gmm.hip): the table interleaves, per 64-row stage, the 32 mean rows and the 32 log-scale rows of the same
// outputs (pad rows / biases ZERO) -> o_q[r] = sum_d mu[r][d]^2 exp(-2 s[r][d]).  y is read (the kernel's prologue gathers a row) but no target
// column means anything in this layout: the caller gathers mu[y] and s[y] itself.
hipError_t launch_score_sigma(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, float* o_q, hipStream_t st) {
  if (R <= 0) return hipSuccess;
  hiertcn_score_out_t out;
  out.rank = nullptr; out.ce = o_q; out.target = nullptr; out.topk_idx = nullptr; out.topk_val = nullptr; out.topk = 0;
  switch (m.Clast) {
    case 64: return launch_score_pipe<64, 22, 1, 3, false>(m, hidden, y, R, &out, st);
    case 128: return launch_score_pipe<128, 22, 1, 3, false>(m, hidden, y, R, &out, st);
    case 256: return launch_score_pipe<256, 22, 1, 3, true>(m, hidden, y, R, &out, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_score(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                        float* logits, hipStream_t st, const int32_t* row_map, const int32_t* n_rows_dev, float* split_scratch) {
  if (R <= 0) return hipSuccess;
  switch (m.Clast) {
    case 64: return launch_score_c<64>(m, hidden, y, R, out, logits, st, row_map, n_rows_dev, split_scratch);
    case 128: return launch_score_c<128>(m, hidden, y, R, out, logits, st, row_map, n_rows_dev, split_scratch);
    case 256: return launch_score_c<256>(m, hidden, y, R, out, logits, st, row_map, n_rows_dev, split_scratch);
    case 512:   // (dense rows: callers zero-fill nothing here, the kernel writes rank 0 / ce 0 on masked rows itself)
      if (logits) return launch_score_512<true, 0, 0>(m, hidden, y, R, nullptr, logits, nullptr, nullptr, 0, -1, st);
      if (out && out->topk > 0 && !ABL_ENV("HIERTCN_TOPK512_V1"))   // lists in the gaps of the seamless two-stage kernel (round 3)
        return out->topk <= 8    ? launch_score_topk<512, 8>(m, hidden, y, R, out, st, row_map, n_rows_dev)
               : out->topk <= 10 ? launch_score_topk<512, 10>(m, hidden, y, R, out, st, row_map, n_rows_dev)
                                 : launch_score_topk<512, 16>(m, hidden, y, R, out, st, row_map, n_rows_dev);
#if HIERTCN_ABLATION   // the kernels the production forms replaced, for A/B runs on one box
      if (out && out->topk > 8) return launch_score_512<false, 0, 16>(m, hidden, y, R, out, nullptr, nullptr, nullptr, 0, -1, st, row_map, n_rows_dev);
      if (out && out->topk > 0) return launch_score_512<false, 0, 8>(m, hidden, y, R, out, nullptr, nullptr, nullptr, 0, -1, st, row_map, n_rows_dev);
      if (getenv("HIERTCN_SCORE512_V1")) return launch_score_512<false, 0, 0>(m, hidden, y, R, out, nullptr, nullptr, nullptr, 0, -1, st);
      if (getenv("HIERTCN_SCORE512_XOR")) return launch_score_pipe<512, 5, 1, 2>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
      if (getenv("HIERTCN_SCORE512_V2")) return launch_score_pipe<512, 5, 1, 2, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);   // per-tile redo, bubble at every tile start
#endif
      return launch_score_pipe<512, 15, 1, 2, true>(m, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
    default: return hipErrorInvalidValue;
  }
}
