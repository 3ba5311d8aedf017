// Training backward of the scoring layer WITHOUT the d-logits matrix (model_tcn.py:41 differentiated by model.py:139-141).
//   g[r][n] = w_r (softmax(z_r)[n] - [n == y_r]),  z = h . Wo + bo;   d h = g . Wo^T,   d Wo = h^T . g,   d bo = sum_r g.
// One kernel template, two roles -- the attention-backward shape (recompute the probabilities tile by tile from the saved
// log-sum-exp, feed them straight into the second product as an MFMA operand, never store them):
//
//   ROWSTAT  (d hidden): a wave keeps 32 scored rows (their hidden vectors: GEMM-1's B operand, registers) and the running
//            d h^T [C x 32 rows] (C/32 accumulator tiles), the workgroup streams the item table Wo^T through the LDS ring;
//   ITEMSTAT (d Wo, d bo): a wave keeps 32 items (their Wo rows) and the running d Wo^T [C x 32 items], the workgroup streams the
//            scored hidden rows; d bo is a lane-local sum.
//
// Per 32 x 32 tile: GEMM-1 (C/16 MFMAs) D[stream][stationary] = stream-vector . stationary-vector, accumulator started from the
// per-stream scalar (bo[n], or ln w_r - lse_r); p = 2^(acc log2e + stationary scalar) is w_r softmax(z_r)[n] exactly (both scalars
// folded), 16 values per lane; pairs of them become bf16 B-operand fragments with no lane movement (an accumulator tile is the next
// MFMA's operand when that product sums over the tile's ROW index: cdna_hip_programming.md 3) and GEMM-2 (C/16 MFMAs)
// out^T[c][stationary] += stream^T[c][stream-vector] . p reads the SAME LDS stage transposed with ds_read_b64_tr_b16.
// One LDS image serves the row reads (ds_read_b128) and the transposed reads conflict-free: chunk' = chunk ^ (((row & 3) << 2) |
// ((row >> 2) & 3)) (cdna_hip_programming.md T10, image (b)), applied on the SOURCE address of the LDS-DMA.
// The one-hot term is one element per row: k_dh_finish / k_dwo_onehot subtract it afterwards.
// Everything is sized by the DEVICE-side count of scored rows (metrics.hip, launch_compact_rows): no host read-back in the step.
#include <string.h>

#include "score_geom.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint2 lds_read_tr16(const unsigned char* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}

// MODE 0  ITEMSTAT  (d Wo, d bo)      stat = Wo^T rows, stream = compact scored rows
// MODE 1  ROWSTAT   (d hidden)        stat = compact scored rows, stream = Wo^T; p normalised by the SAVED log-sum-exp.  With `vote` set it
//                                     is the FIX-UP of mode 2: a 256-row block runs only if one of its rows is flagged in `ovf`
// MODE 2  ROWSTAT + forward statistics, "flash" form: the training step's ONE pass over the logits of a row computes rank, log-sum-exp
//         and target logit (what k_score_pipe computes in evaluation) AND the unnormalised sum_n 2^(z_n log2e - m) Wo[n], which divided
//         by the final sum is softmax . Wo -- d hidden without a second recompute.  The reference point m is fixed (target logit + 64 in
//         log2 units, as in score.hip), so nothing is ever rescaled; a row whose sum overflows fp32 (a logit > 133 nats above the target)
//         is flagged in `ovf` and redone by the fix-up launches (score.hip FIXUP flavour for rank / ce, mode 1 + vote for d hidden).
// Workgroup shape by width.  C <= 256: 8 waves (two per SIMD), 64 stream vectors per LDS stage (two 32-vector tiles), three stages.
// C = 512 (round 3): the stationary operand (128 VGPRs) and the running output (256) of a wave no longer fit two waves per SIMD, so a
// workgroup is 4 waves -- one per SIMD, up to 512 registers each -- of 32 stationary vectors, and a stage is ONE 32-vector tile of 32 KB
// (three stages = 96 KB).  Everything else (LDS image, fragment addressing, the software pipeline) is the same code.
// ONE = one wave per SIMD: what C = 512 needs.  At C = 256 it was measured in round 4 and is slower (mode 0: 150 -> 230 ms, mode 2: 167 -> 232 ms per
// 1024-user step, bit-identical results: profiles/r4_train_notes.md), so only <512, *, true> and <128 / 256, *, false> are instantiated.
// C = 64 (round 4, a narrow `args.tcn_channel`): the dual-use LDS image needs rows of at least 256 bytes (the 4-bit chunk swizzle), so a 128-byte
// vector sits in a 256-byte LDS row (LP = 128 channels of pitch; logical chunks 0..7 spread over the 16 physical ones by the same XOR, the other
// eight positions are filled with a second copy by the DMA and never read): same addressing, half the k-steps and half the output tiles.
template <int C, bool ONE = (C == 512)>
struct SgGeom {
  static constexpr int W = ONE ? 4 : 8;        // waves per workgroup
  static constexpr int SI = ONE ? 32 : 64;     // stream vectors per LDS stage
  static constexpr int TILES = SI / 32;
  static constexpr int THREADS = 64 * W;
  static constexpr int STAT = 32 * W;                 // stationary vectors per workgroup
  static constexpr int LP = C < 128 ? 128 : C;        // channels of LDS row pitch
  static constexpr int STAGE_BYTES = SI * 2 * LP;
};
static bool sg_one(int C, int mode) { (void)mode; return C == 512; }
static int sg_stat(int C, bool one) { return one ? 128 : 256; }
static int sg_si(int C, bool one) { return one ? 32 : 64; }

struct SgradArgs {
  const uint16_t* stat; const float* stat_const; int64_t n_stat_max; const int32_t* n_stat_dev;
  const uint16_t* stream; const float* stream_bias; int n_stages_max; const int32_t* n_stream_dev;
  float* out;            // mode 0: d Wo [NP][C] (+=); mode 1 / 2: d hidden [R][C], written at row row_map[p]
  float* colsum;         // mode 0: d bo
  // modes 1, 2
  const int32_t* row_map; const int32_t* yc; const float* wc;   // mode 1: compact arrays y, w by p; mode 2: y and w by ORIGINAL row
  const uint16_t* outT; int N;                                    // the one-hot term  - w Wo[y]
  int32_t* ovf; int vote;
  int* walk;             // per-XCD stream-position words ([8][32] ints, the handle's; null: every workgroup starts at stage 0)
  // mode 2
  const float* out_bias; int32_t* o_rank; float* o_ce; float* o_tgt;
  // mode 2, small batches: the table split in parts decided on the device (score_geom.h, score_dyn_split).  part [2 parts + 1][part_ld]: per part the
  // row's sum and count, then the target logit; behind them dyn_max_parts slabs [part_ld][C] of the parts' unscaled d hidden by COMPACT row (plain
  // stores: 16.7 M scattered fp32 atomics per launch made this split only 1.7 x faster than no split); k_sgrad_finish adds the parts up, normalises,
  // and writes rank / ce / target / d hidden
  float* part; int64_t part_ld; int dyn_max_parts;
  // mode 0, small catalogues: part y > 0 of the scored rows STORES its d Wo into slab y - 1 of `slabs` ([NP][C] each) instead of adding atomically
  // (k_add_slabs folds them into `out` afterwards)
  float* slabs;
};

template <int C, int MODE, bool ONE = (C == 512)>
__global__ __launch_bounds__((SgGeom<C, ONE>::THREADS), (ONE ? 1 : 2)) void k_sgrad(const SgradArgs a) {
  using G = ScGeom<C>;
  using SG = SgGeom<C, ONE>;
  static_assert(C == 64 || C == 128 || C == 256 || C == 512, "the dual-use LDS image is laid out for 256-, 512- and 1024-byte rows");
  constexpr int NCB = C / 32;
  constexpr int LP = SG::LP;
  constexpr int SI = SG::SI;
  // GEMM-1 as two accumulator chains (even / odd k-steps, added at the end).  C = 512 (one wave per SIMD): nothing else fills the latency of a
  // dependent MFMA chain, fragment reads three ahead (1061 -> 899 ms per step).  C = 256 (two waves per SIMD, 254 of 256 registers in use): the
  // second wave hides most of it already, and round 3's first attempt (four-deep fragment ring) spilled -- 396 vs 344 ms; with the reads one ahead,
  // as the single chain has them, modes 0 / 1 fit and mode 2 spills 9 registers outside the MFMA phases: 337.6 -> 331.1 ms per 1024-user step.
  // -DHIERTCN_SGRAD_ONE_CHAIN_256: the single chain, for A/B.
#ifdef HIERTCN_SGRAD_ONE_CHAIN_256
  constexpr bool TWO_CHAINS = ONE;
#else
  constexpr bool TWO_CHAINS = C >= 256;
#endif
#ifdef HIERTCN_SGRAD_WHOLE_PAIRS   // A/B: a pair's whole epilogue in one MFMA gap (until round 3)
  constexpr bool SPLIT_PAIRS = false;
#else
  constexpr bool SPLIT_PAIRS = NCB >= 8;
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 3 stages of 64 stream vectors + 3 x 64 stream scalars
  const int64_t n_stat = a.n_stat_dev ? (int64_t)a.n_stat_dev[0] : a.n_stat_max;
  int bx = (int)blockIdx.x, by = (int)blockIdx.y, ny = (int)gridDim.y;
  if (MODE == 2 && a.dyn_max_parts > 0) {
    const ScDynSplit d = score_dyn_split(n_stat, SG::STAT, a.n_stages_max, (int)gridDim.x, a.dyn_max_parts);
    by = bx / d.tiles;
    bx -= by * d.tiles;
    ny = d.parts;
    if (by >= ny) return;
  }
  if ((int64_t)bx * SG::STAT >= n_stat) return;
  if (MODE == 1 && a.vote) {   // fix-up launch: only blocks with a flagged row do anything
    const int64_t p = (int64_t)blockIdx.x * SG::STAT + threadIdx.x;
    const int bad = ((int)threadIdx.x < SG::STAT && p < n_stat) ? a.ovf[p] : 0;
    if (!__syncthreads_or(bad)) return;
  }
  int n_stages = a.n_stream_dev ? (int)(((int64_t)a.n_stream_dev[0] + SI - 1) / SI) : a.n_stages_max;
  if (n_stages > a.n_stages_max) n_stages = a.n_stages_max;
  int s_lo = 0;
  if (ny > 1) {   // the stream split in parts (mode 0: the item dimension alone gives N / 256 workgroups): partial sums added atomically
    const int per = (n_stages + ny - 1) / ny;
    s_lo = by * per;
    const int s_hi = (s_lo + per < n_stages) ? s_lo + per : n_stages;
    if (s_hi <= s_lo) {
      if (!(MODE == 0 && a.slabs)) return;
      s_lo = 0;   // a slab part without rows still stores its (zero) tile: k_add_slabs reads every slab
      n_stages = 0;
    } else {
      n_stages = s_hi - s_lo;
    }
  }
  const uint16_t* stream = a.stream + (int64_t)s_lo * SI * C;
  const float* stream_bias = a.stream_bias + (int64_t)s_lo * SI;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t sv = ((int64_t)bx * SG::W + wave) * 32 + lr;   // this lane's stationary vector
  const int64_t svc = sv < n_stat ? sv : n_stat - 1;   // (lanes past the end read the last live vector: finite data, their column is never stored)
  const int64_t srow = (MODE == 2) ? (int64_t)a.row_map[svc] : svc;   // mode 2 reads the hidden rows in place

  uint4 hf[G::KS];
  {
    const uint16_t* sp = a.stat + srow * C + 8 * lh;
#pragma unroll
    for (int s = 0; s < G::KS; s++) hf[s] = *reinterpret_cast<const uint4*>(sp + 16 * s);
  }
  float cst;            // the stationary vector's scalar inside the exponent: (ln w - lse) log2e, bo log2e, or -(reference) in mode 2
  float tlog = 0.f;     // mode 2: the target logit
  int myy = 0;
  if (MODE == 2) {
    // target logit: the same MFMA sequence as the streamed logits on the gathered row Wo^T[y] (diagonal of a 32 x 32 tile), so that
    // z_n > z_target is exactly false for n == y (score.hip file header)
    const int yy = (sv < n_stat) ? a.yc[srow] : 0;
    myy = (yy > 0 && yy < a.N) ? yy : 0;
    f32x16 acc;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) acc[reg] = a.out_bias[__shfl(myy, acc_row(reg, lh))];
    const uint16_t* ap = a.outT + (int64_t)myy * C + 8 * lh;
    if constexpr (TWO_CHAINS) {   // two accumulator chains, even / odd k-steps, added at the end: exactly what g1 does below (bitwise equal logits)
      f32x16 acc2;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) acc2[reg] = 0.f;
#pragma unroll
      for (int s = 0; s < G::KS; s += 2) {
        acc = mfma32(*reinterpret_cast<const uint4*>(ap + 16 * s), hf[s], acc);
        acc2 = mfma32(*reinterpret_cast<const uint4*>(ap + 16 * (s + 1)), hf[s + 1], acc2);
      }
#pragma unroll
      for (int reg = 0; reg < 16; reg++) acc[reg] += acc2[reg];
    } else {
#pragma unroll
      for (int s = 0; s < G::KS; s++) acc = mfma32(*reinterpret_cast<const uint4*>(ap + 16 * s), hf[s], acc);
    }
    const int dreg = (lr & 3) + 4 * (lr >> 3);
    float d = acc[0];
#pragma unroll
    for (int reg = 1; reg < 16; reg++) d = (dreg == reg) ? acc[reg] : d;
    const float other = __shfl_xor(d, 32);
    tlog = (((lr >> 2) & 1) == lh) ? d : other;
    cst = -(tlog * LOG2E_F + 64.0f);
  } else {
    cst = (sv < n_stat) ? a.stat_const[svc] : -INFINITY;   // -inf: every p of this column is 0
  }

  constexpr int PIECES = SG::STAGE_BYTES / 1024;
  constexpr int PPW = PIECES / SG::W;
  constexpr int STAGE_LDS = SG::STAGE_BYTES;
  constexpr int TILE_LDS = STAGE_LDS / SG::TILES;
  constexpr int BIAS_OFF = 3 * STAGE_LDS;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  auto dma_voff = [&](int i) -> unsigned {   // recomputed at every issue: registers are the scarce resource here, not VALU slots
    const int off = (wave * PPW + i) * 1024 + lane * 16;
    const int row = off / (2 * LP), slot = (off % (2 * LP)) >> 4;   // LDS row and physical chunk this lane's 16 bytes land in
    return (unsigned)(row * (2 * C) + (((slot ^ swz(row)) & (C / 8 - 1)) << 4));   // source: the logical chunk (C = 64: two copies of each)
  };
  // XCD-synchronised walk of the stream (score.hip, k_score_pipe: the same reasoning and the same caveats).  Every sum this kernel forms
  // over the stream -- the softmax denominator and count of mode 2, the d hidden / d Wo accumulators -- is independent of the stream
  // order beyond fp32 rounding, so a workgroup may enter the stream at any stage and wrap around.  Workgroups dealt to one XCD
  // (blockIdx.x % 8) start where their running neighbours are, which keeps an XCD's workgroups on the same few MB of the 150 MB item
  // table (mode 2) or of the scored rows (mode 0) however the dispatcher staggers them: without it the bytes through the L2's
  // memory-side port depend on the box (35 GB at 91 % hit on one, 98 GB at 76 % on another, profiles/r2_b vs r2_c).
  // ONE read per workgroup, handed to the waves through LDS (waves reading the moving word themselves could disagree).
  int s0 = 0;
  int* walk_word = nullptr;
  if (a.walk && !(MODE == 1 && a.vote)) {
    walk_word = a.walk + (blockIdx.x & 7) * 32 + 8 + 4 * MODE + (by & 3);
    int* s0_sh = reinterpret_cast<int*>(smem);   // (the stage buffers are not in use yet)
    if (tid == 0) *s0_sh = __hip_atomic_load(walk_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    s0 = __builtin_amdgcn_readfirstlane(*s0_sh);
    __syncthreads();
    s0 = (s0 >= 0 && s0 < n_stages) ? s0 : 0;
  }
  auto stage_issue = [&](int stg_rel, int buf) {   // LDS-DMA by hand (score.hip: why asm); SGPR stage base + 32-bit lane offset
    int stg = s0 + stg_rel;
    stg -= (stg >= n_stages) ? n_stages : 0;
    if (walk_word && wave == 0 && lane == 0 && (stg_rel & 7) == 0) __hip_atomic_store(walk_word, stg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint16_t* sbase = stream + (int64_t)stg * SI * C;
#pragma unroll
    for (int i = 0; i < PPW; i++) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_LDS + (wave * PPW + i) * 1024);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(dma_voff(i)), "s"(sbase), "s"(dst)
                   : "memory");
    }
    if (wave == 0 && lane < SI / 4) {
      const float* src = stream_bias + (int64_t)stg * SI + lane * 4;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + BIAS_OFF + buf * SI * 4);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src), "s"(dst)
                   : "memory");
    }
  };
  const float* sbias = reinterpret_cast<const float*>(smem + BIAS_OFF);

  // GEMM-1 A fragment (row read): stream row lr of the tile, 16-byte chunk 2s + lh
  const int a_k0 = lr * (2 * LP) + ((lh ^ swz(lr)) << 4);
  // GEMM-2 A fragment (transposed read): lane 4q+p of 16-lane group g (h = g >> 1 picks the k half, e = g & 1 the column half of the
  // 32-channel block) supplies the address of stream row 16 s2 + 8 u + 4 h + q, channels 32 cb + 16 e + 4 p .. + 3; u = 0 / 1 are the
  // two halves of the 8-element fragment.  Element j of lane half h then belongs to stream row 16 s2 + 8 (j >> 2) + 4 h + (j & 3):
  // exactly the row of the packed accumulator pair that the B fragment holds in element j.
  const int tg = lane >> 4, th = tg >> 1, te = tg & 1, ti = lane & 15, tq = ti >> 2, tp = ti & 3;
  int tr_a[2];
#pragma unroll
  for (int u = 0; u < 2; u++) {
    const int row = 8 * u + 4 * th + tq;
    const int low = (2 * te + (tp >> 1)) ^ ((2 * u + th) & 3);
    tr_a[u] = row * (2 * LP) + (tq << 6) + (low << 4) + 8 * (tp & 1);   // chunk' = 4 (cb ^ q) + low: the cb part is an XOR with cb << 6
  }

  f32x16 oacc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; cb++)
#pragma unroll
    for (int i = 0; i < 16; i++) oacc[cb][i] = 0.f;
  float cs = 0.f;   // mode 0: column sum (d bo); mode 2: running softmax sum
  int cnt = 0;      // mode 2: #{z > target}

  // One tile = GEMM-1 (C/16 MFMAs into `acc`) -> 16 probabilities per lane -> GEMM-2 (two k-steps of C/32 MFMAs each, from the packed
  // probabilities pk[0..3] / pk[4..7]).  The two waves of a SIMD meet at one barrier per stage, so they run the same part of a tile at the
  // same time; VALU work therefore has to sit beside a wave's OWN MFMAs to be hidden.  Software pipeline, no extra registers:
  //     G1(t)                                            16 MFMAs (dependent chain)
  //     probabilities 0..7 of t   ||  GEMM-2 k-step 1 of tile t-1   (pk[4..7] of t-1 were kept across G1(t); its LDS tile is still resident)
  //     GEMM-2 k-step 0 of t      ||  probabilities 8..15 of t       (-> pk[4..7])
  // The deferred k-step of a stage's last tile reads that stage's buffer during the NEXT iteration, so the stage barrier (and the refill
  // of the oldest buffer behind it) sits behind the first of those overlapped parts instead of at the end of the iteration: still one
  // barrier per stage, still three buffers (s-1 until that barrier, s, s+1; s+2 lands in the buffer of s-1).
  // (Measured alternatives: GEMM-1 of tile t + 1 issued one MFMA at a time between the MFMAs of GEMM-2 of tile t (two tiles in flight, no
  // MFMA-only phase at all): 17 spill accesses per stage at C = 256 and three LDS reads per gap -- 423 vs 337 ms per training step; fragment reads 1 / 2 ahead: no difference; waves 4-7 one part behind waves 0-3 through a rotated per-tile
  // loop with the two halves' barriers at different points: 364 vs 323 ms per training step, the rotated loop costs registers.)
  f32x16 acc;
  uint32_t pk[8];
  auto prob_pair = [&](int i) {   // values i, i + 1 of the tile -> one packed bf16 pair
    const float e0 = __builtin_amdgcn_exp2f(fmaf(acc[i], LOG2E_F, cst));
    const float e1 = __builtin_amdgcn_exp2f(fmaf(acc[i + 1], LOG2E_F, cst));
    if (MODE != 1) cs += e0 + e1;
    if (MODE == 2) cnt += (acc[i] > tlog ? 1 : 0) + (acc[i + 1] > tlog ? 1 : 0);
    pk[i >> 1] = pack2bf(e0, e1);
  };
  // The same pair in two halves for two consecutive MFMA gaps (same operations in the same order: bit-identical results).  A whole pair is
  // 2 fma + 2 exp + 2 add + pack (+ 2 cmp + 2 addc in mode 2) = 36 (52) cycles of vector issue beside an MFMA's 8 -- in one 32-cycle gap it
  // stretches that gap to 44 (60) while the next gap idles at 8; halved, both gaps stay at or near 32 (MI355X_MICROARCH.md, issue costs).
  float e_half = 0.f;
  auto prob_half_a = [&](int i) {
    e_half = __builtin_amdgcn_exp2f(fmaf(acc[i], LOG2E_F, cst));
    if (MODE == 2) cnt += (acc[i] > tlog ? 1 : 0) + (acc[i + 1] > tlog ? 1 : 0);
    asm volatile("" : "+v"(e_half), "+v"(cnt));
  };
  auto prob_half_b = [&](int i) {
    const float e1 = __builtin_amdgcn_exp2f(fmaf(acc[i + 1], LOG2E_F, cst));
    if (MODE != 1) cs += e_half + e1;
    pk[i >> 1] = pack2bf(e_half, e1);
  };
  auto g1 = [&](const unsigned char* tl, const float* tb) {
    int ak = a_k0;   // opaque copy: hipcc otherwise hoists the 16 XOR-ed fragment addresses out of the stage loop and spills them
    asm volatile("" : "+v"(ak));
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(tb + 8 * q + 4 * lh);
#pragma unroll
      for (int j = 0; j < 4; j++) acc[4 * q + j] = b4[j];
    }
    if constexpr (TWO_CHAINS && !ONE) {
      // two waves per SIMD: fragment reads one ahead, as the single chain has them (a deeper ring spills)
      f32x16 acc2;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) acc2[reg] = 0.f;
      uint4 ring[2];
      ring[0] = *reinterpret_cast<const uint4*>(tl + (ak ^ (0 << 5)));
#pragma unroll
      for (int s = 0; s < G::KS; s += 2) {
        ring[1] = *reinterpret_cast<const uint4*>(tl + (ak ^ ((s + 1) << 5)));
        acc = mfma32(ring[0], hf[s], acc);
        if (s + 2 < G::KS) ring[0] = *reinterpret_cast<const uint4*>(tl + (ak ^ ((s + 2) << 5)));
        acc2 = mfma32(ring[1], hf[s + 1], acc2);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int reg = 0; reg < 16; reg++) acc[reg] += acc2[reg];
    } else if constexpr (TWO_CHAINS) {
      // one wave per SIMD: nothing else fills the latency of a dependent MFMA chain, so the 32 k-steps run as TWO independent chains (even / odd
      // k-steps) that are added at the end; fragment reads three ahead
      f32x16 acc2;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) acc2[reg] = 0.f;
      uint4 ring[4];
#pragma unroll
      for (int s = 0; s < 3; s++) ring[s] = *reinterpret_cast<const uint4*>(tl + (ak ^ (s << 5)));
#pragma unroll
      for (int s = 0; s < G::KS; s += 2) {
        if (s + 3 < G::KS) ring[(s + 3) & 3] = *reinterpret_cast<const uint4*>(tl + (ak ^ ((s + 3) << 5)));
        acc = mfma32(ring[s & 3], hf[s], acc);
        if (s + 4 < G::KS) ring[(s + 4) & 3] = *reinterpret_cast<const uint4*>(tl + (ak ^ ((s + 4) << 5)));
        acc2 = mfma32(ring[(s + 1) & 3], hf[s + 1], acc2);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int reg = 0; reg < 16; reg++) acc[reg] += acc2[reg];
    } else {
      uint4 ring[2];
      ring[0] = *reinterpret_cast<const uint4*>(tl + (ak ^ (0 << 5)));
#pragma unroll
      for (int s = 0; s < G::KS; s++) {
        if (s + 1 < G::KS) ring[(s + 1) & 1] = *reinterpret_cast<const uint4*>(tl + (ak ^ ((s + 1) << 5)));
        acc = mfma32(ring[s & 1], hf[s], acc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // one k-step (s2) of GEMM-2 on the tile at `tl`, with `first` probabilities pairs of the CURRENT tile computed in its MFMA gaps
  auto g2 = [&](const unsigned char* tl, int s2, int first, bool run) {
    int tr0 = tr_a[0], tr1 = tr_a[1];
    asm volatile("" : "+v"(tr0), "+v"(tr1));
    auto tr_frag = [&](int cb) -> uint4 {
      const uint2 lo = lds_read_tr16(tl + ((tr0 ^ (cb << 6)) + s2 * 32 * LP));
      const uint2 hi = lds_read_tr16(tl + ((tr1 ^ (cb << 6)) + s2 * 32 * LP));
      return make_uint4(lo.x, lo.y, hi.x, hi.y);
    };
    const uint4 bfrag = make_uint4(pk[4 * s2], pk[4 * s2 + 1], pk[4 * s2 + 2], pk[4 * s2 + 3]);
    uint4 ring[2];
    if (run) ring[0] = tr_frag(0);
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
      if (run) {
        if (cb + 1 < NCB) ring[(cb + 1) & 1] = tr_frag(cb + 1);
        oacc[cb] = mfma32(ring[cb & 1], bfrag, oacc[cb]);
      }
      if constexpr (SPLIT_PAIRS) {   // 4 pairs over the NCB gaps, each pair over two of them
        constexpr int GP = NCB / 4;
        if (cb % GP == 0) prob_half_a(first + 2 * (cb / GP));
        if (cb % GP == GP / 2) prob_half_b(first + 2 * (cb / GP));
      } else {
        if constexpr (NCB < 4) {   // C = 64: two pairs beside each of the two MFMAs
#pragma unroll
          for (int k = 0; k < 4 / NCB; k++) prob_pair(first + 2 * (cb * (4 / NCB) + k));
        } else if ((cb * 4) % NCB == 0) {
          prob_pair(first + 2 * ((cb * 4) / NCB));   // 4 pairs spread over the NCB gaps
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  if (n_stages > 0) stage_issue(0, 0);
  if (n_stages > 1) stage_issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0, pbuf = 0;
  for (int stg = 0; stg < n_stages; stg++) {
    const unsigned char* t0 = smem + buf * STAGE_LDS;
    const float* tb = sbias + buf * SI;
    // ---- tile 0
    g1(t0, tb);
    __builtin_amdgcn_sched_barrier(0);
    g2(smem + pbuf * STAGE_LDS + (SG::TILES - 1) * TILE_LDS, 1, 0, stg > 0);   // probabilities 0..7 || the previous stage's last tile, k-step 1 (pk[4..7] still its own)
    __builtin_amdgcn_sched_barrier(0);
    // stage boundary: everyone is done with the buffer of stage stg - 1, and stage stg + 1 (issued one iteration ago) has landed
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (stg + 2 < n_stages) stage_issue(stg + 2, (buf >= 1) ? buf - 1 : 2);
    g2(t0, 0, 8, true);                                         // k-step 0 of tile 0 || probabilities 8..15
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (SG::TILES == 2) {
      const unsigned char* t1 = t0 + TILE_LDS;
      // ---- tile 1
      g1(t1, tb + 32);
      __builtin_amdgcn_sched_barrier(0);
      g2(t0, 1, 0, true);                                         // probabilities 0..7 of tile 1 || k-step 1 of tile 0
      __builtin_amdgcn_sched_barrier(0);
      g2(t1, 0, 8, true);                                         // k-step 0 of tile 1 || probabilities 8..15
      __builtin_amdgcn_sched_barrier(0);
    }
    pbuf = buf;
    buf = (buf == 2) ? 0 : buf + 1;
  }
  if (n_stages > 0) {   // drain: k-step 1 of the very last tile (its probabilities are in pk[4..7]; nothing left to compute beside it)
    const unsigned char* tl = smem + pbuf * STAGE_LDS + (SG::TILES - 1) * TILE_LDS;
    int tr0 = tr_a[0], tr1 = tr_a[1];
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
      const uint2 lo = lds_read_tr16(tl + ((tr0 ^ (cb << 6)) + 32 * LP));
      const uint2 hi = lds_read_tr16(tl + ((tr1 ^ (cb << 6)) + 32 * LP));
      oacc[cb] = mfma32(make_uint4(lo.x, lo.y, hi.x, hi.y), make_uint4(pk[4], pk[5], pk[6], pk[7]), oacc[cb]);
    }
  }

  // out^T[c][stationary]: register i of tile cb is channel 32 cb + (i & 3) + 8 (i >> 2) + 4 lh of this lane's stationary vector
  if (MODE == 0) {
    if (sv < a.n_stat_max) {
      const bool slab = a.slabs != nullptr;
      float* op = (slab && by > 0 ? a.slabs + (int64_t)(by - 1) * a.n_stat_max * C : a.out) + sv * C;
#pragma unroll
      for (int cb = 0; cb < NCB; cb++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
          float* d = op + 32 * cb + 8 * q + 4 * lh;
          if (ny > 1 && !slab) {
#pragma unroll
            for (int j = 0; j < 4; j++) atomicAdd(d + j, oacc[cb][4 * q + j]);
          } else {
            *reinterpret_cast<float4*>(d) = make_float4(oacc[cb][4 * q], oacc[cb][4 * q + 1], oacc[cb][4 * q + 2], oacc[cb][4 * q + 3]);
          }
        }
      }
    }
    if (a.colsum) {
      const float t = cs + __shfl_xor(cs, 32);
      if (lh == 0 && sv < n_stat) atomicAdd(a.colsum + sv, t);
    }
    return;
  }
  // modes 1, 2: d hidden[row][c] = scale * out^T[c][row] - w Wo[y][c]
  float scale = 1.f, w;
  int yv;
  int64_t drow;
  if (MODE == 2 && a.dyn_max_parts > 0) {
    // one part of the table: the reference point is fixed, so the parts' sums, counts and unscaled d hidden simply add up (k_sgrad_finish)
    const float s_tot = cs + __shfl_xor(cs, 32);
    const int c_tot = cnt + __shfl_xor(cnt, 32);
    if (sv < n_stat) {
      if (lh == 0) {
        a.part[(int64_t)(2 * by) * a.part_ld + sv] = s_tot;
        a.part[(int64_t)(2 * by + 1) * a.part_ld + sv] = __int_as_float(c_tot);
        if (by == 0) a.part[(int64_t)(2 * a.dyn_max_parts) * a.part_ld + sv] = tlog;
      }
      float* op = a.part + (int64_t)(2 * a.dyn_max_parts + 1) * a.part_ld + ((int64_t)by * a.part_ld + sv) * C;
#pragma unroll
      for (int cb = 0; cb < NCB; cb++)
#pragma unroll
        for (int q = 0; q < 4; q++)
          *reinterpret_cast<float4*>(op + 32 * cb + 8 * q + 4 * lh) = make_float4(oacc[cb][4 * q], oacc[cb][4 * q + 1], oacc[cb][4 * q + 2], oacc[cb][4 * q + 3]);
    }
    return;
  }
  if (MODE == 2) {
    const float s_tot = cs + __shfl_xor(cs, 32);   // the two halves saw disjoint items, with the same reference point
    const int c_tot = cnt + __shfl_xor(cnt, 32);
    const bool over = (__float_as_uint(s_tot) & 0x7f800000u) == 0x7f800000u;
    w = (sv < n_stat) ? a.wc[srow] : 0.f;
    scale = myy > 0 ? w / s_tot : 0.f;
    yv = myy;
    drow = srow;
    if (lh == 0 && sv < n_stat) {
      const float lse = (log2f(s_tot) - cst) * LN2_F;   // (m + log2 S) ln 2, m = -cst
      a.o_rank[srow] = myy > 0 ? c_tot : 0;
      a.o_ce[srow] = myy > 0 ? (over ? INFINITY : lse - tlog) : 0.f;   // non-finite: the score.hip fix-up launch redoes this row's tile
      a.o_tgt[srow] = myy > 0 ? tlog : 0.f;
      if (over) a.ovf[sv] = 1;
    }
  } else {
    w = (sv < n_stat) ? a.wc[svc] : 0.f;
    yv = (sv < n_stat) ? a.yc[svc] : 0;
    drow = (int64_t)a.row_map[svc];
  }
  if (sv < n_stat) {
    float* op = a.out + drow * C;
    const bool hot = yv > 0 && yv < a.N;
    const uint16_t* wr = a.outT + (int64_t)(hot ? yv : 0) * C;
    const float wh = hot ? w : 0.f;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int c = 32 * cb + 8 * q + 4 * lh;
        const uint2 t2 = *reinterpret_cast<const uint2*>(wr + c);
        float4 v;
        v.x = oacc[cb][4 * q] * scale - wh * bf2f((uint16_t)(t2.x & 0xffff));
        v.y = oacc[cb][4 * q + 1] * scale - wh * bf2f((uint16_t)(t2.x >> 16));
        v.z = oacc[cb][4 * q + 2] * scale - wh * bf2f((uint16_t)(t2.y & 0xffff));
        v.w = oacc[cb][4 * q + 3] * scale - wh * bf2f((uint16_t)(t2.y >> 16));
        *reinterpret_cast<float4*>(op + c) = v;
      }
    }
  }
}

// compact scored rows: hc[p] = hidden[map[p]], yc, (ln w - lse) log2e, w for p < Rv; zeros / -inf up to the next multiple of 64 (the
// padding rows of the last stage are read as operands).  Grid sized by R (host bound); Rv is read on the device.
__global__ void k_sgrad_gather(const uint16_t* __restrict__ hidden, const int32_t* __restrict__ y, const float* __restrict__ w,
                               const float* __restrict__ ce, const float* __restrict__ tgt, const int32_t* __restrict__ map,
                               const int32_t* __restrict__ n_dev, int C, uint16_t* __restrict__ hc, int32_t* __restrict__ yc, float* __restrict__ wc,
                               float* __restrict__ rb2) {
  const int ch = C >> 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t Rv = n_dev[0], Rpad = (Rv + 63) & ~(int64_t)63;
  const int64_t p = i / ch;
  if (p >= Rpad) return;
  const int c = (int)(i - p * ch);
  if (p >= Rv) {
    *reinterpret_cast<uint4*>(hc + p * C + c * 8) = make_uint4(0, 0, 0, 0);
    if (c == 0) { yc[p] = 0; wc[p] = 0.f; rb2[p] = -INFINITY; }
    return;
  }
  const int64_t r = map[p];
  *reinterpret_cast<uint4*>(hc + p * C + c * 8) = *reinterpret_cast<const uint4*>(hidden + r * C + c * 8);
  if (c == 0) {
    const float wv = w[r];
    yc[p] = y[r];
    wc[p] = wv;
    rb2[p] = wv > 0.f ? logf(wv) - (ce[r] + tgt[r]) : -INFINITY;   // ln w - log-sum-exp (lse = ce + target logit, loss.py:21); natural units: the stream scalar of ITEMSTAT; ROWSTAT scales it by log2e on load
  }
}
__global__ void k_scale_log2e(const float* __restrict__ src, float* __restrict__ dst, int64_t n_max, const int32_t* __restrict__ n_dev, int pad64) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t n = n_dev ? (int64_t)n_dev[0] : n_max;
  if (pad64) n = (n + 63) & ~(int64_t)63;
  if (i < n && i < n_max) dst[i] = src[i] * LOG2E_F;
}

// d Wo[y_p][:] -= w_p h_p[:],  d bo[y_p] -= w_p: one wave per scored row, 256 contiguous bytes per atomic instruction
__global__ __launch_bounds__(256) void k_dwo_onehot(const uint16_t* __restrict__ hc, const int32_t* __restrict__ yc, const float* __restrict__ wc,
                                                    const int32_t* __restrict__ n_dev, int C, int N, float* __restrict__ dwo,
                                                    float* __restrict__ dbias) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= (int64_t)n_dev[0]) return;
  const int yv = yc[p];
  if (yv <= 0 || yv >= N) return;
  const float w = wc[p];
  for (int c = lane; c < C; c += 64) atomicAdd(dwo + (int64_t)yv * C + c, -w * bf2f(hc[p * C + c]));
  if (lane == 0) atomicAdd(dbias + yv, -w);
}

// Small batches, mode 2 split in parts: one wave per compact row adds the parts up, writes rank / ce / target as the unsplit kernel does and turns the
// accumulated sum_n 2^(z_n log2e - m) Wo[n] into d hidden = w softmax . Wo - w Wo[y].
__global__ __launch_bounds__(256) void k_sgrad_finish(const float* __restrict__ part, int64_t part_ld, int max_parts, int grid, int stat, int n_all,
                                                       const int32_t* __restrict__ n_dev, const int32_t* __restrict__ row_map, const int32_t* __restrict__ y,
                                                       const float* __restrict__ w, const uint16_t* __restrict__ outT, int N, int C, float* __restrict__ dh,
                                                       int32_t* __restrict__ o_rank, float* __restrict__ o_ce, float* __restrict__ o_tgt, int32_t* __restrict__ ovf) {
  const int lane = threadIdx.x & 63;
  const int64_t n = n_dev[0];
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n) return;
  int parts = score_dyn_split(n, stat, n_all, grid, max_parts).parts;
  {  // the kernel cuts the stream in ceil(n_all / parts) stages per part: trailing parts without a stage write nothing
    const int per = (n_all + parts - 1) / parts;
    parts = (n_all + per - 1) / per;
  }
  const int64_t srow = row_map[p];
  const int yy = y[srow];
  const int myy = (yy > 0 && yy < N) ? yy : 0;
  float s_tot = 0.f;
  int c_tot = 0;
  for (int k = 0; k < parts; k++) {
    s_tot += part[(int64_t)(2 * k) * part_ld + p];
    c_tot += __float_as_int(part[(int64_t)(2 * k + 1) * part_ld + p]);
  }
  const float tlog = part[(int64_t)(2 * max_parts) * part_ld + p];
  const bool over = (__float_as_uint(s_tot) & 0x7f800000u) == 0x7f800000u;
  const float wv = w[srow];
  const float scale = myy > 0 ? wv / s_tot : 0.f;
  if (lane == 0) {
    const float cst = -(tlog * LOG2E_F + 64.0f);
    const float lse = (log2f(s_tot) - cst) * LN2_F;
    o_rank[srow] = myy > 0 ? c_tot : 0;
    o_ce[srow] = myy > 0 ? (over ? INFINITY : lse - tlog) : 0.f;
    o_tgt[srow] = myy > 0 ? tlog : 0.f;
    if (over) ovf[p] = 1;
  }
  const float wh = myy > 0 ? wv : 0.f;
  const uint16_t* wr = outT + (int64_t)myy * C;
  float* op = dh + srow * C;
  const float* slab = part + (int64_t)(2 * max_parts + 1) * part_ld + p * C;
  for (int c = 4 * lane; c < C; c += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < parts; k++) {
      const float4 t = *reinterpret_cast<const float4*>(slab + (int64_t)k * part_ld * C + c);
      v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
    }
    const uint2 t2 = *reinterpret_cast<const uint2*>(wr + c);
    v.x = v.x * scale - wh * bf2f((uint16_t)(t2.x & 0xffff));
    v.y = v.y * scale - wh * bf2f((uint16_t)(t2.x >> 16));
    v.z = v.z * scale - wh * bf2f((uint16_t)(t2.y & 0xffff));
    v.w = v.w * scale - wh * bf2f((uint16_t)(t2.y >> 16));
    *reinterpret_cast<float4*>(op + c) = v;
  }
}

// out[i] += sum of n_slabs slabs (float4 per thread)
__global__ void k_add_slabs(float* __restrict__ out, const float* __restrict__ slabs, int64_t n4, int n_slabs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  float4 v = reinterpret_cast<float4*>(out)[i];
  for (int k = 0; k < n_slabs; k++) {
    const float4 t = reinterpret_cast<const float4*>(slabs)[(int64_t)k * n4 + i];
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  reinterpret_cast<float4*>(out)[i] = v;
}
// small catalogues (fewer than 128 item blocks: the chip is not filled by the item dimension): row parts of mode 0 through slabs
static int sgrad_dwo_slab_parts(const ModelDev& m, int64_t R) {
  static const bool off = ABL_ENV("HIERTCN_SGRAD0_SLABS_OFF");
  const int stat = sg_stat(m.Clast, sg_one(m.Clast, 0));
  const int blocks = (m.NP + stat - 1) / stat;
  if (off || blocks >= 128) return 1;
  const int stages = (int)((R + 63) / 64);
  int p = 256 / blocks;
  if (p > 4) p = 4;
  while (p > 1 && stages / p < 16) p--;
  return p;
}
size_t sgrad_dwo_slab_floats(const ModelDev& m, int64_t R) {
  const int p = sgrad_dwo_slab_parts(m, R);
  return p > 1 ? (size_t)(p - 1) * (size_t)m.NP * (size_t)m.Clast : 0;
}

#define SG_DYN_MAX_PARTS 16   // (a slab of d hidden per part: 16 x 6,464 rows x 1 KB = 106 MB at the reference's batch and d = 256)
// ... and fewer for larger batches, so that the slabs stay under 1 GiB (the device deals at most by_len = stages / 16 parts, and past 256 live row blocks at most
// grid / blocks)
static int sg_dyn_max_parts(const ModelDev& m, int64_t R) {
  const int64_t p = ((int64_t)1 << 28) / ((R + 64) * (int64_t)m.Clast);
  return p > SG_DYN_MAX_PARTS ? SG_DYN_MAX_PARTS : p < 2 ? 2 : (int)p;
}
// Workgroups of the device-split launch: one round of the chip for fewer than 128 PADDED row blocks (the live ones are fewer still: every (block, part) gets a CU
// of its own); past that -- up to 1024 padded blocks -- eight rounds' worth, of which the device uses rounds x 256 >= blocks x parts (score_dyn_split) and the
// rest exit at once.  Without the split a batch of 138 live blocks is 138 workgroups walking the whole table on 256 CUs, and one of 273 two rounds of which
// the second holds 17 workgroups.
static int sg_dyn_grid(const ModelDev& m, int64_t R) {
  const int max_blocks = m.tune.sgrad_dyn_max_blocks;   // (default 1024)
  const int stat2 = sg_stat(m.Clast, sg_one(m.Clast, 2));
  const int64_t blocks = (R + stat2 - 1) / stat2;
  if (blocks < 128) return SC_DYN_GRID;
  return blocks <= max_blocks ? 8 * SC_DYN_GRID : 0;
}
// floats of the `part` scratch of launch_sgrad_forward (0: no device-side split at this size)
size_t sgrad_forward_part_floats(const ModelDev& m, int64_t R) {
  static const bool off = ABL_ENV("HIERTCN_SGRAD_DYN_OFF");
  if (off || sg_dyn_grid(m, R) == 0 || m.NP / sg_si(m.Clast, sg_one(m.Clast, 2)) < 32) return 0;
  const size_t mp = (size_t)sg_dyn_max_parts(m, R);
  return (2 * mp + 1) * (size_t)(R + 64) + mp * (size_t)(R + 64) * (size_t)m.Clast;
}

bool sgrad_supported(const ModelDev& m) {
  return (m.Clast == 64 && !ABL_ENV("HIERTCN_SGRAD64_OFF")) || m.Clast == 128 || m.Clast == 256 || (m.Clast == 512 && !ABL_ENV("HIERTCN_SGRAD512_OFF"));
}

template <int C, int MODE, bool ONE>
static hipError_t launch_k_sgrad(const SgradArgs& a, dim3 grid, hipStream_t st) {
  using SG = SgGeom<C, ONE>;
  const size_t lds = 3 * (size_t)SG::STAGE_BYTES + 3 * SG::SI * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sgrad<C, MODE, ONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL((k_sgrad<C, MODE, ONE>), grid, dim3(SG::THREADS), lds, st, a);
  return hipGetLastError();
}
template <int MODE>
static hipError_t launch_k_sgrad_c(const ModelDev& m, const SgradArgs& a, dim3 grid, hipStream_t st) {
  switch (m.Clast) {
    case 64: return launch_k_sgrad<64, MODE, false>(a, grid, st);
    case 128: return launch_k_sgrad<128, MODE, false>(a, grid, st);
    case 256: return launch_k_sgrad<256, MODE, false>(a, grid, st);
    case 512: return launch_k_sgrad<512, MODE, true>(a, grid, st);
    default: return hipErrorInvalidValue;
  }
}

// Training forward of the scoring layer + d hidden in one pass (mode 2): rank / ce / target as launch_score would write them for the
// rows listed in row_map (the caller zero-fills the others), d hidden [R][C] (zero-filled by the caller) at those rows, ovf [R] ints
// (zero-filled by the caller) flagged where the running sum overflowed.  w [R]: d loss / d ce per row (model.py:111-117).
hipError_t launch_sgrad_forward(const ModelDev& m, const uint16_t* hidden, const int32_t* y, const float* w, const int32_t* row_map,
                                const int32_t* n_dev, int64_t R, int32_t* o_rank, float* o_ce, float* o_tgt, float* dh, int32_t* ovf,
                                hipStream_t st, float* part) {
  if (R <= 0) return hipSuccess;
  SgradArgs a;
  memset(&a, 0, sizeof(a));
  a.stat = hidden; a.n_stat_max = R; a.n_stat_dev = n_dev;
  a.stream = m.outT; a.stream_bias = m.out_bias; a.n_stages_max = m.NP / sg_si(m.Clast, sg_one(m.Clast, 2));
  a.out = dh; a.row_map = row_map; a.yc = y; a.wc = w; a.outT = m.outT; a.N = m.N; a.ovf = ovf;
  a.out_bias = m.out_bias; a.o_rank = o_rank; a.o_ce = o_ce; a.o_tgt = o_tgt;
  a.walk = m.walk_pos;
  const int stat2 = sg_stat(m.Clast, sg_one(m.Clast, 2));
  if (part && n_dev && row_map && sgrad_forward_part_floats(m, R) > 0) {
    // small batch (the reference's 32 lanes: 25 row blocks, half of them empty when ragged, each walking the whole table): SC_DYN_GRID workgroups dealt
    // as (row block, table part) on the device from the live row count
    a.part = part; a.part_ld = R + 64; a.dyn_max_parts = sg_dyn_max_parts(m, R);
    a.walk = nullptr;
    const int grid = sg_dyn_grid(m, R);
    hipError_t e = launch_k_sgrad_c<2>(m, a, dim3((unsigned)grid), st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_sgrad_finish, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, part, a.part_ld, a.dyn_max_parts, grid, stat2, a.n_stages_max, n_dev,
                       row_map, y, w, m.outT, m.N, m.Clast, dh, o_rank, o_ce, o_tgt, ovf);
    return hipGetLastError();
  }
  return launch_k_sgrad_c<2>(m, a, dim3((unsigned)((R + stat2 - 1) / stat2)), st);
}

// d hidden from the saved log-sum-exp (mode 1) for the compact scored rows; vote != 0: only the 256-row blocks with a row flagged in ovf
hipError_t launch_sgrad_dh(const ModelDev& m, const uint16_t* hc, const float* rb_l2, const int32_t* yc, const float* wc, const int32_t* row_map,
                           const int32_t* n_dev, int64_t R, float* dh, int32_t* ovf, int vote, hipStream_t st) {
  if (R <= 0) return hipSuccess;
  SgradArgs a;
  memset(&a, 0, sizeof(a));
  a.stat = hc; a.stat_const = rb_l2; a.n_stat_max = R + 64; a.n_stat_dev = n_dev;
  a.stream = m.outT; a.stream_bias = m.out_bias; a.n_stages_max = m.NP / sg_si(m.Clast, sg_one(m.Clast, 1));
  a.out = dh; a.row_map = row_map; a.yc = yc; a.wc = wc; a.outT = m.outT; a.N = m.N; a.ovf = ovf; a.vote = vote;
  a.walk = m.walk_pos;
  const int stat1 = sg_stat(m.Clast, sg_one(m.Clast, 1));
  return launch_k_sgrad_c<1>(m, a, dim3((unsigned)((R + stat1 - 1) / stat1)), st);
}

// d Wo / d bo (mode 0), ACCUMULATED into dwo [NP][C] / dbias [NP] (the gradient arena is zeroed at the start of the step), followed by
// the one-hot term.  bias_l2 [NP]: bo log2e (-inf on the padding items); rb_nat [R + 64]: ln w - lse per compact row (-inf padding).
hipError_t launch_sgrad_dwo(const ModelDev& m, const uint16_t* hc, const float* rb_nat, const int32_t* yc, const float* wc, const int32_t* n_dev,
                            int64_t R, const float* bias_l2, float* dwo, float* dbias, hipStream_t st, float* slabs) {
  if (R <= 0) return hipSuccess;
  SgradArgs a;
  memset(&a, 0, sizeof(a));
  a.stat = m.outT; a.stat_const = bias_l2; a.n_stat_max = m.NP;
  const int si = sg_si(m.Clast, sg_one(m.Clast, 0)), stat = sg_stat(m.Clast, sg_one(m.Clast, 0));
  a.stream = hc; a.stream_bias = rb_nat; a.n_stages_max = (int)((R + 63) / 64 * 64 / si); a.n_stream_dev = n_dev;   // (the compact rows are padded to 64)
  a.out = dwo; a.colsum = dbias;
  a.walk = m.walk_pos;
  // row parts so that the grid covers the chip a few times over (the item dimension alone gives N / 256 workgroups)
  const unsigned blocks = (unsigned)((m.NP + stat - 1) / stat);
  unsigned parts = 1;
  double best = 0.0;
  const int min_st = m.tune.sgrad0_min_stages;   // (default 64)
  for (unsigned p = 1; p <= 8 && (p == 1 || a.n_stages_max / (int)p >= min_st); p++) {
    const double eff = ((double)blocks * p / 256.0) / (double)((blocks * p + 255) / 256);
    if (eff > best + 1e-9) { best = eff; parts = p; }
    if (eff >= 0.93) break;
  }
  const int sp = slabs ? sgrad_dwo_slab_parts(m, R) : 1;
  if (sp > 1) {   // (every part stores its whole tile, also a part that finds no live rows behind the device-side count: its accumulators are zero)
    a.slabs = slabs;
    parts = (unsigned)sp;
  }
  hipError_t e = launch_k_sgrad_c<0>(m, a, dim3(blocks, parts), st);
  if (e != hipSuccess) return e;
  if (sp > 1) {
    const int64_t n4 = (int64_t)m.NP * m.Clast / 4;
    hipLaunchKernelGGL(k_add_slabs, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, dwo, slabs, n4, sp - 1);
  }
  hipLaunchKernelGGL(k_dwo_onehot, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, hc, yc, wc, n_dev, m.Clast, m.N, dwo, dbias);
  return hipGetLastError();
}

hipError_t launch_sgrad_prepare(const ModelDev& m, const uint16_t* hidden, const int32_t* y, const float* w, const float* ce, const float* tgt, const int32_t* map,
                                const int32_t* n_dev, int64_t R, uint16_t* hc, int32_t* yc, float* wc, float* rb_nat, float* rb_l2, float* bias_l2,
                                hipStream_t st) {
  if (R <= 0) return hipSuccess;
  const int64_t n = (R + 64) * (m.Clast / 8);
  hipLaunchKernelGGL(k_sgrad_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hidden, y, w, ce, tgt, map, n_dev, m.Clast, hc, yc, wc, rb_nat);
  hipLaunchKernelGGL(k_scale_log2e, dim3((unsigned)((R + 64 + 255) / 256)), dim3(256), 0, st, rb_nat, rb_l2, R + 64, n_dev, 1);
  hipLaunchKernelGGL(k_scale_log2e, dim3((unsigned)((m.NP + 255) / 256)), dim3(256), 0, st, m.out_bias, bias_l2, (int64_t)m.NP, (const int32_t*)nullptr, 0);
  return hipGetLastError();
}
