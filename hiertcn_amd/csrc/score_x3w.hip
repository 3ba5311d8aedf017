// The split-operand scoring kernel (score_x3.hip: HIERTCN_PRECISION_SPLIT16) at a last width of 512 -- BASELINE.json configs[3], the 1 M-item
// catalogue at d = 512 -- model_tcn.py:41 (logits = h . Wo + bo), model.py:105 (row mask), loss.py:21 (cross entropy), loss.py:179 (rank),
// loss.py:120 (ranked top-k), fused, logits never written, every factor of the product carried as hi + lo IEEE halves (22 bits):
//     h . w  ~=  h_hi w_hi + h_hi w_lo + h_lo w_hi          (three f16 MFMAs into one fp32 accumulator per 16 channels)
// exactly the arithmetic of score_x3.hip (same scales, same order of the three products per k-step), in the shape d = 512 needs:
//
//  * a wave's 32 hidden rows, split, are 32 k-steps x 2 x 4 = 256 registers of resident B operand, so a wave owns a SIMD alone (256-thread
//    workgroups, the 512-register file): 4 waves x 32 rows = 128 rows per workgroup;
//  * an item row of the split table is 2 KB ([hi 1 KB | lo 1 KB]); an LDS stage is ONE 32-item tile (64 KB), two of them resident; the row pitch
//    in LDS is 2 KB + 16 B = 129 sixteen-byte chunks (odd), which makes the ds_read_b128 of the 16 lanes of a bank group hit 16 different bank
//    quads; a tile arrives as 64 LDS-DMA pieces of 1 KB (16 per wave), issued ONE per k-step over the first half of the previous tile's phase --
//    alone on its SIMD a wave's DMA issue is not covered by a partner's MFMAs, so it is spread into the MFMA gaps instead of issued as a burst;
//  * a phase = the 96 MFMAs of tile T (fragments read two k-steps ahead through a register ring that runs on across the tile boundary) with the
//    epilogue of tile T - 1's 16 values in their gaps.  The accumulator starts at zero and the bias is added (one fma per value: b 2^(kw+kh) + acc)
//    when the tile is drained, so no phase depends on the NEXT tile's bias; the target logit is formed by the same sequence (MFMAs from zero, then
//    the same fma) and `z_n > z_target` stays exactly false for n == y (loss.py:179);
//  * ONE workgroup barrier per tile, behind the MFMAs of k-step 28: behind it tile T + 1 (issued at k-steps 0..15, every wave having waited for
//    its own pieces) is visible to all waves -- the ring reads three k-steps ahead, so the reads of k-steps 29..31 are the first fragments of that
//    tile -- and every wave has read its last fragment of tile T, whose buffer the next phase refills.
//  * PACED WALK.  A workgroup streams the whole 2 GB table (N = 10^6) and hides the stream's latency a full tile ahead, so nothing couples the
//    32 workgroups of an XCD: left alone they drift apart until no two of them read the same 4 MB of the table at the same time (measured: L2 hit
//    rate 11 %, 1.09 TB per launch through the fabric, 20 % of the wave time in waits).  So the workgroups of an XCD class (blockIdx % 8) publish
//    their walk position in a slot of `pace` every 4 tiles and, every 8 tiles, the one that is more than XW_WINDOW tiles ahead of a NEIGHBOUR (a
//    workgroup within XW_FAR tiles: the same cohort) sleeps until the neighbour has caught up -- bounded: a stale slot costs microseconds, never
//    a hang; placement is assumed for speed only (MI355X_MICROARCH.md, Workgroup dispatch).  The leaders wait for the slowest, the window stays
//    inside the XCD's L2, and one workgroup's miss is 31 workgroups' hit.
// Flavours, outputs, catalogue parts for small batches, ragged row lists: as k_score_x3 (one launcher decides between them: score_x3.hip).
#include "common.h"
#include "exact.h"
#include "score_geom.h"
#include "score_lists.h"
#include "score_x3.h"

#define XW_THREADS 256
#define XW_ROWS 128   // rows per workgroup
#define XW_WINDOW 16  // tiles (1 MB of table) a workgroup may lead a neighbour of its XCD by
#define XW_FAR 1024   // tiles beyond which another workgroup's position is not a neighbour's (a later round of the grid at another place of the table)
#define XW_PACE_TRIES 24   // bounded wait: sleeps of ~2 us
#ifndef XW_DR0
#define XW_DR0 16     // first k-step of the drain (16 values, one per k-step): the half of the phase that issues no LDS-DMA piece
#endif                // (6 .. 16; -DXW_DR0=6 is the first form of the kernel: drain and DMA in the same k-steps)
// Tried and dropped (same box, 153,600 rows x 1,000,000 items, scripts/build_alt.py): the tile through registers instead of LDS-DMA (16-byte global
// loads, ds_write_b128 eight k-steps later, eight pieces in flight per wave): 696 ms against 361; waves 2, 3 issuing their DMA pieces eight k-steps
// behind waves 0, 1 (XW_DMA_LAG): 398 against 388.  Kept: the drain in the DMA-free k-steps (386 -> 373), the ring three k-steps ahead behind MFMA 3
// (369 -> 360), the paced walk (411 -> 386).
// timing experiments (wrong results by design; scripts/build_alt.py builds them into a library of their own, profiles/r5_split16_d512_ablation.txt):
// the kernel without its tile barrier / without its LDS-DMA stream / without its epilogue
#ifndef XW_ABLATE_BARRIER
#define XW_ABLATE_BARRIER 0
#endif
#ifndef XW_ABLATE_DMA
#define XW_ABLATE_DMA 0
#endif
#ifndef XW_ABLATE_EPI
#define XW_ABLATE_EPI 0
#endif
#ifndef XW_DMA_LAG
#define XW_DMA_LAG 0    // k-steps by which waves 2, 3 issue their LDS-DMA pieces behind waves 0, 1 (8 measured 2.4 % SLOWER on one box: 397.7 vs 388.4 ms)
#endif

struct XwGeom {
  static constexpr int CL = 512, KSL = 32, CP = 1024;   // CP: halves per table row [hi | lo]
  static constexpr int ROW_B = 2 * CP;                  // bytes per table row
  static constexpr int PITCH = ROW_B + 16;              // LDS row pitch
  static constexpr int TILE_LDS = 32 * PITCH;           // one 32-item tile
  static constexpr int BIAS_OFF = 2 * TILE_LDS;
  static constexpr int NBIAS = 4;                       // bias slots of 32 floats (tile T's bias is read while tile T + 2's is on its way)
  static constexpr int LDS_BYTES = BIAS_OFF + NBIAS * 32 * 4;
  static constexpr int PPW = 16;                        // 1 KB DMA pieces per wave and tile
};

// FLAVOUR 0 production (software pipeline; KT > 0: ranked lists; a softmax sum past fp32 leaves ce = +inf for the fix-up launch);
//         1 fix-up (running-maximum reference; a row tile is re-done only if flavour 0 left a non-finite ce in it);
//         2 logits (parity / debug: hiertcn_score_logits)
template <int KT, int FLAVOUR>
__global__ __launch_bounds__(XW_THREADS, 1) void k_score_x3w(const float* __restrict__ hidden, const int32_t* __restrict__ y, int64_t R,
                                                             const uint16_t* __restrict__ outS, const float* __restrict__ out_bias, int N, int NP,
                                                             int32_t* __restrict__ o_rank, float* __restrict__ o_ce, float* __restrict__ o_tgt,
                                                             const int32_t* __restrict__ row_map, const int32_t* __restrict__ n_rows_dev,
                                                             int part_stages, float* __restrict__ partials, int* __restrict__ walk_pos,
                                                             int32_t* __restrict__ o_topi, float* __restrict__ o_topv, int topk,
                                                             float* __restrict__ logits, const float* __restrict__ tab_scale, int* __restrict__ pace) {
  using X = XwGeom;
  constexpr int KSL = X::KSL, CP = X::CP, CL = X::CL;
  constexpr bool MAIN = FLAVOUR == 0, FIXUP = FLAVOUR == 1, LOGITS = FLAVOUR == 2;
  static_assert(KT == 0 || MAIN, "lists are built by the production flavour");
  const int64_t Rv = n_rows_dev ? (int64_t)n_rows_dev[0] : R;
  int bx = (int)blockIdx.x, by = (int)blockIdx.y, part_st = part_stages;
  if (part_stages < 0) {   // the catalogue split decided here, from the live row count (score_geom.h)
    const ScDynSplit d = score_dyn_split(Rv, XW_ROWS, NP / SC_STAGE, (int)gridDim.x, -part_stages);
    by = bx / d.tiles;
    bx -= by * d.tiles;
    if (by >= d.parts) return;
    part_st = d.part_stages;
  }
  if ((int64_t)bx * XW_ROWS >= Rv) return;
  if constexpr (FIXUP) {
    const int64_t r = (int64_t)bx * XW_ROWS + threadIdx.x;
    int bad = 0;
    if ((int)threadIdx.x < XW_ROWS && r < Rv) {
      const int64_t src = row_map ? (int64_t)row_map[r] : r;
      bad = (__float_as_uint(o_ce[src]) & 0x7f800000u) == 0x7f800000u ? 1 : 0;
    }
    if (!__syncthreads_or(bad)) return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t row0 = ((int64_t)bx * 4 + wave) * 32;

  // ---- this wave's 32 steps, split, as the two B operands, resident for the whole kernel (256 registers) --------------------------------
  uint4 hhi[KSL], hlo[KSL];
  int myy;
  int64_t mysrc;
  float rs, irs;
  {
    const int64_t r = row0 + lr;
    const int64_t rp = (r < Rv) ? r : (Rv - 1);
    const int64_t rc = row_map ? (int64_t)row_map[rp] : rp;
    mysrc = rc;
    const int yy = (LOGITS || r >= Rv) ? 0 : y[rc];
    myy = (yy > 0 && yy < N) ? yy : 0;
    const float* hp = hidden + rc * CL + 8 * lh;
    float mx = 0.f;
#pragma unroll
    for (int s = 0; s < KSL; s++) {
      const float4 a = *reinterpret_cast<const float4*>(hp + 16 * s), b = *reinterpret_cast<const float4*>(hp + 16 * s + 4);
      mx = fmaxf(mx, fmaxf(fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))), fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w)))));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const int kh = x3_scale_exp(mx);
    const float sh = __builtin_ldexpf(1.0f, kh);
    rs = tab_scale[0] * sh;                           // 2^(kw + kh): what this lane's accumulators are scaled by
    irs = tab_scale[1] * __builtin_ldexpf(1.0f, -kh);
#pragma unroll
    for (int s = 0; s < KSL; s++) x3_split8(hp + 16 * s, sh, hhi[s], hlo[s]);
  }
  const float l2e_s = LOG2E_F * irs;

  // the product of one k-step, in the one order every flavour (and the target logit) uses: score_x3.hip's
#define XW_MFMA3(acc_, ahi_, alo_, s_)            \
  do {                                            \
    acc_ = x3_mfma(ahi_, hhi[s_], acc_);           \
    acc_ = x3_mfma(ahi_, hlo[s_], acc_);           \
    acc_ = x3_mfma(alo_, hhi[s_], acc_);           \
  } while (0)

  X3Row st;
  st.t = 0.f; st.m2 = 0.f; st.s = 0.f; st.c = 0; st.xp = 0.f; st.ep = 0.f;
  if constexpr (!LOGITS) {   // target logits: D[i][j] = Wo^T[y_i] . h_j from zero through the same MFMA sequence, then the same bias fma; keep the diagonal
    f32x16 acc;
#pragma unroll
    for (int reg = 0; reg < 16; reg++) acc[reg] = 0.f;
    const uint16_t* ap = outS + (int64_t)myy * CP + 8 * lh;
#pragma unroll
    for (int s = 0; s < KSL; s++) {
      const uint4 ah = *reinterpret_cast<const uint4*>(ap + 16 * s);
      const uint4 al = *reinterpret_cast<const uint4*>(ap + CL + 16 * s);
      XW_MFMA3(acc, ah, al, s);
    }
#pragma unroll
    for (int reg = 0; reg < 16; reg++) {
      const int yi = __shfl(myy, acc_row(reg, lh));
      acc[reg] = fmaf(out_bias[yi], rs, acc[reg]);   // (column = this lane's own step)
    }
    const int dreg = (lr & 3) + 4 * (lr >> 3);
    float d = acc[0];
#pragma unroll
    for (int reg = 1; reg < 16; reg++) d = (dreg == reg) ? acc[reg] : d;
    const float other = __shfl_xor(d, 32);
    const float t = (((lr >> 2) & 1) == lh) ? d : other;
    st.t = t;
    st.m2 = MAIN ? t * l2e_s + 64.0f : t * l2e_s;   // (score_x3.hip: fixed reference + fix-up launch / running maximum)
  }

  const int n_all = NP / SC_STAGE;
  const int stg_lo = partials ? by * part_st : 0;
  const int stg_hi0 = partials ? stg_lo + part_st : n_all;
  const int n_stages = ((stg_hi0 < n_all) ? stg_hi0 : n_all) - stg_lo;   // 64-item stages of THIS workgroup (<= 0 for a tail part)
  const int n_tiles = n_stages > 0 ? 2 * n_stages : 0;

  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  int s0 = 0;   // per-XCD walk start (score.hip: L2 re-use between the workgroups of one XCD; never with lists, whose ties need index order)
  if (walk_pos && !partials) {
    int* s0_sh = reinterpret_cast<int*>(smem);
    if (tid == 0) *s0_sh = __hip_atomic_load(walk_pos + (blockIdx.x & 7) * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    s0 = __builtin_amdgcn_readfirstlane(*s0_sh);
    __syncthreads();
    s0 = (s0 >= 0 && s0 < n_stages) ? s0 : 0;
  }
  // tile T of this workgroup's walk -> its first item
  auto tile_item0 = [&](int T) {
    int stg = s0 + (T >> 1);
    stg -= (stg >= n_stages) ? n_stages : 0;
    return (stg_lo + stg) * SC_STAGE + (T & 1) * 32;
  };
  // ---- paced walk (file header): this workgroup's slot, and the test "am I too far ahead of a neighbour?" on one wave-wide read of the slots
  int* pace_slot = nullptr;
  const int* pace_row = nullptr;
  if (MAIN && pace && !partials && n_tiles > 4 * XW_WINDOW) {
    int* row = pace + (blockIdx.x & 7) * XW_PACE_INTS;
    int* t_sh = reinterpret_cast<int*>(smem);
    if (tid == 0) *t_sh = atomicAdd(row + XW_PACE_SLOTS, 1);
    __syncthreads();
    const int ticket = __builtin_amdgcn_readfirstlane(*t_sh);
    __syncthreads();
    pace_slot = row + (ticket & (XW_PACE_SLOTS - 1));
    pace_row = row;
  }
  auto tile_abs = [&](int T) {   // position in the table, in tiles
    int stg = s0 + (T >> 1);
    stg -= (stg >= n_stages) ? n_stages : 0;
    return 2 * stg + (T & 1);
  };
  // lanes 0..63 of wave 0 look at one slot each: true if some neighbour is more than XW_WINDOW tiles behind `mine`
  auto pace_ahead = [&](int mine, int other) {
    int lead = mine - other;                    // cyclic, signed
    lead += (lead < -(n_tiles >> 1)) ? n_tiles : 0;
    lead -= (lead > (n_tiles >> 1)) ? n_tiles : 0;
    return other >= 0 && lead > XW_WINDOW && lead <= XW_FAR;
  };
  // ---- item stream: LDS-DMA through inline asm (score.hip, cdna_hip_programming.md 5.7).  Piece i of a wave = table row wave * 8 + i / 2, half i % 2
  const unsigned dma_voff = (unsigned)lane * 16u;
  auto dma_piece = [&](int i, const unsigned char* tile_g, int buf) {
    const int r = wave * 8 + (i >> 1), p = i & 1;
    const unsigned char* src = tile_g + r * X::ROW_B + p * 1024;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + buf * X::TILE_LDS + r * X::PITCH + p * 1024);
    unsigned keep;
    asm volatile("s_nop 4\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(dma_voff), "s"(src), "s"(dst)
                 : "memory");
  };
  auto dma_bias = [&](int item0, int slot) {
    if (wave == 0 && lane < 8) {
      const float* gsrc = out_bias + item0 + lane * 4;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + X::BIAS_OFF + slot * 128);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(gsrc), "s"(dst)
                   : "memory");
    }
  };
  auto tile_gptr = [&](int item0) { return reinterpret_cast<const unsigned char*>(outS) + (int64_t)item0 * X::ROW_B; };
  auto dma_tile_all = [&](int T, int buf) {   // a whole tile at once (prologue; the un-pipelined flavours)
    const int item0 = tile_item0(T);
    const unsigned char* g = tile_gptr(item0);
#pragma unroll
    for (int i = 0; i < X::PPW; i++) dma_piece(i, g, buf);
    dma_bias(item0, T & 3);
  };
  const float* sbias = reinterpret_cast<const float*>(smem + X::BIAS_OFF);
  const int a_k0 = lr * X::PITCH + (lh << 4);   // this lane's w_hi fragment of k-step 0 of item row lr; k-step s: + 32 s; w_lo: + 1024
#define XW_LDS(p_) (*reinterpret_cast<const uint4*>(p_))
#define XW_FH(s_) (a_k0 + ((s_) << 5))
#define XW_FL(s_) (a_k0 + 1024 + ((s_) << 5))

  constexpr int KTA = KT > 0 ? KT : 1;
  TopList<KTA> top;
  if (KT > 0) top_init<KTA>(top);

  if constexpr (MAIN) {
    // one value of the epilogue, software-pipelined over three consecutive slots (score_x3.hip's val)
    auto val = [&](int i, float v) {
      const bool gt = v > st.t;
      if (i >= 2) st.s += st.ep;
      if (i >= 1) st.ep = __builtin_amdgcn_exp2f(st.xp);
      st.xp = fmaf(v, l2e_s, -st.m2);
      st.c += gt ? 1 : 0;
      asm volatile("" : "+v"(st.s), "+v"(st.c), "+v"(st.ep), "+v"(st.xp));
    };
    auto post = [&]() {
      st.s += st.ep;
      st.s += __builtin_amdgcn_exp2f(st.xp);
    };
    uint4 rH[4], rL[4];
    float tk_m = -INFINITY;
    int pace_seen_ = -1;
    // Phase T: MF_: the 96 MFMAs of tile T into x (from zero), fragments two k-steps ahead, tile T + 1's DMA one piece per k-step over
    // k-steps 0..15, the tile barrier behind k-step 28.  DR_: tile T - 1's values in y, in the k-steps without a DMA piece: bias read at k-step 10,
    // bias fma over k-steps 11..14 (and the tile maximum for the lists), the list vote behind k-step 14, one value per k-step over 16..31.
#define XW_STEP(XA, YA, s)                                                                                    \
    {                                                                                                        \
      /* Issue budget of a wave alone on its SIMD: an MFMA holds the issue port for 8 of its 32 cycles, so each of the three gaps of a k-step */ \
      /* hides ~24 cycles of other instructions.  The step's other work is dealt over the three gaps: behind MFMA 1 the                        */ \
      /* compiler's AGPR -> VGPR copies of MFMA 2's operand, behind MFMA 2 the LDS-DMA piece (k-steps 0..15) or the epilogue share (16..31), behind MFMA 3 the fragment reads. */ \
      if (MF_) {                                                                                             \
        if (s == 0 && pace_slot && wave == 0) {   /* paced walk: publish every 4 tiles; every 8 read the neighbours' slots (used at k-step 29) */ \
          if ((T_ & 3) == 0 && lane == 0) __hip_atomic_store(pace_slot, tile_abs(T_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
          if ((T_ & 7) == 4) pace_seen_ = __hip_atomic_load(pace_row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
        }                                                                                                    \
        if (s == KSL - 3 && pace_slot && wave == 0 && (T_ & 7) == 4) {                                       \
          const int mine_ = tile_abs(T_);                                                                    \
          int tries_ = 0;                                                                                    \
          while (__any(pace_ahead(mine_, pace_seen_)) && tries_ < XW_PACE_TRIES) {   /* wave-uniform */        \
            __builtin_amdgcn_s_sleep(64);                                                                    \
            pace_seen_ = __hip_atomic_load(pace_row + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     \
            tries_++;                                                                                        \
          }                                                                                                  \
        }                                                                                                    \
        if (s == 0) {                                                                                        \
          _Pragma("unroll") for (int j = 0; j < 16; j++) XA[j] = 0.f;                                        \
        }                                                                                                    \
        XA = x3_mfma(rH[s & 3], hhi[s], XA);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
      }                                                                                                      \
      if (MF_) {                                                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        XA = x3_mfma(rH[s & 3], hlo[s], XA);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
      }                                                                                                      \
      if (DR_ && !XW_ABLATE_EPI) {   /* the drained tile, in the k-steps that issue no LDS-DMA piece (the second half of the phase) */ \
        if (s == XW_DR0 - 6) {   /* its 16 bias values: read here, used from the next k-step on (no wait on the LDS in front of an fma) */ \
          _Pragma("unroll") for (int q = 0; q < 4; q++) bq_[q] = *reinterpret_cast<const f32x4*>(ybias_ + 8 * q); \
        }                                                                                                    \
        if (s >= XW_DR0 - 5 && s < XW_DR0 - 1) {                                                             \
          const int q_ = (s - (XW_DR0 - 5)) & 3;                                                             \
          _Pragma("unroll") for (int j = 0; j < 4; j++) YA[4 * q_ + j] = fmaf(bq_[q_][j], rs, YA[4 * q_ + j]); \
          if (KT > 0) tk_m = fmaxf(fmaxf(q_ == 0 ? YA[0] : tk_m, YA[4 * q_ + (q_ == 0 ? 1 : 0)]), fmaxf(fmaxf(YA[4 * q_ + 1], YA[4 * q_ + 2]), YA[4 * q_ + 3])); \
        }                                                                                                    \
        if (s >= XW_DR0 && s < XW_DR0 + 16) val(s - XW_DR0, YA[s - XW_DR0]);                                 \
        if (s == KSL - 1) post();                                                                            \
      }                                                                                                      \
      if (MF_) {                                                                                             \
        /* (XW_DMA_LAG > 0: waves 2, 3 issue their pieces that many k-steps behind waves 0, 1 -- an experiment against four simultaneous LDS-DMA  */ \
        /* instructions on the CU's one address path; slower, see the define) */                             \
        if (s < X::PPW + XW_DMA_LAG && more_ && !XW_ABLATE_DMA) {                                                              \
          const int pi_ = s - ((wave & 2) ? XW_DMA_LAG : 0);                                                 \
          if (pi_ >= 0 && pi_ < X::PPW) dma_piece(pi_, gnext_, (T_ + 1) & 1);                                \
          if (s == 0) {                                                                                      \
            dma_bias(nitem0_, (T_ + 1) & 3);                                                                 \
            if (walk_pos && !partials && wave == 0 && lane == 0 && ((T_ + 1) & 15) == 0) {                   \
              int stg_ = s0 + ((T_ + 1) >> 1);                                                               \
              stg_ -= (stg_ >= n_stages) ? n_stages : 0;                                                     \
              __hip_atomic_store(walk_pos + (blockIdx.x & 7) * 32, stg_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); \
            }                                                                                                \
          }                                                                                                  \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        XA = x3_mfma(rL[s & 3], hhi[s], XA);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        /* behind MFMA 3: the fragments of k-step s + 3 (a four-deep ring); the first read of the NEXT tile (s = KSL - 3) sits behind the tile barrier */ \
        if (s == KSL - 3 && !XW_ABLATE_BARRIER) {                                                            \
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                        \
          __syncthreads();                                                                                   \
        }                                                                                                    \
        const unsigned char* src_ = (s + 3 < KSL) ? tile_ : ntile_;                                          \
        rH[(s + 3) & 3] = XW_LDS(src_ + XW_FH((s + 3) % KSL));                                               \
        rL[(s + 3) & 3] = XW_LDS(src_ + XW_FL((s + 3) % KSL));                                               \
      }                                                                                                      \
      __builtin_amdgcn_sched_barrier(0);                                                                     \
    }
#define XW_PHASE(XA, YA, T__, MF, DR)                                                                        \
  do {                                                                                                       \
    constexpr bool MF_ = MF, DR_ = DR;                                                                       \
    const int T_ = (T__);                                                                                    \
    const unsigned char* tile_ = smem + (T_ & 1) * X::TILE_LDS;                                              \
    const unsigned char* ntile_ = smem + ((T_ + 1) & 1) * X::TILE_LDS;                                       \
    const bool more_ = MF_ && T_ + 1 < n_tiles;                                                              \
    const int nitem0_ = more_ ? tile_item0(T_ + 1) : 0;                                                      \
    const unsigned char* gnext_ = tile_gptr(nitem0_);                                                        \
    const float* ybias_ = sbias + ((T_ - 1) & 3) * 32 + 4 * lh;                                              \
    const int dit_ = DR_ ? tile_item0(T_ - 1) + 4 * lh : 0;                                                  \
    f32x4 bq_[4];                                                                                            \
    (void)tile_; (void)ntile_; (void)gnext_; (void)ybias_; (void)dit_; (void)bq_;                            \
    _Pragma("unroll") for (int s = 0; s < XW_DR0 - 1; s++) XW_STEP(XA, YA, s)                                         \
    if (DR_ && KT > 0) {   /* the list vote: every value of the drained tile carries its bias by now */      \
      if (__any(tk_m > top.v[KTA - 1])) top_tile_slow<KTA>(top, YA, dit_);                                   \
    }                                                                                                        \
    _Pragma("unroll") for (int s = XW_DR0 - 1; s < KSL; s++) XW_STEP(XA, YA, s)                                       \
  } while (0)

    if (n_tiles > 0) {
      f32x16 accA, accB;
      dma_tile_all(0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int s = 0; s < 3; s++) {   // first fragments of tile 0
        rH[s] = XW_LDS(smem + XW_FH(s));
        rL[s] = XW_LDS(smem + XW_FL(s));
      }
      XW_PHASE(accA, accB, 0, true, false);
      int T = 1;
      for (; T + 1 < n_tiles; T += 2) {
        XW_PHASE(accB, accA, T, true, true);
        XW_PHASE(accA, accB, T + 1, true, true);
      }
      XW_PHASE(accB, accA, T, true, true);      // the last tile (n_tiles is even)
      XW_PHASE(accA, accB, n_tiles, false, true);   // ... and its drain
      if (pace_slot && tid == 0) __hip_atomic_store(pace_slot, -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // no longer a neighbour
    }
#undef XW_PHASE
#undef XW_STEP
  } else {
    // fix-up / logits: nothing pipelined -- per tile the MFMAs, then the whole epilogue (a rare path and a debug path)
    auto epi_max = [&](const f32x16& v) {   // running-maximum softmax sum (rescale deferred by 2^8) + count: score_x3.hip's
      float x[16];
#pragma unroll
      for (int i = 0; i < 16; i++) x[i] = fmaf(v[i], l2e_s, -st.m2);
      float xm = fmaxf(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])), fmaxf(fmaxf(x[4], x[5]), fmaxf(x[6], x[7])));
      xm = fmaxf(xm, fmaxf(fmaxf(fmaxf(x[8], x[9]), fmaxf(x[10], x[11])), fmaxf(fmaxf(x[12], x[13]), fmaxf(x[14], x[15]))));
      if (__any(xm > 8.0f)) {
        const float d = fmaxf(ceilf(xm), 0.f);
        st.s *= __builtin_amdgcn_exp2f(-d);
        st.m2 += d;
#pragma unroll
        for (int i = 0; i < 16; i++) x[i] -= d;
      }
      float sa = 0.f, sb = 0.f;
      int ca = 0, cb = 0;
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        sa += __builtin_amdgcn_exp2f(x[i]);
        sb += __builtin_amdgcn_exp2f(x[i + 1]);
        ca += (v[i] > st.t) ? 1 : 0;
        cb += (v[i + 1] > st.t) ? 1 : 0;
      }
      st.s += sa + sb;
      st.c += ca + cb;
    };
    auto store_logits = [&](const f32x16& v, int item0) {
      if (row0 + lr >= Rv) return;
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const int it = item0 + 8 * q + j;
          if (it < N) logits[mysrc * N + it] = v[4 * q + j] * irs;
        }
    };
    if (n_tiles > 0) dma_tile_all(0, 0);
    for (int T = 0; T < n_tiles; T++) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();   // tile T has landed for everybody, and everybody is done with tile T - 1, whose buffer tile T + 1 goes into
      if (T + 1 < n_tiles) dma_tile_all(T + 1, (T + 1) & 1);
      const unsigned char* tile = smem + (T & 1) * X::TILE_LDS;
      f32x16 acc;
#pragma unroll
      for (int j = 0; j < 16; j++) acc[j] = 0.f;
#pragma unroll
      for (int s = 0; s < KSL; s++) {
        const uint4 ah = XW_LDS(tile + XW_FH(s));
        const uint4 al = XW_LDS(tile + XW_FL(s));
        XW_MFMA3(acc, ah, al, s);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + (T & 3) * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[4 * q + j] = fmaf(b4[j], rs, acc[4 * q + j]);
      }
      if (LOGITS) store_logits(acc, tile_item0(T) + 4 * lh);
      else epi_max(acc);
    }
  }
#undef XW_FH
#undef XW_FL
#undef XW_LDS
#undef XW_MFMA3
  if constexpr (LOGITS) return;

  // ---- merge the two half-waves (same step, disjoint item subsets) and write: as k_score_x3 ----------------------------------------------
  const int64_t myr = row0 + lr;
  const float l_own = st.m2 + log2f(st.s);
  const float l_o = __shfl_xor(l_own, 32);
  const int c_o = __shfl_xor(st.c, 32);
  const float s_other = __shfl_xor(st.s, 32);
  const float M2 = fmaxf(l_own, l_o);
  const float S = __builtin_amdgcn_exp2f(l_own - M2) + __builtin_amdgcn_exp2f(l_o - M2);
  const int cnt = st.c + c_o;
  if (partials) {
    if (lh == 0 && myr < Rv) {
      float* pp = partials + (int64_t)by * 2 * R;
      pp[myr] = __int_as_float(cnt);
      pp[R + myr] = (n_stages > 0) ? M2 + log2f(S) : -INFINITY;
      if (by == 0 && o_tgt) o_tgt[mysrc] = (myy > 0) ? st.t * irs : 0.f;
    }
  } else if (lh == 0 && myr < Rv) {
    const bool valid = myy > 0;
    float lse = (M2 + log2f(S)) * LN2_F;
    if (MAIN && ((__float_as_uint(st.s) & 0x7f800000u) == 0x7f800000u || (__float_as_uint(s_other) & 0x7f800000u) == 0x7f800000u)) lse = INFINITY;
    if (o_rank) o_rank[mysrc] = valid ? cnt : 0;
    if (o_ce) o_ce[mysrc] = valid ? (lse - st.t * irs) : 0.f;
    if (o_tgt) o_tgt[mysrc] = valid ? st.t * irs : 0.f;
  }
  if constexpr (KT > 0) {
    // the two half-waves' sorted lists (value descending, then index ascending) merged through LDS, which the item stream no longer needs
    float* lv = reinterpret_cast<float*>(smem);
    int* li = reinterpret_cast<int*>(smem + XW_THREADS * KTA * 4);
    top_flush<KTA>(top);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < KTA; k++) { lv[tid * KTA + k] = top.v[k]; li[tid * KTA + k] = top.i[k]; }
    __syncthreads();
    if (lh == 0 && myr < Rv) {
      const bool valid = myy > 0;
      const float* av = lv + tid * KTA; const int* ai = li + tid * KTA;
      const float* bv = lv + (tid + 32) * KTA; const int* bi = li + (tid + 32) * KTA;
      int pa = 0, pb = 0;
      for (int k = 0; k < topk; k++) {
        // masked rows are all-zero logit rows (model.py:105): every value ties at 0 -> indices 0..k-1
        const float va = pa < KTA ? av[pa] : -INFINITY, vb = pb < KTA ? bv[pb] : -INFINITY;
        const int ia = pa < KTA ? ai[pa] : 0x7fffffff, ib = pb < KTA ? bi[pb] : 0x7fffffff;
        const bool ta = (va > vb) || (va == vb && ia < ib);
        o_topi[mysrc * topk + k] = valid ? (ta ? ia : ib) : k;
        o_topv[mysrc * topk + k] = valid ? (ta ? va : vb) * irs : 0.f;
        pa += ta ? 1 : 0;
        pb += ta ? 0 : 1;
      }
    }
  }
}

template <int KT>
static hipError_t launch_x3w_c(const ModelDev& m, const ExactDev& x, const float* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                               hipStream_t st, const int32_t* row_map, const int32_t* n_rows_dev, float* split_scratch) {
  using X = XwGeom;
  size_t lds = X::LDS_BYTES;
  if (KT > 0 && (size_t)XW_THREADS * KT * 8 > lds) lds = (size_t)XW_THREADS * KT * 8;
  const unsigned blocks = (unsigned)((R + XW_ROWS - 1) / XW_ROWS);
  auto kern = k_score_x3w<KT, 0>;
  auto fix = k_score_x3w<0, 1>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(fix), hipFuncAttributeMaxDynamicSharedMemorySize, (int)X::LDS_BYTES)) != hipSuccess) return e;
  int* walk = KT > 0 ? nullptr : m.walk_pos;   // lists: equal values must arrive in index order (tf.nn.top_k), so the walk starts at item 0
#define PACE_ARG m.walk_pace
  const int parts = score_split_parts(R, m.NP, XW_ROWS);
  const int n_all = m.NP / SC_STAGE;
  if (KT == 0 && split_scratch && parts > 1 && (R + XW_ROWS - 1) / XW_ROWS < 128 && out->rank && out->ce && out->target) {
    // small batches: SC_DYN_GRID workgroups dealt as (row tile, catalogue part) on the device from the live row count (score_geom.h)
    hipLaunchKernelGGL(kern, dim3(SC_DYN_GRID), dim3(XW_THREADS), lds, st, hidden, y, R, x.outS, m.out_bias, m.N, m.NP, out->rank, out->ce, out->target,
                       row_map, n_rows_dev, -SC_DYN_MAX_PARTS, split_scratch, (int*)nullptr, (int32_t*)nullptr, (float*)nullptr, 0, (float*)nullptr, (const float*)x.outS_scale, PACE_ARG);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = launch_score_merge(split_scratch, -SC_DYN_MAX_PARTS, R, row_map, n_rows_dev, y, out->target, out->rank, out->ce, st, SC_DYN_GRID, XW_ROWS, n_all)) != hipSuccess) return e;
  } else if (KT == 0 && split_scratch && parts > 1 && out->rank && out->ce && out->target) {
    // mid-size batches: (row tile, catalogue part) workgroups, partial (count, log2-sum) per part added up by score.hip's merge kernel
    const int part_stages = (n_all + parts - 1) / parts;
    hipLaunchKernelGGL(kern, dim3(blocks, parts), dim3(XW_THREADS), lds, st, hidden, y, R, x.outS, m.out_bias, m.N, m.NP, out->rank, out->ce, out->target,
                       row_map, n_rows_dev, part_stages, split_scratch, (int*)nullptr, (int32_t*)nullptr, (float*)nullptr, 0, (float*)nullptr, (const float*)x.outS_scale, PACE_ARG);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if ((e = launch_score_merge(split_scratch, parts, R, row_map, n_rows_dev, y, out->target, out->rank, out->ce, st, 0, XW_ROWS, n_all)) != hipSuccess) return e;
  } else {
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(XW_THREADS), lds, st, hidden, y, R, x.outS, m.out_bias, m.N, m.NP, out->rank, out->ce, out->target, row_map,
                       n_rows_dev, 0, (float*)nullptr, walk, out->topk_idx, out->topk_val, (int)out->topk, (float*)nullptr, (const float*)x.outS_scale, PACE_ARG);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
#undef PACE_ARG
#define PACE_ARG (int*)nullptr
  if (!out->ce) return hipSuccess;
  hipLaunchKernelGGL(fix, dim3(blocks), dim3(XW_THREADS), X::LDS_BYTES, st, hidden, y, R, x.outS, m.out_bias, m.N, m.NP, out->rank, out->ce, out->target,
                     row_map, n_rows_dev, 0, (float*)nullptr, (int*)nullptr, (int32_t*)nullptr, (float*)nullptr, 0, (float*)nullptr, (const float*)x.outS_scale, PACE_ARG);
  return hipGetLastError();
#undef PACE_ARG
}

#define PACE_ARG (int*)nullptr
// hidden fp32 [R][512]; out (fused) or logits (materialised, parity / debug) -- exactly one of them (score_x3.hip's launch_score_x3 dispatches here)
hipError_t launch_score_x3w(const ModelDev& m, const ExactDev& x, const float* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                            float* logits, hipStream_t st, const int32_t* row_map, const int32_t* n_rows_dev, float* split_scratch) {
  if (logits) {
    using X = XwGeom;
    auto kern = k_score_x3w<0, 2>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)X::LDS_BYTES);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((unsigned)((R + XW_ROWS - 1) / XW_ROWS)), dim3(XW_THREADS), X::LDS_BYTES, st, hidden, (const int32_t*)nullptr, R, x.outS,
                       m.out_bias, m.N, m.NP, (int32_t*)nullptr, (float*)nullptr, (float*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, 0,
                       (float*)nullptr, (int*)nullptr, (int32_t*)nullptr, (float*)nullptr, 0, logits, (const float*)x.outS_scale, PACE_ARG);
    return hipGetLastError();
  }
  if (out->topk <= 0) return launch_x3w_c<0>(m, x, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
  if (out->topk <= 10) return launch_x3w_c<10>(m, x, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);   // the reference's recall@10 lists
  return launch_x3w_c<16>(m, x, hidden, y, R, out, st, row_map, n_rows_dev, split_scratch);
}
