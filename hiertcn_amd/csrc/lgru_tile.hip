// HierGRU low level (model_hier.py:67-68 -> model_gru.py:47-55: dynamic_rnn of a MultiRNNCell of GRUCells from a zero state, with
// sequence_length; cell arithmetic customed_gru_cell.py:309-337) of SMALL batches, session-tile stationary: sessions do not interact, so a
// workgroup takes 32 of them through ALL L steps and nl layers on its own -- no launch per cell (the chain of gru_x3.hip: 2 (L + nl - 1)
// dependent launches, ~25 us each at 320 sessions), no exchange between workgroups (the persistent launch of gru_persist.hip: a group barrier
// behind each of 2 (L + nl - 1) phases).  What moves per cell is the WEIGHTS: a wave streams the rows of its 32-unit blocks -- stored
// fragment-major, hi and lo 16-bit parts (the split operands of gru_x3.hip: a.w = a_hi w_hi + a_lo w_hi + a_hi w_lo, 2^-16 per product) --
// from the L2 straight into a REGISTER ring eight k-steps deep (a fragment is one coalesced 1 KB load, used once: no LDS copy; the cell's
// 3 (E + H) / 16 k-steps are fully unrolled, so ring slots are compile-time and the ring runs on into the next cell's first fragments); the
// activations [x | h] of the 32 sessions live in LDS as split 16-bit rows (the MFMA B operand), the fp32 states in registers.
//   D[row = unit][col = session] (swapped orientation, as gru_x3.hip): a lane holds 16 units of one session per tile.
//   wave w owns unit blocks w, w + 4, ...: the r, z and candidate tiles of the same units, so only r * h crosses waves (through LDS).
// Two workgroup barriers per cell.  Applies to E == H in {64, 128, 256}, one or two layers and up to 32,768 sessions, 16-bit storage mode, evaluation and the training
// forward (which also keeps r, z, r * h, the candidate, the output and every state for the backward pass).
#include <stdlib.h>
#include <string.h>

#include "common.h"

#define LT_THREADS 256

struct LgruTileArgs {
  const uint16_t* u16;                       // [M][L][E] projected [e, state] rows, storage type
  const int32_t* seq_len;                    // [M]
  const uint16_t* wgF_hi[HIERTCN_MAX_LAYERS]; const uint16_t* wgF_lo[HIERTCN_MAX_LAYERS];   // [2H/32][K/16][64][8]
  const uint16_t* wcF_hi[HIERTCN_MAX_LAYERS]; const uint16_t* wcF_lo[HIERTCN_MAX_LAYERS];   // [H/32][K/16][64][8]
  const float* bg[HIERTCN_MAX_LAYERS]; const float* bc[HIERTCN_MAX_LAYERS];
  uint16_t* hidden;                          // [M * L][H]: the top layer's outputs, zero past the length
  float* states;                             // training: [(L + 1)][M][HT] (slot 0 zeroed by the caller); null in evaluation
  float* rec;                                // training: per (t, l): gates [M][2H] | r*h [M][H] | cand [M][H] | out [M][H]
  int M, L, H, HT, nl;
};

__device__ __forceinline__ float lt_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float lt_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
__device__ __forceinline__ void lt_split(float v, uint16_t& hi, uint16_t& lo) {
  hi = f2bf(v);
  lo = f2bf(v - bf2f(hi));
}

template <int HB, int NLM>   // H / 32 in {2, 4, 8} (E == H); NLM: compile-time bound on the layers (the states are register arrays)
__global__ __launch_bounds__(LT_THREADS, 1) void k_lgru_tile(const LgruTileArgs a) {
  constexpr int H = 32 * HB, KS = 4 * HB;          // k-steps of a product: (E + H) / 16 with E == H
  constexpr int KSI = KS / 2;                      // ... of which the input half
  constexpr int PITCH = 2 * H + 16;                // bytes per session row of a split activation buffer (rows 4 banks apart)
  constexpr int NBW = (HB + 3) / 4;                // unit blocks per wave: block b = wave + 4 bi
  constexpr int IC = NBW * 3 * KS;                 // k-steps of weight fragments a wave consumes per cell
  constexpr int LT_R = (HB == 2) ? 8 : 16;         // register ring: the fragments of LT_R - 1 k-steps in flight (an L2 round trip under this load is ~1 us:
                                                   // with 7 in flight a k-step took ~280 cycles against the 96 of its three MFMAs)
  static_assert(HB == 2 || HB == 4 || HB == 8, "E == H in {64, 128, 256}");
  static_assert(IC % LT_R == 0, "ring slots must repeat from cell to cell");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nl = a.nl, L = a.L, M = a.M, HT = a.HT;
  const int64_t sess0 = (int64_t)blockIdx.x * 32;
  const int64_t sess = sess0 + lr;
  const bool live = sess < M;
  const int64_t sc = live ? sess : (int64_t)M - 1;
  const int len = a.seq_len[sc];
  const bool has_blocks = wave < HB;               // (H = 64: two of the four waves have no unit block; they help with the row loads only)

  // ---- LDS carve-up
  unsigned char* UB = smem;                                        // [2][32][PITCH]: the projected rows of step t (storage type, no lo part)
  unsigned char* HSh = UB + 2 * 32 * PITCH;                        // [nl][32][PITCH] state, hi parts
  unsigned char* HSl = HSh + nl * 32 * PITCH;                      // [nl][32][PITCH] lo parts
  unsigned char* RHh = HSl + nl * 32 * PITCH;                      // [32][PITCH] r * h
  unsigned char* RHl = RHh + 32 * PITCH;
  for (int i = tid; i < 2 * nl * 32 * PITCH / 16; i += LT_THREADS) reinterpret_cast<uint4*>(HSh)[i] = make_uint4(0, 0, 0, 0);   // zero_state

  // ---- weight stream.  Item i of a cell (i < IC): for every block of the wave its r tile (k-steps 0 .. KS-1) and its z tile, then -- behind the
  // barrier -- every block's candidate tile.  Fragment-major tables: tile row T, k-step s at ((T KS + s) 64 + lane) 8.
  const uint16_t* wg_hi[NLM]; const uint16_t* wg_lo[NLM];
  const uint16_t* wc_hi[NLM]; const uint16_t* wc_lo[NLM];
#pragma unroll
  for (int l = 0; l < NLM; l++) {   // this wave's first block of every table, this lane's 16 bytes of a fragment
    wg_hi[l] = a.wgF_hi[l] + (size_t)wave * KS * 512 + lane * 8; wg_lo[l] = a.wgF_lo[l] + (size_t)wave * KS * 512 + lane * 8;
    wc_hi[l] = a.wcF_hi[l] + (size_t)wave * KS * 512 + lane * 8; wc_lo[l] = a.wcF_lo[l] + (size_t)wave * KS * 512 + lane * 8;
  }
  uint4 rgh[LT_R], rgl[LT_R];   // the ring: item n of the running sequence in slot n % LT_R
  auto frag_load = [&](const uint16_t* ghi, const uint16_t* glo, const uint16_t* chi, const uint16_t* clo, int i, uint4& fh, uint4& fl) {
    // (i is a compile-time constant at every call site once the loops are unrolled)
    if (i < 2 * NBW * KS) {
      const int bi = i / (2 * KS), sel = (i / KS) & 1, ks = i % KS;
      const size_t off = ((size_t)(sel * HB + 4 * bi) * KS + ks) * 512;   // z rows start HB tiles behind the r rows
      fh = *reinterpret_cast<const uint4*>(ghi + off);
      fl = *reinterpret_cast<const uint4*>(glo + off);
    } else {
      const int c = i - 2 * NBW * KS, bi = c / KS, ks = c % KS;
      const size_t off = ((size_t)(4 * bi) * KS + ks) * 512;
      fh = *reinterpret_cast<const uint4*>(chi + off);
      fl = *reinterpret_cast<const uint4*>(clo + off);
    }
  };
  if (has_blocks) {
#pragma unroll
    for (int i = 0; i < LT_R - 1; i++) frag_load(wg_hi[0], wg_lo[0], wc_hi[0], wc_lo[0], i, rgh[i], rgl[i]);
  }

  float hreg[NLM][NBW][16];   // this lane's units of its session, every layer (fp32 state)
#pragma unroll
  for (int l = 0; l < NLM; l++)
#pragma unroll
    for (int b = 0; b < NBW; b++)
#pragma unroll
      for (int i = 0; i < 16; i++) hreg[l][b][i] = 0.f;

  // the projected rows of a step: 32 sessions x H 16-bit values = 32 x (H / 8) chunks of 16 bytes
  constexpr int UCH = 32 * (H / 8), UPT = (UCH + LT_THREADS - 1) / LT_THREADS;
  uint4 ureg[UPT];
  auto u_fetch = [&](int t) {
#pragma unroll
    for (int k = 0; k < UPT; k++) {
      const int c = tid + k * LT_THREADS;
      if (c < UCH) {
        const int row = c / (H / 8), ch = c % (H / 8);
        const int64_t s = (sess0 + row < M) ? sess0 + row : (int64_t)M - 1;
        ureg[k] = *reinterpret_cast<const uint4*>(a.u16 + (s * L + t) * H + ch * 8);
      }
    }
  };
  auto u_store = [&](int buf) {
#pragma unroll
    for (int k = 0; k < UPT; k++) {
      const int c = tid + k * LT_THREADS;
      if (c < UCH) {
        const int row = c / (H / 8), ch = c % (H / 8);
        *reinterpret_cast<uint4*>(UB + (buf * 32 + row) * PITCH + ch * 16) = ureg[k];
      }
    }
  };
  u_fetch(0);
  u_store(0);
  __syncthreads();

  const size_t cellf = (size_t)M * 5 * H;   // floats per (t, l) record
  for (int t = 0; t < L; t++) {
    if (t + 1 < L) u_fetch(t + 1);          // in flight during the step's cells
    const bool active = t < len;
#pragma unroll
    for (int l = 0; l < NLM; l++) {   // (compile-time l: the states are register arrays)
      if (l >= nl) break;
      const unsigned char* Xh = (l == 0) ? UB + (t & 1) * 32 * PITCH : HSh + (l - 1) * 32 * PITCH;
      const unsigned char* Xl = (l == 0) ? nullptr : HSl + (l - 1) * 32 * PITCH;
      const unsigned char* Sh = HSh + l * 32 * PITCH;
      const unsigned char* Sl = HSl + l * 32 * PITCH;
      float* rec = a.rec ? a.rec + ((size_t)t * nl + l) * cellf : nullptr;
      float zreg[NBW][16];
      // the tables of this cell and of the next one (the ring prefetches across the cell boundary)
      const uint16_t* ghi = wg_hi[l]; const uint16_t* glo = wg_lo[l]; const uint16_t* chi = wc_hi[l]; const uint16_t* clo = wc_lo[l];
      const bool wrap = l + 1 >= nl;
      const uint16_t* nghi = wrap ? wg_hi[0] : wg_hi[(l + 1) % NLM]; const uint16_t* nglo = wrap ? wg_lo[0] : wg_lo[(l + 1) % NLM];
      const uint16_t* nchi = wrap ? wc_hi[0] : wc_hi[(l + 1) % NLM]; const uint16_t* nclo = wrap ? wc_lo[0] : wc_lo[(l + 1) % NLM];
      // (opaque copies: the fragment loads do not depend on t, and hipcc would otherwise hoist every one of them out of the step loop -- 512 registers
      // and 1.8 KB of scratch per lane)
      asm volatile("" : "+v"(ghi), "+v"(glo), "+v"(chi), "+v"(clo), "+v"(nghi), "+v"(nglo), "+v"(nchi), "+v"(nclo));
      // one k-step: item i of the cell into `acc`.  The B operand fragment of k-step ks (hi, lo) was read one k-step earlier (a single wave per
      // SIMD has nobody to cover an LDS round trip); this step reads the next one's -- k-step ks + 1 of the same tile, or k-step 0 of the tile's
      // successor in the same phase -- and the ring slot freed by item i - 1 takes item i + LT_R - 1.
      uint4 bqh[2], bql[2];
      auto b_read = [&](const unsigned char* hp, const unsigned char* lp, int ks, uint4& oh, uint4& ol) {   // ks compile-time; second half from (hp2, lp2)
        const int ko = ks * 32 + lh * 16;
        oh = *reinterpret_cast<const uint4*>(hp + lr * PITCH + ko);
        ol = lp ? *reinterpret_cast<const uint4*>(lp + lr * PITCH + ko) : make_uint4(0, 0, 0, 0);
      };
#define LT_BPTR_H(ks_, h2_) ((ks_) < KSI ? Xh : (h2_))
#define LT_BPTR_L(ks_, l2_) ((ks_) < KSI ? Xl : (l2_))
#define LT_KSTEP(acc_, i_, ks_, h2_, l2_, more_)                                                                      \
      {                                                                                                               \
        const int ii_ = (i_), kk_ = (ks_);                                                                           \
        if (ii_ + LT_R - 1 < IC) frag_load(ghi, glo, chi, clo, ii_ + LT_R - 1, rgh[(ii_ + LT_R - 1) % LT_R], rgl[(ii_ + LT_R - 1) % LT_R]);          \
        else frag_load(nghi, nglo, nchi, nclo, ii_ + LT_R - 1 - IC, rgh[(ii_ + LT_R - 1) % LT_R], rgl[(ii_ + LT_R - 1) % LT_R]);                     \
        if (kk_ + 1 < KS) b_read(LT_BPTR_H(kk_ + 1, h2_), LT_BPTR_L(kk_ + 1, l2_), (kk_ + 1) % KSI, bqh[(kk_ + 1) & 1], bql[(kk_ + 1) & 1]);         \
        else if (more_) b_read(Xh, Xl, 0, bqh[0], bql[0]);   /* KS is even: the next tile's k-step 0 lands in slot 0 */                              \
        acc_ = mfma32(rgh[ii_ % LT_R], bqh[kk_ & 1], acc_);                                                          \
        acc_ = mfma32(rgl[ii_ % LT_R], bqh[kk_ & 1], acc_);                                                          \
        if (LT_BPTR_L(kk_, l2_)) acc_ = mfma32(rgh[ii_ % LT_R], bql[kk_ & 1], acc_);                                 \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
      }
      if (has_blocks) {
        // ---------------- gates: the r and z tiles of this wave's units
#pragma unroll
        for (int bi = 0; bi < NBW; bi++) {
          const int blk = wave + 4 * bi;
          f32x16 acc[2];
#pragma unroll
          for (int sel = 0; sel < 2; sel++) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) acc[sel][reg] = a.bg[l][sel * H + 32 * blk + acc_row(reg, lh)];
            if (bi == 0 && sel == 0) b_read(Xh, Xl, 0, bqh[0], bql[0]);   // (every later tile's first fragment is read by its predecessor's last k-step)
#pragma unroll
            for (int ks = 0; ks < KS; ks++) LT_KSTEP(acc[sel], (bi * 2 + sel) * KS + ks, ks, Sh, Sl, !(bi == NBW - 1 && sel == 1))
          }
          // epilogue (customed_gru_cell.py:323-326): r, z = sigmoid; r * h -> LDS (split) for the candidate's recurrent half
#pragma unroll
          for (int q = 0; q < 4; q++) {
            float r4[4], rh4[4];
            uint16_t hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              r4[j] = lt_sigmoid(acc[0][4 * q + j]);
              zreg[bi][4 * q + j] = lt_sigmoid(acc[1][4 * q + j]);
              rh4[j] = r4[j] * hreg[l][bi][4 * q + j];
              lt_split(rh4[j], hi[j], lo[j]);
            }
            const int unit = 32 * blk + 8 * q + 4 * lh;
            *reinterpret_cast<uint2*>(RHh + lr * PITCH + unit * 2) = make_uint2(hi[0] | ((uint32_t)hi[1] << 16), hi[2] | ((uint32_t)hi[3] << 16));
            *reinterpret_cast<uint2*>(RHl + lr * PITCH + unit * 2) = make_uint2(lo[0] | ((uint32_t)lo[1] << 16), lo[2] | ((uint32_t)lo[3] << 16));
            if (rec && live) {
              *reinterpret_cast<float4*>(rec + sess * 2 * H + unit) = make_float4(r4[0], r4[1], r4[2], r4[3]);
              *reinterpret_cast<float4*>(rec + sess * 2 * H + H + unit) = make_float4(zreg[bi][4 * q], zreg[bi][4 * q + 1], zreg[bi][4 * q + 2], zreg[bi][4 * q + 3]);
              *reinterpret_cast<float4*>(rec + (size_t)M * 2 * H + sess * H + unit) = make_float4(rh4[0], rh4[1], rh4[2], rh4[3]);
            }
          }
        }
      }
      __syncthreads();   // r * h of every unit is in LDS
      if (has_blocks) {
        // ---------------- candidate (customed_gru_cell.py:333-336): c = tanh([x, r*h] Wc + b); h' = z h + (1 - z) c
#pragma unroll
        for (int bi = 0; bi < NBW; bi++) {
          const int blk = wave + 4 * bi;
          f32x16 acc;
#pragma unroll
          for (int reg = 0; reg < 16; reg++) acc[reg] = a.bc[l][32 * blk + acc_row(reg, lh)];
          if (bi == 0) b_read(Xh, Xl, 0, bqh[0], bql[0]);
#pragma unroll
          for (int ks = 0; ks < KS; ks++) LT_KSTEP(acc, (2 * NBW + bi) * KS + ks, ks, RHh, RHl, bi + 1 < NBW)
          // dynamic_rnn past the session's length: state through, output zero (model_gru.py:47-53)
#pragma unroll
          for (int q = 0; q < 4; q++) {
            float c4[4], hn[4], st4[4];
            uint16_t hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const float h = hreg[l][bi][4 * q + j], z = zreg[bi][4 * q + j];
              c4[j] = lt_tanh(acc[4 * q + j]);
              hn[j] = z * h + (1.0f - z) * c4[j];
              st4[j] = active ? hn[j] : h;
              hreg[l][bi][4 * q + j] = st4[j];
              lt_split(st4[j], hi[j], lo[j]);
            }
            const int unit = 32 * blk + 8 * q + 4 * lh;
            // the state's split is the h operand of step t + 1 and the next layer's input at this step (for a finished session the next layer's
            // result is discarded, so the carried state standing in for the cell output there changes nothing).  Nobody reads HS[l] in this
            // phase (its readers are the gates phases of this layer and the cells of layer l + 1, all behind a barrier from here).
            *reinterpret_cast<uint2*>(HSh + (l * 32 + lr) * PITCH + unit * 2) = make_uint2(hi[0] | ((uint32_t)hi[1] << 16), hi[2] | ((uint32_t)hi[3] << 16));
            *reinterpret_cast<uint2*>(HSl + (l * 32 + lr) * PITCH + unit * 2) = make_uint2(lo[0] | ((uint32_t)lo[1] << 16), lo[2] | ((uint32_t)lo[3] << 16));
            if (live) {
              if (rec) {
                *reinterpret_cast<float4*>(rec + (size_t)M * 3 * H + sess * H + unit) = make_float4(c4[0], c4[1], c4[2], c4[3]);
                *reinterpret_cast<float4*>(rec + (size_t)M * 4 * H + sess * H + unit) = make_float4(hn[0], hn[1], hn[2], hn[3]);
                *reinterpret_cast<float4*>(a.states + ((size_t)(t + 1) * M + sess) * HT + (size_t)l * H + unit) = make_float4(st4[0], st4[1], st4[2], st4[3]);
              }
              if (l == nl - 1) {
                uint2 o;
                o.x = active ? pack2bf(hn[0], hn[1]) : 0u;
                o.y = active ? pack2bf(hn[2], hn[3]) : 0u;
                *reinterpret_cast<uint2*>(a.hidden + (sess * L + t) * H + unit) = o;
              }
            }
          }
        }
      }
      if (l == nl - 1 && t + 1 < L) u_store((t + 1) & 1);   // (the other buffer: nobody reads it during step t)
      __syncthreads();   // the new state (and, behind the last layer, the next step's rows) are in LDS
    }
  }
#undef LT_KSTEP
#undef LT_BPTR_H
#undef LT_BPTR_L
}

static size_t lt_lds_bytes(const ModelDev& m) {
  const size_t pitch = 2 * (size_t)m.H + 16;
  return 2 * 32 * pitch + 2 * (size_t)m.nl * 32 * pitch + 2 * 32 * pitch;   // rows of step t (x 2) | states hi, lo | r*h hi, lo
}

static int64_t lt_max_sessions(const ModelDev& m) {   // sessions up to which the session-tile kernels run (tune.lgru_tile_max_sessions; default 131072)
  // (measured up to 20,480 sessions of 30 steps: 77.6 vs 80.0 ms per 1024-user evaluation step at d = 256 against the launch chain, 51.9 vs 53.7 at d = 128;
  //  round 4, 40,960 sessions of <= 20 steps, N = 20,778: evaluation 10.3 vs 14.4 ms at d = 256, 4.6 vs 7.2 at d = 128; training step 68 vs 82 and 33 vs 39 ms)
  return m.tune.lgru_tile_max_sessions;
}

bool lgru_tile_applies(const ModelDev& m, int64_t n_sess) {
  if (m.tune.lgru_tile == 0) return false;   // never (A/B runs, tests)
  if (!m.low_gru || !m.lgru_split || !m.lgru_wgF_hi[0] || n_sess <= 0 || n_sess > lt_max_sessions(m)) return false;
  if (m.tune.lgru_fp32_cells == 1) return false;   // (the fp32 cells were asked for)
  return m.E == m.H && (m.H == 64 || m.H == 128 || m.H == 256) && m.nl <= 2 && lt_lds_bytes(m) <= 160 * 1024;
}

hipError_t launch_lgru_tile(const ModelDev& m, const uint16_t* u16, const int32_t* seq_len, int64_t n_sess, int L, uint16_t* hidden, float* states, float* rec,
                            hipStream_t st) {
  LgruTileArgs a;
  memset(&a, 0, sizeof(a));
  a.u16 = u16; a.seq_len = seq_len; a.hidden = hidden; a.states = states; a.rec = rec;
  for (int l = 0; l < m.nl; l++) {
    a.wgF_hi[l] = m.lgru_wgF_hi[l]; a.wgF_lo[l] = m.lgru_wgF_lo[l]; a.wcF_hi[l] = m.lgru_wcF_hi[l]; a.wcF_lo[l] = m.lgru_wcF_lo[l];
    a.bg[l] = m.lgru_bg[l]; a.bc[l] = m.lgru_bc[l];
  }
  a.M = (int)n_sess; a.L = L; a.H = m.H; a.HT = m.HT; a.nl = m.nl;
  const size_t lds = lt_lds_bytes(m);
  const unsigned blocks = (unsigned)((n_sess + 31) / 32);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(LT_THREADS), lds, st, a);
    return hipGetLastError();
  };
  switch (m.H) {
    case 64: return go(k_lgru_tile<2, 2>);
    case 128: return go(k_lgru_tile<4, 2>);
    case 256: return go(k_lgru_tile<8, 2>);
    default: return hipErrorInvalidValue;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward through time of the same cells (model_gru.py:47-55 differentiated by model.py:139-141), same decomposition: a workgroup takes its 32
// sessions from the last step to the first through every layer, the carried state gradients and the gradient handed down to the layer below stay
// in registers (with E == H the input-half tile of a unit block belongs to the wave that owns that block one layer down), the TRANSPOSED weight
// rows stream through the register ring.  Per cell (customed_gru_cell.py:323-336 backwards):
//   d = active ? (carried + d out) : 0;  da_c = d (1-z)(1-c^2);  dh' = d z;  da_z = d (h - c) z (1-z)
//   [d x_c | d(r h)] = da_c . Wc;   da_r = d(r h) h r (1-r);   dh' += d(r h) r
//   [d x_g | d h_g]  = [da_r | da_z] . Wg;   d x = d x_c + d x_g  (-> the layer below, or d u);   carried = active ? dh' + d h_g : carried
// The pre-activation gradients da_c, [da_r | da_z] of every cell are kept: the weight gradients are formed behind the loop, one product per layer
// and operand over all L x M rows (train.hip, lgru_deferred_wgrad).  Two workgroup barriers per cell.
struct LgruTileBwdArgs {
  const float* dH;                           // [M * L][H]: d loss / d (top layer output)
  const int32_t* seq_len;
  const float* states;                       // [(L + 1)][M][HT]
  const float* rec;                          // per (t, l): gates [M][2H] | r*h | cand | out
  const uint16_t* wcT_hi[HIERTCN_MAX_LAYERS]; const uint16_t* wcT_lo[HIERTCN_MAX_LAYERS];   // (Wc)^T fragment-major: [K/32][H/16][64][8]
  const uint16_t* wgT_hi[HIERTCN_MAX_LAYERS]; const uint16_t* wgT_lo[HIERTCN_MAX_LAYERS];   // (Wg)^T: [K/32][2H/16][64][8]
  float* du;                                 // [M * L][E]: gradient of the projected rows
  float* dac_all;                            // [(l L + t)][M][H]
  float* dag_all;                            // [(l L + t)][M][2H]
  int M, L, H, HT, nl;
};

template <int HB, int NLM>
__global__ __launch_bounds__(LT_THREADS, 1) void k_lgru_tile_bwd(const LgruTileBwdArgs a) {
  constexpr int H = 32 * HB;
  constexpr int KSC = 2 * HB, KSG = 4 * HB;        // k-steps (reduction over H resp. 2H) of the two products
  constexpr int PC = 2 * H + 16, PG = 4 * H + 16;  // LDS pitches of the split da_c resp. [da_r | da_z] rows
  constexpr int NBW = (HB + 3) / 4;
  constexpr int IC = NBW * 2 * (KSC + KSG);
  constexpr int LT_R = (HB == 4) ? 16 : 8;         // (H = 256: two blocks per wave leave no room for sixteen)
  static_assert(IC % LT_R == 0, "ring slots must repeat from cell to cell");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nl = a.nl, L = a.L, M = a.M, HT = a.HT;
  const int64_t sess = (int64_t)blockIdx.x * 32 + lr;
  const bool live = sess < M;
  const int64_t sc = live ? sess : (int64_t)M - 1;
  const int len = a.seq_len[sc];
  const bool has_blocks = wave < HB;
  unsigned char* DCh = smem;                 // [32][PC] da_c, hi
  unsigned char* DCl = DCh + 32 * PC;
  unsigned char* DGh = DCl + 32 * PC;        // [32][PG] [da_r | da_z], hi
  unsigned char* DGl = DGh + 32 * PG;

  const uint16_t* tc_hi[NLM]; const uint16_t* tc_lo[NLM];
  const uint16_t* tg_hi[NLM]; const uint16_t* tg_lo[NLM];
#pragma unroll
  for (int l = 0; l < NLM; l++) {
    tc_hi[l] = a.wcT_hi[l] + (size_t)wave * KSC * 512 + lane * 8; tc_lo[l] = a.wcT_lo[l] + (size_t)wave * KSC * 512 + lane * 8;
    tg_hi[l] = a.wgT_hi[l] + (size_t)wave * KSG * 512 + lane * 8; tg_lo[l] = a.wgT_lo[l] + (size_t)wave * KSG * 512 + lane * 8;
  }
  uint4 rgh[LT_R], rgl[LT_R];
  // item i of a cell: for every block (candidate product) its input-half tile then its state-half tile; then the same for the gates product
  auto frag_load = [&](const uint16_t* chi, const uint16_t* clo, const uint16_t* ghi, const uint16_t* glo, int i, uint4& fh, uint4& fl) {
    if (i < 2 * NBW * KSC) {
      const int bi = i / (2 * KSC), part = (i / KSC) & 1, ks = i % KSC;
      const size_t off = ((size_t)(part * HB + 4 * bi) * KSC + ks) * 512;
      fh = *reinterpret_cast<const uint4*>(chi + off);
      fl = *reinterpret_cast<const uint4*>(clo + off);
    } else {
      const int c = i - 2 * NBW * KSC, bi = c / (2 * KSG), part = (c / KSG) & 1, ks = c % KSG;
      const size_t off = ((size_t)(part * HB + 4 * bi) * KSG + ks) * 512;
      fh = *reinterpret_cast<const uint4*>(ghi + off);
      fl = *reinterpret_cast<const uint4*>(glo + off);
    }
  };
  const int ltop = nl - 1;
  if (has_blocks) {
    const uint16_t* c0 = ltop == 0 ? tc_hi[0] : (ltop == 1 ? tc_hi[1 % NLM] : tc_hi[0]);   // (NLM <= 2)
    const uint16_t* c1 = ltop == 0 ? tc_lo[0] : tc_lo[1 % NLM];
    const uint16_t* g0 = ltop == 0 ? tg_hi[0] : tg_hi[1 % NLM];
    const uint16_t* g1 = ltop == 0 ? tg_lo[0] : tg_lo[1 % NLM];
#pragma unroll
    for (int i = 0; i < LT_R - 1; i++) frag_load(c0, c1, g0, g1, i, rgh[i], rgl[i]);
  }

  float dcar[NLM][NBW][16];   // carried state gradients
  float dinu[NBW][16];        // gradient handed down by the layer above at this step
#pragma unroll
  for (int l = 0; l < NLM; l++)
#pragma unroll
    for (int b = 0; b < NBW; b++)
#pragma unroll
      for (int i = 0; i < 16; i++) dcar[l][b][i] = 0.f;   // the low level's final state is dropped (model_gru.py:47)
#pragma unroll
  for (int b = 0; b < NBW; b++)
#pragma unroll
    for (int i = 0; i < 16; i++) dinu[b][i] = 0.f;

  const size_t cellf = (size_t)M * 5 * H;
  for (int t = L - 1; t >= 0; t--) {
    const bool active = t < len;
#pragma unroll
    for (int lq = 0; lq < NLM; lq++) {      // layers top down: l = nl - 1 - lq (compile-time lq, l = a select below)
      if (lq >= nl) break;
      // (the register arrays are indexed by the compile-time POSITION from the top: position lq holds layer nl - 1 - lq)
      const int l = nl - 1 - lq;
      const float* rec = a.rec + ((size_t)t * nl + l) * cellf;
      const float* hp = a.states + ((size_t)t * M + sc) * HT + (size_t)l * H;
      const uint16_t* chi = l == 0 ? tc_hi[0] : tc_hi[1 % NLM]; const uint16_t* clo = l == 0 ? tc_lo[0] : tc_lo[1 % NLM];
      const uint16_t* ghi = l == 0 ? tg_hi[0] : tg_hi[1 % NLM]; const uint16_t* glo = l == 0 ? tg_lo[0] : tg_lo[1 % NLM];
      const int ln = l == 0 ? nl - 1 : l - 1;   // the next cell's layer
      const uint16_t* nchi = ln == 0 ? tc_hi[0] : tc_hi[1 % NLM]; const uint16_t* nclo = ln == 0 ? tc_lo[0] : tc_lo[1 % NLM];
      const uint16_t* nghi = ln == 0 ? tg_hi[0] : tg_hi[1 % NLM]; const uint16_t* nglo = ln == 0 ? tg_lo[0] : tg_lo[1 % NLM];
      asm volatile("" : "+v"(chi), "+v"(clo), "+v"(ghi), "+v"(glo), "+v"(nchi), "+v"(nclo), "+v"(nghi), "+v"(nglo));
      float dhp[NBW][16], dagz[NBW][16], dxc[NBW][16];   // (r and the old state are read again in P2: cheaper than 32 registers per block across a barrier)
      // ---------------- P1: through the blend and the tanh
      if (has_blocks) {
#pragma unroll
        for (int bi = 0; bi < NBW; bi++) {
          const int blk = wave + 4 * bi;
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int unit = 32 * blk + 8 * q + 4 * lh;
            const float4 r4 = *reinterpret_cast<const float4*>(rec + sc * 2 * H + unit);
            const float4 z4 = *reinterpret_cast<const float4*>(rec + sc * 2 * H + H + unit);
            const float4 c4 = *reinterpret_cast<const float4*>(rec + (size_t)M * 3 * H + sc * H + unit);
            const float4 h4 = *reinterpret_cast<const float4*>(hp + unit);
            float4 o4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (lq == 0) o4 = *reinterpret_cast<const float4*>(a.dH + (sc * L + t) * H + unit);
            const float rv[4] = {r4.x, r4.y, r4.z, r4.w}, zv[4] = {z4.x, z4.y, z4.z, z4.w}, cv[4] = {c4.x, c4.y, c4.z, c4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
            const float ov[4] = {o4.x, o4.y, o4.z, o4.w};
            float dac[4];
            uint16_t hi[4], lo[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int e = 4 * q + j;
              const float dout = (lq == 0) ? ov[j] : dinu[bi][e];
              const float d = active ? dcar[lq][bi][e] + dout : 0.f;
              dac[j] = d * (1.f - zv[j]) * (1.f - cv[j] * cv[j]);
              dhp[bi][e] = d * zv[j];
              dagz[bi][e] = d * (hv[j] - cv[j]) * zv[j] * (1.f - zv[j]);
              lt_split(dac[j], hi[j], lo[j]);
            }
            *reinterpret_cast<uint2*>(DCh + lr * PC + unit * 2) = make_uint2(hi[0] | ((uint32_t)hi[1] << 16), hi[2] | ((uint32_t)hi[3] << 16));
            *reinterpret_cast<uint2*>(DCl + lr * PC + unit * 2) = make_uint2(lo[0] | ((uint32_t)lo[1] << 16), lo[2] | ((uint32_t)lo[3] << 16));
            if (live) *reinterpret_cast<float4*>(a.dac_all + (((size_t)l * L + t) * M + sess) * H + unit) = make_float4(dac[0], dac[1], dac[2], dac[3]);
          }
        }
      }
      __syncthreads();   // da_c of every unit is in LDS
      uint4 bqh[2], bql[2];
      auto b_read = [&](const unsigned char* hp_, const unsigned char* lp_, int pitch, int ks, uint4& oh, uint4& ol) {
        const int ko = ks * 32 + lh * 16;
        oh = *reinterpret_cast<const uint4*>(hp_ + lr * pitch + ko);
        ol = *reinterpret_cast<const uint4*>(lp_ + lr * pitch + ko);
      };
#define LB_KSTEP(acc_, i_, ks_, nks_, bh_, bl_, pitch_, more_)                                                        \
      {                                                                                                               \
        const int ii_ = (i_), kk_ = (ks_);                                                                           \
        if (ii_ + LT_R - 1 < IC) frag_load(chi, clo, ghi, glo, ii_ + LT_R - 1, rgh[(ii_ + LT_R - 1) % LT_R], rgl[(ii_ + LT_R - 1) % LT_R]);          \
        else frag_load(nchi, nclo, nghi, nglo, ii_ + LT_R - 1 - IC, rgh[(ii_ + LT_R - 1) % LT_R], rgl[(ii_ + LT_R - 1) % LT_R]);                     \
        if (kk_ + 1 < (nks_)) b_read(bh_, bl_, pitch_, kk_ + 1, bqh[(kk_ + 1) & 1], bql[(kk_ + 1) & 1]);              \
        else if (more_) b_read(bh_, bl_, pitch_, 0, bqh[0], bql[0]);                                                 \
        acc_ = mfma32(rgh[ii_ % LT_R], bqh[kk_ & 1], acc_);                                                          \
        acc_ = mfma32(rgl[ii_ % LT_R], bqh[kk_ & 1], acc_);                                                          \
        acc_ = mfma32(rgh[ii_ % LT_R], bql[kk_ & 1], acc_);                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                                           \
      }
      if (has_blocks) {
        // ---------------- P2: [d x_c | d(r h)] = da_c . Wc; through the reset gate
#pragma unroll
        for (int bi = 0; bi < NBW; bi++) {
          const int blk = wave + 4 * bi;
          f32x16 acc[2];
#pragma unroll
          for (int part = 0; part < 2; part++) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) acc[part][reg] = 0.f;
            if (bi == 0 && part == 0) b_read(DCh, DCl, PC, 0, bqh[0], bql[0]);
#pragma unroll
            for (int ks = 0; ks < KSC; ks++) LB_KSTEP(acc[part], (bi * 2 + part) * KSC + ks, ks, KSC, DCh, DCl, PC, !(bi == NBW - 1 && part == 1))
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int unit = 32 * blk + 8 * q + 4 * lh;
            float dagr[4];
            uint16_t rhi[4], rlo[4], zhi[4], zlo[4];
            const float4 r4 = *reinterpret_cast<const float4*>(rec + sc * 2 * H + unit);
            const float4 h4 = *reinterpret_cast<const float4*>(hp + unit);
            const float rv[4] = {r4.x, r4.y, r4.z, r4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int e = 4 * q + j;
              dxc[bi][e] = acc[0][e];
              const float drh = acc[1][e];
              dagr[j] = drh * hv[j] * rv[j] * (1.f - rv[j]);
              dhp[bi][e] += drh * rv[j];
              lt_split(dagr[j], rhi[j], rlo[j]);
              lt_split(dagz[bi][e], zhi[j], zlo[j]);
            }
            *reinterpret_cast<uint2*>(DGh + lr * PG + unit * 2) = make_uint2(rhi[0] | ((uint32_t)rhi[1] << 16), rhi[2] | ((uint32_t)rhi[3] << 16));
            *reinterpret_cast<uint2*>(DGl + lr * PG + unit * 2) = make_uint2(rlo[0] | ((uint32_t)rlo[1] << 16), rlo[2] | ((uint32_t)rlo[3] << 16));
            *reinterpret_cast<uint2*>(DGh + lr * PG + (H + unit) * 2) = make_uint2(zhi[0] | ((uint32_t)zhi[1] << 16), zhi[2] | ((uint32_t)zhi[3] << 16));
            *reinterpret_cast<uint2*>(DGl + lr * PG + (H + unit) * 2) = make_uint2(zlo[0] | ((uint32_t)zlo[1] << 16), zlo[2] | ((uint32_t)zlo[3] << 16));
            if (live) {
              float* dg = a.dag_all + (((size_t)l * L + t) * M + sess) * 2 * H;
              *reinterpret_cast<float4*>(dg + unit) = make_float4(dagr[0], dagr[1], dagr[2], dagr[3]);
              *reinterpret_cast<float4*>(dg + H + unit) = make_float4(dagz[bi][4 * q], dagz[bi][4 * q + 1], dagz[bi][4 * q + 2], dagz[bi][4 * q + 3]);
            }
          }
        }
      }
      __syncthreads();   // [da_r | da_z] of every unit is in LDS
      if (has_blocks) {
        // ---------------- P3: [d x_g | d h_g] = [da_r | da_z] . Wg
#pragma unroll
        for (int bi = 0; bi < NBW; bi++) {
          const int blk = wave + 4 * bi;
          f32x16 acc[2];
#pragma unroll
          for (int part = 0; part < 2; part++) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) acc[part][reg] = 0.f;
            if (bi == 0 && part == 0) b_read(DGh, DGl, PG, 0, bqh[0], bql[0]);
#pragma unroll
            for (int ks = 0; ks < KSG; ks++) LB_KSTEP(acc[part], 2 * NBW * KSC + (bi * 2 + part) * KSG + ks, ks, KSG, DGh, DGl, PG, !(bi == NBW - 1 && part == 1))
          }
#pragma unroll
          for (int q = 0; q < 4; q++) {
            const int unit = 32 * blk + 8 * q + 4 * lh;
            float dx[4];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              const int e = 4 * q + j;
              dx[j] = dxc[bi][e] + acc[0][e];                                   // zero for a finished session (its da rows are zero)
              dinu[bi][e] = dx[j];
              dcar[lq][bi][e] = active ? dhp[bi][e] + acc[1][e] : dcar[lq][bi][e];
            }
            if (l == 0 && live) *reinterpret_cast<float4*>(a.du + (sess * L + t) * H + unit) = make_float4(dx[0], dx[1], dx[2], dx[3]);
          }
        }
      }
    }
  }
#undef LB_KSTEP
}

static size_t lt_bwd_lds_bytes(const ModelDev& m) { return 2 * 32 * (2 * (size_t)m.H + 16) + 2 * 32 * (4 * (size_t)m.H + 16); }

hipError_t launch_lgru_tile_bwd(const ModelDev& m, const float* dH, const int32_t* seq_len, const float* states, const float* rec, int64_t n_sess, int L, float* du,
                                float* dac_all, float* dag_all, hipStream_t st) {
  LgruTileBwdArgs a;
  memset(&a, 0, sizeof(a));
  a.dH = dH; a.seq_len = seq_len; a.states = states; a.rec = rec; a.du = du; a.dac_all = dac_all; a.dag_all = dag_all;
  for (int l = 0; l < m.nl; l++) {
    a.wcT_hi[l] = m.lgru_wcTF_hi[l]; a.wcT_lo[l] = m.lgru_wcTF_lo[l]; a.wgT_hi[l] = m.lgru_wgTF_hi[l]; a.wgT_lo[l] = m.lgru_wgTF_lo[l];
  }
  a.M = (int)n_sess; a.L = L; a.H = m.H; a.HT = m.HT; a.nl = m.nl;
  const size_t lds = lt_bwd_lds_bytes(m);
  const unsigned blocks = (unsigned)((n_sess + 31) / 32);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(LT_THREADS), lds, st, a);
    return hipGetLastError();
  };
  switch (m.H) {
    case 64: return go(k_lgru_tile_bwd<2, 2>);
    case 128: return go(k_lgru_tile_bwd<4, 2>);
    case 256: return go(k_lgru_tile_bwd<8, 2>);
    default: return hipErrorInvalidValue;
  }
}

// W [N][K] (16-bit, row-major) -> fragment-major copy of its TRANSPOSE [K][N]: WfT[((kb (N / 16) + s) 64 + lane) 8 + j] = W[16 s + 8 (lane >> 5) + j][32 kb + (lane & 31)]
__global__ void k_frag_major_T(const uint16_t* __restrict__ W, uint16_t* __restrict__ WfT, int N, int K) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)(K >> 5) * (N >> 4) * 64;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  const int64_t f = i >> 6;
  const int s = (int)(f % (N >> 4)), kb = (int)(f / (N >> 4));
  uint16_t v[8];
#pragma unroll
  for (int j = 0; j < 8; j++) v[j] = W[(int64_t)(16 * s + 8 * (lane >> 5) + j) * K + 32 * kb + (lane & 31)];
  *reinterpret_cast<uint4*>(WfT + i * 8) = make_uint4(v[0] | ((uint32_t)v[1] << 16), v[2] | ((uint32_t)v[3] << 16), v[4] | ((uint32_t)v[5] << 16), v[6] | ((uint32_t)v[7] << 16));
}

hipError_t launch_frag_major_T(const uint16_t* W, uint16_t* WfT, int N, int K, hipStream_t st) {
  const int64_t total = (int64_t)(K >> 5) * (N >> 4) * 64;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_frag_major_T, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, WfT, N, K);
  return hipGetLastError();
}
