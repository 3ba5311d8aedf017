// HierGRU low level of small batches at H <= 128 (the reference's default width, args.py:51-53), LAYER-PIPELINED with REGISTER-RESIDENT weights
// (round 4; model_gru.py:47-55: dynamic_rnn of a MultiRNNCell of GRUCells from a zero state, with sequence_length; cell arithmetic
// customed_gru_cell.py:309-337).
//
// lgru_tile.hip takes 32 sessions through all steps AND layers in one workgroup and streams each cell's weights (393 KB at H = 128, hi + lo
// parts) from the L2 into a register ring: a k-step then takes ~280 cycles against the 96 of its three MFMAs, and a 320-session batch is 10
// workgroups of 40 sequential cells = 300 us.  Here a workgroup is one (32-session tile, LAYER): 6 H^2 weights x 4 bytes = 393 KB is 384
// registers per lane of the four waves -- the layer's r, z and candidate tiles are loaded ONCE and stay in the 512-register file (one wave
// per SIMD) for all L steps, so a cell is its MFMAs and its epilogue, and layer l + 1 of step t runs beside layer l of step t + 1 on another CU.
// Hand-off between the layers of a tile: the producer copies its new state's split 16-bit rows (what the next layer's MFMA B operand is)
// from LDS to a per-step slot of a global exchange buffer with 16-byte `sc1` stores, every wave waits for its stores (vmcnt(0)), a workgroup
// barrier, then one lane sets the step's flag (`sc1`); the consumer's one lane polls the flag (`sc1` load, bounded), a workgroup barrier, then
// every wave loads the slot with `sc1` loads (MI355X_MICROARCH.md, inter-workgroup visibility: the drained-`sc1` form).  A slot per step: no
// back-pressure, the producer never waits.  The flags are zeroed by a memset node in front of every launch (graph replays included).
// Every wait is bounded: a producer that never arrives (the grid is at most 256 workgroups, producers first in dispatch order, so on an
// otherwise idle device all of them are resident) sets the handle's persistent-launch error word and the launch ends with wrong numbers.
// Same arithmetic as lgru_tile.hip (split operands a.w = a_hi w_hi + a_lo w_hi + a_hi w_lo, fp32 accumulation in the same k order, fp32 state); the
// outputs agree with that kernel's to the last bit except where the compiler contracts the epilogue's blend differently (a handful of 16-bit ulps per
// 24,192 outputs measured).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

#define LP_THREADS 256
#define LP_SPIN_MAX (1u << 22)

struct LgruPipeArgs {
  const uint16_t* u16;                       // [M][L][E] projected [e, state] rows, storage type
  const int32_t* seq_len;                    // [M]
  const uint16_t* wgF_hi[HIERTCN_MAX_LAYERS]; const uint16_t* wgF_lo[HIERTCN_MAX_LAYERS];   // [2H/32][K/16][64][8]
  const uint16_t* wcF_hi[HIERTCN_MAX_LAYERS]; const uint16_t* wcF_lo[HIERTCN_MAX_LAYERS];   // [H/32][K/16][64][8]
  const float* bg[HIERTCN_MAX_LAYERS]; const float* bc[HIERTCN_MAX_LAYERS];
  uint16_t* hidden;                          // [M * L][H]: the top layer's outputs, zero past the length
  float* states;                             // training: [(L + 1)][M][HT] (slot 0 zeroed by the caller); null in evaluation
  float* rec;                                // training: per (t, l): gates [M][2H] | r*h [M][H] | cand [M][H] | out [M][H]
  uint16_t* xch;                             // [nl - 1][L][2 (hi, lo)][M][H]: layer l's state rows of step t, for layer l + 1
  int* flags;                                // [nl - 1][tiles][L]
  unsigned long long* err;                   // the handle's persistent-launch error word
  int M, L, H, HT, nl;
  int test_no_flags;                         // test-hooks build only (fault injection: the producers never signal)
};

__device__ __forceinline__ float lp_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
__device__ __forceinline__ float lp_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(2.8853900817779268f * x)); }
__device__ __forceinline__ void lp_split(float v, uint16_t& hi, uint16_t& lo) {
  hi = f2bf(v);
  lo = f2bf(v - bf2f(hi));
}
typedef unsigned lp_u4 __attribute__((ext_vector_type(4)));   // (inline asm takes native vectors, not HIP's uint4 struct)
__device__ __forceinline__ void lp_st_sc1(void* p, uint4 v) {
  const lp_u4 x = {v.x, v.y, v.z, v.w};
  // (the wait states behind it: a store of more than 64 bits reads its data registers AFTER issue, and the compiler, which sees no store here, may
  // reuse them at once -- without the s_nop the first 8 bytes of the last lanes' chunks arrived as the next address computation's bits)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" ::"v"(p), "v"(x) : "memory");
}

template <int HB>   // H / 32 in {2, 4} (E == H): one unit block per wave
__global__ __launch_bounds__(LP_THREADS, 1) void k_lgru_pipe(const LgruPipeArgs a) {
  constexpr int H = 32 * HB, KS = 4 * HB;          // k-steps of a product: (E + H) / 16 with E == H
  constexpr int KSI = KS / 2;                      // ... of which the input half
  constexpr int PITCH = 2 * H + 16;                // bytes per session row of a split activation buffer (rows 4 banks apart)
  static_assert(HB == 2 || HB == 4, "E == H in {64, 128}: a layer's weights fit the register file");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ int s_dead;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int nl = a.nl, L = a.L, M = a.M, HT = a.HT;
  const int l = (int)blockIdx.y, tile = (int)blockIdx.x, tiles = (int)gridDim.x;
  const int64_t sess0 = (int64_t)tile * 32;
  const int64_t sess = sess0 + lr;
  const bool live = sess < M;
  const int64_t sc = live ? sess : (int64_t)M - 1;
  const int len = a.seq_len[sc];
  const bool has_block = wave < HB;                // (H = 64: two of the four waves own no unit block; they help with the row traffic only)
  const int blk = wave;
  const bool first = l == 0, last = l == nl - 1;

  // ---- LDS carve-up
  unsigned char* XH = smem;                        // [2][32][PITCH]: this layer's input rows of step t (hi parts; layer 0: the projected rows, no lo part)
  unsigned char* XL = XH + 2 * 32 * PITCH;         // [2][32][PITCH] lo parts (layers > 0)
  unsigned char* SH = XL + 2 * 32 * PITCH;         // [32][PITCH] state, hi parts
  unsigned char* SL = SH + 32 * PITCH;
  unsigned char* RHh = SL + 32 * PITCH;            // [32][PITCH] r * h
  unsigned char* RHl = RHh + 32 * PITCH;
  for (int i = tid; i < 2 * 32 * PITCH / 16; i += LP_THREADS) reinterpret_cast<uint4*>(SH)[i] = make_uint4(0, 0, 0, 0);   // zero_state
  if (tid == 0) s_dead = 0;

  // ---- the layer's weights, once: r, z and candidate tiles of this wave's 32 units, every k-step, hi and lo parts (24 KS registers)
  uint4 Wh[3][KS], Wl[3][KS];
  if (has_block) {
#pragma unroll
    for (int sel = 0; sel < 3; sel++) {
      const uint16_t* th = (sel < 2 ? a.wgF_hi[l] : a.wcF_hi[l]) + ((size_t)((sel == 1 ? HB : 0) + blk) * KS * 64 + lane) * 8;
      const uint16_t* tl = (sel < 2 ? a.wgF_lo[l] : a.wcF_lo[l]) + ((size_t)((sel == 1 ? HB : 0) + blk) * KS * 64 + lane) * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        Wh[sel][ks] = *reinterpret_cast<const uint4*>(th + (size_t)ks * 512);
        Wl[sel][ks] = *reinterpret_cast<const uint4*>(tl + (size_t)ks * 512);
      }
    }
    // opaque: these are loads of read-only memory, and hipcc would rather re-load a fragment in front of every use than keep 384 registers live
    // (measured: 25 us per cell, every k-step a bare L2 round trip)
#pragma unroll
    for (int sel = 0; sel < 3; sel++)
#pragma unroll
      for (int ks = 0; ks < KS; ks++) {
        asm volatile("" : "+v"(Wh[sel][ks].x), "+v"(Wh[sel][ks].y), "+v"(Wh[sel][ks].z), "+v"(Wh[sel][ks].w));
        asm volatile("" : "+v"(Wl[sel][ks].x), "+v"(Wl[sel][ks].y), "+v"(Wl[sel][ks].z), "+v"(Wl[sel][ks].w));
      }
  }
  const float* bg_r = a.bg[l] + 32 * blk; const float* bg_z = bg_r + H; const float* bc_c = a.bc[l] + 32 * blk;   // (re-read per cell: 48 registers otherwise)
  float hreg[16];   // this lane's units of its session (fp32 state)
#pragma unroll
  for (int i = 0; i < 16; i++) hreg[i] = 0.f;

  // ---- input rows of a step: 32 sessions x H 16-bit values = 32 x (H / 8) chunks of 16 bytes (x 2 parts behind layer 0)
  constexpr int UCH = 32 * (H / 8), UPT = (UCH + LP_THREADS - 1) / LP_THREADS;
  const size_t xslot = (size_t)2 * M * H;                                    // 16-bit elements of one (layer, step) slot: hi rows, then lo rows
  const uint16_t* xin = first ? nullptr : a.xch + (size_t)(l - 1) * L * xslot;
  uint16_t* xout = last ? nullptr : a.xch + (size_t)l * L * xslot;
  int* fin = first ? nullptr : a.flags + ((size_t)(l - 1) * tiles + tile) * L;
  int* fout = last ? nullptr : a.flags + ((size_t)l * tiles + tile) * L;
  uint4 ureg[UPT];
  auto u_fetch = [&](int t) {   // layer 0: plain loads, in flight during the step's cell
#pragma unroll
    for (int k = 0; k < UPT; k++) {
      const int c = tid + k * LP_THREADS;
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
      const int c = tid + k * LP_THREADS;
      if (c < UCH) {
        const int row = c / (H / 8), ch = c % (H / 8);
        *reinterpret_cast<uint4*>(XH + (buf * 32 + row) * PITCH + ch * 16) = ureg[k];
      }
    }
  };
  // layers > 0: wait for the producer's flag of step t (one lane, bounded), then every wave loads the slot with sc1 loads into LDS buffer `buf`.
  // Contains two workgroup barriers; the first also closes the cell before it.
  auto x_take = [&](int t, int buf) {
    if (tid == 0) {
      unsigned spins = 0;
      while (!s_dead && __hip_atomic_load(fin + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > LP_SPIN_MAX) {   // never in a healthy run: leave a mark and go on rather than hang the device
          __hip_atomic_store(a.err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          s_dead = 1;
        }
      }
    }
    __syncthreads();
    const uint16_t* slot = xin + (size_t)t * xslot;
    lp_u4 vh[UPT], vl[UPT];
    const uint16_t* ph[UPT];
#pragma unroll
    for (int k = 0; k < UPT; k++) {
      const int c = min(tid + k * LP_THREADS, UCH - 1);
      const int row = c / (H / 8), ch = c % (H / 8);
      const int64_t s = (sess0 + row < M) ? sess0 + row : (int64_t)M - 1;
      ph[k] = slot + (size_t)s * H + ch * 8;
    }
    // loads AND their wait in ONE asm statement: under this kernel's register pressure the compiler may move an asm output (to an AGPR) right behind
    // the statement that "produced" it -- before the data of a separately waited-for load has arrived
    static_assert(UPT <= 2, "the statement below is written for one or two chunks per thread");
    const uint16_t* pl0 = ph[0] + (size_t)M * H;
    if constexpr (UPT == 1) {
      asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(vh[0]), "=&v"(vl[0]) : "v"(ph[0]), "v"(pl0) : "memory");
    } else {
      const uint16_t* pl1 = ph[UPT - 1] + (size_t)M * H;
      asm volatile("global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\tglobal_load_dwordx4 %2, %6, off sc1\n\t"
                   "global_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(vh[0]), "=&v"(vl[0]), "=&v"(vh[UPT - 1]), "=&v"(vl[UPT - 1]) : "v"(ph[0]), "v"(pl0), "v"(ph[UPT - 1]), "v"(pl1) : "memory");
    }
#pragma unroll
    for (int k = 0; k < UPT; k++) {
      const int c = tid + k * LP_THREADS;
      if (c < UCH) {
        const int row = c / (H / 8), ch = c % (H / 8);
        *reinterpret_cast<lp_u4*>(XH + (buf * 32 + row) * PITCH + ch * 16) = vh[k];
        *reinterpret_cast<lp_u4*>(XL + (buf * 32 + row) * PITCH + ch * 16) = vl[k];
      }
    }
    __syncthreads();
  };
  // layers below the top: the new state's rows (LDS, complete behind the cell's closing barrier) to the step's slot, then the flag
  auto x_give = [&](int t) {
    uint16_t* slot = xout + (size_t)t * xslot;
#pragma unroll
    for (int k = 0; k < UPT; k++) {
      const int c = tid + k * LP_THREADS;
      if (c < UCH) {
        const int row = c / (H / 8), ch = c % (H / 8);
        if (sess0 + row < M) {
          lp_st_sc1(slot + (size_t)(sess0 + row) * H + ch * 8, *reinterpret_cast<const uint4*>(SH + row * PITCH + ch * 16));
          lp_st_sc1(slot + (size_t)M * H + (size_t)(sess0 + row) * H + ch * 8, *reinterpret_cast<const uint4*>(SL + row * PITCH + ch * 16));
        }
      }
    }
    // (no wait here: the step's flag is set behind the NEXT cell's mid-cell barrier, in front of which every wave waits for its stores -- by then
    // they have long completed; waiting here cost the producer ~2 us per cell)
  };
  // prefetch of the next step's slot by a consumer (flag already seen): compiler-visible agent-scope loads (8 bytes each: `global_load_dwordx2 sc1`),
  // in flight during the candidate phase
  unsigned long long xr[4 * UPT];
  auto x_prefetch = [&](int t) {
    const uint16_t* slot = xin + (size_t)t * xslot;
#pragma unroll
    for (int k = 0; k < UPT; k++) {
      const int c = min(tid + k * LP_THREADS, UCH - 1);
      const int row = c / (H / 8), ch = c % (H / 8);
      const int64_t s = (sess0 + row < M) ? sess0 + row : (int64_t)M - 1;
      const unsigned long long* ph = reinterpret_cast<const unsigned long long*>(slot + (size_t)s * H + ch * 8);
      const unsigned long long* pl = reinterpret_cast<const unsigned long long*>(slot + (size_t)M * H + (size_t)s * H + ch * 8);
      xr[4 * k + 0] = __hip_atomic_load(ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      xr[4 * k + 1] = __hip_atomic_load(ph + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      xr[4 * k + 2] = __hip_atomic_load(pl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      xr[4 * k + 3] = __hip_atomic_load(pl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };
  auto x_prefetch_store = [&](int buf) {
#pragma unroll
    for (int k = 0; k < UPT; k++) {
      const int c = tid + k * LP_THREADS;
      if (c < UCH) {
        const int row = c / (H / 8), ch = c % (H / 8);
        *reinterpret_cast<unsigned long long*>(XH + (buf * 32 + row) * PITCH + ch * 16) = xr[4 * k + 0];
        *reinterpret_cast<unsigned long long*>(XH + (buf * 32 + row) * PITCH + ch * 16 + 8) = xr[4 * k + 1];
        *reinterpret_cast<unsigned long long*>(XL + (buf * 32 + row) * PITCH + ch * 16) = xr[4 * k + 2];
        *reinterpret_cast<unsigned long long*>(XL + (buf * 32 + row) * PITCH + ch * 16 + 8) = xr[4 * k + 3];
      }
    }
  };

  __syncthreads();
  if (first) {
    u_fetch(0);
    u_store(0);
    __syncthreads();
  } else {
    x_take(0, 0);
  }

  const size_t cellf = (size_t)M * 5 * H;   // floats per (t, l) record
  __shared__ int s_next;                    // consumer: the next step's flag as seen at the start of this cell
  auto run = [&](auto first_c) {            // (the layer's position at compile time: no branches in the unrolled k-loops)
  constexpr bool FIRST = decltype(first_c)::value;
  for (int t = 0; t < L; t++) {
    if (FIRST && t + 1 < L) u_fetch(t + 1);          // in flight during the cell
    if (!FIRST && tid == 0) s_next = (t + 1 < L) ? __hip_atomic_load(fin + t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    const bool active = t < len;
    const unsigned char* Xh = XH + (t & 1) * 32 * PITCH;
    const unsigned char* Xl = FIRST ? nullptr : XL + (t & 1) * 32 * PITCH;
    float* rec = a.rec ? a.rec + ((size_t)t * nl + l) * cellf : nullptr;
    float zreg[16];
    uint4 bqh[2], bql[2];
    auto b_read = [&](const unsigned char* hp, const unsigned char* lp, int ks, uint4& oh, uint4& ol) {
      const int ko = ks * 32 + lh * 16;
      oh = *reinterpret_cast<const uint4*>(hp + lr * PITCH + ko);
      ol = lp ? *reinterpret_cast<const uint4*>(lp + lr * PITCH + ko) : make_uint4(0, 0, 0, 0);
    };
    // one product tile: k-steps 0 .. KSI-1 on the input rows, KSI .. KS-1 on (h2, l2); the B fragment of k-step ks + 1 is read one step ahead
#define LP_TILE(acc_, sel_, h2_, l2_)                                                                                  \
    {                                                                                                                   \
      b_read(Xh, Xl, 0, bqh[0], bql[0]);                                                                                \
      _Pragma("unroll") for (int ks = 0; ks < KS; ks++) {                                                               \
        if (ks + 1 < KS) b_read((ks + 1) < KSI ? Xh : (h2_), (ks + 1) < KSI ? Xl : (l2_), (ks + 1) % KSI, bqh[(ks + 1) & 1], bql[(ks + 1) & 1]); \
        acc_ = mfma32(Wh[sel_][ks], bqh[ks & 1], acc_);                                                                 \
        acc_ = mfma32(Wl[sel_][ks], bqh[ks & 1], acc_);                                                                 \
        if (ks >= KSI || !FIRST) acc_ = mfma32(Wh[sel_][ks], bql[ks & 1], acc_);                                        \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
      }                                                                                                                 \
    }
    if (has_block) {
      // ---------------- gates: the r tile, its epilogue, then the z tile of this wave's units (one accumulator tile live at a time: registers)
      f32x16 acc;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) acc[reg] = bg_r[acc_row(reg, lh)];
      LP_TILE(acc, 0, SH, SL)
      // epilogue (customed_gru_cell.py:323-326): r = sigmoid; r * h -> LDS (split) for the candidate's recurrent half
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float r4[4], rh4[4];
        uint16_t hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          r4[j] = lp_sigmoid(acc[4 * q + j]);
          rh4[j] = r4[j] * hreg[4 * q + j];
          lp_split(rh4[j], hi[j], lo[j]);
        }
        const int unit = 32 * blk + 8 * q + 4 * lh;
        *reinterpret_cast<uint2*>(RHh + lr * PITCH + unit * 2) = make_uint2(hi[0] | ((uint32_t)hi[1] << 16), hi[2] | ((uint32_t)hi[3] << 16));
        *reinterpret_cast<uint2*>(RHl + lr * PITCH + unit * 2) = make_uint2(lo[0] | ((uint32_t)lo[1] << 16), lo[2] | ((uint32_t)lo[3] << 16));
        if (rec && live) {
          *reinterpret_cast<float4*>(rec + sess * 2 * H + unit) = make_float4(r4[0], r4[1], r4[2], r4[3]);
          *reinterpret_cast<float4*>(rec + (size_t)M * 2 * H + sess * H + unit) = make_float4(rh4[0], rh4[1], rh4[2], rh4[3]);
        }
      }
#pragma unroll
      for (int reg = 0; reg < 16; reg++) acc[reg] = bg_z[acc_row(reg, lh)];
      LP_TILE(acc, 1, SH, SL)
#pragma unroll
      for (int i = 0; i < 16; i++) zreg[i] = lp_sigmoid(acc[i]);
      if (rec && live) {
#pragma unroll
        for (int q = 0; q < 4; q++)
          *reinterpret_cast<float4*>(rec + sess * 2 * H + H + 32 * blk + 8 * q + 4 * lh) = make_float4(zreg[4 * q], zreg[4 * q + 1], zreg[4 * q + 2], zreg[4 * q + 3]);
      }
    }
    if (!last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's stores of the previous step's slot (issued a gates phase ago) have completed
    __syncthreads();   // r * h of every unit is in LDS
#ifdef HIERTCN_TEST_HOOKS
    if (!a.test_no_flags)
#endif
    if (!last && tid == 0 && t > 0) __hip_atomic_store(fout + t - 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    bool prefetched = false;
    if (!FIRST) {
      prefetched = s_next != 0;   // (uniform: one LDS word, written before the barrier)
      if (prefetched) x_prefetch(t + 1);
    }
    if (has_block) {
      // ---------------- candidate (customed_gru_cell.py:333-336): c = tanh([x, r*h] Wc + b); h' = z h + (1 - z) c
      f32x16 acc;
#pragma unroll
      for (int reg = 0; reg < 16; reg++) acc[reg] = bc_c[acc_row(reg, lh)];
      LP_TILE(acc, 2, RHh, RHl)
      // dynamic_rnn past the session's length: state through, output zero (model_gru.py:47-53)
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float c4[4], hn[4], st4[4];
        uint16_t hi[4], lo[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const float h = hreg[4 * q + j], z = zreg[4 * q + j];
          c4[j] = lp_tanh(acc[4 * q + j]);
          hn[j] = z * h + (1.0f - z) * c4[j];
          st4[j] = active ? hn[j] : h;
          hreg[4 * q + j] = st4[j];
          lp_split(st4[j], hi[j], lo[j]);
        }
        const int unit = 32 * blk + 8 * q + 4 * lh;
        // the state's split is the h operand of step t + 1 and the next layer's input at this step (lgru_tile.hip: for a finished session the next
        // layer's result is discarded, so the carried state standing in for the cell output there changes nothing).  Nobody reads S in this phase.
        *reinterpret_cast<uint2*>(SH + lr * PITCH + unit * 2) = make_uint2(hi[0] | ((uint32_t)hi[1] << 16), hi[2] | ((uint32_t)hi[3] << 16));
        *reinterpret_cast<uint2*>(SL + lr * PITCH + unit * 2) = make_uint2(lo[0] | ((uint32_t)lo[1] << 16), lo[2] | ((uint32_t)lo[3] << 16));
        if (live) {
          if (rec) {
            *reinterpret_cast<float4*>(rec + (size_t)M * 3 * H + sess * H + unit) = make_float4(c4[0], c4[1], c4[2], c4[3]);
            *reinterpret_cast<float4*>(rec + (size_t)M * 4 * H + sess * H + unit) = make_float4(hn[0], hn[1], hn[2], hn[3]);
            *reinterpret_cast<float4*>(a.states + ((size_t)(t + 1) * M + sess) * HT + (size_t)l * H + unit) = make_float4(st4[0], st4[1], st4[2], st4[3]);
          }
          if (last) {
            uint2 o;
            o.x = active ? pack2bf(hn[0], hn[1]) : 0u;
            o.y = active ? pack2bf(hn[2], hn[3]) : 0u;
            *reinterpret_cast<uint2*>(a.hidden + (sess * L + t) * H + unit) = o;
          }
        }
      }
    }
#undef LP_TILE
    if (FIRST && t + 1 < L) u_store((t + 1) & 1);   // (the other buffer: nobody reads it during step t)
    if (!FIRST && prefetched) x_prefetch_store((t + 1) & 1);
    __syncthreads();   // the new state (and the next step's rows, where they were fetched beside the cell) are in LDS
    if (!last) x_give(t);                            // -> the layer above: stores only (reads S: its next writer is this layer's cell t + 1, behind the mid-cell barrier)
    if (!FIRST && !prefetched && t + 1 < L) x_take(t + 1, (t + 1) & 1);   // the producer was not ahead at the start of the cell: wait for it now
  }
  };
  if (first) run(std::true_type{}); else run(std::false_type{});
  if (!last) {   // the last step's slot
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef HIERTCN_TEST_HOOKS
    if (!a.test_no_flags)
#endif
    if (tid == 0) __hip_atomic_store(fout + L - 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

static size_t lp_lds_bytes(const ModelDev& m) {
  const size_t pitch = 2 * (size_t)m.H + 16;
  return 8 * 32 * pitch;   // input rows hi x 2, lo x 2 | state hi, lo | r*h hi, lo
}

bool lgru_pipe_applies(const ModelDev& m, int64_t n_sess) {
  if (m.tune.lgru_pipe == 0) return false;         // never (A/B runs, tests)
  if (m.tune.lgru_persistent == 0) return false;   // set by the shared-GPU guard (distributed.py): no launch that waits on other workgroups
  if (!lgru_tile_applies(m, n_sess)) return false;
  if (!(m.H == 64 || m.H == 128) || m.gru_bar == nullptr) return false;
  const int64_t tiles = (n_sess + 31) / 32;
  return tiles * m.nl <= (m.n_cu > 0 ? m.n_cu : 256);   // every workgroup resident at once (one per CU): a consumer waits for its producer inside the launch
}

// bytes of the exchange slots + flags (0: no pipelined launch for this model; sizes past its range count as the largest one in it)
size_t lgru_pipe_bytes(const ModelDev& m, int64_t n_sess, int L) {
  if (m.H == 256) return lgru_wide_bytes(m, n_sess, L);   // (lgru_wide.hip takes that width: the same scratch pointer carries its slots)
  if (n_sess <= 0 || !(m.H == 64 || m.H == 128) || m.E != m.H || m.nl > 2) return 0;
  if (n_sess > 32 * (256 / m.nl)) n_sess = 32 * (256 / m.nl);   // a workspace sized for a larger batch still serves the smaller ones the launch takes
  const size_t tiles = (size_t)(n_sess + 31) / 32;
  const size_t x = (size_t)(m.nl - 1) * L * 2 * (size_t)n_sess * m.H * 2;
  return ((x + 255) & ~(size_t)255) + ((tiles * m.nl * L * 4 + 255) & ~(size_t)255);
}

hipError_t launch_lgru_pipe(const ModelDev& m, const uint16_t* u16, const int32_t* seq_len, int64_t n_sess, int L, uint16_t* hidden, float* states, float* rec,
                            void* scratch, hipStream_t st) {
  LgruPipeArgs a;
  memset(&a, 0, sizeof(a));
  a.u16 = u16; a.seq_len = seq_len; a.hidden = hidden; a.states = states; a.rec = rec;
  for (int l = 0; l < m.nl; l++) {
    a.wgF_hi[l] = m.lgru_wgF_hi[l]; a.wgF_lo[l] = m.lgru_wgF_lo[l]; a.wcF_hi[l] = m.lgru_wcF_hi[l]; a.wcF_lo[l] = m.lgru_wcF_lo[l];
    a.bg[l] = m.lgru_bg[l]; a.bc[l] = m.lgru_bc[l];
  }
  a.M = (int)n_sess; a.L = L; a.H = m.H; a.HT = m.HT; a.nl = m.nl;
#ifdef HIERTCN_TEST_HOOKS   // libhiertcn_gfx950_testhooks.so only: the shipped libraries read no test switch on this path
  if (getenv("HIERTCN_LGRU_PIPE_TEST_NO_FLAGS")) a.test_no_flags = 1;   // every consumer's bounded wait runs out
#endif
  const unsigned tiles = (unsigned)((n_sess + 31) / 32);
  const size_t xbytes = (((size_t)(m.nl - 1) * L * 2 * (size_t)n_sess * m.H * 2) + 255) & ~(size_t)255;
  a.xch = reinterpret_cast<uint16_t*>(scratch);
  a.flags = reinterpret_cast<int*>(reinterpret_cast<unsigned char*>(scratch) + xbytes);
  a.err = reinterpret_cast<unsigned long long*>(m.gru_bar) + 8 * 16;   // (gru_persist.hip: the word behind the GP_GROUPS = 8 counters, 128 bytes apart)
  hipError_t e;
  if (m.nl > 1 && (e = hipMemsetAsync(a.flags, 0, (size_t)tiles * m.nl * L * 4, st)) != hipSuccess) return e;
  const size_t lds = lp_lds_bytes(m);
  auto go = [&](auto kern) -> hipError_t {
    hipError_t e2 = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e2 != hipSuccess) return e2;
    hipLaunchKernelGGL(kern, dim3(tiles, (unsigned)m.nl), dim3(LP_THREADS), lds, st, a);
    return hipGetLastError();
  };
  switch (m.H) {
    case 64: return go(k_lgru_pipe<2>);
    case 128: return go(k_lgru_pipe<4>);
    default: return hipErrorInvalidValue;
  }
}
