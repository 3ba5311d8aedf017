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

#define SC_THREADS 512
#define SC_STAGE 64  // items per LDS stage (two 32-item MFMA row blocks)

template <int C>
struct ScGeom {
  static constexpr int KS = C / 16;
  static constexpr int CH = C / 8;
  static constexpr int SWZ = (CH < 16 ? CH : 16) - 1;
  static constexpr int STAGE_BYTES = SC_STAGE * 2 * C;
  static constexpr int CHUNKS_PER_THREAD = (SC_STAGE * CH) / SC_THREADS;  // 16-byte chunks each thread stages
  static_assert((SC_STAGE * CH) % SC_THREADS == 0, "stage must split evenly");
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

struct RowState {
  float t;    // target logit
  float m2;   // reference point of the running softmax sum, in log2 units (>= max logit * log2(e) - SC_DEFER)
  float s;    // sum of 2^(z*log2(e) - m2)
  int cnt;    // #{z > t}
};

#define SC_DEFER 8.0f  // rescale only when a logit exceeds the reference by 2^8 (T13): exact in fp32, rarely taken

__device__ __forceinline__ void epilogue(const f32x16& v, RowState& st) {
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
#pragma unroll
  for (int i = 0; i < 16; i++) {
    st.cnt += (v[i] > st.t) ? 1 : 0;
    st.s += __builtin_amdgcn_exp2f(x[i]);
  }
}

template <int C, bool LOGITS>
__global__ __launch_bounds__(SC_THREADS, 2) void k_score(const uint16_t* __restrict__ hidden, const int32_t* __restrict__ y,
                                                         int64_t R, const uint16_t* __restrict__ outT,
                                                         const float* __restrict__ out_bias, int N, int NP,
                                                         int32_t* __restrict__ o_rank, float* __restrict__ o_ce,
                                                         float* __restrict__ o_tgt, float* __restrict__ logits) {
  using G = ScGeom<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 stages + 2 x 64 bias floats
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t row0 = ((int64_t)blockIdx.x * 8 + wave) * 64;
  // Waves w and w+4 share a SIMD.  Waves 4-7 run half a tile behind waves 0-3 (they defer the epilogue of a
  // stage's second item block past the barrier), so that on every SIMD one wave's MFMA burst runs beside the
  // other's VALU epilogue instead of both bursts colliding after each barrier (MI355X_MICROARCH.md, two waves
  // per SIMD, item 9).
  const bool late = !LOGITS && wave >= 4;

  // ---- this wave's 64 steps as the B operand, resident for the whole kernel -------------------
  uint4 hf[2][G::KS];
  int64_t myrow[2];
  int myy[2];
#pragma unroll
  for (int cbk = 0; cbk < 2; cbk++) {
    int64_t r = row0 + cbk * 32 + lr;
    myrow[cbk] = r;
    const int64_t rc = (r < R) ? r : (R - 1);
    int yy = (LOGITS || r >= R) ? 0 : y[rc];
    myy[cbk] = (yy > 0 && yy < N) ? yy : 0;
    const uint16_t* hp = hidden + rc * C + 8 * lh;
#pragma unroll
    for (int s = 0; s < G::KS; s++) hf[cbk][s] = *reinterpret_cast<const uint4*>(hp + 16 * s);
  }

  RowState st[2];
  if (!LOGITS) {
    // ---- target logits: D[i][j] = bo[y_i] + Wo^T[y_i] . h_j, keep the diagonal ----------------
#pragma unroll
    for (int cbk = 0; cbk < 2; cbk++) {
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
        acc1[4 * q + j] = b4[j];
      }
    }
    const unsigned char* arow = smem + buf * G::STAGE_BYTES + (ib * 32 + lr) * (2 * C);
#pragma unroll
    for (int s = 0; s < G::KS; s++) {
      const uint4 a = *reinterpret_cast<const uint4*>(arow + (((2 * s + lh) ^ (lr & G::SWZ)) << 4));
      acc0 = mfma32(a, hf[0][s], acc0);
      acc1 = mfma32(a, hf[1][s], acc1);
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
          if (myrow[0] < R) logits[myrow[0] * N + it + j] = acc0[4 * q + j];
          if (myrow[1] < R) logits[myrow[1] * N + it + j] = acc1[4 * q + j];
        }
      }
    }
  };

  stage_issue(0, 0);
  __syncthreads();  // (hipcc drains vmcnt(0) here because an LDS-DMA is in flight)

  for (int stg = 0; stg < n_stages; stg++) {
    const int buf = stg & 1;
    // buf^1 was last read (by MFMA operand / bias reads) before the barrier that ended stage stg-1
    if (stg + 1 < n_stages) stage_issue(stg + 1, buf ^ 1);
    if (late && stg > 0) {  // deferred epilogue of the previous stage's second item block
      epilogue(acc0, st[0]);
      epilogue(acc1, st[1]);
    }
    mfma_tile(buf, 0);
    if (LOGITS) {
      store_logits(stg, 0);
    } else {
      epilogue(acc0, st[0]);
      epilogue(acc1, st[1]);
    }
    mfma_tile(buf, 1);
    if (LOGITS) {
      store_logits(stg, 1);
    } else if (!late) {
      epilogue(acc0, st[0]);
      epilogue(acc1, st[1]);
    }
    __syncthreads();
  }
  if (late) {
    epilogue(acc0, st[0]);
    epilogue(acc1, st[1]);
  }

  if (!LOGITS) {
#pragma unroll
    for (int cbk = 0; cbk < 2; cbk++) {
      // merge the two half-waves (same step, disjoint item subsets)
      const float m_o = __shfl_xor(st[cbk].m2, 32);
      const float s_o = __shfl_xor(st[cbk].s, 32);
      const int c_o = __shfl_xor(st[cbk].cnt, 32);
      const float M2 = fmaxf(st[cbk].m2, m_o);
      const float S = st[cbk].s * __builtin_amdgcn_exp2f(st[cbk].m2 - M2) + s_o * __builtin_amdgcn_exp2f(m_o - M2);
      const int cnt = st[cbk].cnt + c_o;
      if (lh == 0 && myrow[cbk] < R) {
        const bool valid = myy[cbk] > 0;
        const float lse = (M2 + log2f(S)) * LN2_F;
        if (o_rank) o_rank[myrow[cbk]] = valid ? cnt : 0;
        if (o_ce) o_ce[myrow[cbk]] = valid ? (lse - st[cbk].t) : 0.f;
        if (o_tgt) o_tgt[myrow[cbk]] = valid ? st[cbk].t : 0.f;
      }
    }
  }
}

template <int C>
static hipError_t launch_score_c(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R,
                                 const hiertcn_score_out_t* out, float* logits, hipStream_t st) {
  using G = ScGeom<C>;
  const size_t lds = 2 * G::STAGE_BYTES + 2 * SC_STAGE * sizeof(float);
  const unsigned blocks = (unsigned)((R + 511) / 512);
  hipError_t e;
  if (logits) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_score<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_score<C, true>), dim3(blocks), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP,
                       (int32_t*)nullptr, (float*)nullptr, (float*)nullptr, logits);
  } else {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_score<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL((k_score<C, false>), dim3(blocks), dim3(SC_THREADS), lds, st, hidden, y, R, m.outT, m.out_bias, m.N, m.NP,
                       out->rank, out->ce, out->target, (float*)nullptr);
  }
  return hipGetLastError();
}

hipError_t launch_score(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                        float* logits, hipStream_t st) {
  if (R <= 0) return hipSuccess;
  switch (m.Clast) {
    case 64: return launch_score_c<64>(m, hidden, y, R, out, logits, st);
    case 128: return launch_score_c<128>(m, hidden, y, R, out, logits, st);
    case 256: return launch_score_c<256>(m, hidden, y, R, out, logits, st);
    default: return hipErrorInvalidValue;
  }
}
