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

struct RowState {
  float t;    // target logit
  float m2;   // running max * log2(e)
  float m;    // running max
  float s;    // sum of exp(z - m)
  int cnt;    // #{z > t}
};

__device__ __forceinline__ void epilogue(const f32x16& v, RowState& st) {
  float tm = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
  tm = fmaxf(tm, fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
  tm = fmaxf(tm, fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])));
  tm = fmaxf(tm, fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15])));
  if (__any(tm > st.m)) {  // wave-uniform, rare once the running max has settled
    const float mn = fmaxf(st.m, tm);
    st.s *= __builtin_amdgcn_exp2f((st.m - mn) * LOG2E_F);
    st.m = mn;
    st.m2 = mn * LOG2E_F;
  }
#pragma unroll
  for (int i = 0; i < 16; i++) {
    st.cnt += (v[i] > st.t) ? 1 : 0;
    st.s += __builtin_amdgcn_exp2f(fmaf(v[i], LOG2E_F, -st.m2));
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
  float* sbias = reinterpret_cast<float*>(smem + 2 * G::STAGE_BYTES);
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t row0 = ((int64_t)blockIdx.x * 8 + wave) * 64;

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
      st[cbk].m = t;
      st[cbk].m2 = t * LOG2E_F;
      st[cbk].s = 0.f;
      st[cbk].cnt = 0;
    }
  }

  // ---- stream the item table ------------------------------------------------------------------
  const int n_stages = NP / SC_STAGE;
  uint4 pre[G::CHUNKS_PER_THREAD];
  float pre_b = 0.f;
  auto stage_load = [&](int stg) {
    const uint16_t* src = outT + (int64_t)stg * SC_STAGE * C;
#pragma unroll
    for (int i = 0; i < G::CHUNKS_PER_THREAD; i++) {
      const int id = tid + i * SC_THREADS;
      pre[i] = *reinterpret_cast<const uint4*>(src + (int64_t)id * 8);
    }
    if (tid < SC_STAGE) pre_b = out_bias[stg * SC_STAGE + tid];
  };
  auto stage_write = [&](int buf) {
    unsigned char* dst = smem + buf * G::STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < G::CHUNKS_PER_THREAD; i++) {
      const int id = tid + i * SC_THREADS;
      const int row = id / G::CH, ch = id % G::CH;
      *reinterpret_cast<uint4*>(dst + row * (2 * C) + ((ch ^ (row & G::SWZ)) << 4)) = pre[i];
    }
    if (tid < SC_STAGE) sbias[buf * SC_STAGE + tid] = pre_b;
  };

  stage_load(0);
  stage_write(0);
  __syncthreads();

  for (int stg = 0; stg < n_stages; stg++) {
    const int buf = stg & 1;
    if (stg + 1 < n_stages) stage_load(stg + 1);  // global loads in flight under the MFMAs (T14)
    const unsigned char* tile = smem + buf * G::STAGE_BYTES;
#pragma unroll
    for (int ib = 0; ib < 2; ib++) {
      f32x16 acc0, acc1;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + buf * SC_STAGE + ib * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int j = 0; j < 4; j++) {
          acc0[4 * q + j] = b4[j];
          acc1[4 * q + j] = b4[j];
        }
      }
      const unsigned char* arow = tile + (ib * 32 + lr) * (2 * C);
#pragma unroll
      for (int s = 0; s < G::KS; s++) {
        const uint4 a = *reinterpret_cast<const uint4*>(arow + (((2 * s + lh) ^ (lr & G::SWZ)) << 4));
        acc0 = mfma32(a, hf[0][s], acc0);
        acc1 = mfma32(a, hf[1][s], acc1);
      }
      if (LOGITS) {
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
      } else {
        epilogue(acc0, st[0]);
        epilogue(acc1, st[1]);
      }
    }
    if (stg + 1 < n_stages) stage_write(buf ^ 1);
    __syncthreads();
  }

  if (!LOGITS) {
#pragma unroll
    for (int cbk = 0; cbk < 2; cbk++) {
      // merge the two half-waves (same step, disjoint item subsets)
      const float m_o = __shfl_xor(st[cbk].m, 32);
      const float s_o = __shfl_xor(st[cbk].s, 32);
      const int c_o = __shfl_xor(st[cbk].cnt, 32);
      const float M = fmaxf(st[cbk].m, m_o);
      const float S = st[cbk].s * __builtin_amdgcn_exp2f((st[cbk].m - M) * LOG2E_F) +
                      s_o * __builtin_amdgcn_exp2f((m_o - M) * LOG2E_F);
      const int cnt = st[cbk].cnt + c_o;
      if (lh == 0 && myrow[cbk] < R) {
        const bool valid = myy[cbk] > 0;
        const float lse = M + logf(S);
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
