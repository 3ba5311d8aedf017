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
#include "score_geom.h"

typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint2 lds_read_tr16(const unsigned char* p) {
  const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
  return __builtin_bit_cast(uint2, v);
}

template <int C, bool ROWSTAT>
__global__ __launch_bounds__(SC_THREADS, 2) void k_sgrad(const uint16_t* __restrict__ stat, const float* __restrict__ stat_const,
                                                         int64_t n_stat_max, const int32_t* __restrict__ n_stat_dev,
                                                         const uint16_t* __restrict__ stream, const float* __restrict__ stream_bias,
                                                         int n_stages_max, const int32_t* __restrict__ n_stream_dev,
                                                         float* __restrict__ out, float* __restrict__ colsum) {
  using G = ScGeom<C>;
  static_assert(C == 128 || C == 256, "the dual-use LDS image is laid out for 256- and 512-byte rows");
  constexpr int NCB = C / 32;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 3 stages of 64 stream vectors + 3 x 64 stream scalars
  const int64_t n_stat = n_stat_dev ? (int64_t)n_stat_dev[0] : n_stat_max;
  if ((int64_t)blockIdx.x * 256 >= n_stat) return;
  int n_stages = n_stream_dev ? (int)(((int64_t)n_stream_dev[0] + SC_STAGE - 1) / SC_STAGE) : n_stages_max;
  if (n_stages > n_stages_max) n_stages = n_stages_max;
  int s_lo = 0;
  if (gridDim.y > 1) {   // the stream split in parts (ITEMSTAT: the item dimension alone gives N / 256 workgroups): partial sums added atomically
    const int per = (n_stages + (int)gridDim.y - 1) / (int)gridDim.y;
    s_lo = (int)blockIdx.y * per;
    const int s_hi = (s_lo + per < n_stages) ? s_lo + per : n_stages;
    if (s_hi <= s_lo) return;
    n_stages = s_hi - s_lo;
  }
  stream += (int64_t)s_lo * SC_STAGE * C;
  stream_bias += (int64_t)s_lo * SC_STAGE;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t sv = ((int64_t)blockIdx.x * 8 + wave) * 32 + lr;   // this lane's stationary vector
  const int64_t svc = sv < n_stat ? sv : n_stat - 1;   // (lanes past the end read the last live vector: finite data, their column is never stored)

  uint4 hf[G::KS];
  {
    const uint16_t* sp = stat + svc * C + 8 * lh;
#pragma unroll
    for (int s = 0; s < G::KS; s++) hf[s] = *reinterpret_cast<const uint4*>(sp + 16 * s);
  }
  const float cst = (sv < n_stat) ? stat_const[svc] : -INFINITY;   // (ln w - lse) log2e, or bo log2e; -inf: every p of this column is 0

  constexpr int PIECES = G::STAGE_BYTES / 1024;
  constexpr int PPW = PIECES / 8;
  constexpr int STAGE_LDS = G::STAGE_BYTES;
  constexpr int TILE_LDS = STAGE_LDS / 2;
  constexpr int BIAS_OFF = 3 * STAGE_LDS;
  const unsigned lds_base = (unsigned)(size_t)((__attribute__((address_space(3))) unsigned char*)smem);
  auto swz = [](int row) { return ((row & 3) << 2) | ((row >> 2) & 3); };
  auto dma_voff = [&](int i) -> unsigned {   // recomputed at every issue: registers are the scarce resource here, not VALU slots
    const int off = (wave * PPW + i) * 1024 + lane * 16;
    const int row = off / (2 * C), slot = (off % (2 * C)) >> 4;
    return (unsigned)(row * (2 * C) + ((slot ^ swz(row)) << 4));
  };
  auto stage_issue = [&](int stg, int buf) {   // LDS-DMA by hand (score.hip: why asm); SGPR stage base + 32-bit lane offset
    const uint16_t* sbase = stream + (int64_t)stg * SC_STAGE * C;
#pragma unroll
    for (int i = 0; i < PPW; i++) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + buf * STAGE_LDS + (wave * PPW + i) * 1024);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(dma_voff(i)), "s"(sbase), "s"(dst)
                   : "memory");
    }
    if (wave == 0 && lane < SC_STAGE / 4) {
      const float* src = stream_bias + (int64_t)stg * SC_STAGE + lane * 4;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + BIAS_OFF + buf * SC_STAGE * 4);
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep)
                   : "v"(src), "s"(dst)
                   : "memory");
    }
  };
  const float* sbias = reinterpret_cast<const float*>(smem + BIAS_OFF);

  // GEMM-1 A fragment (row read): stream row lr of the tile, 16-byte chunk 2s + lh
  const int a_k0 = lr * (2 * C) + ((lh ^ swz(lr)) << 4);
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
    tr_a[u] = row * (2 * C) + (tq << 6) + (low << 4) + 8 * (tp & 1);   // chunk' = 4 (cb ^ q) + low: the cb part is an XOR with cb << 6
  }

  f32x16 oacc[NCB];
#pragma unroll
  for (int cb = 0; cb < NCB; cb++)
#pragma unroll
    for (int i = 0; i < 16; i++) oacc[cb][i] = 0.f;
  float cs = 0.f;

  if (n_stages > 0) stage_issue(0, 0);
  if (n_stages > 1) stage_issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int stg = 0; stg < n_stages; stg++) {
    const int buf2 = (buf >= 1) ? buf - 1 : 2;   // (buf + 2) % 3
    if (stg + 2 < n_stages) stage_issue(stg + 2, buf2);
#pragma unroll
    for (int tile = 0; tile < 2; tile++) {
      const unsigned char* tl = smem + buf * STAGE_LDS + tile * TILE_LDS;
      // opaque copies: hipcc otherwise hoists the 16 + 16 XOR-ed fragment addresses out of the stage loop and spills them
      int ak = a_k0, tr0 = tr_a[0], tr1 = tr_a[1];
      asm volatile("" : "+v"(ak), "+v"(tr0), "+v"(tr1));
      f32x16 acc;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(sbias + buf * SC_STAGE + tile * 32 + 8 * q + 4 * lh);
#pragma unroll
        for (int j = 0; j < 4; j++) acc[4 * q + j] = b4[j];
      }
#pragma unroll
      for (int s = 0; s < G::KS; s++) {
        const uint4 a = *reinterpret_cast<const uint4*>(tl + (ak ^ (s << 5)));
        acc = mfma32(a, hf[s], acc);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the three parts of a tile apart: overlapping them costs registers this kernel does not have
      uint32_t pk[8];
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        const float e0 = __builtin_amdgcn_exp2f(fmaf(acc[i], LOG2E_F, cst));
        const float e1 = __builtin_amdgcn_exp2f(fmaf(acc[i + 1], LOG2E_F, cst));
        if (!ROWSTAT) cs += e0 + e1;
        pk[i >> 1] = pack2bf(e0, e1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s2 = 0; s2 < 2; s2++) {
        const uint4 bfrag = make_uint4(pk[4 * s2], pk[4 * s2 + 1], pk[4 * s2 + 2], pk[4 * s2 + 3]);
#pragma unroll
        for (int cb = 0; cb < NCB; cb++) {
          const uint2 lo = lds_read_tr16(tl + ((tr0 ^ (cb << 6)) + s2 * 32 * C));
          const uint2 hi = lds_read_tr16(tl + ((tr1 ^ (cb << 6)) + s2 * 32 * C));
          oacc[cb] = mfma32(make_uint4(lo.x, lo.y, hi.x, hi.y), bfrag, oacc[cb]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    buf = (buf == 2) ? 0 : buf + 1;
  }

  // out^T[c][stationary]: register i of tile cb is channel 32 cb + (i & 3) + 8 (i >> 2) + 4 lh of this lane's stationary vector
  if (sv < n_stat_max && (ROWSTAT ? sv < n_stat : true)) {
    float* op = out + sv * C;
#pragma unroll
    for (int cb = 0; cb < NCB; cb++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        float* d = op + 32 * cb + 8 * q + 4 * lh;
        if (gridDim.y > 1) {
#pragma unroll
          for (int j = 0; j < 4; j++) atomicAdd(d + j, oacc[cb][4 * q + j]);
        } else {
          *reinterpret_cast<float4*>(d) = make_float4(oacc[cb][4 * q], oacc[cb][4 * q + 1], oacc[cb][4 * q + 2], oacc[cb][4 * q + 3]);
        }
      }
    }
  }
  if (!ROWSTAT && colsum) {
    const float t = cs + __shfl_xor(cs, 32);
    if (lh == 0 && sv < n_stat) atomicAdd(colsum + sv, t);
  }
}

// compact scored rows: hc[p] = hidden[map[p]], yc, (ln w - lse) log2e, w for p < Rv; zeros / -inf up to the next multiple of 64 (the
// padding rows of the last stage are read as operands).  Grid sized by R (host bound); Rv is read on the device.
__global__ void k_sgrad_gather(const uint16_t* __restrict__ hidden, const int32_t* __restrict__ y, const float* __restrict__ w,
                               const float* __restrict__ lse, const int32_t* __restrict__ map, const int32_t* __restrict__ n_dev, int C,
                               uint16_t* __restrict__ hc, int32_t* __restrict__ yc, float* __restrict__ wc, float* __restrict__ rb2) {
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
    rb2[p] = wv > 0.f ? logf(wv) - lse[r] : -INFINITY;   // natural units: the stream scalar of ITEMSTAT; ROWSTAT scales it by log2e on load
  }
}
__global__ void k_scale_log2e(const float* __restrict__ src, float* __restrict__ dst, int64_t n_max, const int32_t* __restrict__ n_dev, int pad64) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t n = n_dev ? (int64_t)n_dev[0] : n_max;
  if (pad64) n = (n + 63) & ~(int64_t)63;
  if (i < n && i < n_max) dst[i] = src[i] * LOG2E_F;
}

// d hidden[map[p]][:] = dHc[p][:] - w_p Wo[y_p][:]   (the one-hot term of g; dH zeroed beforehand: padded steps have no gradient)
__global__ void k_dh_finish(const float* __restrict__ dhc, const int32_t* __restrict__ map, const int32_t* __restrict__ n_dev,
                            const int32_t* __restrict__ yc, const float* __restrict__ wc, const uint16_t* __restrict__ outT, int C, int N,
                            float* __restrict__ dh) {
  const int ch = C >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t p = i / ch;
  if (p >= (int64_t)n_dev[0]) return;
  const int c = (int)(i - p * ch) * 4;
  float4 v = *reinterpret_cast<const float4*>(dhc + p * C + c);
  const int yv = yc[p];
  if (yv > 0 && yv < N) {
    const uint2 wr = *reinterpret_cast<const uint2*>(outT + (int64_t)yv * C + c);
    const float w = wc[p];
    v.x -= w * bf2f((uint16_t)(wr.x & 0xffff)); v.y -= w * bf2f((uint16_t)(wr.x >> 16));
    v.z -= w * bf2f((uint16_t)(wr.y & 0xffff)); v.w -= w * bf2f((uint16_t)(wr.y >> 16));
  }
  *reinterpret_cast<float4*>(dh + (int64_t)map[p] * C + c) = v;
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

bool sgrad_supported(const ModelDev& m) { return m.Clast == 128 || m.Clast == 256; }

template <int C>
static hipError_t launch_sgrad_c(const ModelDev& m, const uint16_t* hc, const float* rb_nat, const float* rb_l2, const int32_t* n_dev, int64_t R,
                                 const float* bias_l2, float* dhc, float* dwo, float* dbias, hipStream_t st) {
  using G = ScGeom<C>;
  const size_t lds = 3 * (size_t)G::STAGE_BYTES + 3 * SC_STAGE * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sgrad<C, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_sgrad<C, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)) != hipSuccess)
    return e;
  // d hidden: rows stationary (workgroups past the device-side row count exit at once), the whole item table streamed
  hipLaunchKernelGGL((k_sgrad<C, true>), dim3((unsigned)((R + 255) / 256)), dim3(SC_THREADS), lds, st, hc, rb_l2, R + 64, n_dev, m.outT, m.out_bias,
                     m.NP / SC_STAGE, (const int32_t*)nullptr, dhc, (float*)nullptr);
  // d Wo / d bo: items stationary, the scored rows streamed; row parts so that the grid covers the chip a few times over
  const unsigned blocks = (unsigned)((m.NP + 255) / 256);
  unsigned parts = 1;
  const int st_max = (int)((R + 63) / 64);
  {
    double best = 0.0;
    for (unsigned p = 1; p <= 8 && (p == 1 || st_max / (int)p >= 64); p++) {
      const double eff = ((double)blocks * p / 256.0) / (double)((blocks * p + 255) / 256);
      if (eff > best + 1e-9) { best = eff; parts = p; }
      if (eff >= 0.93) break;
    }
  }
  hipLaunchKernelGGL((k_sgrad<C, false>), dim3(blocks, parts), dim3(SC_THREADS), lds, st, m.outT, bias_l2, (int64_t)m.NP, (const int32_t*)nullptr, hc, rb_nat,
                     st_max, n_dev, dwo, dbias);
  return hipGetLastError();
}

// The scoring layer's backward from the compact scored rows (k_sgrad_gather): dhc [R + 64][C] fp32 (compact), d outT [NP][C] and d bo [NP]
// ACCUMULATED into (the gradient arena is zeroed at the start of the step).  bias_l2 [NP]: bo log2e (-inf on the padding items).
hipError_t launch_sgrad(const ModelDev& m, const uint16_t* hc, const float* rb_nat, const float* rb_l2, const int32_t* n_dev, int64_t R,
                        const float* bias_l2, float* dhc, float* dwo, float* dbias, hipStream_t st) {
  if (R <= 0) return hipSuccess;
  switch (m.Clast) {
    case 128: return launch_sgrad_c<128>(m, hc, rb_nat, rb_l2, n_dev, R, bias_l2, dhc, dwo, dbias, st);
    case 256: return launch_sgrad_c<256>(m, hc, rb_nat, rb_l2, n_dev, R, bias_l2, dhc, dwo, dbias, st);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_sgrad_prepare(const ModelDev& m, const uint16_t* hidden, const int32_t* y, const float* w, const float* lse, const int32_t* map,
                                const int32_t* n_dev, int64_t R, uint16_t* hc, int32_t* yc, float* wc, float* rb_nat, float* rb_l2, float* bias_l2,
                                hipStream_t st) {
  if (R <= 0) return hipSuccess;
  const int64_t n = (R + 64) * (m.Clast / 8);
  hipLaunchKernelGGL(k_sgrad_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, hidden, y, w, lse, map, n_dev, m.Clast, hc, yc, wc, rb_nat);
  hipLaunchKernelGGL(k_scale_log2e, dim3((unsigned)((R + 64 + 255) / 256)), dim3(256), 0, st, rb_nat, rb_l2, R + 64, n_dev, 1);
  hipLaunchKernelGGL(k_scale_log2e, dim3((unsigned)((m.NP + 255) / 256)), dim3(256), 0, st, m.out_bias, bias_l2, (int64_t)m.NP, (const int32_t*)nullptr, 0);
  return hipGetLastError();
}

hipError_t launch_sgrad_finish(const ModelDev& m, const float* dhc, const uint16_t* hc, const int32_t* map, const int32_t* n_dev, const int32_t* yc,
                               const float* wc, int64_t R, float* dh, float* dwo, float* dbias, hipStream_t st) {
  if (R <= 0) return hipSuccess;
  const int64_t n = R * (m.Clast / 4);
  hipLaunchKernelGGL(k_dh_finish, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, dhc, map, n_dev, yc, wc, m.outT, m.Clast, m.N, dh);
  hipLaunchKernelGGL(k_dwo_onehot, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, hc, yc, wc, n_dev, m.Clast, m.N, dwo, dbias);
  return hipGetLastError();
}
