// Layer-at-a-time low level for the shapes the fused kernel (tcn.hip) does not take: the single-level
// model_tcn (model.py:71-73, model_tcn.py:13-44: T <= 500 steps per user, six blocks with dilations
// 1..32 and channels [128,128,128,128,256,256] -> a biased Dense residual where Cin != Cout,
// customized_tcn_cell.py:102-106,123-124) and any model_hier configuration with L > 32 or unequal
// widths.  Activations live in HBM/L2 as bf16 rows [sequence * L + t][C] between layers (two
// ping-pong buffers): with a receptive field of 1 + (K-1)(2^n - 1) = 253 steps for six blocks a
// single fused LDS slab would recompute a 252-step halo per 32-step tile.
//
// One wave = one 32 (output channels) x 32 (time steps) tile.  A = weights [tap][cout][cin] rows,
// B = activation rows shifted by (K-1-j)*dil steps (zero operand left of the sequence start,
// customized_tcn_cell.py:47-48), D[row = cout][col = t]; taps outermost, k-steps inner -- the SAME
// accumulation order as the fused kernel, so both produce bitwise identical activations.
#include "common.h"
#include <cstdlib>

// RB = time blocks (32 steps each) per wave: the wave's weight fragment of a (tap, k-step) then feeds RB MFMAs instead of one -- at RB = 1
// every MFMA pulls 2 KB through L2 (a weight fragment and an activation fragment, nothing reused inside a wave), which is what bounds
// the kernel on long sequences (single-level model_tcn: T <= 500); the activation fragments of a time block are shared by the four waves
// of a workgroup (consecutive channel blocks) through the CU's L1.  Accumulation order per output is unchanged (bias, then taps, then
// k-steps), so results are bit-identical for every RB.
template <int RB>
__global__ __launch_bounds__(256) void k_tcn_layer(LayerArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int ncb = a.Cout >> 5;
  const int ntt = (a.L + 31) >> 5;
  const int ntg = (ntt + RB - 1) / RB;                 // groups of RB time blocks
  const int64_t tile = (int64_t)blockIdx.x * 4 + wave;
  const int cb = (int)(tile % ncb);
  const int64_t rest = tile / ncb;
  const int tg = (int)(rest % ntg);
  const int64_t seq = rest / ntg;
  if (seq >= a.n_seq) return;
  const int64_t row0 = seq * a.L;
  const int ks = a.Cin >> 4;

  f32x16 acc[RB], racc[RB];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int c0 = cb * 32 + 8 * q + 4 * lh;
    f32x4 init = {0.f, 0.f, 0.f, 0.f}, rinit = {0.f, 0.f, 0.f, 0.f};
    if (a.seq_init) init = *reinterpret_cast<const f32x4*>(a.seq_init + seq * a.Cout + c0);
    if (a.bias) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(a.bias + c0);
#pragma unroll
      for (int j = 0; j < 4; j++) init[j] += b4[j];
    }
    if (a.bres) rinit = *reinterpret_cast<const f32x4*>(a.bres + c0);
#pragma unroll
    for (int r = 0; r < RB; r++)
#pragma unroll
      for (int j = 0; j < 4; j++) { acc[r][4 * q + j] = init[j]; racc[r][4 * q + j] = rinit[j]; }
  }

  const uint4 zero = make_uint4(0, 0, 0, 0);
  // per trip: one weight fragment and the RB activation fragments it multiplies, loads before MFMAs
  auto k_run = [&](const uint16_t* arow, int back, f32x16* c) {
    const uint16_t* brow[RB];
    bool ok[RB];
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const int t = (tg * RB + r) * 32 + lr, tsrc = t - back;
      ok[r] = tsrc >= 0 && t < a.L;
      brow[r] = a.in + (row0 + (ok[r] ? tsrc : 0)) * a.Cin + 8 * lh;
    }
    for (int s = 0; s < ks; s += 2) {   // (Cin is a multiple of 32: ks is even)
      uint4 av[2], bv[2][RB];
#pragma unroll
      for (int u = 0; u < 2; u++) {
        av[u] = *reinterpret_cast<const uint4*>(arow + 16 * (s + u));
#pragma unroll
        for (int r = 0; r < RB; r++) bv[u][r] = ok[r] ? *reinterpret_cast<const uint4*>(brow[r] + 16 * (s + u)) : zero;
      }
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int r = 0; r < RB; r++) c[r] = mfma32(av[u], bv[u][r], c[r]);
    }
  };
  for (int j = 0; j < a.taps; j++)   // o[t] += x[t - (K-1-j)*dil] . W[j]
    k_run(a.W + ((int64_t)j * a.Cout + cb * 32 + lr) * a.Cin + 8 * lh, (a.taps - 1 - j) * a.dil, acc);
  if (!a.linear && a.Wres)           // res(x) = x . Wr + br  (customized_tcn_cell.py:106,124)
    k_run(a.Wres + ((int64_t)cb * 32 + lr) * a.Cin + 8 * lh, 0, racc);
#pragma unroll
  for (int r = 0; r < RB; r++) {
    const int t = (tg * RB + r) * 32 + lr;
    if (t >= a.L) continue;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c0 = cb * 32 + 8 * q + 4 * lh;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = acc[r][4 * q + j];
      if (!a.linear) {
        float rr[4];
        if (a.Wres) {
#pragma unroll
          for (int j = 0; j < 4; j++) rr[j] = racc[r][4 * q + j];
        } else {   // identity residual (Cin == Cout)
          const uint2 res = *reinterpret_cast<const uint2*>(a.in + (row0 + t) * a.Cin + c0);
          rr[0] = bf2f((uint16_t)(res.x & 0xffff)); rr[1] = bf2f((uint16_t)(res.x >> 16));
          rr[2] = bf2f((uint16_t)(res.y & 0xffff)); rr[3] = bf2f((uint16_t)(res.y >> 16));
        }
        if (a.c_out) {   // training: keep relu(conv + b) -- the backward pass needs its sign pattern
          uint2 co;
          co.x = pack2bf(fmaxf(v[0], 0.f), fmaxf(v[1], 0.f));
          co.y = pack2bf(fmaxf(v[2], 0.f), fmaxf(v[3], 0.f));
          *reinterpret_cast<uint2*>(a.c_out + (row0 + t) * a.Cout + c0) = co;
        }
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = fmaxf(fmaxf(v[j], 0.f) + rr[j], 0.f);
      }
      uint2 o;
      o.x = pack2bf(v[0], v[1]);
      o.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<uint2*>(a.out + (row0 + t) * a.Cout + c0) = o;
    }
  }
}

// ids [R] -> rows [R][E] bf16: the one-hot "lookup" of model.py:56-61 + model_tcn.py:35 / model_hier.py:50
// (row 0 of the table is stored as zeros).  HBM/L2-bound gather, 16 B per lane.
__global__ __launch_bounds__(256) void k_embed_rows(const uint16_t* __restrict__ emb, const int32_t* __restrict__ ids, int64_t R, int E, int N,
                                                    uint16_t* __restrict__ out) {
  const int ch = E >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * ch) return;
  const int64_t r = i / ch;
  const int c = (int)(i - r * ch);
  int id = ids[r];
  id = (unsigned)id < (unsigned)N ? id : 0;   // out of range = null id: tf.one_hot gives a zero row (model.py:59)
  *reinterpret_cast<uint4*>(out + r * E + c * 8) = *reinterpret_cast<const uint4*>(emb + (int64_t)id * E + c * 8);
}

hipError_t launch_embed_rows(const ModelDev& m, const int32_t* ids, int64_t R, uint16_t* out, hipStream_t st) {
  const int64_t n = R * (m.E / 8);
  hipLaunchKernelGGL(k_embed_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m.emb, ids, R, m.E, m.N, out);
  return hipGetLastError();
}

static int max_width(const ModelDev& m) {
  int w = m.E;
  for (int i = 0; i < m.n_blocks; i++) w = m.C[i] > w ? m.C[i] : w;
  return w;
}

size_t tcn_layers_workspace_bytes(const ModelDev& m, int64_t n_sessions, int L) {
  // two ping-pong activation buffers, each 256-byte aligned
  const size_t one = (((size_t)n_sessions * L * max_width(m) * sizeof(uint16_t)) + 255) & ~(size_t)255;
  return 2 * one;
}

hipError_t launch_tcn_layer(const LayerArgs& a, hipStream_t st) {
  const int ntt = (a.L + 31) / 32;
  // time blocks per wave: as many as a sequence has, up to four, while the grid still fills the chip a few times over
  const int64_t tiles1 = a.n_seq * ntt * (a.Cout / 32);
  int rb = ntt >= 4 ? 4 : (ntt >= 2 ? 2 : 1);
  while (rb > 1 && tiles1 / rb < 4 * 1024) rb >>= 1;
  if (getenv("HIERTCN_TCN_LAYER_RB1")) rb = 1;   // A/B
  const int64_t tiles = a.n_seq * ((ntt + rb - 1) / rb) * (a.Cout / 32);
  const dim3 grid((unsigned)((tiles + 3) / 4));
  switch (rb) {
    case 4: hipLaunchKernelGGL(k_tcn_layer<4>, grid, dim3(256), 0, st, a); break;
    case 2: hipLaunchKernelGGL(k_tcn_layer<2>, grid, dim3(256), 0, st, a); break;
    default: hipLaunchKernelGGL(k_tcn_layer<1>, grid, dim3(256), 0, st, a); break;
  }
  return hipGetLastError();
}

hipError_t launch_tcn_layers(const ModelDev& m, const int32_t* x, const float* sproj, int64_t n_sessions, int L, uint16_t* hidden,
                             void* workspace, hipStream_t st) {
  const int64_t R = n_sessions * L;
  const size_t one = tcn_layers_workspace_bytes(m, n_sessions, L) / 2;
  uint16_t* buf[2] = {reinterpret_cast<uint16_t*>(workspace), reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(workspace) + one)};
  int cur = 0;
  {
    const int64_t n = R * (m.E / 8);
    hipLaunchKernelGGL(k_embed_rows, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m.emb, x, R, m.E, m.N, buf[cur]);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  int cin = m.E;
  if (m.hier) {   // [e, state] . Wp = e . Wp[:E] + state . Wp[E:]  (model_hier.py:54-55 -> model_tcn.py:35), no bias, no activation
    LayerArgs a = {};
    a.in = buf[cur]; a.out = buf[cur ^ 1]; a.W = m.projT; a.seq_init = sproj;
    a.Cin = m.E; a.Cout = m.E; a.taps = 1; a.dil = 0; a.L = L; a.n_seq = n_sessions; a.linear = 1;
    hipError_t e = launch_tcn_layer(a, st);
    if (e != hipSuccess) return e;
    cur ^= 1;
  }
  for (int i = 0; i < m.n_blocks; i++) {
    LayerArgs a = {};
    a.in = buf[cur];
    a.out = (i == m.n_blocks - 1) ? hidden : buf[cur ^ 1];
    a.W = m.convT[i]; a.bias = m.conv_bias[i]; a.Wres = m.resT[i]; a.bres = m.res_bias[i];
    a.Cin = cin; a.Cout = m.C[i]; a.taps = m.K; a.dil = 1 << i; a.L = L; a.n_seq = n_sessions; a.linear = 0;
    hipError_t e = launch_tcn_layer(a, st);
    if (e != hipSuccess) return e;
    cur ^= 1;
    cin = m.C[i];
  }
  return hipSuccess;
}
