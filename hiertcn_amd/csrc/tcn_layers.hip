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
#include <cstdio>
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


// ---------------------------------------------------------------------------------------------------------------------------------
// LDS-tiled flavour for long sequences (single-level model_tcn: T <= 500, six blocks, customized_tcn_cell.py:147-161 with
// args.py:310-311).  The kernel above feeds every MFMA from two un-staged global loads and re-reads the activations K * (Cout / 32)
// times; this one gives a workgroup a (sequence, TT-step) tile and ALL its output channels:
//   * the activation slab -- rows [t0 - (K-1)*dil, t0 + TT) x Cin, zero left of the sequence start (customized_tcn_cell.py:47-48) and
//     right of its end -- is staged in LDS ONCE (row pitch Cin*2 + 16 bytes: the 16 lanes of a ds_read_b128 phase hit 16 distinct bank
//     groups) and serves every tap, every channel block and the identity residual: activations cross L2 once (plus the halo);
//   * a wave owns one 32-channel block and RB = 4 time blocks (four accumulators): per (tap, k-step) one weight fragment feeds four MFMAs
//     whose B operands are ds_read_b128 from the slab at a tap-dependent row offset;
//   * the wave's weight fragments come from a FRAGMENT-MAJOR copy of the kernels ([channel block][tap][k-step][lane][8]: one contiguous,
//     fully coalesced 1 KB per fragment, the whole stream of a wave one linear run) through an 8-deep register ring issued eight k-steps
//     ahead -- L2 latency is covered by the ring, not by occupancy.
// Accumulation order per output (bias, taps outermost, k-steps inner, zero operands outside the sequence) is the other two kernels', so
// the three low-level paths stay bitwise identical (tests/test_single_level.py).
// ---------------------------------------------------------------------------------------------------------------------------------
struct LdsGeom { int cbw, tgw, halo, pitch, rows; int seq_L; };   // seq_L > 0: flat rows (one tap: no row depends on another), seq_init by row / seq_L

template <int RB>
__global__ __launch_bounds__(512) void k_tcn_layer_lds(LayerArgs a, LdsGeom g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char slab[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int cbi = wave % g.cbw, tgi = wave / g.cbw;
  const int cb = blockIdx.y * g.cbw + cbi;
  const int TT = g.tgw * RB * 32;
  const int ntile = (a.L + TT - 1) / TT;
  const int64_t seq = blockIdx.x / ntile;
  const int t0 = (int)(blockIdx.x - seq * ntile) * TT;
  const int64_t row0 = seq * a.L;
  const int ks = a.Cin >> 4;
  const int nthr = blockDim.x;

  {  // stage the slab: a contiguous run of rows in global memory -> padded rows in LDS
    const int cpr = a.Cin >> 3;                 // 16-byte chunks per row
    const int total = g.rows * cpr;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    for (int c0 = tid; c0 < total; c0 += 4 * nthr) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = c0 + u * nthr;
        const int row = c / cpr, col = c - row * cpr;
        const int t = t0 - g.halo + row;
        v[u] = zero;
        if (c < total && t >= 0 && t < a.L) v[u] = *reinterpret_cast<const uint4*>(a.in + (row0 + t) * a.Cin + col * 8);
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int c = c0 + u * nthr;
        const int row = c / cpr, col = c - row * cpr;
        if (c < total) *reinterpret_cast<uint4*>(slab + row * g.pitch + col * 16) = v[u];
      }
    }
  }

  f32x16 acc[RB], racc[RB];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int c0 = cb * 32 + 8 * q + 4 * lh;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f}, rinit = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) bias4 = *reinterpret_cast<const f32x4*>(a.bias + c0);
    if (a.bres) rinit = *reinterpret_cast<const f32x4*>(a.bres + c0);
#pragma unroll
    for (int r = 0; r < RB; r++) {
      f32x4 init = {0.f, 0.f, 0.f, 0.f};
      if (a.seq_init) {
        int64_t sq = seq;
        if (g.seq_L > 0) {   // flat rows: this lane's row belongs to sequence row / seq_L
          const int64_t row = (int64_t)t0 + (tgi * RB + r) * 32 + lr;
          sq = (row < a.L ? row : (int64_t)a.L - 1) / g.seq_L;
        }
        init = *reinterpret_cast<const f32x4*>(a.seq_init + sq * a.Cout + c0);
      }
#pragma unroll
      for (int j = 0; j < 4; j++) { acc[r][4 * q + j] = init[j] + bias4[j]; racc[r][4 * q + j] = rinit[j]; }
    }
  }
  // B-operand base of time block r: slab row (tgi*RB + r)*32 + lr (+ halo - back per tap), byte column 16*lh (+ 32 per k-step)
  const unsigned char* bbase = slab + ((tgi * RB) * 32 + lr) * g.pitch + 16 * lh;
  const int blk = 32 * g.pitch;

  // one weight stream: n_steps fragments of 1 KB, `tap_rows` slab rows further per tap.  Fragments travel in batches of eight through
  // two register sets: the loads of batch g+1 are issued in one run BEFORE the 32 MFMAs of batch g, so they have that long (1,000+
  // cycles per wave) to arrive.  The step code is branch-free (clamped prefetch index, scalar selects for the tap wrap): with control
  // flow between the steps hipcc can no longer count outstanding loads and waits for vmcnt(0) behind every step.
  auto run = [&](const uint16_t* wstream, int n_taps, int first_back, int tap_rows, f32x16* c) {
    const int n_steps = n_taps * ks;
    const int n_full = n_steps & ~7;
    const uint16_t* wp = wstream + lane * 8;
    int s = 0;
    int rowoff = (g.halo - first_back) * g.pitch;
    const int tap_bytes = tap_rows * g.pitch;
    auto step = [&](const uint4 av) {
      uint4 bv[RB];
#pragma unroll
      for (int r = 0; r < RB; r++) bv[r] = *reinterpret_cast<const uint4*>(bbase + r * blk + rowoff + 32 * s);
#pragma unroll
      for (int r = 0; r < RB; r++) c[r] = mfma32(av, bv[r], c[r]);
      s++;
      const bool wrap = s == ks;
      s = wrap ? 0 : s;
      rowoff += wrap ? tap_bytes : 0;
    };
    auto load8 = [&](uint4* w, int g0) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int idx = g0 + u < n_steps ? g0 + u : n_steps - 1;
        w[u] = *reinterpret_cast<const uint4*>(wp + (int64_t)idx * 512);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the whole batch of loads ahead of the MFMAs that follow (hipcc sinks them to their uses otherwise)
    };
    auto comp8 = [&](const uint4* w) {
#pragma unroll
      for (int u = 0; u < 8; u++) step(w[u]);
    };
    auto tail = [&](const uint4* w) {   // fewer than eight fragments left, already loaded
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (n_full + u < n_steps) step(w[u]);
    };
    uint4 wa[8], wb[8];
    load8(wa, 0);
    int i0 = 0;
    for (; i0 + 16 <= n_full; i0 += 16) {
      load8(wb, i0 + 8);
      comp8(wa);
      load8(wa, i0 + 16);
      comp8(wb);
    }
    if (i0 < n_full) {
      load8(wb, i0 + 8);
      comp8(wa);
      tail(wb);
    } else {
      tail(wa);
    }
  };
  __syncthreads();
  // taps: o[t] += x[t - (K-1-j)*dil] . W[j]  (j = 0 reaches furthest back)
  run(a.Wf + (int64_t)cb * a.taps * ks * 512, a.taps, (a.taps - 1) * a.dil, a.dil, acc);
  if (!a.linear && a.Wresf) run(a.Wresf + (int64_t)cb * ks * 512, 1, 0, 0, racc);   // res(x) = x . Wr + br (customized_tcn_cell.py:106,124)

#pragma unroll
  for (int r = 0; r < RB; r++) {
    const int tl = (tgi * RB + r) * 32 + lr;
    const int t = t0 + tl;
    if (t >= a.L) continue;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int c0 = cb * 32 + 8 * q + 4 * lh;
      float v[4];
#pragma unroll
      for (int j = 0; j < 4; j++) v[j] = acc[r][4 * q + j];
      if (!a.linear) {
        float rr[4];
        if (a.Wresf) {
#pragma unroll
          for (int j = 0; j < 4; j++) rr[j] = racc[r][4 * q + j];
        } else {   // identity residual (Cin == Cout): the row is in the slab
          const uint2 res = *reinterpret_cast<const uint2*>(slab + (tl + g.halo) * g.pitch + c0 * 2);
          rr[0] = bf2f((uint16_t)(res.x & 0xffff)); rr[1] = bf2f((uint16_t)(res.x >> 16));
          rr[2] = bf2f((uint16_t)(res.y & 0xffff)); rr[3] = bf2f((uint16_t)(res.y >> 16));
        }
        if (a.c_out) {
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

// [taps][Cout][Cin] -> fragment-major [Cout/32][taps][Cin/16][64 lanes][8]: lane l of fragment (cb, j, s) holds W[j][32 cb + (l & 31)][16 s + 8 (l >> 5) ..+8]
__global__ void k_frag_major(const uint16_t* __restrict__ W, uint16_t* __restrict__ Wf, int taps, int Cout, int Cin) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int ks = Cin >> 4, ncb = Cout >> 5;
  if (i >= (int64_t)ncb * taps * ks * 64) return;
  const int l = (int)(i & 63);
  int64_t f = i >> 6;
  const int s = (int)(f % ks); f /= ks;
  const int j = (int)(f % taps);
  const int cb = (int)(f / taps);
  *reinterpret_cast<uint4*>(Wf + i * 8) =
      *reinterpret_cast<const uint4*>(W + ((int64_t)j * Cout + cb * 32 + (l & 31)) * Cin + 16 * s + 8 * (l >> 5));
}

hipError_t launch_frag_major(const uint16_t* W, uint16_t* Wf, int taps, int Cout, int Cin, hipStream_t st) {
  const int64_t n = (int64_t)(Cout >> 5) * taps * (Cin >> 4) * 64;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_frag_major, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, W, Wf, taps, Cout, Cin);
  return hipGetLastError();
}

hipError_t refresh_frag_major(const ModelDev& m, hipStream_t st) {
  hipError_t e;
  if (m.hier && m.projF && (e = launch_frag_major(m.projT, m.projF, 1, m.E, m.E, st)) != hipSuccess) return e;
  for (int i = 0, cin = m.E; i < m.n_blocks && !m.low_gru; cin = m.C[i], i++) {
    if (m.convF[i] && (e = launch_frag_major(m.convT[i], m.convF[i], m.K, m.C[i], cin, st)) != hipSuccess) return e;
    if (m.resF[i] && m.resT[i] && (e = launch_frag_major(m.resT[i], m.resF[i], 1, m.C[i], cin, st)) != hipSuccess) return e;
  }
  return hipSuccess;
}

// geometry of the LDS-tiled kernel for this layer, or cbw = 0 when it does not apply (then the un-staged kernel runs)
static LdsGeom lds_geom(const LayerArgs& a) {
  LdsGeom g = {0, 0, 0, 0, 0};
  if (!a.Wf || (a.Wres && !a.Wresf) || ABL_ENV("HIERTCN_TCN_LAYER_V1")) return g;   // (ablation build: the un-staged kernel everywhere)
  const int ncb = a.Cout >> 5;
  if (ncb & (ncb - 1)) return g;
  if (a.L < 96 && !(a.taps == 1 && a.n_seq * (int64_t)a.L >= 4096 && a.n_seq * (int64_t)a.L < ((int64_t)1 << 31)))
    return g;   // short sequences: the un-staged kernel (tiles of 128 steps would be mostly padding) -- unless the layer has ONE tap (the 'emb'
                // projection, a Dense): then no row depends on another and the rows of all sequences tile as one flat run
  const int cbw = ncb < 8 ? ncb : 8;
  const int halo = (a.taps - 1) * a.dil;
  const int pitch = a.Cin * 2 + 16;
  for (int tgw = 8 / cbw; tgw >= 1; tgw >>= 1) {
    const int rows = tgw * 128 + halo;
    const int64_t Leff = a.L < 96 ? a.n_seq * (int64_t)a.L : a.L;   // (flat rows when the sequences are short: see above)
    if ((int64_t)rows * pitch <= 160 * 1024 - 512 && (tgw == 1 || (tgw - 1) * 128 < Leff)) {
      g.cbw = cbw; g.tgw = tgw; g.halo = halo; g.pitch = pitch; g.rows = rows;
      g.seq_L = (a.L < 96) ? a.L : 0;
      return g;
    }
  }
  return g;
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

// rows [n_sess * L][E + HT] = E[x_t] (zero row for a null / out-of-range id) | state of the session, rounded to the storage type
// (model_hier.py:50,54-55: dense 'emb' of the one-hot row, the carried state tiled over the steps, concatenated)
__global__ __launch_bounds__(256) void k_embed_concat(const uint16_t* __restrict__ emb, const int32_t* __restrict__ ids, const float* __restrict__ state,
                                                      int64_t R, int L, int E, int HT, int N, uint16_t* __restrict__ out) {
  const int W = E + HT, ch = W >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * ch) return;
  const int64_t r = i / ch;
  const int c = (int)(i - r * ch) * 8;
  uint4 v;
  if (c < E) {
    int id = ids[r];
    id = (unsigned)id < (unsigned)N ? id : 0;
    v = *reinterpret_cast<const uint4*>(emb + (int64_t)id * E + c);
  } else {
    const float* sp = state + (r / L) * HT + (c - E);
    v = make_uint4(pack2bf(sp[0], sp[1]), pack2bf(sp[2], sp[3]), pack2bf(sp[4], sp[5]), pack2bf(sp[6], sp[7]));
  }
  *reinterpret_cast<uint4*>(out + r * W + c) = v;
}

hipError_t launch_embed_concat(const ModelDev& m, const int32_t* ids, const float* state, int64_t n_sess, int L, uint16_t* out, hipStream_t st) {
  if ((m.E & 7) || (m.HT & 7)) return hipErrorInvalidValue;
  const int64_t R = n_sess * L;
  const int64_t n = R * ((m.E + m.HT) / 8);
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(k_embed_concat, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, m.emb, ids, state, R, L, m.E, m.HT, m.N, out);
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

static hipError_t launch_tcn_layer_lds(const LayerArgs& a, const LdsGeom& g, hipStream_t st) {
  const int TT = g.tgw * 128;
  const size_t lds = (size_t)g.rows * g.pitch;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_tcn_layer_lds<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const dim3 grid((unsigned)(a.n_seq * ((a.L + TT - 1) / TT)), (unsigned)((a.Cout >> 5) / g.cbw));
  hipLaunchKernelGGL(k_tcn_layer_lds<4>, grid, dim3(64 * g.cbw * g.tgw), lds, st, a, g);
  return hipGetLastError();
}

hipError_t launch_tcn_layer(const LayerArgs& a, hipStream_t st) {
  const LdsGeom g = lds_geom(a);
  if (ABL_ENV("HIERTCN_DEBUG_LAYER"))
    fprintf(stderr, "launch_tcn_layer: L %d n_seq %lld Cin %d Cout %d taps %d Wf %p Wres %p Wresf %p -> cbw %d tgw %d seq_L %d\n", a.L, (long long)a.n_seq, a.Cin,
            a.Cout, a.taps, (const void*)a.Wf, (const void*)a.Wres, (const void*)a.Wresf, g.cbw, g.tgw, g.seq_L);
  if (g.cbw > 0 && g.seq_L > 0) {   // flat rows: one "sequence" of n_seq * L rows
    LayerArgs f = a;
    f.L = (int)(a.n_seq * (int64_t)a.L);
    f.n_seq = 1;
    return launch_tcn_layer_lds(f, g, st);
  }
  if (g.cbw > 0) return launch_tcn_layer_lds(a, g, st);
  const int ntt = (a.L + 31) / 32;
  // time blocks per wave: as many as a sequence has, up to four, while the grid still fills the chip a few times over
  const int64_t tiles1 = a.n_seq * ntt * (a.Cout / 32);
  int rb = ntt >= 4 ? 4 : (ntt >= 2 ? 2 : 1);
  while (rb > 1 && tiles1 / rb < 4 * 1024) rb >>= 1;
  if (ABL_ENV("HIERTCN_TCN_LAYER_RB1")) rb = 1;   // A/B (ablation build)
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
    a.in = buf[cur]; a.out = buf[cur ^ 1]; a.W = m.projT; a.Wf = m.projF; a.seq_init = sproj;
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
    a.Wf = m.convF[i]; a.Wresf = m.resF[i];
    a.Cin = cin; a.Cout = m.C[i]; a.taps = m.K; a.dil = 1 << i; a.L = L; a.n_seq = n_sessions; a.linear = 0;
    hipError_t e = launch_tcn_layer(a, st);
    if (e != hipSuccess) return e;
    cur ^= 1;
    cin = m.C[i];
  }
  return hipSuccess;
}
