// The EXACT numerics mode (hiertcn_set_precision(h, HIERTCN_PRECISION_FP32)): the whole path in fp32, as the reference
// computes it (model_tcn.py:35-41, customized_dense_layer.py:155-172, customized_convolution_layer.py:171-198, loss.py:21,120,179) --
// nothing is stored in 16 bits, every product runs on v_mfma_f32_32x32x2_f32 (an fp32 multiply-add chain), the logits of a row chunk are
// MATERIALISED and rank / cross entropy / ranked top-k are taken on the materialised fp32 values, exactly the arrays the reference's
// graph holds.  This is the mode that meets BASELINE.json's "fp32 logits within 1e-3" on a TRAINED model (|z| ~ 10-30), where the 16-bit
// storage of the production path costs ~2e-3 of the logit scale.  It is built for agreement, not for speed: one LDS-tiled
// GEMM kernel serves the embedding projection, every causal conv tap, the Dense down-sample and the scoring layer.
#include <math.h>

#include "common.h"
#include "exact.h"

typedef float f32x16x __attribute__((ext_vector_type(16)));

// ---------------------------------------------------------------------------------------------------------------------------------
// C[M][N] = epi( sum_taps A_tap[M][Cin] . Wt[tap][N][Cin]^T ) -- both operands row-major with the reduction index contiguous.
// Workgroup = 4 waves = a 128 x 128 tile, each wave four 32 x 32 accumulators; 32-deep k chunks staged through LDS (pitch 36 floats:
// the 16 lanes of a ds_read_b128 phase hit 16 distinct bank groups), the next chunk's global loads in flight during the MFMAs.
// A rows: plain, causally shifted inside sequences of L rows (conv tap j reads row t - (taps-1-j)*dil, zero before the sequence start:
// customized_tcn_cell.py:46-49), or gathered from an item table by id (model.py:59-61: ids <= 0 / >= n_tab are zero rows).
// ---------------------------------------------------------------------------------------------------------------------------------
#define XT 128
#define XK 32
#define XP 36

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_xgemm(const XGemm g) {   // (three workgroups per CU: <= 170 registers; four would spill 74)
  __shared__ float As[XT * XP];
  __shared__ float Bs[XT * XP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 31, lh = lane >> 5;
  const int64_t m0 = (int64_t)blockIdx.y * XT;
  const int n0 = blockIdx.x * XT;
  const int wm = wave >> 1, wn = wave & 1;
  const int ldrow = tid >> 3, ldc4 = (tid & 7) * 4;   // this thread stages rows ldrow + 32p, floats [ldc4, ldc4 + 4) of a chunk

  // per staged A row: source row index for tap shift 0 (or -1: zero row) and the position inside its sequence
  int64_t a_src[4];
  int a_t[4];
#pragma unroll
  for (int p = 0; p < 4; p++) {
    const int64_t gm = m0 + ldrow + 32 * p;
    a_src[p] = -1; a_t[p] = 0;
    if (gm < g.M) {
      if (g.ids) {
        const int id = g.ids[gm];
        if (id > 0 && id < g.n_tab) a_src[p] = id;
      } else {
        a_src[p] = gm;
        a_t[p] = g.L > 0 ? (int)(gm % g.L) : 0;
      }
    }
  }
  const int kchunks = g.Cin / XK;
  const int n_it = g.taps * kchunks;
  float4 ra[4], rb[4];
  // the staged rows' addresses are derived ONCE PER TAP (row index, causal shift, 64-bit row offset, the bounds tests) and advanced by one chunk per
  // trip: re-derived per load they were ~240 vector instructions per 32-deep chunk beside its 64 MFMAs (3.7 VALU per MFMA, profiles/r5_xgemm_pmc.json)
  const float* pa[4];
  const float* pb[4];
  auto fetch = [&](int it) {
    const int j = it / kchunks, kc = it - j * kchunks;
    if (kc == 0) {
      const int sh = (g.taps - 1 - j) * g.dil;
#pragma unroll
      for (int p = 0; p < 4; p++) {
        pa[p] = (a_src[p] >= 0 && (g.ids || a_t[p] - sh >= 0)) ? g.A + (a_src[p] - (g.ids ? 0 : sh)) * g.lda + ldc4 : nullptr;
        const int gn = n0 + ldrow + 32 * p;
        pb[p] = gn < g.N ? g.Wt + ((int64_t)j * g.N + gn) * g.Cin + ldc4 : nullptr;
      }
    }
#pragma unroll
    for (int p = 0; p < 4; p++) {
      ra[p] = pa[p] ? *reinterpret_cast<const float4*>(pa[p] + kc * XK) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[p] = pb[p] ? *reinterpret_cast<const float4*>(pb[p] + kc * XK) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  f32x16x acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int q = 0; q < 16; q++) acc[i][j][q] = 0.f;

  fetch(0);
  for (int it = 0; it < n_it; it++) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; p++) {
      *reinterpret_cast<float4*>(&As[(ldrow + 32 * p) * XP + ldc4]) = ra[p];
      *reinterpret_cast<float4*>(&Bs[(ldrow + 32 * p) * XP + ldc4]) = rb[p];
    }
    __syncthreads();
    if (it + 1 < n_it) fetch(it + 1);
#pragma unroll
    for (int s = 0; s < XK / 8; s++) {
      // k order inside an 8-deep step: MFMA c (0..3) consumes k = 8s + c (lane half 0) and 8s + 4 + c (lane half 1)
      const float4 a0 = *reinterpret_cast<const float4*>(&As[(64 * wm + lr) * XP + 8 * s + 4 * lh]);
      const float4 a1 = *reinterpret_cast<const float4*>(&As[(64 * wm + 32 + lr) * XP + 8 * s + 4 * lh]);
      const float4 b0 = *reinterpret_cast<const float4*>(&Bs[(64 * wn + lr) * XP + 8 * s + 4 * lh]);
      const float4 b1 = *reinterpret_cast<const float4*>(&Bs[(64 * wn + 32 + lr) * XP + 8 * s + 4 * lh]);
#define XSTEP(c)                                                                      \
  acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b0.c, acc[0][0], 0, 0, 0);   \
  acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.c, b1.c, acc[0][1], 0, 0, 0);   \
  acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b0.c, acc[1][0], 0, 0, 0);   \
  acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.c, b1.c, acc[1][1], 0, 0, 0);
      XSTEP(x) XSTEP(y) XSTEP(z) XSTEP(w)
#undef XSTEP
    }
  }
  // epilogue: D[row = m][col = n]; lane holds column lr of 16 rows per tile.  Row-major over the lane's 32 rows: one 64-bit row offset (and, for the
  // projection, one 32-bit division for the sequence's row of seq_init) per ROW, the two columns' stores behind it -- the column-major form derived
  // them per element behind per-element branches: ~10,000 instructions per wave and tile, 20 us of every tile's 34 at K = 256
  // (scripts/xgemm_k_sweep.py: 0.36 ms + 3.5 us K per 768 x 300,000 chunk whatever the row stride)
  int nn[2];
  float bn[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    nn[j] = n0 + 64 * wn + 32 * j + lr;
    bn[j] = (g.bias && nn[j] < g.N) ? g.bias[nn[j]] : 0.f;
  }
  const bool block = g.epi == XEPI_BLOCK;
#pragma unroll
  for (int i = 0; i < 2; i++) {
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const int64_t m = m0 + 64 * wm + 32 * i + acc_row(q, lh);
      if (m >= g.M) continue;
      float* orow = g.out + m * g.ldo;
      const float* srow = g.seq_init ? g.seq_init + (int64_t)((int)m / g.Lseq) * g.N : nullptr;   // (m < 65535 x 128 rows per launch: launch_xgemm)
      float* o2row = (block && g.out2) ? g.out2 + m * g.ldo : nullptr;
      const float* rrow = block ? g.res + m * g.ldres : nullptr;
#pragma unroll
      for (int j = 0; j < 2; j++) {
        if (nn[j] >= g.N) continue;
        float v = acc[i][j][q];
        if (srow) v += srow[nn[j]];
        v += bn[j];
        if (block) {   // customized_tcn_cell.py:117-126: relu(relu(conv1(x)) + res(x)), dropout rate 0
          v = fmaxf(v, 0.f);
          if (o2row) o2row[nn[j]] = v;
          v = fmaxf(v + rrow[nn[j]], 0.f);
        }
        orow[nn[j]] = v;
      }
    }
  }
}

hipError_t launch_xgemm(const XGemm& g, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0) return hipSuccess;
  if (g.Cin % XK != 0 || (g.seq_init && g.L > 0 && g.L != g.Lseq)) return hipErrorInvalidValue;
  // gridDim.y <= 65535: at most 8.3 M rows per launch, cut at a sequence boundary (the kernel takes t = row % L from the launch's own rows)
  const int64_t unit = g.seq_init ? g.Lseq : (g.L > 0 ? g.L : 1);
  const int64_t rows_per_launch = ((int64_t)65535 * XT / unit) * unit;
  for (int64_t r0 = 0; r0 < g.M; r0 += rows_per_launch) {
    XGemm p = g;
    p.M = (g.M - r0 < rows_per_launch) ? g.M - r0 : rows_per_launch;
    if (r0 > 0) {
      if (g.ids) p.ids = g.ids + r0; else p.A = g.A + r0 * g.lda;
      p.out = g.out + r0 * g.ldo;
      if (g.out2) p.out2 = g.out2 + r0 * g.ldo;
      if (g.res) p.res = g.res + r0 * g.ldres;
      if (g.seq_init) p.seq_init = g.seq_init + (r0 / g.Lseq) * (int64_t)g.N;
    }
    dim3 grid((unsigned)((g.N + XT - 1) / XT), (unsigned)((p.M + XT - 1) / XT));
    hipLaunchKernelGGL(k_xgemm, grid, dim3(256), 0, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// session mean-pool on the fp32 item table (model_hier.py:83-85): one wave per session, targets summed in step order
// ---------------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_xsession_pool(const float* __restrict__ emb, const float* __restrict__ emb_bias,
                                                       const int32_t* __restrict__ y, int64_t n_sessions, int L, int E, int N,
                                                       float* __restrict__ pooled) {
  const int lane = threadIdx.x & 63;
  const int64_t sess = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (sess >= n_sessions) return;
  const int32_t* ys = y + sess * L;
  for (int c = lane; c < E; c += 64) {
    float s = 0.f;
    int cnt = 0;
    for (int t = 0; t < L; t++) {
      const int id = ys[t];
      if (id > 0 && id < N) { s += emb[(int64_t)id * E + c]; cnt++; }
    }
    pooled[sess * E + c] = s / (float)cnt + emb_bias[c];   // cnt == 0 -> 0/0 = NaN, as the reference (model_hier.py:84)
  }
}

// single-level model_tcn: the 'emb' dense on a one-hot row is the row itself (model.py:59-61, model_tcn.py:35)
__global__ void k_xembed_rows(const float* __restrict__ emb, const int32_t* __restrict__ ids, int64_t R, int E, int N, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int e4 = E >> 2;
  if (i >= R * e4) return;
  const int64_t r = i / e4;
  const int c = (int)(i - r * e4) * 4;
  const int id = ids[r];
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (id > 0 && id < N) v = *reinterpret_cast<const float4*>(emb + (int64_t)id * E + c);
  *reinterpret_cast<float4*>(out + r * E + c) = v;
}

hipError_t launch_xembed_rows(const float* emb, const int32_t* ids, int64_t R, int E, int N, float* out, hipStream_t st) {
  if (R <= 0) return hipSuccess;
  const int64_t n4 = R * (E >> 2);
  hipLaunchKernelGGL(k_xembed_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, emb, ids, R, E, N, out);
  return hipGetLastError();
}

hipError_t launch_xsession_pool(const ModelDev& m, const ExactDev& x, const int32_t* y, int64_t n_sessions, int L, float* pooled, hipStream_t st) {
  if (n_sessions <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_xsession_pool, dim3((unsigned)((n_sessions + 3) / 4)), dim3(256), 0, st, x.emb, m.emb_bias, y, n_sessions, L, m.E, m.N, pooled);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------------------
// low level: ids -> last TemporalBlock activations, fp32 [R][Clast]
// workspace: two ping-pong activation buffers + one residual buffer, each [R][Cmax] fp32
// ---------------------------------------------------------------------------------------------------------------------------------
static int xmaxc(const ModelDev& m) {
  int c = m.E;
  for (int i = 0; i < m.n_blocks; i++) c = m.C[i] > c ? m.C[i] : c;
  return c;
}
static size_t xal(size_t b) { return (b + 255) & ~(size_t)255; }

size_t exact_tcn_workspace_bytes(const ModelDev& m, int64_t n_sessions, int L) {
  if (m.low_gru)   // projected rows (fp32) + lengths + two state buffers + one (gates | r*h | cand | out) record per layer
    return xal((size_t)n_sessions * L * m.E * 4) + xal((size_t)n_sessions * 4) + 2 * xal((size_t)n_sessions * m.HT * 4) +
           xal((size_t)m.nl * n_sessions * 5 * m.H * 4) + (n_sessions <= 1024 ? xal(low_gru_persist_floats(m, n_sessions, L) * 4) : 0);
  return 3 * xal((size_t)n_sessions * L * xmaxc(m) * sizeof(float));
}

// HierGRU low level (model_hier.py:67-68 -> model_gru.py:16-63) in fp32: the 'emb' projection of [e, state] as a gathered GEMM, then the
// same exact-fp32 GRU cell launches as the production path (pool_gru.hip) with fp32 output rows
static hipError_t launch_exact_low_gru(const ModelDev& m, const ExactDev& x, const int32_t* ids, const float* sproj, int64_t n_sess, int L,
                                       float* hidden, void* workspace, hipStream_t st) {
  const int64_t R = n_sess * L;
  unsigned char* w = reinterpret_cast<unsigned char*>(workspace);
  auto take = [&](size_t bytes) { unsigned char* p = w; w += xal(bytes); return p; };
  LowGruBufs b = {};
  b.uf = reinterpret_cast<float*>(take((size_t)R * m.E * 4));
  b.seq_len = reinterpret_cast<int32_t*>(take((size_t)n_sess * 4));
  b.states = reinterpret_cast<float*>(take(2 * xal((size_t)n_sess * m.HT * 4)));
  b.n_states = 2;
  b.rec = reinterpret_cast<float*>(take((size_t)m.nl * n_sess * 5 * m.H * 4));
  b.keep_all = 0;
  b.hidden32 = hidden;
  b.persist = n_sess <= 1024 ? reinterpret_cast<float*>(take(low_gru_persist_floats(m, n_sess, L) * 4)) : nullptr;
  hipError_t e;
  if (!m.hier) {   // single-level model_gru: the embedded ids are the stack's input (model_gru.py:46 on one-hot rows = the lookup)
    e = launch_xembed_rows(x.emb, ids, R, m.E, m.N, b.uf, st);
  } else {
    XGemm g = {};
    g.A = x.emb; g.lda = m.E; g.ids = ids; g.n_tab = m.N;
    g.Wt = x.projT; g.taps = 1; g.Cin = m.E; g.M = R; g.N = m.E;
    g.seq_init = sproj; g.Lseq = L;
    g.out = b.uf; g.ldo = m.E; g.epi = XEPI_LINEAR;
    e = launch_xgemm(g, st);
  }
  if (e != hipSuccess) return e;
  return launch_low_gru_bufs(m, ids, sproj, n_sess, L, nullptr, b, st);
}

hipError_t launch_exact_tcn(const ModelDev& m, const ExactDev& x, const int32_t* ids, const float* sproj, int64_t n_sessions, int L, float* hidden,
                            void* workspace, hipStream_t st) {
  const int64_t R = n_sessions * L;
  if (R <= 0) return hipSuccess;
  if (m.low_gru) return launch_exact_low_gru(m, x, ids, sproj, n_sessions, L, hidden, workspace, st);
  const size_t bufb = xal((size_t)R * xmaxc(m) * sizeof(float));
  float* buf[2] = {reinterpret_cast<float*>(workspace), reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + bufb)};
  float* resb = reinterpret_cast<float*>(reinterpret_cast<unsigned char*>(workspace) + 2 * bufb);
  hipError_t e;
  float* cur = buf[0];
  if (m.hier) {   // model_hier.py:50,54-55 -> model_tcn.py:35: [e, state] . Wp = e . Wp[:E] + state . Wp[E:] (the scan's sproj), no bias
    XGemm g = {};
    g.A = x.emb; g.lda = m.E; g.ids = ids; g.n_tab = m.N;
    g.Wt = x.projT; g.taps = 1; g.dil = 0; g.L = 0; g.Cin = m.E; g.M = R; g.N = m.E;
    g.seq_init = sproj; g.Lseq = L;
    g.out = cur; g.ldo = m.E; g.epi = XEPI_LINEAR;
    if ((e = launch_xgemm(g, st)) != hipSuccess) return e;
  } else {
    const int64_t n4 = R * (m.E >> 2);
    hipLaunchKernelGGL(k_xembed_rows, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, x.emb, ids, R, m.E, m.N, cur);
    if ((e = hipGetLastError()) != hipSuccess) return e;
  }
  for (int i = 0, cin = m.E; i < m.n_blocks; cin = m.C[i], i++) {
    const int cout = m.C[i];
    float* nxt = (i == m.n_blocks - 1) ? hidden : (cur == buf[0] ? buf[1] : buf[0]);
    const float* res = cur;
    int64_t ldres = cin;
    if (cin != cout) {   // customized_tcn_cell.py:102-106,123-124: a biased Dense on the residual path
      XGemm r = {};
      r.A = cur; r.lda = cin; r.Wt = x.resT[i]; r.taps = 1; r.Cin = cin; r.M = R; r.N = cout; r.bias = m.res_bias[i];
      r.out = resb; r.ldo = cout; r.epi = XEPI_LINEAR;
      if ((e = launch_xgemm(r, st)) != hipSuccess) return e;
      res = resb; ldres = cout;
    }
    XGemm g = {};
    g.A = cur; g.lda = cin; g.Wt = x.convT[i]; g.taps = m.K; g.dil = 1 << i; g.L = L; g.Cin = cin; g.M = R; g.N = cout;
    g.bias = m.conv_bias[i]; g.res = res; g.ldres = ldres; g.out = nxt; g.ldo = cout; g.epi = XEPI_BLOCK;
    if ((e = launch_xgemm(g, st)) != hipSuccess) return e;
    cur = nxt;
  }
  return hipSuccess;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// scoring on materialised fp32 logits (model_tcn.py:41, model.py:105, loss.py:21,179,120)
// ---------------------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float xblock_max(float v, float* sh) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// one workgroup per row.  rank = #{n : z_n > z_y} (strict, loss.py:179); ce = logsumexp(z) - z_y with the row maximum as the reference
// point ([TF-sem] softmax_cross_entropy_with_logits subtracts the row maximum); rows without a target: 0 / 0 / 0 (model.py:105).
__global__ __launch_bounds__(256) void k_xrow_stats(const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ y, int N,
                                                    int32_t* __restrict__ o_rank, float* __restrict__ o_ce, float* __restrict__ o_tgt) {
  __shared__ float shf[4];
  __shared__ int shi[4];
  __shared__ double shd[4];
  const int64_t r = blockIdx.x;
  const int yi = y[r];
  if (yi <= 0 || yi >= N) {
    if (threadIdx.x == 0) {
      if (o_rank) o_rank[r] = 0;
      if (o_ce) o_ce[r] = 0.f;
      if (o_tgt) o_tgt[r] = 0.f;
    }
    return;
  }
  const float* z = logits + r * ld;
  const float tz = z[yi];
  // ONE pass over the row (it is read from HBM: a chunk of materialised logits does not fit any cache): count, running maximum and the sum
  // of exponentials relative to the running maximum (rescaled when the maximum moves), per thread; then the threads' (max, sum) pairs are
  // merged at the row maximum.  The sum is carried in fp64: N fp32 terms would put ~1e-6 of rounding into ce, this keeps it at the logit error.
  // (eight strided elements per trip: eight loads in flight, ONE test of the running maximum and one fp64 add per eight terms -- the per-element
  //  form, a load + a data-dependent branch + an fp64 update each, ran at 1.4 TB/s: 0.65 ms per 768-row chunk of a 300,000-item catalogue)
  float mx = -INFINITY;
  double s = 0.0;
  int cnt = 0;
  for (int n0 = threadIdx.x; n0 < N; n0 += 8 * 256) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = (n0 + 256 * j < N) ? z[n0 + 256 * j] : -INFINITY;   // (-inf: counts nothing, adds exp(-inf) = 0)
    float m8 = v[0];
#pragma unroll
    for (int j = 1; j < 8; j++) m8 = fmaxf(m8, v[j]);
#pragma unroll
    for (int j = 0; j < 8; j++) cnt += v[j] > tz ? 1 : 0;
    if (m8 > mx) {
      s *= (double)expf(mx - m8);           // (mx = -inf on the first trip: exp(-inf) = 0 and s = 0)
      mx = m8;
    }
    float p = 0.f;                          // eight terms <= 1 each: an fp32 partial sum carries ~1e-7 of its own size into the fp64 total
#pragma unroll
    for (int j = 0; j < 8; j++) p += expf(v[j] - mx);
    s += (double)p;
  }
  const float row_max = xblock_max(mx, shf);
  s *= (double)expf(mx - row_max);           // a thread that saw no element has mx = -inf, s = 0: 0 * 0
  for (int o = 32; o > 0; o >>= 1) {
    cnt += __shfl_xor(cnt, o);
    s += __shfl_xor(s, o);
  }
  if ((threadIdx.x & 63) == 0) { shi[threadIdx.x >> 6] = cnt; shd[threadIdx.x >> 6] = s; }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (o_rank) o_rank[r] = shi[0] + shi[1] + shi[2] + shi[3];
    const double tot = (shd[0] + shd[1]) + (shd[2] + shd[3]);
    if (o_ce) o_ce[r] = (float)((double)row_max + log(tot) - (double)tz);
    if (o_tgt) o_tgt[r] = tz;
  }
}

// ranked top-k of a materialised row ([TF-sem] tf.nn.top_k, loss.py:120: values descending, equal values in index order): k rounds of a
// block-wide arg-max over the items that come AFTER the previous pick in that order.  Rows without a target are all-zero rows
// (model.py:105): 0..k-1 with value 0.
__global__ __launch_bounds__(256) void k_xrow_topk(const float* __restrict__ logits, int64_t ld, const int32_t* __restrict__ y, int N, int k,
                                                   int32_t* __restrict__ o_idx, float* __restrict__ o_val) {
  __shared__ float shv[4];
  __shared__ int shx[4];
  const int64_t r = blockIdx.x;
  const int yi = y[r];
  if (yi <= 0 || yi >= N) {
    if ((int)threadIdx.x < k) { o_idx[r * k + threadIdx.x] = threadIdx.x; o_val[r * k + threadIdx.x] = 0.f; }
    return;
  }
  const float* z = logits + r * ld;
  float pv = INFINITY;
  int px = -1;
  for (int i = 0; i < k; i++) {
    float bv = -INFINITY;
    int bx = 0x7fffffff;
    for (int n = threadIdx.x; n < N; n += 256) {
      const float v = z[n];
      const bool after = (v < pv) || (v == pv && n > px);
      if (after && (v > bv || (v == bv && n < bx))) { bv = v; bx = n; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o);
      const int ox = __shfl_xor(bx, o);
      if (ov > bv || (ov == bv && ox < bx)) { bv = ov; bx = ox; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { shv[threadIdx.x >> 6] = bv; shx[threadIdx.x >> 6] = bx; }
    __syncthreads();
    bv = shv[0]; bx = shx[0];
#pragma unroll
    for (int w = 1; w < 4; w++)
      if (shv[w] > bv || (shv[w] == bv && shx[w] < bx)) { bv = shv[w]; bx = shx[w]; }
    if (threadIdx.x == 0) { o_idx[r * k + i] = bx; o_val[r * k + i] = bv; }
    pv = bv; px = bx;
  }
}

// rows of materialised logits per chunk: 2 GiB of scratch, at most 8192 rows.  (1 GiB / 4096 until round 5: the GEMM's rate grows with the number of
// tile rounds of a launch -- scripts/xgemm_tile_sweep.py: 78 TFLOP/s at 3 rounds of the chip, 92 at 18 (768 rows x 300,000 items), 102 at 37.)
int64_t exact_chunk_rows(int N) {
  int64_t rows = ((int64_t)2 << 30) / ((int64_t)N * 4);
  rows &= ~(int64_t)127;
  if (rows < 128) rows = 128;
  if (rows > 8192) rows = 8192;
  return rows;
}

hipError_t launch_exact_logits(const ModelDev& m, const ExactDev& x, const float* hidden, int64_t R, float* logits, hipStream_t st) {
  XGemm g = {};
  g.A = hidden; g.lda = m.Clast; g.Wt = x.outT; g.taps = 1; g.Cin = m.Clast; g.M = R; g.N = m.N; g.bias = m.out_bias;
  g.out = logits; g.ldo = m.N; g.epi = XEPI_LINEAR;
  return launch_xgemm(g, st);
}

hipError_t launch_xrow_stats(const float* logits, int64_t ld, const int32_t* y, int64_t R, int N, int32_t* o_rank, float* o_ce, float* o_tgt, hipStream_t st) {
  if (R <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_xrow_stats, dim3((unsigned)R), dim3(256), 0, st, logits, ld, y, N, o_rank, o_ce, o_tgt);
  return hipGetLastError();
}

hipError_t launch_exact_score(const ModelDev& m, const ExactDev& x, const float* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                              hipStream_t st) {
  const int64_t chunk = x.chunk_rows;
  for (int64_t r0 = 0; r0 < R; r0 += chunk) {
    const int64_t rc = R - r0 < chunk ? R - r0 : chunk;
    hipError_t e = launch_exact_logits(m, x, hidden + r0 * m.Clast, rc, x.logits, st);
    if (e != hipSuccess) return e;
    if (out->rank || out->ce || out->target) {
      hipLaunchKernelGGL(k_xrow_stats, dim3((unsigned)rc), dim3(256), 0, st, x.logits, (int64_t)m.N, y + r0, m.N, out->rank ? out->rank + r0 : nullptr,
                         out->ce ? out->ce + r0 : nullptr, out->target ? out->target + r0 : nullptr);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (out->topk > 0) {
      hipLaunchKernelGGL(k_xrow_topk, dim3((unsigned)rc), dim3(256), 0, st, x.logits, (int64_t)m.N, y + r0, m.N, out->topk,
                         out->topk_idx + r0 * out->topk, out->topk_val + r0 * out->topk);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
  }
  return hipSuccess;
}
