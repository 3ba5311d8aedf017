// Training step of the HierTCN path: model.py:102-117 (masked lane-mean cross
// entropy) differentiated through model_tcn.py:41 (scoring), customized_tcn_cell.py:109-127 (blocks),
// model_tcn.py:35 / model_hier.py:50,54-55 (projection, item embedding), model_hier.py:83-93 (session mean-pool,
// GRU across the S in-batch slots, mask reset) and applied with TF-1.x Adam (model.py:139-141).  No gradient
// crosses the batch boundary: state_in is data (run_hier_xing.py:291,301).
//
// First build, correctness first.  The d-logits matrix [rows, N] is never held whole: the scoring kernel
// re-computes it 64-item stage by stage and writes bf16 chunks G[rows, NC]; the two products per chunk
// (d hidden += G . Wo_chunk^T, d Wo_chunk = hidden^T . G) and every other backward matrix product are plain
// GEMMs on the library's own MFMA kernels (gemm.hip: 16-bit inputs / fp32 accumulate for the low level, exact fp32 for the GRU); the pieces
// around them (masks of the relus, causal shifts, scatter-adds into the item table, gate derivatives, Adam)
// are hand-written, bandwidth-bound kernels.  Gradients pass straight through the bf16 roundings of the
// forward pass.  Both model_hier and the single-level model_tcn (per-block widths, Dense down-sample residual).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "train.h"
#include "exact.h"
#include "gemm.h"

static thread_local char g_terr[512] = "";
const char* train_last_error() { return g_terr; }
static int tfail(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_terr, sizeof(g_terr), fmt, ap);
  va_end(ap);
  return -1;
}
#define T_HIP(expr)                                                                                      \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return tfail("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)
static size_t al64(size_t n) { return (n + 63) & ~(size_t)63; }

size_t train_layout(const ModelDev& m, TrainCtx& t) {
  size_t off = 0;
  auto seg = [&](TrainSeg& s, size_t n) { s.off = off; s.n = n; off += al64(n); };
  seg(t.emb, (size_t)m.N * m.E);
  seg(t.outT, (size_t)m.NP * m.Clast);
  seg(t.out_bias, (size_t)m.NP);
  if (m.hier) {
    seg(t.emb_bias, (size_t)m.E);
    seg(t.projT, (size_t)m.E * m.E);
    seg(t.projS, (size_t)m.E * m.HT);
  }
  for (int i = 0, cin = m.E; i < m.n_blocks && !m.low_gru; cin = m.C[i], i++) {
    seg(t.convT[i], (size_t)m.K * m.C[i] * cin);
    seg(t.conv_bias[i], (size_t)m.C[i]);
    if (cin != m.C[i]) {
      seg(t.resT[i], (size_t)m.C[i] * cin);
      seg(t.res_bias[i], (size_t)m.C[i]);
    }
  }
  for (int l = 0; l < m.nl && m.hier; l++) {   // the across-session GRU (model_hier.py:36-37); the single-level model_gru has none
    const int in = (l == 0) ? m.E : m.H;
    seg(t.wg[l], (size_t)(in + m.H) * 2 * m.H);
    seg(t.bg[l], (size_t)2 * m.H);
    seg(t.wc[l], (size_t)(in + m.H) * m.H);
    seg(t.bc[l], (size_t)m.H);
  }
  for (int l = 0; l < m.nl && m.low_gru; l++) {
    const int in = (l == 0) ? m.E : m.H;
    seg(t.lwg[l], (size_t)(in + m.H) * 2 * m.H);
    seg(t.lbg[l], (size_t)2 * m.H);
    seg(t.lwc[l], (size_t)(in + m.H) * m.H);
    seg(t.lbc[l], (size_t)m.H);
  }
  t.total = off;
  return off;
}

hipError_t train_enable(ModelDev& m, TrainCtx& t) {
#ifdef HIERTCN_F16
  // d-logits of ~1e-10 underflow in half precision; the f16 build is an inference build (no loss scaling here)
  return hipErrorNotSupported;
#endif
  train_layout(m, t);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&t.arena), 4 * t.total * sizeof(float));
  if (e != hipSuccess) return e;
  if ((e = hipMemset(t.arena, 0, 4 * t.total * sizeof(float))) != hipSuccess) return e;
  {  // [tap][Cin][Cout] / [Ein][Eout] copies of the 16-bit kernels: the k-contiguous B operand of d a_in = d c . W^T (gemm16_nt)
    size_t n = m.hier ? (size_t)m.E * m.E : 0;
    for (int i = 0, cin = m.E; i < m.n_blocks && !m.low_gru; cin = m.C[i], i++) n += (size_t)m.K * cin * m.C[i] + (cin != m.C[i] ? (size_t)cin * m.C[i] : 0);
    if ((e = hipMalloc(reinterpret_cast<void**>(&t.dlay), (n ? n : 8) * sizeof(uint16_t))) != hipSuccess) return e;
    uint16_t* p = t.dlay;
    if (m.hier) { t.projD = p; p += (size_t)m.E * m.E; }
    for (int i = 0, cin = m.E; i < m.n_blocks && !m.low_gru; cin = m.C[i], i++) {
      t.convD[i] = p; p += (size_t)m.K * cin * m.C[i];
      if (cin != m.C[i]) { t.resD[i] = p; p += (size_t)cin * m.C[i]; }
    }
  }
  // the fp32-stored tensors ARE their masters from now on (the conversion kernels of set_weights write into the arena)
  float* M = t.master();
  m.out_bias = M + t.out_bias.off;
  if (m.hier) {
    m.emb_bias = M + t.emb_bias.off;
    m.projS = M + t.projS.off;
  }
  for (int i = 0; i < m.n_blocks && !m.low_gru; i++) {
    m.conv_bias[i] = M + t.conv_bias[i].off;
    if (t.resT[i].n) m.res_bias[i] = M + t.res_bias[i].off;
  }
  for (int l = 0; l < m.nl; l++) {
    if (m.hier) {
      m.gru_wg[l] = M + t.wg[l].off; m.gru_bg[l] = M + t.bg[l].off;
      m.gru_wc[l] = M + t.wc[l].off; m.gru_bc[l] = M + t.bc[l].off;
    }
    if (m.low_gru) {
      m.lgru_wg[l] = M + t.lwg[l].off; m.lgru_bg[l] = M + t.lbg[l].off;
      m.lgru_wc[l] = M + t.lwc[l].off; m.lgru_bc[l] = M + t.lbc[l].off;
    }
  }
  t.enabled = true;
  t.step = 0;
  return hipSuccess;
}

void train_destroy(TrainCtx& t) {
  if (t.arena) (void)hipFree(t.arena);
  if (t.dlay) (void)hipFree(t.dlay);
  t.arena = nullptr; t.dlay = nullptr; t.enabled = false;
}

// ---------------------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------------------
__global__ void k_cast_bf16(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n, int64_t zero_first) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
#pragma unroll
    for (int j = 0; j < 4; j++)
      if (i + j < n) dst[i + j] = (i + j < zero_first) ? (uint16_t)0 : f2bf(src[i + j]);
  }
}
static hipError_t cast_bf16(const float* src, uint16_t* dst, int64_t n, int64_t zero_first, hipStream_t st) {
  if (n <= 0) return hipSuccess;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_cast_bf16, dim3((unsigned)blocks), dim3(256), 0, st, src, dst, n, zero_first);
  return hipGetLastError();
}

hipError_t train_refresh_bf16(const ModelDev& m, const TrainCtx& t, hipStream_t st) {
  const float* M = t.master();
  hipError_t e;
  if ((e = cast_bf16(M + t.emb.off, m.emb, (int64_t)t.emb.n, m.E, st)) != hipSuccess) return e;  // row 0 = null id -> zeros (model.py:59)
  if ((e = cast_bf16(M + t.outT.off, m.outT, (int64_t)t.outT.n, 0, st)) != hipSuccess) return e;
  if (m.hier && (e = cast_bf16(M + t.projT.off, m.projT, (int64_t)t.projT.n, 0, st)) != hipSuccess) return e;
  for (int i = 0; i < m.n_blocks && !m.low_gru; i++) {
    if ((e = cast_bf16(M + t.convT[i].off, m.convT[i], (int64_t)t.convT[i].n, 0, st)) != hipSuccess) return e;
    if (t.resT[i].n && (e = cast_bf16(M + t.resT[i].off, m.resT[i], (int64_t)t.resT[i].n, 0, st)) != hipSuccess) return e;
  }
  // the transposed copies the backward's d a_in products read (masters are [tap][Cout][Cin] / [Eout][Ein])
  if (m.hier && (e = launch_transpose_to_bf16(M + t.projT.off, t.projD, m.E, m.E, m.E, st)) != hipSuccess) return e;
  for (int i = 0, cin = m.E; i < m.n_blocks && !m.low_gru; cin = m.C[i], i++) {
    const int cout = m.C[i];
    if ((e = launch_transpose_to_bf16(M + t.convT[i].off, t.convD[i], cout, cin, cin, st, m.K)) != hipSuccess) return e;   // the K taps in one launch
    if (t.resT[i].n && (e = launch_transpose_to_bf16(M + t.resT[i].off, t.resD[i], cout, cin, cin, st)) != hipSuccess) return e;
  }
  if ((e = refresh_lgru_split(m, st)) != hipSuccess) return e;
  return refresh_frag_major(m, st);
}

hipError_t train_fill_masters(const ModelDev& m, TrainCtx& t, const hiertcn_weights_t* w, hipStream_t st) {
  float* M = t.master();
  hipError_t e;
  const int E = m.E;
  if ((e = hipMemcpyAsync(M + t.emb.off, w->emb_kernel, t.emb.n * sizeof(float), hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
  if ((e = hipMemsetAsync(M + t.outT.off, 0, t.outT.n * sizeof(float), st)) != hipSuccess) return e;
  if ((e = launch_transpose_f32(w->out_kernel, M + t.outT.off, m.Clast, m.N, st)) != hipSuccess) return e;  // [C][N] -> [N][C]
  if (m.hier && (e = launch_transpose_f32(w->proj_kernel, M + t.projT.off, E, E, st)) != hipSuccess) return e;  // rows [0,E) -> [Eout][Ein]
  for (int i = 0, cin = E; i < m.n_blocks && !m.low_gru; cin = m.C[i], i++) {
    const int cout = m.C[i];
    for (int j = 0; j < m.K; j++)
      if ((e = launch_transpose_f32(w->conv_kernel[i] + (size_t)j * cin * cout, M + t.convT[i].off + (size_t)j * cin * cout, cin, cout, st)) !=
          hipSuccess)
        return e;
    if (t.resT[i].n && (e = launch_transpose_f32(w->res_kernel[i], M + t.resT[i].off, cin, cout, st)) != hipSuccess) return e;  // [Cin][Cout] -> [Cout][Cin]
  }
  t.step = 0;
  if ((e = hipMemsetAsync(t.mom1(), 0, 2 * t.total * sizeof(float), st)) != hipSuccess) return e;  // m and v
  return train_refresh_bf16(m, t, st);   // every 16-bit copy from the masters, incl. the transposed ones only the backward reads
}

// act[b] = number of valid steps of lane b; users += 1 for every lane with at least one (model.py:112-113)
__global__ __launch_bounds__(256) void k_lane_act(const int32_t* __restrict__ y, int T, float* __restrict__ act, float* __restrict__ users) {
  __shared__ int red[256];
  const int b = blockIdx.x;
  int c = 0;
  for (int t = threadIdx.x; t < T; t += 256) c += y[(int64_t)b * T + t] > 0 ? 1 : 0;
  red[threadIdx.x] = c;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    act[b] = (float)red[0];
    if (red[0] > 0) atomicAdd(users, 1.0f);
  }
}

// w[r] = d loss / d ce[r] = mask / ((act_b + 1e-6) * users) (model.py:111-117);  lse[r] = ce[r] + target[r]
__global__ void k_row_weights(const int32_t* __restrict__ y, const float* __restrict__ act, const float* __restrict__ users,
                              const float* __restrict__ ce, const float* __restrict__ tgt, int64_t R, int T, float* __restrict__ w,
                              float* __restrict__ lse, int64_t r0 = 0) {   // rows [r0, R): the fp32 training mode's row chunks
  const int64_t r = r0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  const bool valid = y[r] > 0;
  const float u = users[0];
  w[r] = (valid && u > 0.f) ? 1.0f / ((act[r / T] + 1e-6f) * u) : 0.f;
  lse[r] = valid ? ce[r] + tgt[r] : 0.f;
}

// compact row p = row map[p]: hidden row, target id, weight and log-sum-exp gathered into dense arrays
__global__ void k_gather_scored(const uint16_t* __restrict__ hidden, const int32_t* __restrict__ y, const float* __restrict__ w,
                                const float* __restrict__ lse, const int32_t* __restrict__ map, int64_t Rv, int C, uint16_t* __restrict__ hc,
                                int32_t* __restrict__ yc, float* __restrict__ wc, float* __restrict__ lc) {
  const int ch = C >> 3;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Rv * ch) return;
  const int64_t p = i / ch;
  const int c = (int)(i - p * ch);
  const int64_t r = map[p];
  *reinterpret_cast<uint4*>(hc + p * C + c * 8) = *reinterpret_cast<const uint4*>(hidden + r * C + c * 8);
  if (c == 0) { yc[p] = y[r]; wc[p] = w[r]; lc[p] = lse[r]; }
}
// dst[map[p]][:] = src[p][:]  (dst zeroed beforehand: padded steps have no gradient)
__global__ void k_scatter_rows_f32(const float* __restrict__ src, const int32_t* __restrict__ map, int64_t Rv, int C, float* __restrict__ dst) {
  const int ch = C >> 2;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Rv * ch) return;
  const int64_t p = i / ch;
  const int c = (int)(i - p * ch) * 4;
  *reinterpret_cast<float4*>(dst + (int64_t)map[p] * C + c) = *reinterpret_cast<const float4*>(src + p * C + c);
}

// out[c] += sum over rows of bf16 src[r][c]  (d bias of a TemporalBlock / of the scoring layer from a G chunk).  512 rows per workgroup, four
// independent partial sums per thread: at 4096 rows and one dependent chain of 512 loads per thread a [19,200 x 128] call (the reference's 32
// lanes) ran 134 us on 20 workgroups.
#define COLSUM16_ROWS 512
__global__ __launch_bounds__(256) void k_colsum_bf16(const uint16_t* __restrict__ src, int64_t rows, int64_t ld, int cols,
                                                     float* __restrict__ out) {
  __shared__ float red[8][64];
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;  // 32 column pairs x 8 row lanes
  const int c0 = blockIdx.x * 64 + cx * 2;
  const int64_t r0 = (int64_t)blockIdx.y * COLSUM16_ROWS;
  const int64_t r1 = (r0 + COLSUM16_ROWS < rows) ? r0 + COLSUM16_ROWS : rows;
  float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f, u0 = 0.f, u1 = 0.f, v0 = 0.f, v1 = 0.f;
  if (c0 < cols) {
    int64_t r = r0 + ry;
    for (; r + 24 < r1; r += 32) {
      const uint32_t a = *reinterpret_cast<const uint32_t*>(src + r * ld + c0);
      const uint32_t b = *reinterpret_cast<const uint32_t*>(src + (r + 8) * ld + c0);
      const uint32_t c = *reinterpret_cast<const uint32_t*>(src + (r + 16) * ld + c0);
      const uint32_t d = *reinterpret_cast<const uint32_t*>(src + (r + 24) * ld + c0);
      s0 += bf2f((uint16_t)(a & 0xffff)); s1 += bf2f((uint16_t)(a >> 16));
      t0 += bf2f((uint16_t)(b & 0xffff)); t1 += bf2f((uint16_t)(b >> 16));
      u0 += bf2f((uint16_t)(c & 0xffff)); u1 += bf2f((uint16_t)(c >> 16));
      v0 += bf2f((uint16_t)(d & 0xffff)); v1 += bf2f((uint16_t)(d >> 16));
    }
    for (; r < r1; r += 8) {
      const uint32_t a = *reinterpret_cast<const uint32_t*>(src + r * ld + c0);
      s0 += bf2f((uint16_t)(a & 0xffff)); s1 += bf2f((uint16_t)(a >> 16));
    }
  }
  red[ry][cx * 2] = (s0 + t0) + (u0 + v0);
  red[ry][cx * 2 + 1] = (s1 + t1) + (u1 + v1);
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) s += red[i][threadIdx.x];
    const int c = blockIdx.x * 64 + threadIdx.x;
    if (c < cols) atomicAdd(out + c, s);
  }
}

// out[c] += sum over rows of fp32 src[r * ld + c].  256 rows per workgroup: a [20,480 x 256] gate gradient is 320 workgroups (it was 40 at 2048
// rows each: 86 us per call on a quarter of the chip, 17 ms of the HierGRU training step over its 200 calls), four independent partial sums
// per thread so that the loads of a thread overlap.
#define COLSUM_ROWS 256
__global__ __launch_bounds__(256) void k_colsum_f32(const float* __restrict__ src, int64_t rows, int64_t ld, int cols, float* __restrict__ out) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ry = threadIdx.x >> 6;
  __shared__ float red[4][64];
  const int64_t r0 = (int64_t)blockIdx.y * COLSUM_ROWS;
  const int64_t r1 = (r0 + COLSUM_ROWS < rows) ? r0 + COLSUM_ROWS : rows;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < cols) {
    int64_t r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      s0 += src[r * ld + c]; s1 += src[(r + 4) * ld + c]; s2 += src[(r + 8) * ld + c]; s3 += src[(r + 12) * ld + c];
    }
    for (; r < r1; r += 4) s0 += src[r * ld + c];
  }
  red[ry][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x < 64 && c < cols) atomicAdd(out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
static hipError_t colsum_f32(const float* src, int64_t rows, int64_t ld, int cols, float* out, hipStream_t st) {
  hipLaunchKernelGGL(k_colsum_f32, dim3((cols + 63) / 64, (unsigned)((rows + COLSUM_ROWS - 1) / COLSUM_ROWS)), dim3(256), 0, st, src, rows, ld, cols, out);
  return hipGetLastError();
}
// several column sums over the SAME number of rows in one launch (the bias gradients of a level's cells: 2 nl of them, each a launch of its own before)
struct ColsumDesc { const float* src; int64_t ld; int cols; float* out; };
struct ColsumGroup { ColsumDesc d[2 * HIERTCN_MAX_LAYERS]; int n; };
__global__ __launch_bounds__(256) void k_colsum_f32_group(const ColsumGroup g, int64_t rows) {
  const ColsumDesc& d = g.d[blockIdx.z];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int ry = threadIdx.x >> 6;
  __shared__ float red[4][64];
  const int64_t r0 = (int64_t)blockIdx.y * COLSUM_ROWS;
  const int64_t r1 = (r0 + COLSUM_ROWS < rows) ? r0 + COLSUM_ROWS : rows;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < d.cols) {
    int64_t r = r0 + ry;
    for (; r + 12 < r1; r += 16) {
      s0 += d.src[r * d.ld + c]; s1 += d.src[(r + 4) * d.ld + c]; s2 += d.src[(r + 8) * d.ld + c]; s3 += d.src[(r + 12) * d.ld + c];
    }
    for (; r < r1; r += 4) s0 += d.src[r * d.ld + c];
  }
  red[ry][threadIdx.x & 63] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (threadIdx.x < 64 && c < d.cols) atomicAdd(d.out + c, red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
}
static hipError_t colsum_f32_group(const ColsumGroup& g, int64_t rows, hipStream_t st) {
  if (g.n <= 0) return hipSuccess;
  int most = 0;
  for (int i = 0; i < g.n; i++) most = g.d[i].cols > most ? g.d[i].cols : most;
  hipLaunchKernelGGL(k_colsum_f32_group, dim3((most + 63) / 64, (unsigned)((rows + COLSUM_ROWS - 1) / COLSUM_ROWS), (unsigned)g.n), dim3(256), 0, st, g, rows);
  return hipGetLastError();
}

// TemporalBlock backward, pointwise part (customized_tcn_cell.py:120-127): out = relu(c + res), c = relu(conv + b)
//   ds = da * [out > 0]   (grad of the pre-activation sum; also the grad of the identity residual)
//   dc = ds * [c > 0]     (grad of conv + b)
__global__ void k_block_bwd(const float* __restrict__ da, const uint16_t* __restrict__ aout, const uint16_t* __restrict__ cact, int64_t n,
                            float* __restrict__ ds, uint16_t* __restrict__ dc) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float4 g = *reinterpret_cast<const float4*>(da + i);
  const uint2 ao = *reinterpret_cast<const uint2*>(aout + i);
  const uint2 co = *reinterpret_cast<const uint2*>(cact + i);
  const float gs[4] = {g.x, g.y, g.z, g.w};
  const uint16_t a4[4] = {(uint16_t)(ao.x & 0xffff), (uint16_t)(ao.x >> 16), (uint16_t)(ao.y & 0xffff), (uint16_t)(ao.y >> 16)};
  const uint16_t c4[4] = {(uint16_t)(co.x & 0xffff), (uint16_t)(co.x >> 16), (uint16_t)(co.y & 0xffff), (uint16_t)(co.y >> 16)};
  float s4[4], d4[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    s4[j] = bf2f(a4[j]) > 0.f ? gs[j] : 0.f;
    d4[j] = bf2f(c4[j]) > 0.f ? s4[j] : 0.f;
  }
  *reinterpret_cast<float4*>(ds + i) = make_float4(s4[0], s4[1], s4[2], s4[3]);
  uint2 o;
  o.x = pack2bf(d4[0], d4[1]);
  o.y = pack2bf(d4[2], d4[3]);
  *reinterpret_cast<uint2*>(dc + i) = o;
}

// table[ids[r]][:] += rows[r][:] (fp32 atomics; id 0 = null id has no embedding: model.py:59).  One wave per row, lane e
// takes channels e, e + 64, ...: every atomic instruction of the wave covers 256 contiguous bytes of the destination row
// (a thread-per-float4 layout issues four strided atomics per lane and takes 4 x as long on the Zipf-distributed ids).
__global__ __launch_bounds__(256) void k_scatter_add_rows(float* __restrict__ table, const int32_t* __restrict__ ids, const float* __restrict__ rows,
                                                          int64_t R, int E, int N) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int id = ids[r];
  if (id <= 0 || id >= N) return;   // null / out-of-range ids have no embedding row (the arena's next segment starts right behind the table)
  const float* src = rows + r * E;
  float* d = table + (int64_t)id * E;
  for (int e = lane; e < E; e += 64) atomicAdd(d + e, src[e]);
}

// dst[(b*(S+1) + s)][e] = sum_t src[((b*S + s)*L + t)][e]; row s == S of every lane = 0  (d sproj, laid out like the state history)
__global__ void k_segment_sum(const float* __restrict__ src, int B, int S, int L, int E, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * (S + 1) * E) return;
  const int e = (int)(i % E);
  const int64_t bs = i / E;
  const int s = (int)(bs % (S + 1));
  const int64_t b = bs / (S + 1);
  float acc = 0.f;
  if (s < S) {
    const float* p = src + ((b * S + s) * L) * (int64_t)E + e;
    for (int t = 0; t < L; t++) acc += p[(int64_t)t * E];
  }
  dst[i] = acc;
}

// session mean-pool backward (model_hier.py:83-85): table[y[sess][t]] += dpooled[sess] / cnt for valid t
__global__ __launch_bounds__(256) void k_pool_bwd(float* __restrict__ table, const int32_t* __restrict__ y, const float* __restrict__ dpooled,
                                                  int64_t n_sessions, int L, int E, int N) {
  const int64_t sess = blockIdx.x;
  if (sess >= n_sessions) return;
  int cnt = 0;
  for (int t = 0; t < L; t++) cnt += (y[sess * L + t] > 0 && y[sess * L + t] < N) ? 1 : 0;
  if (cnt == 0) return;
  const float inv = 1.0f / (float)cnt;
  for (int e = threadIdx.x; e < E; e += 256) {
    const float g = dpooled[sess * E + e] * inv;
    for (int t = 0; t < L; t++) {
      const int id = y[sess * L + t];
      if (id > 0 && id < N) atomicAdd(table + (int64_t)id * E + e, g);
    }
  }
}

// ---- GRU backward, elementwise pieces (customed_gru_cell.py:323-337 differentiated) ----------------------
// dhp: grad of the cell's (unmasked) output h' [B][H] (row stride ldd); extra: optional second grad term added to it.
// gates [B][2H] = (r, u); cand c; h = old state slice (row stride ldh).
//   dac = dh' (1-u) (1-c^2);  dag[:, H:] = dh' (h - c) u (1-u);  dhpart = dh' u
__global__ void k_gru_bwd_a(const float* __restrict__ dhp, int ldd, const float* __restrict__ extra, const float* __restrict__ h, int ldh,
                            const float* __restrict__ gates, const float* __restrict__ cand, int B, int H, float* __restrict__ dac,
                            float* __restrict__ dag, float* __restrict__ dhpart) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int64_t b = i / H;
  const int k = (int)(i - b * H);
  const float d = dhp[b * ldd + k] + (extra ? extra[i] : 0.f);
  const float u = gates[b * 2 * H + H + k], c = cand[i], hv = h[b * ldh + k];
  dac[i] = d * (1.f - u) * (1.f - c * c);
  dag[b * 2 * H + H + k] = d * (hv - c) * u * (1.f - u);
  dhpart[i] = d * u;
}
// the same with k_gru_out_grad folded in (no layer above: the top layer of a slot): dh' = (carried + state-projection path) * mask
__global__ void k_gru_bwd_a_top(const float* __restrict__ dcar, int ldc, const float* __restrict__ dsb, int64_t ld_dsb, const float* __restrict__ mask, int ldm,
                                const float* __restrict__ h, int ldh, const float* __restrict__ gates, const float* __restrict__ cand, int B, int H,
                                float* __restrict__ dac, float* __restrict__ dag, float* __restrict__ dhpart) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int64_t b = i / H;
  const int k = (int)(i - b * H);
  const float d = (dcar[b * ldc + k] + dsb[b * ld_dsb + k]) * mask[b * ldm];
  const float u = gates[b * 2 * H + H + k], c = cand[i], hv = h[b * ldh + k];
  dac[i] = d * (1.f - u) * (1.f - c * c);
  dag[b * 2 * H + H + k] = d * (hv - c) * u * (1.f - u);
  dhpart[i] = d * u;
}
// drh = dcat_c[:, in:] (grad of r*h);  dag[:, :H] = drh h r (1-r);  dhpart += drh r
__global__ void k_gru_bwd_b(const float* __restrict__ dcat_c, int K, int in, const float* __restrict__ h, int ldh,
                            const float* __restrict__ gates, int B, int H, float* __restrict__ dag, float* __restrict__ dhpart) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int64_t b = i / H;
  const int k = (int)(i - b * H);
  const float drh = dcat_c[b * K + in + k];
  const float r = gates[b * 2 * H + k], hv = h[b * ldh + k];
  dag[b * 2 * H + k] = drh * hv * r * (1.f - r);
  dhpart[i] += drh * r;
}
// din = dcat_c[:, :in] + dcat_g[:, :in] -> dst_in (row stride ldi);  dh = dhpart + dcat_g[:, in:] -> dst_h (row stride ldo)
__global__ void k_gru_bwd_c(const float* __restrict__ dcat_c, const float* __restrict__ dcat_g, int K, int in, const float* __restrict__ dhpart,
                            int B, int H, float* __restrict__ dst_in, int ldi, float* __restrict__ dst_h, int ldo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = in > H ? in : H;
  if (i >= (int64_t)B * W) return;
  const int64_t b = i / W;
  const int k = (int)(i - b * W);
  if (k < in) dst_in[b * ldi + k] = dcat_c[b * K + k] + dcat_g[b * K + k];
  if (k < H) dst_h[b * ldo + k] = dhpart[b * H + k] + dcat_g[b * K + in + k];
}
// HierGRU low level under dynamic_rnn (model_gru.py:47-53): a row past its sequence_length neither produced an output nor moved
// its state at this step -- no gradient enters the cell (d = 0) and the carried state gradient passes through unchanged.
__global__ void k_lgru_bwd_a(const float* __restrict__ dcar, int ldd, const float* __restrict__ dout, int64_t ld_dout,
                             const int32_t* __restrict__ seq_len, int tstep, const float* __restrict__ h, int ldh,
                             const float* __restrict__ gates, const float* __restrict__ cand, int M, int H, float* __restrict__ dac,
                             float* __restrict__ dag, float* __restrict__ dhpart) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * H) return;
  const int64_t b = i / H;
  const int k = (int)(i - b * H);
  const bool active = tstep < seq_len[b];
  const float d = active ? dcar[b * ldd + k] + dout[b * ld_dout + k] : 0.f;
  const float u = gates[b * 2 * H + H + k], c = cand[i], hv = h[b * ldh + k];
  dac[i] = d * (1.f - u) * (1.f - c * c);
  dag[b * 2 * H + H + k] = d * (hv - c) * u * (1.f - u);
  dhpart[i] = d * u;
}
__global__ void k_lgru_bwd_c(const float* __restrict__ dcat_c, const float* __restrict__ dcat_g, int K, int in, const float* __restrict__ dhpart,
                             int M, int H, float* __restrict__ dst_in, int64_t ldi, float* __restrict__ dst_h, int ldo,
                             const float* __restrict__ dcar, const int32_t* __restrict__ seq_len, int tstep) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int W = in > H ? in : H;
  if (i >= (int64_t)M * W) return;
  const int64_t b = i / W;
  const int k = (int)(i - b * W);
  const bool active = tstep < seq_len[b];
  if (k < in) dst_in[b * ldi + k] = dcat_c[b * K + k] + dcat_g[b * K + k];   // zero for an inactive row (its dac / dag rows are zero)
  if (k < H) dst_h[b * ldo + k] = active ? dhpart[b * H + k] + dcat_g[b * K + in + k] : dcar[b * ldo + k];
}
// grad of the unmasked cell outputs of slot s: (carried + state-projection path) * mask  (model_hier.py:93)
__global__ void k_gru_out_grad(const float* __restrict__ carried, const float* __restrict__ dsb, int64_t ld_dsb, const float* __restrict__ mask,
                               int ldm, int B, int HT, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * HT) return;
  const int64_t b = i / HT;
  const int k = (int)(i - b * HT);
  out[i] = (carried[i] + dsb[b * ld_dsb + k]) * mask[b * ldm];
}

__global__ void k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m1, float* __restrict__ m2, size_t n, float lr_t,
                       float b1, float b2, float eps, float gscale) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    const float gv = g[i] * gscale;
    const float a = b1 * m1[i] + (1.f - b1) * gv;
    const float v = b2 * m2[i] + (1.f - b2) * gv * gv;
    m1[i] = a;
    m2[i] = v;
    const float upd = lr_t * a / (sqrtf(v) + eps);
    if (upd != 0.f) p[i] -= upd;   // (keeps -inf pad biases -inf: -inf - 0 would also be fine, NaN from inf*0 never arises)
  }
}

hipError_t train_adam(const ModelDev& m, TrainCtx& t, float lr, float beta1, float beta2, float eps, float grad_scale, hipStream_t st) {
  t.step += 1;
  const double tt = (double)t.step;
  const float lr_t = (float)(lr * sqrt(1.0 - pow((double)beta2, tt)) / (1.0 - pow((double)beta1, tt)));
  size_t blocks = (t.total + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, st, t.master(), t.grad(), t.mom1(), t.mom2(), t.total, lr_t, beta1, beta2, eps,
                     grad_scale);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  return train_refresh_bf16(m, t, st);
}

hipError_t train_export(const ModelDev& m, const TrainCtx& t, const hiertcn_weights_t* d, int which, hipStream_t st) {
  const float* A = which == 0 ? t.master() : t.grad();
  hipError_t e;
  const int E = m.E, H = m.H;
  auto cp = [&](const float* dst, const TrainSeg& s, size_t n) -> hipError_t {
    return dst ? hipMemcpyAsync(const_cast<float*>(dst), A + s.off, n * sizeof(float), hipMemcpyDeviceToDevice, st) : hipSuccess;
  };
  auto tr = [&](const float* dst, const float* src, int64_t rows, int64_t cols) -> hipError_t {  // src [rows][cols] -> dst [cols][rows]
    return dst ? launch_transpose_f32(src, const_cast<float*>(dst), rows, cols, st) : hipSuccess;
  };
  if ((e = cp(d->emb_kernel, t.emb, t.emb.n)) != hipSuccess) return e;
  if ((e = tr(d->out_kernel, A + t.outT.off, m.N, m.Clast)) != hipSuccess) return e;   // [N][C] -> [C][N]
  if ((e = cp(d->out_bias, t.out_bias, (size_t)m.N)) != hipSuccess) return e;
  for (int i = 0, cin = E; i < m.n_blocks && !m.low_gru; cin = m.C[i], i++) {
    const int cout = m.C[i];
    for (int j = 0; j < m.K; j++)
      if (d->conv_kernel[i] &&
          (e = tr(d->conv_kernel[i] + (size_t)j * cin * cout, A + t.convT[i].off + (size_t)j * cin * cout, cout, cin)) != hipSuccess)
        return e;
    if ((e = cp(d->conv_bias[i], t.conv_bias[i], (size_t)cout)) != hipSuccess) return e;
    if (t.resT[i].n) {
      if ((e = tr(d->res_kernel[i], A + t.resT[i].off, cout, cin)) != hipSuccess) return e;
      if ((e = cp(d->res_bias[i], t.res_bias[i], (size_t)cout)) != hipSuccess) return e;
    }
  }
  if (m.hier) {
    if ((e = cp(d->emb_bias, t.emb_bias, (size_t)E)) != hipSuccess) return e;
    if (d->proj_kernel) {
      if ((e = tr(d->proj_kernel, A + t.projT.off, E, E)) != hipSuccess) return e;                          // rows [0,E)
      if ((e = tr(d->proj_kernel + (size_t)E * E, A + t.projS.off, E, m.HT)) != hipSuccess) return e;       // [E][HT] -> [HT][E]
    }
    for (int l = 0; l < m.nl; l++) {
      const int K = ((l == 0) ? E : H) + H;
      if ((e = tr(d->gru_gate_kernel[l], A + t.wg[l].off, 2 * H, K)) != hipSuccess) return e;
      if ((e = cp(d->gru_gate_bias[l], t.bg[l], (size_t)2 * H)) != hipSuccess) return e;
      if ((e = tr(d->gru_cand_kernel[l], A + t.wc[l].off, H, K)) != hipSuccess) return e;
      if ((e = cp(d->gru_cand_bias[l], t.bc[l], (size_t)H)) != hipSuccess) return e;
    }
  }
  if (m.low_gru) {
    for (int l = 0; l < m.nl; l++) {
      const int K = ((l == 0) ? E : H) + H;
      {
        if ((e = tr(d->low_gate_kernel[l], A + t.lwg[l].off, 2 * H, K)) != hipSuccess) return e;
        if ((e = cp(d->low_gate_bias[l], t.lbg[l], (size_t)2 * H)) != hipSuccess) return e;
        if ((e = tr(d->low_cand_kernel[l], A + t.lwc[l].off, H, K)) != hipSuccess) return e;
        if ((e = cp(d->low_cand_bias[l], t.lbc[l], (size_t)H)) != hipSuccess) return e;
      }
    }
  }
  return hipSuccess;
}


// ---------------------------------------------------------------------------------------------------------
// data-parallel helpers
// ---------------------------------------------------------------------------------------------------------
hipError_t train_count_users(const int32_t* y, int B, int T, float* act_scratch, float* users, hipStream_t st) {
  hipError_t e = hipMemsetAsync(users, 0, sizeof(float), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_lane_act, dim3(B), dim3(256), 0, st, y, T, act_scratch, users);
  return hipGetLastError();
}

// Row-sparse exchange of the item table's gradient: only rows whose id occurs in the batch are non-zero (scatter-adds of the
// input embedding and of the session mean-pool).  flags[id] = 1 for every id seen -> ascending list (the compaction of metrics.hip).
__global__ void k_flag_ids(const int32_t* __restrict__ ids, int64_t n, int N, int32_t* __restrict__ flags) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int id = ids[i];
  if (id > 0 && id < N) flags[id] = 1;   // (benign race: every writer stores the same value)
}
size_t train_rows_workspace_bytes(const ModelDev& m) {
  const size_t n = ((size_t)m.N + 255) & ~(size_t)255;
  return n * 4 /* flags */ + (n / 1024 + 2 + 255) / 256 * 256 * 4 /* block counts */;
}
hipError_t train_rows_list(const ModelDev& m, const int32_t* x, const int32_t* y, int64_t n_ids, int32_t* ids_out, int32_t* count_out, void* ws,
                           hipStream_t st) {
  const size_t n = ((size_t)m.N + 255) & ~(size_t)255;
  int32_t* flags = reinterpret_cast<int32_t*>(ws);
  int32_t* scratch = flags + n;
  hipError_t e = hipMemsetAsync(flags, 0, n * 4, st);
  if (e != hipSuccess) return e;
  const unsigned blocks = (unsigned)((n_ids + 255) / 256);
  if (x && n_ids > 0) hipLaunchKernelGGL(k_flag_ids, dim3(blocks), dim3(256), 0, st, x, n_ids, m.N, flags);
  if (y && n_ids > 0) hipLaunchKernelGGL(k_flag_ids, dim3(blocks), dim3(256), 0, st, y, n_ids, m.N, flags);
  if ((e = hipGetLastError()) != hipSuccess) return e;
  return launch_compact_rows(flags, (int64_t)m.N, ids_out, count_out, scratch, st);   // ids_out must hold min(N - 1, 2 n_ids) entries
}
// one wave per listed row; lane e takes channels e, e + 64, ...
__global__ __launch_bounds__(256) void k_rows_op(float* __restrict__ table, int E, int op, const int32_t* __restrict__ ids,
                                                 const int32_t* __restrict__ count, int64_t cap, float* __restrict__ rows) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t n = count[0] < cap ? count[0] : cap;
  if (i >= n) return;
  float* d = table + (int64_t)ids[i] * E;
  float* r = rows ? rows + i * E : nullptr;
  for (int e = lane; e < E; e += 64) {
    if (op == 0) r[e] = d[e];
    else if (op == 1) d[e] = 0.f;
    else d[e] += r[e];
  }
}
hipError_t train_rows_op(const ModelDev& m, const TrainCtx& t, int op, const int32_t* ids, const int32_t* count, int64_t cap, float* rows,
                         hipStream_t st) {
  if (cap <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_rows_op, dim3((unsigned)((cap + 3) / 4)), dim3(256), 0, st, t.grad() + t.emb.off, m.E, op, ids, count, cap, rows);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------
// workspace
// ---------------------------------------------------------------------------------------------------------
struct TrainWs {
  size_t pooled, sproj, gru_ws, gru_save, act[2 * HIERTCN_MAX_BLOCKS + 2], rank, ce, tgt, lse, w, lane_act, users;
  size_t dA, dB, dsf, dcb, dub, gchunk, map, nvalid, cscratch, hc, yc, wc, lc, dsp_ext, dsb_ext, dpooled, dstate[2], dhp, dac, dag, dhpart, dcat_c, dcat_g, din_up;
  size_t total;
  size_t lg_x3;   // split-operand scratch of the HierGRU cells (gru_x3.hip)
  size_t lg_pipe;   // exchange slots + flags of the layer-pipelined HierGRU forward (lgru_pipe.hip)
  size_t lg_bwd;    // d x slots + counters of the pipelined backward (lgru_wide.hip, k_lgru_wide_bwd)
  size_t lg_uf, lg_len, lg_save, lg_dcar[2], lg_dac, lg_dag, lg_dhpart, lg_dcat_c, lg_dcat_g, lg_din_up;   // HierGRU low level
  size_t splitk;  // partial outputs of the split-K weight-gradient GEMMs
  size_t splitk_floats;
  size_t rb;      // ln w - lse per compact row (+ one padding stage of -inf), k_dlogits / k_sgrad
  size_t rb2;     // the same x log2e (k_sgrad, rows stationary)
  size_t bl2;     // bo x log2e [NP] (k_sgrad, items stationary)
  size_t ovf;     // per compact row: the forward pass's running sum overflowed (k_sgrad mode 2 -> fix-up launches)
  size_t sgpart;  // small batches: per-part sums / counts of k_sgrad mode 2 (score_bwd.hip, sgrad_forward_part_floats)
  int sgpart_on;
  size_t sgslab;  // small catalogues: d Wo slabs of the row parts of k_sgrad mode 0 (sgrad_dwo_slab_floats)
  int sgslab_on;
  int fused;      // 1: the G-free scoring backward (score_bwd.hip); 0: the d-logits chunks + two products per chunk (gemm.hip)
  int nc_stages;  // stages (of 64 items) per G chunk (item-chunked path: C_last = 512)
  int rc_stages;  // stages (of 64 rows) per G chunk (row-chunked path, score_grad.hip)
  // fp32 training mode (m.train_fp32): fp32 activations a^0..a^nb, c^0..c^(nb-1), embedded inputs, materialised logits [R][NP], two shift scratches
  size_t f_act[HIERTCN_MAX_BLOCKS + 1], f_c[HIERTCN_MAX_BLOCKS], f_emb, f_logits, f_sh[2];
};

// The scoring layer's backward: score_bwd.hip (no d-logits matrix, no host read-back) where the kernel is built for the
// width; HIERTCN_TRAIN_LEGACY=1 selects the previous path (G chunks + two products per chunk on gemm.hip's kernels) for A/B runs.
// HierGRU low level: keep the pre-activation gradients of every (step, layer) cell and form the weight gradients behind the BPTT loop (as the
// across-session scan always does) while that costs at most 8 GiB (the reference's 32 lanes = 640 sessions: 118 MB).  That is also what lets the
// session-tile backward run (lgru_tile.hip / lgru_wide.hip keep every cell's pre-activation gradients).  Round 4: 1 -> 8 GiB -- at 1 GiB batches past
// ~870 lanes x 10 slots x 20 steps (d = 256) fell back to the per-cell launch chain: 27.1 ms per training step at 1024 lanes against 20.0 (2048: 44.2 -> 36.3;
// d = 128, 2048: 26.9 -> 17.0).  tune.lgru_defer_max_gib overrides.
static bool lgru_deferred_wgrad(const ModelDev& m, size_t M, int L) {
  if (m.tune.lgru_defer_max_gib < 0) return false;   // -1: never defer (the per-cell weight-gradient launches)
  const size_t gib = (size_t)m.tune.lgru_defer_max_gib;
  return (size_t)m.nl * L * M * 3 * m.H * 4 <= (gib << 30) && !ABL_ENV("HIERTCN_LGRU_WGRAD_PER_CELL");
}

// score_bwd.hip's fused scoring backward covers every last width hiertcn_create takes.  The implementation it replaced (bf16 d-logits chunks from
// score_grad.hip / k_score + two 16-bit GEMMs per chunk) exists in the ablation build only, behind HIERTCN_TRAIN_LEGACY=1, for A/B runs.
static bool fused_scoring_backward(const ModelDev& m) { return sgrad_supported(m) && !ABL_ENV("HIERTCN_TRAIN_LEGACY"); }

// fp32 training mode: rows of materialised logits per chunk of the scoring layer (tune.train_fp32_chunk_mib MiB of [rows][NP] floats, at least 64 rows)
static int64_t fp32_chunk_rows(const ModelDev& m, int64_t R) {
  const int64_t mib = m.tune.train_fp32_chunk_mib > 0 ? m.tune.train_fp32_chunk_mib : 8192;
  int64_t rows = (mib << 20) / ((int64_t)m.NP * 4);
  if (rows < 64) rows = 64;
  return rows < R ? rows : R;
}

static TrainWs train_ws(const ModelDev& m, int B, int S, int L) {
  TrainWs w;
  memset(&w, 0, sizeof(w));
  w.fused = fused_scoring_backward(m) ? 1 : 0;
  size_t off = 0;
  auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~(size_t)255; return o; };
  const size_t R = (size_t)B * S * L, E = (size_t)m.E, HT = (size_t)m.HT, H = (size_t)m.H;
  size_t C = E;   // widest activation
  for (int i = 0; i < m.n_blocks; i++) C = (size_t)m.C[i] > C ? (size_t)m.C[i] : C;
  w.pooled = take((size_t)B * S * E * 4);
  w.sproj = take((size_t)B * S * E * 4);
  w.gru_ws = take(m.hier ? gru_workspace_bytes(m, B, S) : 0);
  w.gru_save = take(m.hier ? gru_save_floats(m, B, S) * 4 : 0);
  // activations kept for the backward pass: [0] embedded inputs, [1] projection output a^0, then per block: c_k, a^k
  for (int i = 0; i < 2 + 2 * m.n_blocks; i++) w.act[i] = take(R * C * 2);
  w.rank = take(R * 4); w.ce = take(R * 4); w.tgt = take(R * 4); w.lse = take(R * 4); w.w = take(R * 4);
  w.lane_act = take((size_t)B * 4);
  w.users = take(256);
  w.dA = take(R * C * 4); w.dB = take(R * C * 4); w.dsf = take((R + 64) * C * 4);
  w.dcb = take(R * C * 2); w.dub = take(R * C * 2);
#if HIERTCN_ABLATION   // the legacy scoring backward's d-logits chunk (HIERTCN_TRAIN_LEGACY=1)
  // G chunk: at most 16 GiB of bf16 (the two products per chunk run 12 % faster than with 4 GiB chunks: fewer, larger
  // GEMMs -- profiles/r1_train_notes.md), at least one 64-item stage
  size_t budget = (size_t)16 << 30;
  size_t st_cap = 4096;
  if (const char* gb = getenv("HIERTCN_TRAIN_CHUNK_GIB")) {  // experiments: larger chunks (288 GB of HBM allow it)
    const long v = atol(gb);
    if (v >= 1 && v <= 128) budget = (size_t)v << 30;
  }
  size_t st = budget / (R * 64 * 2);
  const size_t nst = (size_t)m.NP / 64;
  if (st < 1) st = 1;
  if (st > nst) st = nst;
  if (st > st_cap) st = st_cap;
  if (const char* ov = getenv("HIERTCN_TRAIN_CHUNK_STAGES")) {  // tests: force several chunks on a small catalogue
    const long v = atol(ov);
    if (v >= 1 && (size_t)v < st) st = (size_t)v;
  }
  w.nc_stages = (int)st;
  size_t gbytes = R * st * 64 * 2;
  if (dlogits_supported(m)) {
    // row-chunked path: G[rows of the chunk, padded to 64][all items, padded to 512]
    const size_t ldg = (size_t)dlogits_ld(m);
    size_t rs = budget / (ldg * 2 * 64);
    const size_t rst = (R + 63) / 64;
    if (rs < 1) rs = 1;
    if (rs > rst) rs = rst;
    if (const char* ov = getenv("HIERTCN_TRAIN_CHUNK_STAGES")) {
      const long v = atol(ov);
      if (v >= 1 && (size_t)v < rs) rs = (size_t)v;
    }
    w.rc_stages = (int)rs;
    gbytes = rs * 64 * ldg * 2;
  }
  w.gchunk = take(w.fused ? 256 : gbytes);
#else
  w.gchunk = take(256);
#endif
  w.map = take(R * 4); w.nvalid = take(256); w.cscratch = take((R / 1024 + 2) * 4);
  w.hc = take((R + 64) * C * 2); w.yc = take(R * 4); w.wc = take(R * 4); w.lc = take(R * 4);
  w.rb = take((R + 64) * 4);
  w.rb2 = take((R + 64) * 4);
  w.bl2 = take((size_t)m.NP * 4);
  w.ovf = take((R + 64) * 4);
  w.sgpart_on = (w.fused && sgrad_forward_part_floats(m, (int64_t)R) > 0) ? 1 : 0;
  w.sgpart = take(w.sgpart_on ? sgrad_forward_part_floats(m, (int64_t)R) * 4 : 0);
  w.sgslab_on = (w.fused && sgrad_dwo_slab_floats(m, (int64_t)R) > 0) ? 1 : 0;
  w.sgslab = take(w.sgslab_on ? sgrad_dwo_slab_floats(m, (int64_t)R) * 4 : 0);
  {  // partial tiles of the row-sliced weight-gradient GEMMs (gemm.hip): up to 64 slices of the largest [out x in] gradient
    const size_t gw = (size_t)2 * H * (size_t)(E > H ? E : H), cw = (size_t)C * C;
    w.splitk_floats = (size_t)64 * (gw > cw ? gw : cw);
    // (the grouped launch of a level's 4 nl weight-gradient products slices all of them at once: four times the room where that stays under 256 MB)
    if (w.splitk_floats * 4 <= ((size_t)64 << 20)) w.splitk_floats *= 4;
    w.splitk = take(w.splitk_floats * 4);
  }
  w.dsp_ext = take((size_t)B * (S + 1) * E * 4);
  w.dsb_ext = take((size_t)B * (S + 1) * HT * 4);
  w.dpooled = take((size_t)B * S * E * 4);
  w.dstate[0] = take((size_t)B * HT * 4); w.dstate[1] = take((size_t)B * HT * 4);
  w.dhp = take((size_t)B * HT * 4);
  // the gate / candidate pre-activation gradients of EVERY (slot, layer) cell of the across-session scan: [nl][S][B][H] and [nl][S][B][2H] -- the
  // weight gradients are formed behind the scan in one product per layer and operand instead of two per cell
  w.dac = take((size_t)m.nl * S * B * H * 4); w.dag = take((size_t)m.nl * S * B * 2 * H * 4); w.dhpart = take((size_t)2 * B * H * 4);   // (two: the fused epilogue of a cell's last product writes the next cell's while this cell's is read)
  const size_t Kmax = (E > H ? E : H) + H;
  w.dcat_c = take((size_t)B * Kmax * 4); w.dcat_g = take((size_t)B * Kmax * 4); w.din_up = take((size_t)B * H * 4);
  if (m.low_gru) {
    const size_t M = (size_t)B * S;
    w.lg_uf = take(R * E * 4); w.lg_len = take(M * 4);
    w.lg_x3 = take(low_gru_x3_elems(m, (int64_t)M) * 2);
    w.lg_save = take(low_gru_save_floats(m, (int64_t)M, L) * 4);
    w.lg_pipe = take(lgru_pipe_bytes(m, (int64_t)M, L));
    w.lg_bwd = take(lgru_wide_bwd_bytes(m, (int64_t)M, L));
    w.lg_dcar[0] = take(M * HT * 4); w.lg_dcar[1] = take(M * HT * 4);
    const size_t cells = lgru_deferred_wgrad(m, M, L) ? (size_t)m.nl * L : 1;
    w.lg_dac = take(cells * M * H * 4); w.lg_dag = take(cells * M * 2 * H * 4); w.lg_dhpart = take((size_t)m.nl * M * H * 4);   // (per layer: train_lgru_bptt_wavefront)
    w.lg_dcat_c = take((size_t)m.nl * M * Kmax * 4); w.lg_dcat_g = take(M * Kmax * 4); w.lg_din_up = take((size_t)m.nl * M * H * 4);
  }
  if (m.train_fp32) {
    for (int i = 0; i <= m.n_blocks; i++) w.f_act[i] = take(R * C * 4);
    for (int i = 0; i < m.n_blocks; i++) w.f_c[i] = take(R * C * 4);
    w.f_emb = take(R * E * 4);
    w.f_logits = take((size_t)fp32_chunk_rows(m, (int64_t)R) * (size_t)m.NP * 4);   // one row chunk of logits (the reference's own batch: all of it, 532 MB)
    w.f_sh[0] = take(R * C * 4); w.f_sh[1] = take(R * C * 4);
  }
  w.total = off;
  return w;
}

// ---------------------------------------------------------------------------------------------------------
// fp32 training mode (hiertcn_set_precision(FP32) before hiertcn_train_enable; model_hier with either low level -- TemporalConvNet or the GRU stack
// of model_gru.py:46-55, the reference's default (args.py:40) -- and the single-level model_tcn): the step the
// reference takes (model.py:135-141 on fp32 tensors, loss.py:20-21 on materialised logits) -- fp32 forward on the MASTERS (exact.hip's GEMM
// kernel, activations kept in fp32), logits [R][NP] materialised once, d logits formed in place, every product of the backward an exact-fp32
// MFMA GEMM (gemm.hip).  Sized for the reference's own batch (32 lanes x 10 x 20 x 20,778 items: 532 MB of logits -- what TensorFlow holds
// too); the throughput path stays the 16-bit one.  Gradients agree with fp64 autograd on the UNROUNDED weights to ~1e-5 (tests/test_gpu_train_fp32.py).
// ---------------------------------------------------------------------------------------------------------
// g[r][n] = w_r (exp(z[r][n] - lse_r) - [n == y_r]) in place over [R][ld]; rows without a target have w = 0
__global__ void k_xdlogits(float* __restrict__ z, int64_t ld, int64_t R, const int32_t* __restrict__ y, const float* __restrict__ w,
                           const float* __restrict__ lse) {
  const int64_t r = blockIdx.x;   // (rows on x: a chunk of a small catalogue can hold more than 65,535 of them)
  const float wr = w[r], l = lse[r];
  const int yr = y[r];
  float* zr = z + r * ld;
  for (int64_t n = ((int64_t)blockIdx.y * blockDim.x + threadIdx.x) * 4; n < ld; n += (int64_t)gridDim.y * blockDim.x * 4) {
    float4 v = *reinterpret_cast<float4*>(zr + n);
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = wr == 0.f ? 0.f : wr * (expf(o[j] - l) - ((int)(n + j) == yr ? 1.f : 0.f));
    *reinterpret_cast<float4*>(zr + n) = make_float4(o[0], o[1], o[2], o[3]);
  }
}
// TemporalBlock backward, pointwise part, on fp32 activations: ds = da [out > 0], dc = ds [c > 0]
__global__ void k_block_bwd_f32(const float* __restrict__ da, const float* __restrict__ aout, const float* __restrict__ cact, int64_t n,
                                float* __restrict__ ds, float* __restrict__ dc) {
  const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= n) return;
  const float4 g = *reinterpret_cast<const float4*>(da + i), a = *reinterpret_cast<const float4*>(aout + i), c = *reinterpret_cast<const float4*>(cact + i);
  const float4 s4 = make_float4(a.x > 0.f ? g.x : 0.f, a.y > 0.f ? g.y : 0.f, a.z > 0.f ? g.z : 0.f, a.w > 0.f ? g.w : 0.f);
  *reinterpret_cast<float4*>(ds + i) = s4;
  *reinterpret_cast<float4*>(dc + i) = make_float4(c.x > 0.f ? s4.x : 0.f, c.y > 0.f ? s4.y : 0.f, c.z > 0.f ? s4.z : 0.f, c.w > 0.f ? s4.w : 0.f);
}
// dst[r] = src[r - shift] inside sequences of L rows (zero where r - shift leaves the sequence); shift > 0: the conv tap's delayed input
// (customized_tcn_cell.py:46-49), shift < 0: the gradient that flows back to an earlier step
__global__ void k_shift_rows_f32(const float* __restrict__ src, int64_t R, int C, int L, int shift, float* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c4 = C >> 2;
  if (i >= R * c4) return;
  const int64_t r = i / c4;
  const int c = (int)(i - r * c4) * 4;
  const int t = (int)(r % L) - shift;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (t >= 0 && t < L) v = *reinterpret_cast<const float4*>(src + (r - shift) * C + c);
  *reinterpret_cast<float4*>(dst + r * C + c) = v;
}
static hipError_t shift_rows_f32(const float* src, int64_t R, int C, int L, int shift, float* dst, hipStream_t st) {
  hipLaunchKernelGGL(k_shift_rows_f32, dim3((unsigned)((R * (C >> 2) + 255) / 256)), dim3(256), 0, st, src, R, C, L, shift, dst);
  return hipGetLastError();
}
// the masters in exact.hip's terms: same layouts ([N][E], [Eout][Ein], [tap][Cout][Cin], [Cout][Cin], [NP][C]), so the fp32 kernels read them as they are
static ExactDev fp32_master_view(const ModelDev& m, const TrainCtx& t) {
  ExactDev x;
  memset(&x, 0, sizeof(x));
  float* M = t.master();
  x.emb = M + t.emb.off;
  x.projT = m.hier ? M + t.projT.off : nullptr;
  for (int i = 0; i < m.n_blocks; i++) {
    x.convT[i] = M + t.convT[i].off;
    x.resT[i] = t.resT[i].n ? M + t.resT[i].off : nullptr;
  }
  x.outT = M + t.outT.off;
  return x;
}

size_t train_workspace_bytes(const ModelDev& m, int B, int S, int L) { return train_ws(m, B, S, L).total; }

// phases: TRAIN_BEGIN = forward + loss + backward of the scoring layer (d out_kernel / d out_bias are final behind it: a data-parallel
// caller starts their collective there), TRAIN_FINISH = the rest of the backward pass; both = the whole step.
// users_global (device, optional): the user count of the GLOBAL batch (all ranks) that normalises the loss (model.py:113-117);
// null = this batch's own count.
int train_forward_backward(const ModelDev& m, TrainCtx& t, const int32_t* x, const int32_t* y_pool, const float* mask, const float* state_in, int B,
                           int S, int L, void* wsp, float* state_out, double* metrics8, double* metric_scratch, hipStream_t st, int phases,
                           const float* users_global, const int32_t* y_score) {
  // y: what the scoring layer, the loss, the metrics and the row weights read (train.h); y_pool: what the session mean-pool reads (model_hier.py:83-85)
  const int32_t* y = y_score ? y_score : y_pool;
  const bool do_begin = (phases & TRAIN_BEGIN) != 0, do_finish = (phases & TRAIN_FINISH) != 0;
  if (!t.enabled) return tfail("training is not enabled on this handle (hiertcn_train_enable before hiertcn_set_weights)");
  const TrainWs w = train_ws(m, B, S, L);
  unsigned char* ws = reinterpret_cast<unsigned char*>(wsp);
  const int E = m.E, H = m.H, HT = m.HT, nb = m.n_blocks, CL = m.Clast;
  int Cmax = E;
  for (int i = 0; i < nb; i++) Cmax = m.C[i] > Cmax ? m.C[i] : Cmax;
  const int64_t n_sess = (int64_t)B * S, R = n_sess * L;
    if (!m.hier && S != 1) return tfail("train step: the single-level model takes one sequence per lane (S = 1)");
  float* G = t.grad();
  if (do_begin) T_HIP(hipMemsetAsync(G, 0, t.total * sizeof(float), st));
#define F32(o) reinterpret_cast<float*>(ws + (o))
#define BF(o) reinterpret_cast<uint16_t*>(ws + (o))

  // ================= forward (activations kept) ==========================================================
  float* pooled = F32(w.pooled);
  float* sproj = F32(w.sproj);
  float* hist = F32(w.gru_ws);   // [B][S+1][HT] at the head of the GRU workspace (pool_gru.hip)
  float* gsave = F32(w.gru_save);
  uint16_t* a_emb = BF(w.act[0]);
  uint16_t* a0 = m.hier ? BF(w.act[1]) : a_emb;   // input of block 0: the projection's output, or the embedded ids themselves
  // equal widths and L <= 32: the fused low-level kernel of the evaluation path, writing the kept activations on its way
  // (bitwise the layer-at-a-time kernels' values; HIERTCN_FORCE_LAYERED=1 takes those instead)
  const bool F = m.train_fp32 != 0;   // the fp32 training mode (see fp32_master_view above)
  const bool fused_fw = !F && tcn_fused_applies(m, L);
  if (do_begin && F) {
    // ---- forward on the masters, everything kept in fp32 (model_hier.py:39-93 as exact.hip computes it) ----
    const ExactDev xv = fp32_master_view(m, t);
    if (!m.hier) {   // single-level model_tcn (model.py:71-73): block 0 reads the embedded ids; model_gru (:66-67): the GRU stack does (lb.uf)
      T_HIP(launch_xembed_rows(xv.emb, x, R, E, m.N, m.low_gru ? F32(w.lg_uf) : F32(w.f_act[0]), st));
    } else {
    T_HIP(launch_xsession_pool(m, xv, y_pool, n_sess, L, pooled, st));
    T_HIP(launch_gru_scan(m, pooled, mask, state_in, B, S, sproj, state_out, nullptr, hist, st, gsave));
    T_HIP(launch_xembed_rows(xv.emb, x, R, E, m.N, F32(w.f_emb), st));
    }
    if (m.hier) {   // the 'emb' projection of [e, state] (model_tcn.py:35 / model_gru.py:46): block 0's input, or the low-level GRU's (HierGRU: kept as lb.uf)
      XGemm g = {};
      g.A = F32(w.f_emb); g.lda = E; g.Wt = xv.projT; g.taps = 1; g.Cin = E; g.M = R; g.N = E;
      g.seq_init = sproj; g.Lseq = L; g.out = m.low_gru ? F32(w.lg_uf) : F32(w.f_act[0]); g.ldo = E; g.epi = XEPI_LINEAR;
      T_HIP(launch_xgemm(g, st));
    }
    for (int i = 0, cin = E; i < nb && !m.low_gru; cin = m.C[i], i++) {
      const int cout = m.C[i];
      const float* res = F32(w.f_act[i]);
      int64_t ldres = cin;
      if (cin != cout) {   // customized_tcn_cell.py:102-106: a biased Dense on the residual path
        XGemm r = {};
        r.A = F32(w.f_act[i]); r.lda = cin; r.Wt = xv.resT[i]; r.taps = 1; r.Cin = cin; r.M = R; r.N = cout; r.bias = m.res_bias[i];
        r.out = F32(w.f_sh[0]); r.ldo = cout; r.epi = XEPI_LINEAR;
        T_HIP(launch_xgemm(r, st));
        res = F32(w.f_sh[0]); ldres = cout;
      }
      XGemm g = {};
      g.A = F32(w.f_act[i]); g.lda = cin; g.Wt = xv.convT[i]; g.taps = m.K; g.dil = 1 << i; g.L = L; g.Cin = cin; g.M = R; g.N = cout;
      g.bias = m.conv_bias[i]; g.res = res; g.ldres = ldres; g.out = F32(w.f_act[i + 1]); g.out2 = F32(w.f_c[i]); g.ldo = cout; g.epi = XEPI_BLOCK;
      T_HIP(launch_xgemm(g, st));
    }
  }
  if (do_begin && !F) {
  if (!m.low_gru && !fused_fw) T_HIP(launch_embed_rows(m, x, R, a_emb, st));
  if (m.hier) {
    T_HIP(launch_session_pool(m, y_pool, n_sess, L, pooled, st));
    T_HIP(launch_gru_scan(m, pooled, mask, state_in, B, S, sproj, state_out, nullptr, hist, st, gsave));
    if (!m.low_gru && !fused_fw) {
      LayerArgs a = {};
      a.in = a_emb; a.out = a0; a.W = m.projT; a.Wf = m.projF; a.seq_init = sproj;
      a.Cin = E; a.Cout = E; a.taps = 1; a.dil = 0; a.L = L; a.n_seq = n_sess; a.linear = 1;
      T_HIP(launch_tcn_layer(a, st));
    }
  }
  }
  LowGruBufs lb;
  memset(&lb, 0, sizeof(lb));
  if (m.low_gru) {   // HierGRU: embedding + projection + the GRU stack over the L steps, every state and gate kept (model_gru.py:46-55)
    lb.a_emb = a_emb; lb.u16 = a0; lb.uf = F32(w.lg_uf); lb.seq_len = reinterpret_cast<int32_t*>(ws + w.lg_len);
    lb.states = F32(w.lg_save); lb.n_states = L + 1;
    lb.x3 = BF(w.lg_x3);
    lb.pipe = lgru_pipe_bytes(m, n_sess, L) ? ws + w.lg_pipe : nullptr;
    lb.rec = lb.states + (size_t)(L + 1) * n_sess * HT; lb.keep_all = 1;
    if (F) {   // fp32 mode: lb.uf holds the fp32 projection already; the exact-fp32 cell chain (pool_gru.hip) on the masters, records kept, fp32 output rows
      lb.hidden32 = F32(w.f_act[nb]); lb.x3 = nullptr; lb.pipe = nullptr; lb.u16 = nullptr; lb.a_emb = nullptr;
    }
    if (do_begin) T_HIP(launch_low_gru_bufs(m, x, sproj, n_sess, L, F ? nullptr : BF(w.act[2]), lb, st));
  }
  auto a_of = [&](int k) -> uint16_t* { return k == 0 ? a0 : BF(w.act[1 + 2 * k]); };   // a^k: output of block k-1
  if (fused_fw && do_begin) {
    TcnSaves sv;
    memset(&sv, 0, sizeof(sv));
    sv.emb = a_emb;
    if (m.hier) sv.a[0] = a0;
    for (int i = 0; i < nb; i++) {
      sv.c[i] = BF(w.act[2 + 2 * i]);
      if (i + 1 < nb) sv.a[i + 1] = a_of(i + 1);   // (the last block's output is the kernel's own result)
    }
    T_HIP(launch_tcn_forward(m, x, m.hier ? sproj : nullptr, n_sess, L, a_of(nb), nullptr, st, &sv));
  }
  for (int i = 0, cin = E; i < nb && !m.low_gru && !fused_fw && do_begin && !F; cin = m.C[i], i++) {
    LayerArgs a = {};
    a.in = a_of(i);
    a.c_out = BF(w.act[2 + 2 * i]);
    a.out = a_of(i + 1);
    a.W = m.convT[i]; a.bias = m.conv_bias[i]; a.Wres = m.resT[i]; a.bres = m.res_bias[i]; a.Wf = m.convF[i]; a.Wresf = m.resF[i];
    a.Cin = cin; a.Cout = m.C[i]; a.taps = m.K; a.dil = 1 << i; a.L = L; a.n_seq = n_sess; a.linear = 0;
    T_HIP(launch_tcn_layer(a, st));
  }
  const uint16_t* hidden = m.low_gru ? BF(w.act[2]) : a_of(nb);
  hiertcn_score_out_t so;
  memset(&so, 0, sizeof(so));
  so.rank = reinterpret_cast<int32_t*>(ws + w.rank); so.ce = F32(w.ce); so.target = F32(w.tgt);
  // Only the rows that carry a target are scored and differentiated (a ragged batch is mostly padding): the device builds
  // their ascending list; its length comes back to the host once per step because it sizes the chunk products below.
  int32_t* rmap = reinterpret_cast<int32_t*>(ws + w.map);
  int32_t* nvalid_d = reinterpret_cast<int32_t*>(ws + w.nvalid);
  float* dH = F32(w.dA);   // [R][C_last] fp32
  if (do_begin && F) {
    // ---- scoring layer on MATERIALISED fp32 logits (model_tcn.py:41, model.py:105, loss.py:20-21,179), pad columns carry bias -inf ----
    const ExactDev xv = fp32_master_view(m, t);
    const float* hid32 = F32(w.f_act[nb]);
    float* Z = F32(w.f_logits);
    const int64_t ldz = m.NP;
    // the row weights d loss / d ce depend on the mask only (model.py:111-117): lane activity and the user count first, then the layer in ROW CHUNKS of
    // at most tune.train_fp32_chunk_mib MiB of logits (the reference's own batch is one chunk; larger batches take several -- d Wo^T and d bo accumulate)
    float* users = F32(w.users);
    T_HIP(hipMemsetAsync(users, 0, sizeof(float), st));
    hipLaunchKernelGGL(k_lane_act, dim3(B), dim3(256), 0, st, y, S * L, F32(w.lane_act), users);
    T_HIP(hipGetLastError());
    const int64_t RC = fp32_chunk_rows(m, R);
    for (int64_t r0 = 0; r0 < R; r0 += RC) {
      const int64_t rc = R - r0 < RC ? R - r0 : RC;
      const float* hc32 = hid32 + r0 * CL;
      {
        XGemm g = {};
        g.A = hc32; g.lda = CL; g.Wt = xv.outT; g.taps = 1; g.Cin = CL; g.M = rc; g.N = m.NP; g.bias = m.out_bias; g.out = Z; g.ldo = ldz; g.epi = XEPI_LINEAR;
        T_HIP(launch_xgemm(g, st));
      }
      T_HIP(launch_xrow_stats(Z, ldz, y + r0, rc, m.N, so.rank + r0, so.ce + r0, so.target + r0, st));
      hipLaunchKernelGGL(k_row_weights, dim3((unsigned)((rc + 255) / 256)), dim3(256), 0, st, y, F32(w.lane_act), users_global ? users_global : users, so.ce,
                         so.target, r0 + rc, S * L, F32(w.w), F32(w.lse), r0);
      T_HIP(hipGetLastError());
      // d loss / d logits in place; then the three products: d Wo^T [NP][C] += g^T . h, d bo += column sums, d h = g . Wo^T
      hipLaunchKernelGGL(k_xdlogits, dim3((unsigned)rc, 8), dim3(256), 0, st, Z, ldz, rc, y + r0, F32(w.w) + r0, F32(w.lse) + r0);
      T_HIP(hipGetLastError());
      T_HIP(gemm32_tn(Z, ldz, hc32, CL, G + t.outT.off, CL, m.NP, CL, (int)rc, r0 == 0 ? 0.f : 1.f, F32(w.splitk), w.splitk_floats, st));
      T_HIP(colsum_f32(Z, rc, ldz, m.NP, G + t.out_bias.off, st));
      T_HIP(gemm32_nn(Z, ldz, xv.outT, CL, dH + r0 * CL, CL, (int)rc, CL, m.NP, 0.f, st, F32(w.splitk), w.splitk_floats));   // (few output tiles, K = NP: sliced)
    }
    if (metrics8) T_HIP(launch_metrics_scratch(m, so.rank, so.ce, y, B, S * L, metrics8, metric_scratch, st));
  }
  if (do_begin && !F) {
  T_HIP(launch_compact_rows(y, R, rmap, nvalid_d, reinterpret_cast<int32_t*>(ws + w.cscratch), st));
  T_HIP(hipMemsetAsync(ws + w.rank, 0, w.lse - w.rank, st));   // rank, ce, target: three consecutive slices of the workspace (train_ws), one fill
  float* users = F32(w.users);
  if (w.fused) {
    // score_bwd.hip, mode 2: ONE pass over the logits gives rank / ce / target (the evaluation kernel's outputs) and d hidden; the row
    // weights d loss / d ce depend on the mask only and are needed first
    T_HIP(hipMemsetAsync(users, 0, sizeof(float), st));
    hipLaunchKernelGGL(k_lane_act, dim3(B), dim3(256), 0, st, y, S * L, F32(w.lane_act), users);
    hipLaunchKernelGGL(k_row_weights, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, y, F32(w.lane_act), users_global ? users_global : users, so.ce,
                       so.target, R, S * L, F32(w.w), F32(w.lse));
    T_HIP(hipGetLastError());
    T_HIP(hipMemsetAsync(dH, 0, (size_t)R * CL * 4, st));
    T_HIP(hipMemsetAsync(ws + w.ovf, 0, (size_t)R * 4, st));
    T_HIP(launch_sgrad_forward(m, hidden, y, F32(w.w), rmap, nvalid_d, R, so.rank, so.ce, so.target, dH, reinterpret_cast<int32_t*>(ws + w.ovf), st,
                                w.sgpart_on ? F32(w.sgpart) : nullptr));
    T_HIP(launch_score_fixup_only(m, hidden, y, R, &so, st, rmap, nvalid_d));   // rank / ce of row tiles whose sum overflowed (pathological inputs)
  } else
#if HIERTCN_ABLATION
  T_HIP(launch_score(m, hidden, y, R, &so, nullptr, st, rmap, nvalid_d));
#else
  return tfail("train step: no scoring backward for a last width of %d", CL);   // (not reached: hiertcn_create takes the widths score_bwd.hip covers)
#endif
  if (metrics8) T_HIP(launch_metrics_scratch(m, so.rank, so.ce, y, B, S * L, metrics8, metric_scratch, st));
#if HIERTCN_ABLATION
  int64_t Rv = 0;
  if (!w.fused) {   // legacy path only: the row count sizes the chunk products below
    int32_t nvalid_h = 0;
    T_HIP(hipMemcpyAsync(&nvalid_h, nvalid_d, sizeof(int32_t), hipMemcpyDeviceToHost, st));
    T_HIP(hipStreamSynchronize(st));
    Rv = nvalid_h;
  }
#endif

  // ================= d loss / d logits, scoring layer =====================================================
#if HIERTCN_ABLATION
  if (!w.fused) {   // (the fused path took the row weights before its forward pass and reads ce + target, the log-sum-exp, where it needs it)
    T_HIP(hipMemsetAsync(users, 0, sizeof(float), st));
    hipLaunchKernelGGL(k_lane_act, dim3(B), dim3(256), 0, st, y, S * L, F32(w.lane_act), users);
    hipLaunchKernelGGL(k_row_weights, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, y, F32(w.lane_act), users_global ? users_global : users, so.ce, so.target, R, S * L,
                       F32(w.w), F32(w.lse));
    T_HIP(hipGetLastError());
    T_HIP(hipMemsetAsync(dH, 0, (size_t)R * CL * 4, st));
  }
#endif
  if (w.fused) {
    // score_bwd.hip: everything is sized by R on the host and by the device-side row count inside the kernels.  d hidden came out of
    // the forward pass; what is left is d Wo / d bo from the saved log-sum-exp (and d hidden again for row blocks that overflowed).
    uint16_t* hc = BF(w.hc);
    int32_t* yc = reinterpret_cast<int32_t*>(ws + w.yc);
    T_HIP(launch_sgrad_prepare(m, hidden, y, F32(w.w), so.ce, so.target, rmap, nvalid_d, R, hc, yc, F32(w.wc), F32(w.rb), F32(w.rb2), F32(w.bl2), st));
    T_HIP(launch_sgrad_dh(m, hc, F32(w.rb2), yc, F32(w.wc), rmap, nvalid_d, R, dH, reinterpret_cast<int32_t*>(ws + w.ovf), 1, st));
    T_HIP(launch_sgrad_dwo(m, hc, F32(w.rb), yc, F32(w.wc), nvalid_d, R, F32(w.bl2), G + t.outT.off, G + t.out_bias.off, st,
                            w.sgslab_on ? F32(w.sgslab) : nullptr));
  }
#if HIERTCN_ABLATION
  else if (Rv > 0) {
    uint16_t* hc = BF(w.hc);
    int32_t* yc = reinterpret_cast<int32_t*>(ws + w.yc);
    float* dHc = F32(w.dsf);   // [Rv][C_last] fp32 (the block loop's scratch, free until then)
    hipLaunchKernelGGL(k_gather_scored, dim3((unsigned)((Rv * (CL / 8) + 255) / 256)), dim3(256), 0, st, hidden, y, F32(w.w), F32(w.lse), rmap, Rv, CL,
                       hc, yc, F32(w.wc), F32(w.lc));
    T_HIP(hipGetLastError());
    T_HIP(hipMemsetAsync(dHc, 0, (size_t)Rv * CL * 4, st));
    uint16_t* Gc = BF(w.gchunk);
    const int n_stages = m.NP / 64;
    if (dlogits_supported(m)) {
      // Row chunks (score_grad.hip): G[chunk rows][all items] from the item-stationary kernel, which also leaves d bo behind;
      // then the two plain products of the chunk.
      const int64_t ldg = dlogits_ld(m);
      const int64_t Rpad = (Rv + 63) & ~(int64_t)63;
      T_HIP(hipMemsetAsync(hc + Rv * CL, 0, (size_t)(Rpad - Rv) * CL * 2, st));   // the padding rows of the last stage are read as operands
      T_HIP(launch_dlogits_rowbias(F32(w.wc), F32(w.lc), Rv, Rpad, F32(w.rb), st));
      const int64_t rows_per = (int64_t)w.rc_stages * 64;
      for (int64_t r0 = 0; r0 < Rv; r0 += rows_per) {
        const int64_t r1 = (r0 + rows_per < Rv) ? r0 + rows_per : Rv;
        const int rows = (int)(r1 - r0);
        T_HIP(launch_dlogits_rows(m, hc, F32(w.rb), yc, F32(w.wc), F32(w.lc), so.target, rmap, r0, r1, Gc, G + t.out_bias.off, st));
        // dH[rows][C] = G[rows][NP] . outT[NP][C]   == col-major: dH^T[C x rows] = outT^T[C x NP] . G^T[NP x rows] (ld = ldg)
        T_HIP(gemm16_nn(Gc, ldg, m.outT, CL, dHc + r0 * CL, CL, rows, CL, m.NP, 0.f, st));
        // d outT[NP][C] += G^T[NP x rows] . hidden[rows][C]   == col-major [C x NP] += hidden^T[C x rows] . (G^T[NP x rows])^T
        T_HIP(gemm16_tn(Gc, ldg, hc + r0 * CL, CL, G + t.outT.off, 0, m.NP, CL, rows, 1, nullptr, 0, r0 == 0 ? 0.f : 1.f, nullptr, 0, st));
      }
    } else
    for (int s0 = 0; s0 < n_stages; s0 += w.nc_stages) {
      const int s1 = (s0 + w.nc_stages < n_stages) ? s0 + w.nc_stages : n_stages;
      const int NC = (s1 - s0) * 64;
      T_HIP(launch_score_grad_chunk(m, hc, yc, Rv, F32(w.lc), F32(w.wc), s0, s1, Gc, st));
      // d bo[n] = sum_r G[r][n]
      hipLaunchKernelGGL(k_colsum_bf16, dim3(NC / 64, (unsigned)((Rv + COLSUM16_ROWS - 1) / COLSUM16_ROWS)), dim3(256), 0, st, Gc, Rv, (int64_t)NC, NC,
                         G + t.out_bias.off + (size_t)s0 * 64);
      T_HIP(hipGetLastError());
      // dH[Rv][C] += G[Rv][NC] . outT_chunk[NC][C]   == col-major: dH^T[C x Rv] += outT_chunk^T[C x NC] . G^T[NC x Rv]
      T_HIP(gemm16_nn(Gc, NC, m.outT + (size_t)s0 * 64 * CL, CL, dHc, CL, Rv, CL, NC, 1.f, st));
      // d outT_chunk[NC][C] = G^T[NC x Rv] . hidden[Rv][C]   == col-major [C x NC] = hidden^T[C x Rv] . (G^T[NC x Rv])^T
      T_HIP(gemm16_tn(Gc, NC, hc, CL, G + t.outT.off + (size_t)s0 * 64 * CL, 0, NC, CL, Rv, 1, nullptr, 0, 0.f, nullptr, 0, st));
    }
    hipLaunchKernelGGL(k_scatter_rows_f32, dim3((unsigned)((Rv * (CL / 4) + 255) / 256)), dim3(256), 0, st, dHc, rmap, Rv, CL, dH);
    T_HIP(hipGetLastError());
  }
#endif

  }   // do_begin
  if (!do_finish) return 0;

  // ================= TemporalBlocks, last to first =========================================================
  float* da = dH;              // grad of a^{k+1}, fp32 [R][Cout]
  float* dprev = F32(w.dB);    // grad of a^k
  float* dsf = F32(w.dsf);
  uint16_t* dcb = BF(w.dcb);
  uint16_t* dub = BF(w.dub);
  if (m.low_gru) {
    // ============ HierGRU low level: BPTT over the L steps of every session (model_gru.py:47-55), last step first ============
    const int M = (int)n_sess;
    const float* lst = lb.states;
    const float* lrec = lb.rec;
    const size_t lcell = (size_t)n_sess * 5 * H;
    float* du = dsf;                       // d u [R][E]: grad of the projected [e, state] rows
    float* lcar = F32(w.lg_dcar[0]);
    float* lnext = F32(w.lg_dcar[1]);
    // (lcar: the launch chain's carried gradient; the one-launch forms keep theirs in registers -- zeroed below only where the chain runs)
    float* dac_all = F32(w.lg_dac);
    float* dag_all = F32(w.lg_dag);
    const bool deferred = lgru_deferred_wgrad(m, (size_t)M, L);
    float* dhpart = F32(w.lg_dhpart);
    float* dcat_c = F32(w.lg_dcat_c);
    float* dcat_g = F32(w.lg_dcat_g);
    float* din_up = F32(w.lg_din_up);
    const unsigned gMH = (unsigned)(((int64_t)M * H + 255) / 256);
    // small batches: the whole BPTT as one launch, a workgroup per 32 sessions (lgru_tile.hip); it keeps the cells' pre-activation gradients, the
    // weight gradients follow below as always in this regime
    const bool tile_bwd = !F && deferred && lgru_tile_applies(m, (int64_t)M) && m.tune.lgru_tile_bwd != 0;   // (fp32 mode: the exact-fp32 launch chain below)
    if (!tile_bwd) T_HIP(hipMemsetAsync(lcar, 0, (size_t)M * HT * 4, st));   // the low level's final state is dropped (model_gru.py:47)
    if (tile_bwd && lgru_wide_bwd_applies(m, (int64_t)M) && lgru_wide_bwd_bytes(m, (int64_t)M, L) > 0)   // H = 128 / 256: HB workgroups per (tile, layer), weights in registers, layers pipelined
      T_HIP(launch_lgru_wide_bwd(m, dH, lb.seq_len, lst, lrec, (int64_t)M, L, du, dac_all, dag_all, ws + w.lg_bwd, st));
    else if (tile_bwd) T_HIP(launch_lgru_tile_bwd(m, dH, lb.seq_len, lst, lrec, (int64_t)M, L, du, dac_all, dag_all, st));
    // the launch chain (the fp32 mode; batches the one-launch forms do not take): with every cell's pre-activation gradients kept and few enough rows for the
    // skinny products, the (step, layer) wavefront with fused epilogues -- 3 (L + nl - 1) dependent launches; otherwise five launches per cell, one after the other
    const bool wave_bwd = !tile_bwd && deferred && M <= gemm32_skinny_max_rows();
    if (wave_bwd) {
      LgruBptt bp;
      memset(&bp, 0, sizeof(bp));
      bp.nl = m.nl; bp.L = L; bp.M = M; bp.H = H; bp.in0 = E;
      for (int l = 0; l < m.nl; l++) { bp.wc[l] = m.lgru_wc[l]; bp.wg[l] = m.lgru_wg[l]; }
      bp.seq_len = lb.seq_len; bp.states = lst; bp.recs = lrec;
      bp.d_out = dH; bp.ld_d_out = (int64_t)L * H;
      bp.d_in0 = du; bp.ld_d_in0 = (int64_t)L * E;
      bp.dac_all = dac_all; bp.dag_all = dag_all;
      bp.car[0] = lcar; bp.car[1] = lnext; bp.dhpart = dhpart; bp.din_up = din_up;
      bp.dcat_c = dcat_c; bp.dcat_pitch = (size_t)M * ((size_t)(E > H ? E : H) + H);
      T_HIP(train_lgru_bptt_wavefront(bp, st));
    }
    for (int tt = L - 1; tt >= 0 && !tile_bwd && !wave_bwd; tt--) {
      for (int l = m.nl - 1; l >= 0; l--) {
        const int in = (l == 0) ? E : H;
        const int K = in + H;
        const float* rec = lrec + ((size_t)tt * m.nl + l) * lcell;
        const float* gates = rec;
        const float* rh = rec + (size_t)M * 2 * H;
        const float* cand = rec + (size_t)M * 3 * H;
        const float* hold = lst + (size_t)tt * n_sess * HT + (size_t)l * H;   // row stride HT
        const float* inp = (l == 0) ? lb.uf + (size_t)tt * E : lrec + ((size_t)tt * m.nl + l - 1) * lcell + (size_t)M * 4 * H;
        const int ldin = (l == 0) ? L * E : H;
        const float* dout = (l == m.nl - 1) ? dH + (size_t)tt * H : din_up;
        const int64_t ld_dout = (l == m.nl - 1) ? (int64_t)L * H : H;
        float* dac = deferred ? dac_all + ((size_t)l * L + tt) * M * H : dac_all;
        float* dag = deferred ? dag_all + ((size_t)l * L + tt) * M * 2 * H : dag_all;
        hipLaunchKernelGGL(k_lgru_bwd_a, dim3(gMH), dim3(256), 0, st, lcar + (size_t)l * H, HT, dout, ld_dout, lb.seq_len, tt, hold, HT, gates, cand,
                           M, H, dac, dag, dhpart);
        T_HIP(hipGetLastError());
        T_HIP(gemm32_nn(dac, H, m.lgru_wc[l], K, dcat_c, K, M, K, H, 0.f, st));
        if (!deferred) {
          T_HIP(gemm32_tn(dac, H, inp, ldin, G + t.lwc[l].off, K, H, in, M, 1.f, F32(w.splitk), w.splitk_floats, st));
          T_HIP(gemm32_tn(dac, H, rh, H, G + t.lwc[l].off + in, K, H, H, M, 1.f, F32(w.splitk), w.splitk_floats, st));
          T_HIP(colsum_f32(dac, M, H, H, G + t.lbc[l].off, st));
        }
        hipLaunchKernelGGL(k_gru_bwd_b, dim3(gMH), dim3(256), 0, st, dcat_c, K, in, hold, HT, gates, M, H, dag, dhpart);
        T_HIP(hipGetLastError());
        T_HIP(gemm32_nn(dag, 2 * H, m.lgru_wg[l], K, dcat_g, K, M, K, 2 * H, 0.f, st));
        if (!deferred) {
          T_HIP(gemm32_tn(dag, 2 * H, inp, ldin, G + t.lwg[l].off, K, 2 * H, in, M, 1.f, F32(w.splitk), w.splitk_floats, st));
          T_HIP(gemm32_tn(dag, 2 * H, hold, HT, G + t.lwg[l].off + in, K, 2 * H, H, M, 1.f, F32(w.splitk), w.splitk_floats, st));
          T_HIP(colsum_f32(dag, M, 2 * H, 2 * H, G + t.lbg[l].off, st));
        }
        float* dst_in = (l == 0) ? du + (size_t)tt * E : din_up;
        const int64_t ldi = (l == 0) ? (int64_t)L * E : H;
        const int Wd = in > H ? in : H;
        hipLaunchKernelGGL(k_lgru_bwd_c, dim3((unsigned)(((int64_t)M * Wd + 255) / 256)), dim3(256), 0, st, dcat_c, dcat_g, K, in, dhpart, M, H,
                           dst_in, ldi, lnext + (size_t)l * H, HT, lcar + (size_t)l * H, lb.seq_len, tt);
        T_HIP(hipGetLastError());
      }
      float* tmp = lcar; lcar = lnext; lnext = tmp;
    }
    if (deferred) {   // the weight gradients of all L * nl cells: one product per layer and operand over the L * M rows (rows (t, m): slab t, session m)
      Gemm32TnDesc wd[4 * HIERTCN_MAX_LAYERS];
      int nw = 0;
      ColsumGroup cg;
      memset(&cg, 0, sizeof(cg));
      for (int l = 0; l < m.nl; l++) {
        const int in = (l == 0) ? E : H;
        const int K = in + H;
        const float* dac = dac_all + (size_t)l * L * M * H;
        const float* dag = dag_all + (size_t)l * L * M * 2 * H;
        const float* inp = (l == 0) ? lb.uf : lrec + (size_t)(l - 1) * lcell + (size_t)M * 4 * H;
        const int64_t in_ld = (l == 0) ? (int64_t)L * E : H, in_ss = (l == 0) ? (int64_t)E : (int64_t)((size_t)m.nl * lcell);
        const float* rh = lrec + (size_t)l * lcell + (size_t)M * 2 * H;
        const float* hold = lst + (size_t)l * H;   // states 0 .. L-1 are contiguous rows (t, m) of HT floats
        const int LM = L * M;
        wd[nw++] = Gemm32TnDesc{dac, H, 0, 0, inp, in_ld, M, in_ss, G + t.lwc[l].off, K, H, in, LM, 1.f};
        wd[nw++] = Gemm32TnDesc{dac, H, 0, 0, rh, H, M, (int64_t)((size_t)m.nl * lcell), G + t.lwc[l].off + in, K, H, H, LM, 1.f};
        wd[nw++] = Gemm32TnDesc{dag, 2 * H, 0, 0, inp, in_ld, M, in_ss, G + t.lwg[l].off, K, 2 * H, in, LM, 1.f};
        wd[nw++] = Gemm32TnDesc{dag, 2 * H, 0, 0, hold, HT, 0, 0, G + t.lwg[l].off + in, K, 2 * H, H, LM, 1.f};
        cg.d[cg.n++] = ColsumDesc{dac, H, H, G + t.lbc[l].off};
        cg.d[cg.n++] = ColsumDesc{dag, 2 * H, 2 * H, G + t.lbg[l].off};
      }
      T_HIP(colsum_f32_group(cg, (int64_t)L * M, st));
      for (int i0 = 0; i0 < nw; i0 += 8) {   // (one grouped launch + one grouped sum for the 4 nl products instead of 4 nl of each)
        const int n = nw - i0 < 8 ? nw - i0 : 8;
        if (gemm32_tn_slabs_batch_applies(wd + i0, n)) {
          T_HIP(gemm32_tn_slabs_batch(wd + i0, n, F32(w.splitk), w.splitk_floats, st));
        } else {
          for (int i = i0; i < i0 + n; i++)
            T_HIP(gemm32_tn_slabs(wd[i].A, wd[i].lda, wd[i].a_div, wd[i].a_ss, wd[i].B, wd[i].ldb, wd[i].b_div, wd[i].b_ss, wd[i].C, wd[i].ldc, wd[i].N1, wd[i].N2,
                                  wd[i].Mred, wd[i].beta, F32(w.splitk), w.splitk_floats, st));
        }
      }
    }
    da = du;
  }
  for (int i = nb - 1; i >= 0 && F && !m.low_gru; i--) {   // fp32 mode: the same block backward on fp32 activations, every product an exact-fp32 GEMM
    const ExactDev xv = fp32_master_view(m, t);
    const int cin = (i == 0) ? E : m.C[i - 1], cout = m.C[i];
    const float* a_in = F32(w.f_act[i]);
    const int dil = 1 << i;
    const int64_t nOut = R * cout;
    float* dc = F32(w.f_sh[0]);
    float* sh = F32(w.f_sh[1]);
    hipLaunchKernelGGL(k_block_bwd_f32, dim3((unsigned)((nOut / 4 + 255) / 256)), dim3(256), 0, st, da, F32(w.f_act[i + 1]), F32(w.f_c[i]), nOut, dsf, dc);
    T_HIP(hipGetLastError());
    T_HIP(colsum_f32(dc, R, cout, cout, G + t.conv_bias[i].off, st));
    if (t.resT[i].n) {
      T_HIP(colsum_f32(dsf, R, cout, cout, G + t.res_bias[i].off, st));
      T_HIP(gemm32_tn(dsf, cout, a_in, cin, G + t.resT[i].off, cin, cout, cin, (int)R, 0.f, F32(w.splitk), w.splitk_floats, st));   // d resT[Cout][Cin] = ds^T . a_in
      T_HIP(gemm32_nn(dsf, cout, xv.resT[i], cin, dprev, cin, (int)R, cin, cout, 0.f, st));                                       // d a_in = ds . resT
    } else {
      T_HIP(hipMemcpyAsync(dprev, dsf, (size_t)nOut * 4, hipMemcpyDeviceToDevice, st));   // identity residual
    }
    for (int j = 0; j < m.K; j++) {   // tap j reads a_in[t - s_j], s_j = (K-1-j) dil (customized_tcn_cell.py:46-49)
      const int sj = (m.K - 1 - j) * dil;
      if (sj >= L) continue;   // the tap only ever saw the zero padding
      const float* a_sh = a_in;
      const float* dc_sh = dc;
      if (sj > 0) {
        T_HIP(shift_rows_f32(a_in, R, cin, L, sj, sh, st));
        a_sh = sh;
      }
      // d W_j[Cout][Cin] = sum_t dc[t]^T . a_in[t - s_j]
      T_HIP(gemm32_tn(dc, cout, a_sh, cin, G + t.convT[i].off + (size_t)j * cout * cin, cin, cout, cin, (int)R, 0.f, F32(w.splitk), w.splitk_floats, st));
      if (sj > 0) {
        T_HIP(shift_rows_f32(dc, R, cout, L, -sj, sh, st));
        dc_sh = sh;
      }
      // d a_in[tau] += dc[tau + s_j] . W_j      (W_j = master [Cout][Cin]: the k-major operand as it is)
      T_HIP(gemm32_nn(dc_sh, cout, xv.convT[i] + (size_t)j * cout * cin, cin, dprev, cin, (int)R, cin, cout, 1.f, st));
    }
    float* tmp = da; da = dprev; dprev = tmp;
  }
  for (int i = nb - 1; i >= 0 && !m.low_gru && !F; i--) {
    const int cin = (i == 0) ? E : m.C[i - 1], cout = m.C[i];
    const uint16_t* a_in = a_of(i);
    const uint16_t* c_act = BF(w.act[2 + 2 * i]);
    const uint16_t* a_out = a_of(i + 1);
    const int dil = 1 << i;
    const int64_t nOut = R * cout;
    hipLaunchKernelGGL(k_block_bwd, dim3((unsigned)((nOut / 4 + 255) / 256)), dim3(256), 0, st, da, a_out, c_act, nOut, dsf, dcb);
    T_HIP(hipGetLastError());
    // conv bias: sum over rows of dc
    hipLaunchKernelGGL(k_colsum_bf16, dim3((cout + 63) / 64, (unsigned)((R + COLSUM16_ROWS - 1) / COLSUM16_ROWS)), dim3(256), 0, st, dcb, R, (int64_t)cout, cout,
                       G + t.conv_bias[i].off);
    T_HIP(hipGetLastError());
    if (m.resT[i]) {
      // Dense down-sample on the residual path (customized_tcn_cell.py:102-106,123-124): res = a_in . Wr + br
      T_HIP(cast_bf16(dsf, dub, nOut, 0, st));
      T_HIP(colsum_f32(dsf, R, cout, cout, G + t.res_bias[i].off, st));
      // d resT[Cout][Cin] = ds^T[Cout x R] . a_in[R][Cin]   == col-major [Cin x Cout] = a_in^T[Cin x R] . (ds^T)^T
      T_HIP(gemm16_tn(dub, cout, a_in, cin, G + t.resT[i].off, 0, cout, cin, R, 1, nullptr, 0, 0.f, F32(w.splitk), w.splitk_floats, st));
      // d a_in[R][Cin] = ds[R][Cout] . resT[Cout][Cin]   == col-major [Cin x R] = resT(col-major [Cin x Cout]) . ds^T[Cout x R]
      T_HIP(gemm16_nt(dub, cout, t.resD[i], cout, dprev, cin, R, cin, cout, 1, nullptr, 0, 0.f, st));
    } else {
      // identity residual: d a_in starts as ds
      T_HIP(hipMemcpyAsync(dprev, dsf, (size_t)nOut * 4, hipMemcpyDeviceToDevice, st));
    }
    {
      // o[t] = sum_j a_in[t - s_j] . W_j with s_j = (K-1-j) dil (customized_tcn_cell.py:46-49), all taps in one launch each:
      //   d W_j^T[Cout][Cin] = sum_t dc[t]^T . a_in[t - s_j]          (row-sliced, partial tiles summed in slice order)
      //   d a_in[tau]       += sum_j dc[tau + s_j] . W_j^T            (W kept as [tap][Cin][Cout] for this product: TrainCtx::convD)
      int back[8], fwd[8];
      for (int j = 0; j < m.K; j++) { fwd[j] = (m.K - 1 - j) * dil; back[j] = -fwd[j]; }
      T_HIP(gemm16_tn(dcb, cout, a_in, cin, G + t.convT[i].off, (int64_t)cin * cout, cout, cin, R, m.K, back, L, 0.f, F32(w.splitk), w.splitk_floats, st));
      T_HIP(gemm16_nt(dcb, cout, t.convD[i], cout, dprev, cin, R, cin, cout, m.K, fwd, L, 1.f, st));
    }
    float* tmp = da; da = dprev; dprev = tmp;
  }
  if (!m.hier) {
    // single-level model_tcn / model_gru (model.py:56-57,66-67,71-73): block 0 / the GRU stack reads the embedded ids directly
    hipLaunchKernelGGL(k_scatter_add_rows, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, G + t.emb.off, x, da, R, E, m.N);
    T_HIP(hipGetLastError());
#undef F32
#undef BF
    return 0;
  }
  const int64_t nRC = R * E;
#define F32(o) reinterpret_cast<float*>(ws + (o))
#define BF(o) reinterpret_cast<uint16_t*>(ws + (o))
  // ================= projection u = e . Wp[:E] + sproj (model_hier.py:54-55 -> model_tcn.py:35) ============
  if (F) {
    const ExactDev xv = fp32_master_view(m, t);
    T_HIP(gemm32_tn(da, E, F32(w.f_emb), E, G + t.projT.off, E, E, E, (int)R, 0.f, F32(w.splitk), w.splitk_floats, st));   // d projT[Eout][Ein] = du^T . e
    T_HIP(gemm32_nn(da, E, xv.projT, E, dprev, E, (int)R, E, E, 0.f, st));                                                  // d e = du . projT
  } else {
  T_HIP(cast_bf16(da, dub, nRC, 0, st));
  // d projT[Eout][Ein] = du^T[Eout x R] . e[R][Ein]   == col-major [Ein x Eout] = e^T[Ein x R] . (du^T)^T
  T_HIP(gemm16_tn(dub, E, a_emb, E, G + t.projT.off, 0, E, E, R, 1, nullptr, 0, 0.f, F32(w.splitk), w.splitk_floats, st));
  // d e[R][Ein] = du[R][Eout] . projT[Eout][Ein]   == col-major [Ein x R] = projT(col-major [Ein x Eout]) . du^T[Eout x R]
  T_HIP(gemm16_nt(dub, E, t.projD, E, dprev, E, R, E, E, 1, nullptr, 0, 0.f, st));
  }
  hipLaunchKernelGGL(k_scatter_add_rows, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, G + t.emb.off, x, dprev, R, E, m.N);
  T_HIP(hipGetLastError());
  // d sproj, laid out [B][S+1][E] like the state history (slot S = 0)
  float* dsp = F32(w.dsp_ext);
  hipLaunchKernelGGL(k_segment_sum, dim3((unsigned)(((int64_t)B * (S + 1) * E + 255) / 256)), dim3(256), 0, st, da, B, S, L, E, dsp);
  T_HIP(hipGetLastError());
  const int Mext = B * (S + 1);
  // d projS[E][HT] = dsproj^T[E x M] . state_before[M][HT]   == col-major [HT x E] = hist^T[HT x M] . (dsp^T[E x M])^T
  T_HIP(gemm32_tn(dsp, E, hist, HT, G + t.projS.off, HT, E, HT, Mext, 0.f, F32(w.splitk), w.splitk_floats, st));
  // d state_before[M][HT] = dsproj[M][E] . projS[E][HT]   == col-major [HT x M] = projS(col-major [HT x E]) . dsp^T[E x M]
  float* dsb = F32(w.dsb_ext);
  T_HIP(gemm32_nn(dsp, E, m.projS, HT, dsb, HT, Mext, HT, E, 0.f, st));

  // ================= GRU through the S slots, last to first (customed_gru_cell.py:323-337, model_hier.py:91-93) =====
  float* dcar = F32(w.dstate[0]);
  float* dnext = F32(w.dstate[1]);
  if (!gru_bptt_persist_applies(m, B, S)) T_HIP(hipMemsetAsync(dcar, 0, (size_t)B * HT * 4, st));   // (the persistent launch keeps the carried gradient in LDS)
  float* dhp = F32(w.dhp);
  float* dac_all = F32(w.dac);
  float* dag_all = F32(w.dag);
  float* dhpart = F32(w.dhpart);
  float* dcat_c = F32(w.dcat_c);
  float* dcat_g = F32(w.dcat_g);
  float* din_up = F32(w.din_up);
  float* dpooled = F32(w.dpooled);
  const int64_t hs = (int64_t)(S + 1) * HT;
  const size_t cell = (size_t)B * 5 * H;
  const unsigned gBH = (unsigned)(((int64_t)B * H + 255) / 256);
  const bool bptt_fused = B <= m.tune.skinny_max_rows && m.tune.gru_bptt_fused != 0;
  const bool bptt_persist = gru_bptt_persist_applies(m, B, S);
  if (bptt_persist)   // small batches: the whole backward scan as ONE persistent launch (gru_persist.hip, k_gru_bptt_persist)
    T_HIP(launch_gru_bptt_persist(m, gsave, hist, mask, dsb, B, S, dac_all, dag_all, dpooled, st));
  for (int s = S - 1; s >= 0 && !bptt_persist; s--) {
    if (!bptt_fused) {
      hipLaunchKernelGGL(k_gru_out_grad, dim3((unsigned)(((int64_t)B * HT + 255) / 256)), dim3(256), 0, st, dcar, dsb + (size_t)(s + 1) * HT, hs,
                         mask + s, S, B, HT, dhp);
      T_HIP(hipGetLastError());
    }
    for (int l = m.nl - 1; l >= 0; l--) {
      const int in = (l == 0) ? E : H;
      const int K = in + H;
      const float* rec = gsave + ((size_t)s * m.nl + l) * cell;
      const float* gates = rec;
      const float* rh = rec + (size_t)B * 2 * H;
      const float* cand = rec + (size_t)B * 3 * H;
      const float* hold = hist + (size_t)s * HT + (size_t)l * H;   // row stride hs
      const float* inp = (l == 0) ? pooled + (size_t)s * E : gsave + ((size_t)s * m.nl + (l - 1)) * cell + (size_t)B * 4 * H;
      (void)rh; (void)inp;
      float* dac = dac_all + ((size_t)l * S + s) * B * H;        // this cell's slots: kept for the weight gradients behind the scan
      float* dag = dag_all + ((size_t)l * S + s) * B * 2 * H;
      float* dst_in = (l == 0) ? dpooled + (size_t)s * E : din_up;
      const int ldi = (l == 0) ? S * E : H;
      if (bptt_fused) {
        // a handful of lanes: the elementwise kernel behind each product is its epilogue (gemm.hip, gemm32_nn_gru_bwd), the top layer's first step takes
        // k_gru_out_grad in, and a lower layer's first step rides on the epilogue of the layer above: 2 nl + 1 launches per slot, not 5 nl + 1
        float* dhp_cell = dhpart + (size_t)((m.nl - 1 - l) & 1) * B * H;
        if (l == m.nl - 1) {
          hipLaunchKernelGGL(k_gru_bwd_a_top, dim3(gBH), dim3(256), 0, st, dcar + (size_t)l * H, HT, dsb + (size_t)(s + 1) * HT + (size_t)l * H, hs, mask + s, S,
                             hold, (int)hs, gates, cand, B, H, dac, dag, dhp_cell);
          T_HIP(hipGetLastError());
        }
        GruBwdEpi ep;
        memset(&ep, 0, sizeof(ep));
        ep.in = in; ep.H = H; ep.K = K;
        ep.gates = gates; ep.h = hold; ep.ldh = (int)hs; ep.dag = dag; ep.dhpart = dhp_cell; ep.dcat_c = dcat_c;
        ep.dst_in = dst_in; ep.ldi = ldi; ep.dst_h = dnext + (size_t)l * H; ep.ldo = HT;
        T_HIP(gemm32_nn_gru_bwd(1, dac, H, m.gru_wc[l], K, B, K, H, ep, st));        // d[in | r*h] = dac . wc, then k_gru_bwd_b's arithmetic
        if (l > 0) {   // the cell below: its k_gru_out_grad + k_gru_bwd_a in this product's epilogue
          const float* rec2 = gsave + ((size_t)s * m.nl + (l - 1)) * cell;
          ep.na_dcar = dcar + (size_t)(l - 1) * H; ep.na_ldc = HT;
          ep.na_dsb = dsb + (size_t)(s + 1) * HT + (size_t)(l - 1) * H; ep.na_ld_dsb = hs;
          ep.na_mask = mask + s; ep.na_ldm = S;
          ep.na_gates = rec2; ep.na_cand = rec2 + (size_t)B * 3 * H; ep.na_h = hist + (size_t)s * HT + (size_t)(l - 1) * H; ep.na_ldh = (int)hs;
          ep.na_dac = dac_all + ((size_t)(l - 1) * S + s) * B * H;
          ep.na_dag = dag_all + ((size_t)(l - 1) * S + s) * B * 2 * H;
          ep.na_dhpart = dhpart + (size_t)((m.nl - l) & 1) * B * H;
        }
        T_HIP(gemm32_nn_gru_bwd(2, dag, 2 * H, m.gru_wg[l], K, B, K, 2 * H, ep, st));  // d[in | h] = dag . wg, then k_gru_bwd_c's
        continue;
      }
      hipLaunchKernelGGL(k_gru_bwd_a, dim3(gBH), dim3(256), 0, st, dhp + (size_t)l * H, HT, (l < m.nl - 1) ? din_up : nullptr, hold, (int)hs,
                         gates, cand, B, H, dac, dag, dhpart);
      T_HIP(hipGetLastError());
      // d[in | r*h] [B][K] = dac[B][H] . wc[H][K]   == col-major [K x B] = wc(col-major [K x H]) . dac^T[H x B]
      T_HIP(gemm32_nn(dac, H, m.gru_wc[l], K, dcat_c, K, B, K, H, 0.f, st));
      hipLaunchKernelGGL(k_gru_bwd_b, dim3(gBH), dim3(256), 0, st, dcat_c, K, in, hold, (int)hs, gates, B, H, dag, dhpart);
      T_HIP(hipGetLastError());
      // d[in | h] [B][K] = dag[B][2H] . wg[2H][K]
      T_HIP(gemm32_nn(dag, 2 * H, m.gru_wg[l], K, dcat_g, K, B, K, 2 * H, 0.f, st));
      const int Wd = in > H ? in : H;
      hipLaunchKernelGGL(k_gru_bwd_c, dim3((unsigned)(((int64_t)B * Wd + 255) / 256)), dim3(256), 0, st, dcat_c, dcat_g, K, in, dhpart, B, H, dst_in,
                         ldi, dnext + (size_t)l * H, HT);
      T_HIP(hipGetLastError());
    }
    float* tmp = dcar; dcar = dnext; dnext = tmp;
  }
  // Weight gradients of the scan, behind it: d wc[H][K] = sum over ALL (slot, lane) rows of dac^T . [in | r*h], d wg likewise with [in | h] -- one
  // product per layer and operand over S * B rows instead of one per cell over B (at the reference's 32 lanes a cell's product is 8 workgroups of 20
  // chunks: 161 launches of 14 us were a third of the training step).  The operands stay where the forward left them: row (s, b) of `in` / r*h / h is
  // addressed as slab s (the slot's record, the slot's column of `pooled` / `hist`) + lane b (gemm32_tn_slabs).
  {
    // (small batches: the 4 nl products are 4-16 output tiles each behind a reduction of S B = 320 rows -- one grouped launch instead of 4 nl)
    Gemm32TnDesc wd[4 * HIERTCN_MAX_LAYERS];
    int nw = 0;
    ColsumGroup cg;
    memset(&cg, 0, sizeof(cg));
    for (int l = 0; l < m.nl; l++) {
      const int in = (l == 0) ? E : H;
      const int K = in + H;
      const float* dac = dac_all + (size_t)l * S * B * H;
      const float* dag = dag_all + (size_t)l * S * B * 2 * H;
      const float* inp = (l == 0) ? pooled : gsave + (size_t)(l - 1) * cell + (size_t)B * 4 * H;
      const int64_t in_ld = (l == 0) ? (int64_t)S * E : H, in_ss = (l == 0) ? (int64_t)E : (int64_t)m.nl * cell;
      const float* rh = gsave + (size_t)l * cell + (size_t)B * 2 * H;
      const float* hold = hist + (size_t)l * H;
      const int SB = S * B;
      wd[nw++] = Gemm32TnDesc{dac, H, 0, 0, inp, in_ld, B, in_ss, G + t.wc[l].off, K, H, in, SB, 1.f};
      wd[nw++] = Gemm32TnDesc{dac, H, 0, 0, rh, H, B, (int64_t)((size_t)m.nl * cell), G + t.wc[l].off + in, K, H, H, SB, 1.f};
      wd[nw++] = Gemm32TnDesc{dag, 2 * H, 0, 0, inp, in_ld, B, in_ss, G + t.wg[l].off, K, 2 * H, in, SB, 1.f};
      wd[nw++] = Gemm32TnDesc{dag, 2 * H, 0, 0, hold, hs, B, (int64_t)HT, G + t.wg[l].off + in, K, 2 * H, H, SB, 1.f};
      cg.d[cg.n++] = ColsumDesc{dac, H, H, G + t.bc[l].off};
      cg.d[cg.n++] = ColsumDesc{dag, 2 * H, 2 * H, G + t.bg[l].off};
    }
    T_HIP(colsum_f32_group(cg, (int64_t)S * B, st));
    for (int i0 = 0; i0 < nw; i0 += 8) {
      const int n = nw - i0 < 8 ? nw - i0 : 8;
      if (gemm32_tn_slabs_batch_applies(wd + i0, n)) {
        T_HIP(gemm32_tn_slabs_batch(wd + i0, n, F32(w.splitk), w.splitk_floats, st));
      } else {
        for (int i = i0; i < i0 + n; i++)
          T_HIP(gemm32_tn_slabs(wd[i].A, wd[i].lda, wd[i].a_div, wd[i].a_ss, wd[i].B, wd[i].ldb, wd[i].b_div, wd[i].b_ss, wd[i].C, wd[i].ldc, wd[i].N1, wd[i].N2,
                                wd[i].Mred, wd[i].beta, F32(w.splitk), w.splitk_floats, st));
      }
    }
  }
  // ================= session mean-pool + 'emb' bias (model_hier.py:83-85) ================================
  T_HIP(colsum_f32(dpooled, n_sess, E, E, G + t.emb_bias.off, st));
  hipLaunchKernelGGL(k_pool_bwd, dim3((unsigned)n_sess), dim3(256), 0, st, G + t.emb.off, y_pool, dpooled, n_sess, L, E, m.N);
  T_HIP(hipGetLastError());
#undef F32
#undef BF
  return 0;
}

// ---- small pieces of the fp32 training step that the mixture model's training step (gmm_train.hip) composes as well -------------------------
hipError_t train_block_bwd_f32(const float* da, const float* aout, const float* cact, int64_t n, float* ds, float* dc, hipStream_t st) {
  hipLaunchKernelGGL(k_block_bwd_f32, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, da, aout, cact, n, ds, dc);
  return hipGetLastError();
}
hipError_t train_shift_rows_f32(const float* src, int64_t R, int C, int L, int shift, float* dst, hipStream_t st) { return shift_rows_f32(src, R, C, L, shift, dst, st); }
hipError_t train_colsum_f32(const float* src, int64_t rows, int64_t ld, int cols, float* out, hipStream_t st) { return colsum_f32(src, rows, ld, cols, out, st); }
hipError_t train_lane_act(const int32_t* y, int B, int T, float* act, float* users, hipStream_t st) {
  hipError_t e = hipMemsetAsync(users, 0, sizeof(float), st);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_lane_act, dim3(B), dim3(256), 0, st, y, T, act, users);
  return hipGetLastError();
}
hipError_t train_adam_flat(float* p, const float* g, float* m1, float* m2, size_t n, float lr, float beta1, float beta2, float eps, float grad_scale,
                           int64_t step, hipStream_t st) {
  const double tt = (double)step;
  const float lr_t = (float)(lr * sqrt(1.0 - pow((double)beta2, tt)) / (1.0 - pow((double)beta1, tt)));
  size_t blocks = (n + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m1, m2, n, lr_t, beta1, beta2, eps, grad_scale);
  return hipGetLastError();
}
// first pointwise step of a cell's backward under dynamic_rnn's sequence_length (k_lgru_bwd_a) in the form the FUSED products of gemm.hip need: an inactive
// row (t >= sequence_length) neither produced an output nor moved its state -- its dac / dag rows are zero, and its carried state gradient passes through as
// dhpart, which the second product's epilogue then writes out unchanged (0 . W + dhpart).  Up to two cells per launch (blockIdx.y).
struct LgruBwdA {
  const float* dcar; int ldd; const float* dout; int64_t ld_dout; int tstep; const float* h; int ldh; const float* gates; const float* cand;
  float* dac; float* dag; float* dhpart;
};
struct LgruBwdAPair { LgruBwdA a[2]; };
__global__ void k_lgru_bwd_a_fused(const LgruBwdAPair p, const int32_t* __restrict__ seq_len, int M, int H) {
  const LgruBwdA& a = p.a[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)M * H) return;
  const int64_t b = i / H;
  const int k = (int)(i - b * H);
  const bool active = a.tstep < seq_len[b];
  const float car = a.dcar[b * a.ldd + k];
  const float d = active ? car + a.dout[b * a.ld_dout + k] : 0.f;
  const float u = a.gates[b * 2 * H + H + k], c = a.cand[i], hv = a.h[b * a.ldh + k];
  a.dac[i] = d * (1.f - u) * (1.f - c * c);
  a.dag[b * 2 * H + H + k] = d * (hv - c) * u * (1.f - u);
  a.dag[b * 2 * H + k] = 0.f;                    // (the r half is written by the first product's epilogue for every row; zero here in case M exceeds a tile edge)
  a.dhpart[i] = active ? d * u : car;
}

hipError_t train_lgru_bptt_wavefront(const LgruBptt& b, hipStream_t st) {
  const int nl = b.nl, L = b.L, M = b.M, H = b.H, HT = b.H * b.nl;
  if (M <= 0 || M > gemm32_skinny_max_rows() || nl < 1 || nl > HIERTCN_MAX_LAYERS) return hipErrorInvalidValue;
  const size_t cell = (size_t)M * 5 * H;
  hipError_t e;
  // the carried gradient alternates between two buffers by the parity of t: cell (t, l) reads slot (t + 1) & 1, writes slot t & 1, columns of layer l; the
  // final state is dropped (model_gru.py:47 / model_gru_gmm.py:47: outputs only), so both start from zero
  for (int i = 0; i < 2; i++)
    if ((e = hipMemsetAsync(b.car[i], 0, (size_t)M * HT * 4, st)) != hipSuccess) return e;
  for (int tau = 0; tau < L + nl - 1; tau++) {
    GruBwdProd p1[HIERTCN_MAX_LAYERS], p2[HIERTCN_MAX_LAYERS];
    LgruBwdA aa[HIERTCN_MAX_LAYERS];
    int n = 0;
    for (int l = nl - 1; l >= 0; l--) {
      const int tt = (L - 1) - (tau - (nl - 1 - l));
      if (tt < 0 || tt >= L) continue;
      const int in = (l == 0) ? b.in0 : H, Kc = in + H;
      float* din_l = b.din_up + (size_t)l * M * H;                               // written by layer l (its input's gradient), read by layer l - 1
      float* d_in = (l == 0 && b.in0) ? b.d_in0 + (size_t)tt * b.in0 : din_l;
      const int64_t ld_d_in = (l == 0 && b.in0) ? b.ld_d_in0 : H;
      const float* rec = b.recs + ((size_t)tt * nl + l) * cell;
      const float* gates = rec;
      const float* cand = rec + (size_t)M * 3 * H;
      const float* hold = b.states + (size_t)tt * M * HT + (size_t)l * H;       // row stride HT
      float* dac = b.dac_all + ((size_t)l * L + tt) * M * H;
      float* dag = b.dag_all + ((size_t)l * L + tt) * M * 2 * H;
      float* dhp = b.dhpart + (size_t)l * M * H;
      LgruBwdA& a = aa[n];
      a.dcar = b.car[(tt + 1) & 1] + (size_t)l * H; a.ldd = HT;
      a.dout = (l == nl - 1) ? b.d_out + (size_t)tt * H : b.din_up + (size_t)(l + 1) * M * H;
      a.ld_dout = (l == nl - 1) ? b.ld_d_out : H;
      a.tstep = tt; a.h = hold; a.ldh = HT; a.gates = gates; a.cand = cand; a.dac = dac; a.dag = dag; a.dhpart = dhp;
      GruBwdEpi ep;
      memset(&ep, 0, sizeof(ep));
      ep.in = in; ep.H = H; ep.K = Kc;
      ep.gates = gates; ep.h = hold; ep.ldh = HT; ep.dag = dag; ep.dhpart = dhp; ep.dcat_c = b.dcat_c + (size_t)l * b.dcat_pitch;
      ep.dst_in = d_in; ep.ldi = (int)ld_d_in; ep.dst_h = b.car[tt & 1] + (size_t)l * H; ep.ldo = HT;
      p1[n].A = dac; p1[n].lda = H; p1[n].B = b.wc[l]; p1[n].ldb = Kc; p1[n].M = M; p1[n].N = Kc; p1[n].K = H; p1[n].ep = ep;            // d [in | r*h] = dac . wc, then step b
      p2[n].A = dag; p2[n].lda = 2 * H; p2[n].B = b.wg[l]; p2[n].ldb = Kc; p2[n].M = M; p2[n].N = Kc; p2[n].K = 2 * H; p2[n].ep = ep;    // d [in | h] = dag . wg, then step c
      n++;
    }
    for (int i = 0; i < n; i += 2) {
      const int k = (n - i) > 1 ? 2 : 1;
      LgruBwdAPair pa;
      pa.a[0] = aa[i]; pa.a[1] = aa[i + k - 1];
      hipLaunchKernelGGL(k_lgru_bwd_a_fused, dim3((unsigned)(((int64_t)M * H + 255) / 256), k), dim3(256), 0, st, pa, b.seq_len, M, H);
      if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    for (int i = 0; i < n; i += 2)
      if ((e = gemm32_nn_gru_bwd_n(1, p1 + i, (n - i) > 1 ? 2 : 1, st)) != hipSuccess) return e;
    for (int i = 0; i < n; i += 2)
      if ((e = gemm32_nn_gru_bwd_n(2, p2 + i, (n - i) > 1 ? 2 : 1, st)) != hipSuccess) return e;
  }
  return hipSuccess;
}

// the pointwise pieces of a GRU cell's backward under dynamic_rnn's sequence_length (k_lgru_bwd_a / k_gru_bwd_b / k_lgru_bwd_c above)
hipError_t train_lgru_bwd_a(const float* dcar, int ldd, const float* dout, int64_t ld_dout, const int32_t* seq_len, int tstep, const float* h, int ldh,
                            const float* gates, const float* cand, int M, int H, float* dac, float* dag, float* dhpart, hipStream_t st) {
  hipLaunchKernelGGL(k_lgru_bwd_a, dim3((unsigned)(((int64_t)M * H + 255) / 256)), dim3(256), 0, st, dcar, ldd, dout, ld_dout, seq_len, tstep, h, ldh, gates, cand,
                     M, H, dac, dag, dhpart);
  return hipGetLastError();
}
hipError_t train_gru_bwd_b(const float* dcat_c, int K, int in, const float* h, int ldh, const float* gates, int M, int H, float* dag, float* dhpart,
                           hipStream_t st) {
  hipLaunchKernelGGL(k_gru_bwd_b, dim3((unsigned)(((int64_t)M * H + 255) / 256)), dim3(256), 0, st, dcat_c, K, in, h, ldh, gates, M, H, dag, dhpart);
  return hipGetLastError();
}
hipError_t train_lgru_bwd_c(const float* dcat_c, const float* dcat_g, int K, int in, const float* dhpart, int M, int H, float* dst_in, int64_t ldi, float* dst_h,
                            int ldo, const float* dcar, const int32_t* seq_len, int tstep, hipStream_t st) {
  const int Wd = in > H ? in : H;
  hipLaunchKernelGGL(k_lgru_bwd_c, dim3((unsigned)(((int64_t)M * Wd + 255) / 256)), dim3(256), 0, st, dcat_c, dcat_g, K, in, dhpart, M, H, dst_in, ldi, dst_h, ldo,
                     dcar, seq_len, tstep);
  return hipGetLastError();
}
