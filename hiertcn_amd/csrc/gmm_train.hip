// Training step of the mixture-output model (BASELINE.json configs[4]; model_tcn_gmm.py:27-44 + the likelihood losses of loss.py:247-257 under
// model.py:111-117's lane mean and model.py:135-141's Adam).  C-ABI: hiertcn_gmm_train_* / hiertcn_gmm_adam_* in include/hiertcn.h (ABI 12).
//
// The reference never trains these models on XING (calc_loss_gmm has no caller and model.py:105-132 cannot build the graph), so
// the step is defined by the build's own decode (oracle/gmm_oracle.py) and checked against fp64 autograd of that decode (tests/test_gpu_gmm_train.py).
//
// Everything is fp32 on the masters (the fp32 training mode's recipe, train.hip): exact-fp32 MFMA GEMMs (gemm.hip, exact.hip), nothing rounded to
// 16 bits.  What makes the step cheap is the same algebra as the forward's Gram form (gmm_gram.hip): for a one-hot target
//     d2_m = |y - mu_m|^2 = h^T G_m h + 2 h.v_m + |b_m|^2 - 2 mu_m[y] + 1,     G_m = W_m W_m^T [C x C],  v_m = W_m b_m,
// so neither the [rows, M D] means nor their gradient are ever formed.  With g_rm = d loss / d d2_m * 2 per (row, mixture):
//     d h_r      = sum_m g_rm (G_m h_r + v_m - W_m[:, y_r]) + W_pi d a_r
//     d W_m^T    = W_m^T S_m + b_m s_m^T - scatter(y_r <- g_rm h_r),       S_m = sum_r g_rm h_r h_r^T [C x C],  s_m = sum_r g_rm h_r
//     d b_m      = W_m^T s_m + (sum_r g_rm) b_m - scatter(y_r <- g_rm)
// i.e. per mixture two [rows x C x C] products, one [C x D x C] product (G_m) and one [D x C x C] product (d W_m^T) instead of two
// [rows x D x C] ones: at the reference's batch (6,400 rows, D = 20,778, C = 256, M = 5) 0.14 TFLOP instead of 0.7, and no 2.7 GB matrix.
// The trunk is model_tcn_gmm's: block 0 reads one-hot rows, so its conv taps and its Dense residual are row GATHERS of [K + 1][N][C0] tables in
// the forward and row SCATTER-ADDS in the backward; blocks 1.. are train.hip's fp32 TemporalBlock forward / backward.
//
// model_gru_gmm's trunk (trunk 1, model_gru_gmm.py:33-48) is the GRU stack on the same one-hot rows: the forward keeps every cell's record
// (pool_gru.hip: launch_onehot_gru_records), the backward runs through the T steps with train.hip's cell pieces (two exact-fp32 products + three
// pointwise launches per cell), the weight gradients of all T x nl cells follow as one product per layer and operand, and layer 0's input half -- a
// gathered table row in the forward -- takes its gradient as a row scatter-add.
//
// A LEARNED scale (args.sigma None: M (1 + 2 D) outputs, sigma_d = exp(s_d)) has no such identity -- sum_d mu_d^2 exp(-2 s_d) is no quadratic form of the
// hidden row -- so that variant materialises mu and s per mixture ([2 M][rows][D] fp32, 5.3 GB at the reference's batch), forms their gradients in place
// and takes three exact-fp32 products per table (k_gtl_rows / k_gtl_grad below).
//
// model_hier with a mixture low level (model_hier.py:39-93 around trunk 2 = tcn_gmm or trunk 3 = gru_gmm) trains as ONE step on this handle (hiertcn_gmm_train_step_hier): the handle
// then also keeps the HIGH level's masters (item table, emb bias, across-session cells), runs session mean-pool -> across-session scan with every record
// kept (pool_gru.hip's launch_gru_scan on a view of the masters) -> the [e, state] rows -> the step above with block 0 as an ordinary TemporalBlock on
// those rows (or the GRU stack with a dense layer-0 input: launch_rows_gru_records) -> and sends the rows' gradient back: the item table's rows by scatter-add, the states through the S slots (three launches per cell, the
// user-boundary mask in the first), the pool.
//
// Built for: trunks 0 and 1 (the models on ids) and trunks 2 and 3 together with the high level, both scale variants, loss modes gmm_prob and gmm_prob_max:
// every mixture model the reference's model.py:74-77 / model_hier.py:64-69 dispatch to.
#include <math.h>
#include <string.h>

#include "exact.h"
#include "gemm.h"
#include "gmm.h"
#include "train.h"

#define GT_MP 16   // the prior logits' gradient is kept [rows][16] (num_mixture <= 16; the fp32 GEMM wants multiples of 4)

struct GSeg {
  size_t off = 0, n = 0;   // floats, inside each arena
};

struct GmmTrain {
  float* arena = nullptr;   // [4][total], REFERENCE (TF) layouts: master | grad | m | v -- a checkpoint is a copy, Adam is elementwise
  size_t total = 0;
  GSeg conv[HIERTCN_MAX_BLOCKS], conv_b[HIERTCN_MAX_BLOCKS], res[HIERTCN_MAX_BLOCKS], res_b[HIERTCN_MAX_BLOCKS], head_w, head_b;
  GSeg gk[HIERTCN_MAX_LAYERS], gb[HIERTCN_MAX_LAYERS], ck[HIERTCN_MAX_LAYERS], cb[HIERTCN_MAX_LAYERS];   // trunk 1: gate / candidate kernel [in + H][2H | H], bias
  // trunk 2 as model_hier's low level: the HIGH level trains with it (hE = 0: a handle that trains on dense rows it is handed is not built)
  int hE = 0, hH = 0, hnl = 0, hN = 0;      // emb width, across-session GRU width / layers, item table rows
  GSeg hemb, hemb_b, hgk[HIERTCN_MAX_LAYERS], hgb[HIERTCN_MAX_LAYERS], hck[HIERTCN_MAX_LAYERS], hcb[HIERTCN_MAX_LAYERS];
  float* hwg[HIERTCN_MAX_LAYERS] = {};      // across-session cells, output-major [2H][in + H] / [H][in + H] (the scan's B operands), and their gradients
  float* hwc[HIERTCN_MAX_LAYERS] = {};
  float* dhwg[HIERTCN_MAX_LAYERS] = {};
  float* dhwc[HIERTCN_MAX_LAYERS] = {};
  int64_t step = 0;         // Adam time step
  bool stale = false;       // the handle's evaluation tables (gmm.hip) are older than the masters: hiertcn_gmm_train_sync
  // compute-layout copies of the masters, re-derived at the start of every step, and the gradients in the same layouts
  float* work = nullptr;    // one allocation behind everything below
  float* convT[HIERTCN_MAX_BLOCKS] = {};    // [K][Cout][Cin], blocks >= 1
  float* resT[HIERTCN_MAX_BLOCKS] = {};     // [Cout][Cin]
  float* dconvT[HIERTCN_MAX_BLOCKS] = {};
  float* dresT[HIERTCN_MAX_BLOCKS] = {};
  float* wg[HIERTCN_MAX_LAYERS] = {};       // trunk 1, output-major: layer 0 [2H][H] (the state half; the input half is the master's rows), layers >= 1 [2H][2H]
  float* wc[HIERTCN_MAX_LAYERS] = {};       // layer 0 [H][H], layers >= 1 [H][2H]
  float* dwg[HIERTCN_MAX_LAYERS] = {};
  float* dwc[HIERTCN_MAX_LAYERS] = {};
  float *Wt = nullptr, *bt = nullptr, *pw = nullptr, *pb = nullptr;         // [M][DP][C] item-major (pad rows zero), [M][DP], [GT_MP][C], [GT_MP]
  float *dWt = nullptr, *dbt = nullptr, *dpw = nullptr, *dpb = nullptr;
  float *Gm = nullptr, *vm = nullptr, *cm = nullptr, *Sm = nullptr, *sm = nullptr, *gs = nullptr;   // [C][M C] (G_m side by side), [M][C], [M], [M C][C] (S_m stacked), [M][C], [M]
  float* splitk = nullptr;
  size_t splitk_floats = 0;
  float* master() const { return arena; }
  float* grad() const { return arena + total; }
  float* mom1() const { return arena + 2 * total; }
  float* mom2() const { return arena + 3 * total; }
};

void gmm_train_destroy(GmmTrain* t) {
  if (!t) return;
  if (t->arena) (void)hipFree(t->arena);
  if (t->work) (void)hipFree(t->work);
  delete t;
}

static size_t up64(size_t n) { return (n + 63) & ~(size_t)63; }

// ---- layout transposes between the reference's tensors and the compute layouts --------------------------------------------------------------
// dst[j * ld_dst + i] = src[i * ld_src + j]   (i < rows, j < cols), 32 x 32 tiles through LDS: both sides coalesced whatever the strides
__global__ __launch_bounds__(256) void k_gt_transpose(const float* __restrict__ src, int64_t ld_src, float* __restrict__ dst, int64_t ld_dst, int rows,
                                                      int cols) {
  __shared__ float tile[32][33];
  const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int i = i0 + k, j = j0 + tx;
    tile[k][tx] = (i < rows && j < cols) ? src[(int64_t)i * ld_src + j] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int j = j0 + k, i = i0 + tx;
    if (i < rows && j < cols) dst[(int64_t)j * ld_dst + i] = tile[tx][k];
  }
}
static hipError_t gt_transpose(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int rows, int cols, hipStream_t st) {
  hipLaunchKernelGGL(k_gt_transpose, dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)), dim3(256), 0, st, src, ld_src, dst, ld_dst, rows, cols);
  return hipGetLastError();
}

// ---- block 0 on ids, fp32: a0 = relu(relu(b + sum_j W_j[id[t - s_j]]) + Wr[id[t]] + br), c0 = relu(conv) kept for the backward --------------------
// (customized_tcn_cell.py:46-49,102-106,120-127 on one_hot(id) * sign(id), model.py:56-61: a one-hot row times a kernel is that kernel's row)
__global__ __launch_bounds__(256) void k_gt_block0(const float* __restrict__ conv, const float* __restrict__ res, const float* __restrict__ b0,
                                                   const float* __restrict__ br0, const int32_t* __restrict__ ids, int64_t R, int T, int N, int C0, int K,
                                                   float* __restrict__ a0, float* __restrict__ c0) {
  const int ch = C0 >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * ch) return;
  const int64_t r = i / ch;
  const int c = (int)(i - r * ch) * 4;
  const int t = (int)(r % T);
  float4 acc = *reinterpret_cast<const float4*>(b0 + c), rs = *reinterpret_cast<const float4*>(br0 + c);
  for (int j = 0; j < K; j++) {
    const int s = K - 1 - j;                 // tap j reads id[t - (K-1-j)] (dilation 1)
    if (t - s < 0) continue;                 // causal left padding: zeros
    const int id = ids[r - s];
    if (id <= 0 || id >= N) continue;        // null / out-of-range id: an all-zero one-hot row
    const float4 v = *reinterpret_cast<const float4*>(conv + ((int64_t)j * N + id) * C0 + c);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  const int id = ids[r];
  if (id > 0 && id < N) {
    const float4 v = *reinterpret_cast<const float4*>(res + (int64_t)id * C0 + c);
    rs.x += v.x; rs.y += v.y; rs.z += v.z; rs.w += v.w;
  }
  const float4 cc = make_float4(fmaxf(acc.x, 0.f), fmaxf(acc.y, 0.f), fmaxf(acc.z, 0.f), fmaxf(acc.w, 0.f));
  *reinterpret_cast<float4*>(c0 + r * C0 + c) = cc;
  *reinterpret_cast<float4*>(a0 + r * C0 + c) = make_float4(fmaxf(cc.x + rs.x, 0.f), fmaxf(cc.y + rs.y, 0.f), fmaxf(cc.z + rs.z, 0.f), fmaxf(cc.w + rs.w, 0.f));
}
// its backward: the rows the forward gathered receive ds (residual table) / dc (tap tables) -- scatter-adds into the [K][N][C0] and [N][C0] gradients
__global__ __launch_bounds__(256) void k_gt_block0_bwd(const float* __restrict__ ds, const float* __restrict__ dc, const int32_t* __restrict__ ids, int64_t R,
                                                       int T, int N, int C0, int K, float* __restrict__ dconv, float* __restrict__ dres) {
  const int ch = C0 >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * ch) return;
  const int64_t r = i / ch;
  const int c = (int)(i - r * ch) * 4;
  const int t = (int)(r % T);
  const float4 g = *reinterpret_cast<const float4*>(dc + r * C0 + c);
  if (g.x != 0.f || g.y != 0.f || g.z != 0.f || g.w != 0.f)
    for (int j = 0; j < K; j++) {
      const int s = K - 1 - j;
      if (t - s < 0) continue;
      const int id = ids[r - s];
      if (id <= 0 || id >= N) continue;
      float* d = dconv + ((int64_t)j * N + id) * C0 + c;
      atomicAdd(d, g.x); atomicAdd(d + 1, g.y); atomicAdd(d + 2, g.z); atomicAdd(d + 3, g.w);
    }
  const int id = ids[r];
  if (id > 0 && id < N) {
    const float4 q = *reinterpret_cast<const float4*>(ds + r * C0 + c);
    float* d = dres + (int64_t)id * C0 + c;
    if (q.x != 0.f) atomicAdd(d, q.x);
    if (q.y != 0.f) atomicAdd(d + 1, q.y);
    if (q.z != 0.f) atomicAdd(d + 2, q.z);
    if (q.w != 0.f) atomicAdd(d + 3, q.w);
  }
}

// trunk 1, layer 0: the input half of [x, h] . W on a one-hot row is the table row W[id] (model_gru_gmm.py:33-48 on one_hot(id) * sign(id)); its gradient is a
// scatter-add of the cell's pre-activation gradients into that row.  dac [L][M][H] (candidate), dag [L][M][2H] (gates); rows (t, m); zero where a row was inactive.
__global__ __launch_bounds__(256) void k_gt_gru_in_scatter(const float* __restrict__ dac, const float* __restrict__ dag, const int32_t* __restrict__ ids, int M,
                                                           int L, int N, int H, float* __restrict__ dgk, float* __restrict__ dck) {
  const int ch = H >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)L * M * ch) return;
  const int64_t row = i / ch;                  // (t, m)
  const int c = (int)(i - row * ch) * 4;
  const int t = (int)(row / M), m = (int)(row - (int64_t)t * M);
  const int id = ids[(int64_t)m * L + t];
  if (id <= 0 || id >= N) return;
  const float4 a = *reinterpret_cast<const float4*>(dac + row * H + c);
  const float4 g0 = *reinterpret_cast<const float4*>(dag + row * 2 * H + c);
  const float4 g1 = *reinterpret_cast<const float4*>(dag + row * 2 * H + H + c);
  float* dc = dck + (int64_t)id * H + c;
  float* dg = dgk + (int64_t)id * 2 * H + c;
  if (a.x != 0.f) atomicAdd(dc, a.x);
  if (a.y != 0.f) atomicAdd(dc + 1, a.y);
  if (a.z != 0.f) atomicAdd(dc + 2, a.z);
  if (a.w != 0.f) atomicAdd(dc + 3, a.w);
  if (g0.x != 0.f) atomicAdd(dg, g0.x);
  if (g0.y != 0.f) atomicAdd(dg + 1, g0.y);
  if (g0.z != 0.f) atomicAdd(dg + 2, g0.z);
  if (g0.w != 0.f) atomicAdd(dg + 3, g0.w);
  if (g1.x != 0.f) atomicAdd(dg + H, g1.x);
  if (g1.y != 0.f) atomicAdd(dg + H + 1, g1.y);
  if (g1.z != 0.f) atomicAdd(dg + H + 2, g1.z);
  if (g1.w != 0.f) atomicAdd(dg + H + 3, g1.w);
}

// ---- model_hier around the dense-row trunk (model_hier.py:39-93 with model_low_type 'tcn_gmm'): the rows, and the pieces of its backward ------------------
// rows[(b, s, t)] = [ emb[x_bst] (zeros for the null id) | state before slot s of user b ]   (model_hier.py:50,54-55); hist [B][S + 1][HT]
__global__ __launch_bounds__(256) void k_gt_hier_rows(const float* __restrict__ emb, const int32_t* __restrict__ x, const float* __restrict__ hist, int64_t R,
                                                      int S, int L, int E, int HT, int N, float* __restrict__ rows) {
  const int W = E + HT, ch = W >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * ch) return;
  const int64_t r = i / ch;
  const int c = (int)(i - r * ch) * 4;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < E) {
    const int id = x[r];
    if (id > 0 && id < N) v = *reinterpret_cast<const float4*>(emb + (int64_t)id * E + c);
  } else {
    const int64_t bs = r / L, b = bs / S, sl = bs - b * S;
    v = *reinterpret_cast<const float4*>(hist + (b * (S + 1) + sl) * HT + (c - E));
  }
  *reinterpret_cast<float4*>(rows + r * W + c) = v;
}
// backward of the rows: the item table's rows take d rows[:, :E] (scatter-add); d (state before slot s) = sum_t d rows[(b, s, t), E:] -> dsb [B][S + 1][HT]
// (slot S stays zero: the state behind the last slot reaches no row)
__global__ __launch_bounds__(256) void k_gt_hier_rows_bwd(const float* __restrict__ drows, const int32_t* __restrict__ x, int64_t n_sess, int S, int L, int E,
                                                          int HT, int N, float* __restrict__ demb, float* __restrict__ dsb) {
  const int W = E + HT, ch = W >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_sess * ch) return;
  const int64_t bs = i / ch;
  const int c = (int)(i - bs * ch) * 4;
  if (c < E) {
    for (int t = 0; t < L; t++) {
      const int64_t r = bs * L + t;
      const int id = x[r];
      if (id <= 0 || id >= N) continue;
      const float4 g = *reinterpret_cast<const float4*>(drows + r * W + c);
      float* d = demb + (int64_t)id * E + c;
      if (g.x != 0.f) atomicAdd(d, g.x);
      if (g.y != 0.f) atomicAdd(d + 1, g.y);
      if (g.z != 0.f) atomicAdd(d + 2, g.z);
      if (g.w != 0.f) atomicAdd(d + 3, g.w);
    }
  } else {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t = 0; t < L; t++) {
      const float4 g = *reinterpret_cast<const float4*>(drows + (bs * L + t) * W + c);
      acc.x += g.x; acc.y += g.y; acc.z += g.z; acc.w += g.w;
    }
    const int64_t b = bs / S, sl = bs - b * S;
    *reinterpret_cast<float4*>(dsb + (b * (S + 1) + sl) * HT + (c - E)) = acc;
  }
}
// first pointwise step of an across-session cell's backward (customed_gru_cell.py:323-337 differentiated; model_hier.py:93's mask): the gradient of the
// cell's unmasked output is (carried from slot s + 1 + what slot s + 1's rows sent to the state they read) * mask[s] + what the layer above sent down
struct GtHlBwdA {
  const float* dcar; int ldc; const float* dsb_next; int64_t ld_dsb; const float* mask; int ldm; const float* din_up; const float* h; int64_t ldh;
  const float* gates; const float* cand; float* dac; float* dag; float* dhpart;
};
struct GtHlBwdAPair { GtHlBwdA a[2]; };   // the cells of one anti-diagonal (slot, layer): blockIdx.y
__global__ void k_gt_hl_bwd_a(const GtHlBwdAPair p, int B, int H) {
  const GtHlBwdA& a = p.a[blockIdx.y];
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)B * H) return;
  const int64_t b = i / H;
  const int k = (int)(i - b * H);
  const float d = (a.dcar[b * a.ldc + k] + a.dsb_next[b * a.ld_dsb + k]) * a.mask[b * a.ldm] + (a.din_up ? a.din_up[i] : 0.f);
  const float u = a.gates[b * 2 * H + H + k], c = a.cand[i], hv = a.h[b * a.ldh + k];
  a.dac[i] = d * (1.f - u) * (1.f - c * c);
  a.dag[b * 2 * H + H + k] = d * (hv - c) * u * (1.f - u);
  a.dag[b * 2 * H + k] = 0.f;
  a.dhpart[i] = d * u;
}
// backward of the session mean-pool + bias (model_hier.py:83-85): d emb[y_t] += d pooled / cnt for the session's cnt targets (the bias: column sums)
__global__ __launch_bounds__(256) void k_gt_pool_bwd(const float* __restrict__ dpooled, const int32_t* __restrict__ y, int64_t n_sess, int L, int E, int N,
                                                     float* __restrict__ demb) {
  const int ch = E >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n_sess * ch) return;
  const int64_t bs = i / ch;
  const int c = (int)(i - bs * ch) * 4;
  int cnt = 0;
  for (int t = 0; t < L; t++) { const int id = y[bs * L + t]; cnt += (id > 0 && id < N) ? 1 : 0; }
  if (cnt == 0) return;
  float4 g = *reinterpret_cast<const float4*>(dpooled + bs * E + c);
  const float inv = 1.0f / (float)cnt;
  g.x *= inv; g.y *= inv; g.z *= inv; g.w *= inv;
  for (int t = 0; t < L; t++) {
    const int id = y[bs * L + t];
    if (id <= 0 || id >= N) continue;
    float* d = demb + (int64_t)id * E + c;
    atomicAdd(d, g.x); atomicAdd(d + 1, g.y); atomicAdd(d + 2, g.z); atomicAdd(d + 3, g.w);
  }
}

// ---- head ------------------------------------------------------------------------------------------------------------------------------------------
// v_m = W_m b_m and |b_m|^2 in two fixed-order stages (no atomics: these enter the FORWARD, where 1 / (2 sigma^2) = 50 multiplies every last-bit
// difference of d2_m before the soft-max over the mixtures -- the loss and the responsibilities are then repeatable run to run):
//   part[m][z][c] = sum_{d in slice z} Wt[m][d][c] bt[m][d],  part[m][z][C] = sum_{d in slice z} bt[m][d]^2;   then the slices in order
__global__ __launch_bounds__(256) void k_gt_vc_part(const float* __restrict__ Wt, const float* __restrict__ bt, int DP, int C, int dslice,
                                                    float* __restrict__ part) {
  const int m = blockIdx.y, z = blockIdx.z, nz = gridDim.z;
  const int c = blockIdx.x * 64 + (threadIdx.x & 63);
  const int q = threadIdx.x >> 6;
  const int d0 = z * dslice, d1 = (d0 + dslice < DP) ? d0 + dslice : DP;
  const float* W = Wt + (size_t)m * DP * C;
  const float* b = bt + (size_t)m * DP;
  float acc = 0.f, sq = 0.f;
  for (int d = d0 + q; d < d1; d += 4) {
    const float bd = b[d];
    if (c < C) acc = fmaf(W[(size_t)d * C + c], bd, acc);
    sq = fmaf(bd, bd, sq);
  }
  __shared__ float red[4][64], rsq[4];
  red[q][threadIdx.x & 63] = acc;
  if ((threadIdx.x & 63) == 0) rsq[q] = sq;
  __syncthreads();
  float* out = part + ((size_t)m * nz + z) * (C + 1);
  if (q == 0 && c < C) out[c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
  if (blockIdx.x == 0 && threadIdx.x == 0) out[C] = (rsq[0] + rsq[1]) + (rsq[2] + rsq[3]);
}
__global__ void k_gt_vc_sum(const float* __restrict__ part, int nz, int C, float* __restrict__ vm, float* __restrict__ cm) {
  const int m = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  float acc = 0.f;
  for (int z = 0; z < nz; z++) acc += part[((size_t)m * nz + z) * (C + 1) + c];
  if (c < C) vm[(size_t)m * C + c] = acc;
  else cm[m] = acc;
}

struct HeadRows {
  const float* H;        // [R][C] hidden rows
  const float* X;        // [R][M C]: columns [m C, (m + 1) C) = H . G_m  (ONE product H . [G_1 | ... | G_M])
  const float* Wt;       // [M][DP][C]
  const float* bt;       // [M][DP]
  const float* pw;       // [GT_MP][C]
  const float* pb;       // [GT_MP]
  const float* vm;       // [M][C]
  const float* cm;       // [M]
  const int32_t* y;      // [R]
  const float* act;      // [B] valid steps per lane
  const float* users;    // [1]
  int64_t R;
  int T, C, M, D, DP, mode;
  float inv_s2;          // 1 / sigma^2
  double cst_over_D;     // (-D log sigma - D/2 log 2 pi) / D
  float* loss;           // [R] per-step loss (0 on rows without a target), unweighted: what hiertcn_gmm_loss returns
  float* dH;             // [R][C]
  float* GH;             // [R][M C]: columns [m C, (m + 1) C) = g_rm h_r  (so that S = GH^T . H is ONE product, [M C][C])
  float* gv;             // [M][R]    = g_rm
  float* dA;             // [R][GT_MP] = d loss / d prior logits
};
// one wave per row: the M (prior logit, h^T G h, h.v, mu[y]) dot products, the loss, its gradient w.r.t. the row
template <int CPL>   // channels per lane = C / 64
__global__ __launch_bounds__(256) void k_gt_head_rows(const HeadRows a) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.R) return;
  const int C = a.C, M = a.M;
  const int yy = a.y[r];
  const bool valid = yy > 0 && yy < a.D;
  float h[CPL];
#pragma unroll
  for (int k = 0; k < CPL; k++) h[k] = a.H[r * C + lane + 64 * k];
  float pl[16], d2[16];
  for (int m = 0; m < M; m++) {
    const float* xr = a.X + ((size_t)r * M + m) * C;
    const float* wy = a.Wt + ((size_t)m * a.DP + (valid ? yy : 0)) * C;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; k++) {
      const int c = lane + 64 * k;
      s0 = fmaf(h[k], a.pw[(size_t)m * C + c], s0);
      s1 = fmaf(h[k], xr[c], s1);
      s2 = fmaf(h[k], a.vm[(size_t)m * C + c], s2);
      s3 = fmaf(h[k], wy[c], s3);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s0 += __shfl_xor(s0, off); s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); s3 += __shfl_xor(s3, off);
    }
    pl[m] = s0 + a.pb[m];
    const float sq = s1 + 2.f * s2 + a.cm[m];                                        // |mu_m|^2
    const float tg = valid ? s3 + a.bt[(size_t)m * a.DP + yy] : 0.f;                 // mu_m[y]
    d2[m] = sq - 2.f * tg + 1.f;
  }
  // log pi = log_softmax(prior logits)
  float mx = -INFINITY;
  for (int m = 0; m < M; m++) mx = fmaxf(mx, pl[m]);
  float se = 0.f;
  for (int m = 0; m < M; m++) se += expf(pl[m] - mx);
  const float lse_pi = mx + logf(se);
  const float u = a.users[0];
  const float w = (valid && u > 0.f) ? 1.0f / ((a.act[r / a.T] + 1e-6f) * u) : 0.f;   // d (batch loss) / d loss_r (model.py:111-117)
  const float invD = 1.0f / (float)a.D;
  float gpi[16], g[16];   // d L / d log pi_m;  2 d L / d d2_m
  float loss = 0.f;
  if (a.mode == 0) {      // gmm_prob (loss.py:248): -log_prob / D, log_prob = LSE_m(log pi_m - d2_m / (2 sigma^2)) + cst
    float z[16], zmax = -INFINITY;
    for (int m = 0; m < M; m++) { z[m] = (pl[m] - lse_pi) - 0.5f * a.inv_s2 * d2[m]; zmax = fmaxf(zmax, z[m]); }
    float s = 0.f;
    for (int m = 0; m < M; m++) { z[m] = expf(z[m] - zmax); s += z[m]; }
    loss = (float)(-((double)(zmax + logf(s)) * (double)invD + a.cst_over_D));
    for (int m = 0; m < M; m++) {
      const float rho = z[m] / s;
      gpi[m] = -w * rho * invD;
      g[m] = w * rho * a.inv_s2 * invD;
    }
  } else {                // gmm_prob_max (loss.py:249-257): -max_m (log pi_m + mvn_m / D); the gradient goes to the (first) arg max
    float best = -INFINITY;
    int mb = 0;
    for (int m = 0; m < M; m++) {
      const float v = (pl[m] - lse_pi) - 0.5f * a.inv_s2 * d2[m] * invD;
      if (v > best) { best = v; mb = m; }
    }
    loss = (float)(-((double)best + a.cst_over_D));
    for (int m = 0; m < M; m++) { gpi[m] = (m == mb) ? -w : 0.f; g[m] = (m == mb) ? w * a.inv_s2 * invD : 0.f; }
  }
  if (!valid) loss = 0.f;
  float gsum = 0.f;
  for (int m = 0; m < M; m++) gsum += gpi[m];
  float da[16];
  for (int m = 0; m < M; m++) da[m] = gpi[m] - expf(pl[m] - lse_pi) * gsum;      // log_softmax backward
  // d h = sum_m g_m (G_m h + v_m - W_m[:, y]) + sum_j da_j pi_w[j];   GH_m = g_m h
  float dh[CPL];
#pragma unroll
  for (int k = 0; k < CPL; k++) dh[k] = 0.f;
  for (int m = 0; m < M; m++) {
    const float* xr = a.X + ((size_t)r * M + m) * C;
    const float* wy = a.Wt + ((size_t)m * a.DP + (valid ? yy : 0)) * C;
    float* gh = a.GH + ((size_t)r * M + m) * C;
#pragma unroll
    for (int k = 0; k < CPL; k++) {
      const int c = lane + 64 * k;
      dh[k] += g[m] * (xr[c] + a.vm[(size_t)m * C + c] - wy[c]) + da[m] * a.pw[(size_t)m * C + c];
      gh[c] = g[m] * h[k];
    }
  }
#pragma unroll
  for (int k = 0; k < CPL; k++) a.dH[r * C + lane + 64 * k] = dh[k];
  if (lane == 0) {
    a.loss[r] = loss;
    for (int m = 0; m < M; m++) a.gv[(size_t)m * a.R + r] = g[m];
  }
  if (lane < GT_MP) a.dA[r * GT_MP + lane] = lane < M ? da[lane] : 0.f;
}

// dWt[m][d][:] += bt[m][d] s_m[:];  dbt[m][d] = Wt[m][d] . s_m + gs[m] bt[m][d]   -- one wave per (m, d) row, d < D
__global__ __launch_bounds__(256) void k_gt_head_fix(const float* __restrict__ Wt, const float* __restrict__ bt, const float* __restrict__ sm,
                                                     const float* __restrict__ gs, int M, int D, int DP, int C, float* __restrict__ dWt,
                                                     float* __restrict__ dbt) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (int64_t)M * D) return;
  const int m = (int)(row / D), d = (int)(row - (int64_t)m * D);
  const size_t o = ((size_t)m * DP + d) * C;
  const float bd = bt[(size_t)m * DP + d];
  float dot = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float s = sm[(size_t)m * C + c];
    dot = fmaf(Wt[o + c], s, dot);
    dWt[o + c] += bd * s;
  }
  for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off);
  if (lane == 0) dbt[(size_t)m * DP + d] = dot + gs[m] * bd;
}
// the target column of every row: dWt[m][y_r][:] -= g_rm h_r, dbt[m][y_r] -= g_rm   (GH holds g_rm h_r)
__global__ __launch_bounds__(256) void k_gt_head_scatter(const float* __restrict__ GH, const float* __restrict__ gv, const int32_t* __restrict__ y, int64_t R,
                                                         int M, int D, int DP, int C, float* __restrict__ dWt, float* __restrict__ dbt) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int yy = y[r];
  if (yy <= 0 || yy >= D) return;
  for (int m = 0; m < M; m++) {
    const float g = gv[(size_t)m * R + r];
    if (g == 0.f) continue;
    const float* gh = GH + ((size_t)r * M + m) * C;
    float* dst = dWt + ((size_t)m * DP + yy) * C;
    for (int c = lane; c < C; c += 64) atomicAdd(dst + c, -gh[c]);
    if (lane == 0) atomicAdd(dbt + (size_t)m * DP + yy, -g);
  }
}
// gs[m] = sum_r gv[m][r]   (one workgroup per mixture, fixed-order tree)
__global__ __launch_bounds__(256) void k_gt_rowsum(const float* __restrict__ gv, int64_t R, float* __restrict__ gs) {
  __shared__ float red[256];
  const float* v = gv + (size_t)blockIdx.x * R;
  float acc = 0.f;
  for (int64_t r = threadIdx.x; r < R; r += 256) acc += v[r];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) gs[blockIdx.x] = red[0];
}
// head gradients from the compute layouts into the reference layout: dW[c][m] = dpw[m][c], db[m] = dpb[m], db[M + m D + d] = dbt[m][d]
__global__ void k_gt_head_small(const float* __restrict__ dpw, const float* __restrict__ dpb, const float* __restrict__ dbt, int M, int Tn, int D, int DP, int C,
                                int64_t n_out, float* __restrict__ dW, float* __restrict__ db) {   // Tn tables: M mean blocks (+ M log-scale blocks)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (int64_t)M * C) {
    const int m = (int)(i / C), c = (int)(i - (int64_t)m * C);
    dW[(int64_t)c * n_out + m] = dpw[(size_t)m * C + c];
  }
  if (i < M) db[i] = dpb[i];
  if (i < (int64_t)Tn * D) {
    const int m = (int)(i / D), d = (int)(i - (int64_t)m * D);
    db[M + i] = dbt[(size_t)m * DP + d];
  }
}

// ---- a learned scale (args.sigma None: head outputs M (1 + 2 D), sigma_d = exp(s_d); the decode of oracle/gmm_oracle.py) ---------------------------------
//   log N_m(y) = -1/2 [ Q_m + (1 - 2 mu_m[y]) exp(-2 s_m[y]) ] - sum_d s_d - D/2 log 2 pi,   Q_m = sum_d mu_d^2 exp(-2 s_d)
// Q_m is no quadratic form of the hidden row, so this path MATERIALISES mu and s (MUS: [2 M][R][DP] fp32: 5.3 GB at the reference's 6,400 rows) with
// exact-fp32 products, turns them into their gradients in place and pushes those through three more products per table.
struct LearnedRows {
  const float* H; const float* MUS; const float* pw; const float* pb; const int32_t* y; const float* act; const float* users;
  int64_t R;
  int T, C, M, D, DP, mode;
  double cst_over_D;     // (-D/2 log 2 pi) / D
  float* loss; float* gl; float* dA; float* dH;
};
// pass 1, one wave per row: Q_m and sum_d s_d over the D outputs, the target's mean and log-scale, the loss, gl_m = d L / d log N_m, d prior logits, and
// dH initialised with the prior's share  sum_j da_j pi_w[j]
template <int CPL>
__global__ __launch_bounds__(256) void k_gtl_rows(const LearnedRows a) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= a.R) return;
  const int C = a.C, M = a.M, D = a.D;
  const int yy = a.y[r];
  const bool valid = yy > 0 && yy < D;
  float h[CPL];
#pragma unroll
  for (int k = 0; k < CPL; k++) h[k] = a.H[r * C + lane + 64 * k];
  float pl[16], lp[16];
  for (int m = 0; m < M; m++) {
    const float* mu = a.MUS + ((size_t)m * a.R + r) * a.DP;
    const float* sc = a.MUS + ((size_t)(M + m) * a.R + r) * a.DP;
    float q = 0.f, ss = 0.f, s0 = 0.f;
    for (int d = lane; d < D; d += 64) {
      const float u = mu[d], sv = sc[d];
      q = fmaf(u * u, expf(-2.f * sv), q);
      ss += sv;
    }
#pragma unroll
    for (int k = 0; k < CPL; k++) s0 = fmaf(h[k], a.pw[(size_t)m * C + lane + 64 * k], s0);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { q += __shfl_xor(q, off); ss += __shfl_xor(ss, off); s0 += __shfl_xor(s0, off); }
    pl[m] = s0 + a.pb[m];
    const float tg = valid ? mu[yy] : 0.f, sy = valid ? sc[yy] : 0.f;
    lp[m] = -0.5f * (q + (valid ? (1.f - 2.f * tg) * expf(-2.f * sy) : 0.f)) - ss;      // log N_m(y) - cst
  }
  float mx = -INFINITY;
  for (int m = 0; m < M; m++) mx = fmaxf(mx, pl[m]);
  float se = 0.f;
  for (int m = 0; m < M; m++) se += expf(pl[m] - mx);
  const float lse_pi = mx + logf(se);
  const float u = a.users[0];
  const float w = (valid && u > 0.f) ? 1.0f / ((a.act[r / a.T] + 1e-6f) * u) : 0.f;
  const float invD = 1.0f / (float)D;
  float gpi[16], gl[16];
  float loss = 0.f;
  if (a.mode == 0) {
    float z[16], zmax = -INFINITY;
    for (int m = 0; m < M; m++) { z[m] = (pl[m] - lse_pi) + lp[m]; zmax = fmaxf(zmax, z[m]); }
    float s = 0.f;
    for (int m = 0; m < M; m++) { z[m] = expf(z[m] - zmax); s += z[m]; }
    loss = (float)(-((double)(zmax + logf(s)) * (double)invD + a.cst_over_D));
    for (int m = 0; m < M; m++) { const float rho = z[m] / s; gpi[m] = -w * rho * invD; gl[m] = -w * rho * invD; }
  } else {
    float best = -INFINITY;
    int mb = 0;
    for (int m = 0; m < M; m++) {
      const float v = (pl[m] - lse_pi) + lp[m] * invD;
      if (v > best) { best = v; mb = m; }
    }
    loss = (float)(-((double)best + a.cst_over_D));
    for (int m = 0; m < M; m++) { gpi[m] = (m == mb) ? -w : 0.f; gl[m] = (m == mb) ? -w * invD : 0.f; }
  }
  if (!valid) loss = 0.f;
  float gsum = 0.f;
  for (int m = 0; m < M; m++) gsum += gpi[m];
  float da[16];
  for (int m = 0; m < M; m++) da[m] = gpi[m] - expf(pl[m] - lse_pi) * gsum;
#pragma unroll
  for (int k = 0; k < CPL; k++) {
    const int c = lane + 64 * k;
    float acc = 0.f;
    for (int m = 0; m < M; m++) acc = fmaf(da[m], a.pw[(size_t)m * C + c], acc);
    a.dH[r * C + c] = acc;
  }
  if (lane == 0) {
    a.loss[r] = loss;
    for (int m = 0; m < M; m++) a.gl[(size_t)m * a.R + r] = gl[m];
  }
  if (lane < GT_MP) a.dA[r * GT_MP + lane] = lane < M ? da[lane] : 0.f;
}
// pass 2, in place: mu <- d L / d mu = gl (y_d - mu_d) e^{-2 s_d};  s <- d L / d s = gl ((y_d - mu_d)^2 e^{-2 s_d} - 1);  0 in the pad columns d >= D
__global__ __launch_bounds__(256) void k_gtl_grad(float* __restrict__ MUS, const float* __restrict__ gl, const int32_t* __restrict__ y, int64_t R, int M, int D,
                                                  int DP) {
  const int d4 = DP >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)M * R * d4) return;
  const int64_t mr = i / d4;
  const int d0 = (int)(i - mr * d4) * 4;
  const int m = (int)(mr / R);
  const int64_t r = mr - (int64_t)m * R;
  const float g = gl[mr];
  const int yy = y[r];
  float* pm = MUS + (size_t)mr * DP + d0;
  float* ps = MUS + ((size_t)(M + m) * R + r) * DP + d0;
  float4 u = *reinterpret_cast<float4*>(pm), sv = *reinterpret_cast<float4*>(ps);
  float uo[4] = {u.x, u.y, u.z, u.w}, so[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int d = d0 + j;
    if (d >= D || g == 0.f) { uo[j] = 0.f; so[j] = 0.f; continue; }
    const float e = expf(-2.f * so[j]);
    const float diff = ((d == yy) ? 1.f : 0.f) - uo[j];
    uo[j] = g * diff * e;
    so[j] = g * (diff * diff * e - 1.f);
  }
  *reinterpret_cast<float4*>(pm) = make_float4(uo[0], uo[1], uo[2], uo[3]);
  *reinterpret_cast<float4*>(ps) = make_float4(so[0], so[1], so[2], so[3]);
}

// ---- the handle's training state -------------------------------------------------------------------------------------------------------------------
static void gt_layout(const hiertcn_gmm_ctx* g, GmmTrain* t) {
  size_t off = 0;
  auto seg = [&](GSeg& s, size_t n) { s.off = off; s.n = n; off += up64(n); };
  const int K = g->K;
  for (int l = 0; l < g->nl; l++) {   // trunks 1 / 3 (nb == 0): rnn.GRUCell layouts, in = item_num (one-hot rows) or input_dim (dense rows) for layer 0, H above
    const size_t in = (l == 0) ? (g->trunk == 3 ? (size_t)g->cfg.input_dim : (size_t)g->N) : (size_t)g->H, H = (size_t)g->H;
    seg(t->gk[l], (in + H) * 2 * H); seg(t->gb[l], 2 * H); seg(t->ck[l], (in + H) * H); seg(t->cb[l], H);
  }
  for (int i = 0, cin = (g->trunk == 2 ? g->cfg.input_dim : g->N); i < g->nb; cin = g->cfg.channels[i], i++) {
    const int cout = g->cfg.channels[i];
    seg(t->conv[i], (size_t)K * cin * cout);
    seg(t->conv_b[i], cout);
    if (cin != cout) { seg(t->res[i], (size_t)cin * cout); seg(t->res_b[i], cout); }
  }
  const size_t n_out = (size_t)g->M * (1 + (g->learned ? 2 : 1) * (size_t)g->D);   // a learned scale: M (1 + 2 D) outputs (model_tcn_gmm.py:39-40)
  seg(t->head_w, (size_t)g->Clast * n_out);
  seg(t->head_b, n_out);
  if (t->hE) {   // hier/emb/{kernel,bias}, hier/multi_rnn_cell/cell_l/gru_cell/...
    seg(t->hemb, (size_t)t->hN * t->hE);
    seg(t->hemb_b, t->hE);
    for (int l = 0; l < t->hnl; l++) {
      const size_t in = (l == 0) ? (size_t)t->hE : (size_t)t->hH, H = (size_t)t->hH;
      seg(t->hgk[l], (in + H) * 2 * H); seg(t->hgb[l], 2 * H); seg(t->hck[l], (in + H) * H); seg(t->hcb[l], H);
    }
  }
  t->total = off;
}

static int cmax_of(const hiertcn_gmm_ctx* g) {
  int w = 0;
  for (int i = 0; i < g->nb; i++) w = g->cfg.channels[i] > w ? g->cfg.channels[i] : w;
  return w;
}

struct GtWs {
  size_t act[HIERTCN_MAX_BLOCKS], cact[HIERTCN_MAX_BLOCKS], resb, X, GH, dA, gv, loss, dH, bufA, bufB, ds, dc, sh, lane_act, users, total;
  // trunk 1 (GRU stack on one-hot rows): the forward's records and the backward-through-time's buffers
  size_t hid, seq_len, states, recs, dac_all, dag_all, dcar[2], dhpart, dcat_c, dcat_g, din_up;
  size_t MUS;   // a learned scale: the materialised means and log-scales, [2 M][R][DP] fp32 (then, in place, their gradients)
};
static GtWs gt_ws(const hiertcn_gmm_ctx* g, int64_t B, int T) {
  GtWs w;
  memset(&w, 0, sizeof(w));
  const size_t R = (size_t)B * T;
  size_t off = 0;
  auto take = [&](size_t floats) { const size_t o = off; off += ((floats * 4) + 255) & ~(size_t)255; return o; };
  const size_t C = (size_t)g->Clast;
  const bool gru = g->trunk == 1 || g->trunk == 3;
  size_t cm = gru ? C : (size_t)cmax_of(g);
  if (g->trunk == 2 && (size_t)g->cfg.input_dim > cm) cm = (size_t)g->cfg.input_dim;   // (block 0's shifted input rows)
  if (g->learned) {
    w.MUS = take((size_t)2 * g->M * R * g->DP);
  } else {
    w.X = take((size_t)g->M * R * C);
    w.GH = take((size_t)g->M * R * C);
  }
  w.dA = take(R * GT_MP);
  w.gv = take((size_t)g->M * R);
  w.loss = take(R);
  w.dH = take(R * cm);
  w.lane_act = take((size_t)B);
  w.users = take(64);
  if (gru) {
    const size_t H = (size_t)g->H, HT = H * g->nl, M = (size_t)B, L = (size_t)T, nl = (size_t)g->nl;
    const size_t Kmax = (g->trunk == 3 && (size_t)g->cfg.input_dim + H > 2 * H) ? (size_t)g->cfg.input_dim + H : 2 * H;
    w.hid = take(R * H);
    w.seq_len = take(M);
    w.states = take((L + 1) * M * HT);
    w.recs = take(L * nl * M * 5 * H);
    w.dac_all = take(nl * L * M * H);
    w.dag_all = take(nl * L * M * 2 * H);
    w.dcar[0] = take(M * HT);
    w.dcar[1] = take(M * HT);
    w.dhpart = take(nl * M * H);      // (per layer: the cells of one anti-diagonal of the backward run in one launch)
    w.dcat_c = take(nl * M * Kmax);
    w.dcat_g = take(M * Kmax);
    w.din_up = take(nl * M * H);
  } else {
    for (int i = 0; i < g->nb; i++) { w.act[i] = take(R * g->cfg.channels[i]); w.cact[i] = take(R * g->cfg.channels[i]); }
    w.resb = take(R * cm);
    w.bufA = take(R * cm);
    w.bufB = take(R * cm);
    w.ds = take(R * cm);
    w.dc = take(R * cm);
    w.sh = take(R * cm);
  }
  w.total = off;
  return w;
}

#define GT_READY(g)                                                                                                    \
  if (!(g)) return gfail(HIERTCN_ERR_INVALID, "null handle");                                                          \
  if (!(g)->train) return gfail(HIERTCN_ERR_STATE, "hiertcn_gmm_train_enable has not been called on this handle");     \
  TuneScope tune_(nullptr)

static void master_ptrs(const hiertcn_gmm_ctx* g, const GmmTrain* t, const float* base, hiertcn_gmm_weights_t* w) {
  memset(w, 0, sizeof(*w));
  for (int i = 0; i < g->nb; i++) {
    w->conv_kernel[i] = base + t->conv[i].off;
    w->conv_bias[i] = base + t->conv_b[i].off;
    if (t->res[i].n) { w->res_kernel[i] = base + t->res[i].off; w->res_bias[i] = base + t->res_b[i].off; }
  }
  for (int l = 0; l < g->nl; l++) {
    w->gru_gate_kernel[l] = base + t->gk[l].off; w->gru_gate_bias[l] = base + t->gb[l].off;
    w->gru_cand_kernel[l] = base + t->ck[l].off; w->gru_cand_bias[l] = base + t->cb[l].off;
  }
  w->head_weights = base + t->head_w.off;
  w->head_biases = base + t->head_b.off;
  if (t->hE) {
    w->hl_emb_kernel = base + t->hemb.off; w->hl_emb_bias = base + t->hemb_b.off;
    for (int l = 0; l < t->hnl; l++) {
      w->hl_gate_kernel[l] = base + t->hgk[l].off; w->hl_gate_bias[l] = base + t->hgb[l].off;
      w->hl_cand_kernel[l] = base + t->hck[l].off; w->hl_cand_bias[l] = base + t->hcb[l].off;
    }
  }
}

// compute-layout copies of the masters (start of a step): conv / residual kernels of blocks >= 1 output-major, the head item-major
static int gt_refresh(hiertcn_gmm_ctx* g, hipStream_t st) {
  GmmTrain* t = g->train;
  const float* P = t->master();
  const int K = g->K, C = g->Clast, M = g->M, D = g->D, DP = g->DP;
  const int b0 = g->trunk == 2 ? 0 : 1;   // dense input rows: block 0 is an ordinary TemporalBlock with Cin = input_dim
  for (int i = b0, cin = (b0 ? g->C0 : g->cfg.input_dim); i < g->nb; cin = g->cfg.channels[i], i++) {
    const int cout = g->cfg.channels[i];
    for (int j = 0; j < K; j++)
      G_HIP(gt_transpose(P + t->conv[i].off + (size_t)j * cin * cout, cout, t->convT[i] + (size_t)j * cin * cout, cin, cin, cout, st));
    if (t->res[i].n) G_HIP(gt_transpose(P + t->res[i].off, cout, t->resT[i], cin, cin, cout, st));
  }
  for (int l = 0; l < t->hnl; l++) {   // the across-session cells, output-major
    const int H = t->hH, Kc = ((l == 0) ? t->hE : H) + H;
    G_HIP(gt_transpose(P + t->hgk[l].off, 2 * H, t->hwg[l], Kc, Kc, 2 * H, st));
    G_HIP(gt_transpose(P + t->hck[l].off, H, t->hwc[l], Kc, Kc, H, st));
  }
  for (int l = 0; l < g->nl; l++) {   // trunk 1: the state half of layer 0 / the whole kernels above, output-major (the cells' B operand); trunk 3: whole kernels
    const int H = g->H, in_tf = (l == 0) ? g->N : H;
    const int Kc = (l == 0) ? (g->trunk == 3 ? g->cfg.input_dim + H : H) : 2 * H;
    const size_t skip = (l == 0 && g->trunk == 1) ? (size_t)in_tf : 0;   // one-hot input: rows [N, N + H) only -- the input half stays where it is (item-major rows of the master)
    G_HIP(gt_transpose(P + t->gk[l].off + skip * 2 * H, 2 * H, t->wg[l], Kc, Kc, 2 * H, st));
    G_HIP(gt_transpose(P + t->ck[l].off + skip * H, H, t->wc[l], Kc, Kc, H, st));
  }
  const int Tn = g->learned ? 2 * M : M;   // item-major tables: the M mean blocks, then (learned scale) the M log-scale blocks
  const int64_t n_out = (int64_t)M + (int64_t)Tn * D;
  const float* W = P + t->head_w.off;
  const float* b = P + t->head_b.off;
  for (int m = 0; m < Tn; m++) {
    G_HIP(gt_transpose(W + M + (int64_t)m * D, n_out, t->Wt + (size_t)m * DP * C, C, C, D, st));   // Wt[m][d][c] = W[c][M + m D + d]
    G_HIP(hipMemcpyAsync(t->bt + (size_t)m * DP, b + M + (size_t)m * D, (size_t)D * 4, hipMemcpyDeviceToDevice, st));
  }
  G_HIP(gt_transpose(W, n_out, t->pw, C, C, M, st));                                               // pw[m][c] = W[c][m]
  G_HIP(hipMemcpyAsync(t->pb, b, (size_t)M * 4, hipMemcpyDeviceToDevice, st));
  return HIERTCN_OK;
}

extern "C" {

int hiertcn_gmm_train_enable(hiertcn_gmm_handle_t g, const hiertcn_gmm_weights_t* w, void* stream) {
  if (!g || !w) return gfail(HIERTCN_ERR_INVALID, "null argument");
  if (g->train) return gfail(HIERTCN_ERR_STATE, "training is already enabled on this handle");
  int hE = 0;
  if (g->trunk == 2 || g->trunk == 3) {   // model_hier.py:39-93 around the dense-row trunk: the high level (item table, emb bias, across-session cells) trains with it
    const int H = g->cfg.hidden_dim, nl = g->cfg.num_layer;
    if (H < 32 || (H & 31) || H > 1024 || nl < 1 || nl > HIERTCN_MAX_LAYERS || g->cfg.item_num < 2)
      return gfail(HIERTCN_ERR_INVALID, "training a dense-row trunk as model_hier's low level: cfg.hidden_dim (a multiple of 32), cfg.num_layer and cfg.item_num describe the high level");
    hE = g->cfg.input_dim - H * nl;
    if (hE < 32 || (hE & 31)) return gfail(HIERTCN_ERR_INVALID, "input_dim %d - hidden_dim %d x num_layer %d = emb width %d: a multiple of 32", g->cfg.input_dim, H, nl, hE);
    if (!w->hl_emb_kernel || !w->hl_emb_bias)
      return gfail(HIERTCN_ERR_INVALID, "training a dense-row trunk means training model_hier around it: hl_emb_kernel / hl_emb_bias (hier/emb/{kernel,bias}) are required");
    for (int l = 0; l < nl; l++)
      if (!w->hl_gate_kernel[l] || !w->hl_gate_bias[l] || !w->hl_cand_kernel[l] || !w->hl_cand_bias[l])
        return gfail(HIERTCN_ERR_INVALID, "training a dense-row trunk: the across-session cell %d (hl_gate / hl_cand kernel and bias) is required", l);
  }
  if (g->M > GT_MP) return gfail(HIERTCN_ERR_INVALID, "num_mixture %d: the training step takes up to 16 mixtures", g->M);
  if (g->Clast != 64 && g->Clast != 128 && g->Clast != 256 && g->Clast != 512) return gfail(HIERTCN_ERR_INVALID, "last width %d: 64, 128, 256, 512", g->Clast);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  TuneScope tune_(nullptr);
  if (!w->head_weights || !w->head_biases) return gfail(HIERTCN_ERR_INVALID, "the head (fully_connected weights / biases) is required");
  for (int l = 0; l < g->nl; l++)
    if (!w->gru_gate_kernel[l] || !w->gru_gate_bias[l] || !w->gru_cand_kernel[l] || !w->gru_cand_bias[l])
      return gfail(HIERTCN_ERR_INVALID, "GRU layer %d: gate / candidate kernel and bias are required", l);
  for (int i = 0, cin = (g->trunk == 2 ? g->cfg.input_dim : g->N); i < g->nb; cin = g->cfg.channels[i], i++) {
    if (!w->conv_kernel[i] || !w->conv_bias[i]) return gfail(HIERTCN_ERR_INVALID, "conv_kernel[%d]/conv_bias[%d] missing", i, i);
    if ((cin != g->cfg.channels[i]) != (w->res_kernel[i] != nullptr && w->res_bias[i] != nullptr))
      return gfail(HIERTCN_ERR_INVALID, "block %d: a Dense residual (kernel and bias) exists exactly when Cin != Cout", i);
  }
  GmmTrain* t = new GmmTrain();
  if (hE) { t->hE = hE; t->hH = g->cfg.hidden_dim; t->hnl = g->cfg.num_layer; t->hN = g->cfg.item_num; }
  gt_layout(g, t);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&t->arena), 4 * t->total * sizeof(float));
  if (e != hipSuccess) { gmm_train_destroy(t); return gfail(HIERTCN_ERR_HIP, "hipMalloc of the training arenas (%zu MB) failed: %s", 4 * t->total * 4 >> 20, hipGetErrorString(e)); }
  // working set
  const size_t C = (size_t)g->Clast, M = (size_t)g->M, DP = (size_t)g->DP;
  size_t off = 0;
  auto take = [&](size_t floats) { const size_t o = off; off += up64(floats); return o; };
  size_t o_convT[HIERTCN_MAX_BLOCKS] = {}, o_resT[HIERTCN_MAX_BLOCKS] = {}, o_dconvT[HIERTCN_MAX_BLOCKS] = {}, o_dresT[HIERTCN_MAX_BLOCKS] = {};
  const int blk0 = g->trunk == 2 ? 0 : 1;
  for (int i = blk0; i < g->nb; i++) {
    o_convT[i] = take(t->conv[i].n); o_dconvT[i] = take(t->conv[i].n);
    if (t->res[i].n) { o_resT[i] = take(t->res[i].n); o_dresT[i] = take(t->res[i].n); }
  }
  size_t o_hwg[HIERTCN_MAX_LAYERS] = {}, o_hwc[HIERTCN_MAX_LAYERS] = {}, o_dhwg[HIERTCN_MAX_LAYERS] = {}, o_dhwc[HIERTCN_MAX_LAYERS] = {};
  for (int l = 0; l < t->hnl; l++) {
    o_hwg[l] = take(t->hgk[l].n); o_dhwg[l] = take(t->hgk[l].n); o_hwc[l] = take(t->hck[l].n); o_dhwc[l] = take(t->hck[l].n);
  }
  size_t o_wg[HIERTCN_MAX_LAYERS] = {}, o_wc[HIERTCN_MAX_LAYERS] = {}, o_dwg[HIERTCN_MAX_LAYERS] = {}, o_dwc[HIERTCN_MAX_LAYERS] = {};
  for (int l = 0; l < g->nl; l++) {
    const size_t H = (size_t)g->H, Kc = (l == 0) ? (g->trunk == 3 ? (size_t)g->cfg.input_dim + H : H) : 2 * H;
    o_wg[l] = take(2 * H * Kc); o_dwg[l] = take(2 * H * Kc); o_wc[l] = take(H * Kc); o_dwc[l] = take(H * Kc);
  }
  const size_t Tn = g->learned ? 2 * M : M;
  const size_t o_Wt = take(Tn * DP * C), o_bt = take(Tn * DP), o_pw = take(GT_MP * C), o_pb = take(GT_MP);
  const size_t o_dWt = take(Tn * DP * C), o_dbt = take(Tn * DP), o_dpw = take(GT_MP * C), o_dpb = take(GT_MP);
  const size_t o_Gm = take(M * C * C), o_vm = take(M * C), o_cm = take(64), o_Sm = take(M * C * C), o_sm = take(M * C), o_gs = take(64);
  size_t sk = 128 * C * C;
  if (sk > ((size_t)16 << 20)) sk = (size_t)16 << 20;
  const size_t o_sk = take(sk);
  e = hipMalloc(reinterpret_cast<void**>(&t->work), off * sizeof(float));
  if (e != hipSuccess) { gmm_train_destroy(t); return gfail(HIERTCN_ERR_HIP, "hipMalloc of the training work set (%zu MB) failed: %s", off * 4 >> 20, hipGetErrorString(e)); }
  for (int i = blk0; i < g->nb; i++) {
    t->convT[i] = t->work + o_convT[i]; t->dconvT[i] = t->work + o_dconvT[i];
    if (t->res[i].n) { t->resT[i] = t->work + o_resT[i]; t->dresT[i] = t->work + o_dresT[i]; }
  }
  for (int l = 0; l < t->hnl; l++) { t->hwg[l] = t->work + o_hwg[l]; t->dhwg[l] = t->work + o_dhwg[l]; t->hwc[l] = t->work + o_hwc[l]; t->dhwc[l] = t->work + o_dhwc[l]; }
  for (int l = 0; l < g->nl; l++) { t->wg[l] = t->work + o_wg[l]; t->dwg[l] = t->work + o_dwg[l]; t->wc[l] = t->work + o_wc[l]; t->dwc[l] = t->work + o_dwc[l]; }
  t->Wt = t->work + o_Wt; t->bt = t->work + o_bt; t->pw = t->work + o_pw; t->pb = t->work + o_pb;
  t->dWt = t->work + o_dWt; t->dbt = t->work + o_dbt; t->dpw = t->work + o_dpw; t->dpb = t->work + o_dpb;
  t->Gm = t->work + o_Gm; t->vm = t->work + o_vm; t->cm = t->work + o_cm; t->Sm = t->work + o_Sm; t->sm = t->work + o_sm; t->gs = t->work + o_gs;
  t->splitk = t->work + o_sk; t->splitk_floats = sk;
  hipError_t e2 = hipMemsetAsync(t->work, 0, off * sizeof(float), st);   // (the pad rows d >= D of the item-major tables stay zero for good)
  if (e2 == hipSuccess) e2 = hipMemsetAsync(t->arena, 0, 4 * t->total * sizeof(float), st);
  if (e2 != hipSuccess) { gmm_train_destroy(t); return gfail(HIERTCN_ERR_HIP, "hipMemsetAsync failed: %s", hipGetErrorString(e2)); }
  g->train = t;
  auto fill = [&]() -> int {
  // masters <- the caller's tensors (reference layouts: plain copies)
  hiertcn_gmm_weights_t dst;
  master_ptrs(g, t, t->master(), &dst);
  for (int i = 0; i < g->nb; i++) {
    G_HIP(hipMemcpyAsync(const_cast<float*>(dst.conv_kernel[i]), w->conv_kernel[i], t->conv[i].n * 4, hipMemcpyDeviceToDevice, st));
    G_HIP(hipMemcpyAsync(const_cast<float*>(dst.conv_bias[i]), w->conv_bias[i], t->conv_b[i].n * 4, hipMemcpyDeviceToDevice, st));
    if (t->res[i].n) {
      G_HIP(hipMemcpyAsync(const_cast<float*>(dst.res_kernel[i]), w->res_kernel[i], t->res[i].n * 4, hipMemcpyDeviceToDevice, st));
      G_HIP(hipMemcpyAsync(const_cast<float*>(dst.res_bias[i]), w->res_bias[i], t->res_b[i].n * 4, hipMemcpyDeviceToDevice, st));
    }
  }
  for (int l = 0; l < g->nl; l++) {
    G_HIP(hipMemcpyAsync(const_cast<float*>(dst.gru_gate_kernel[l]), w->gru_gate_kernel[l], t->gk[l].n * 4, hipMemcpyDeviceToDevice, st));
    G_HIP(hipMemcpyAsync(const_cast<float*>(dst.gru_gate_bias[l]), w->gru_gate_bias[l], t->gb[l].n * 4, hipMemcpyDeviceToDevice, st));
    G_HIP(hipMemcpyAsync(const_cast<float*>(dst.gru_cand_kernel[l]), w->gru_cand_kernel[l], t->ck[l].n * 4, hipMemcpyDeviceToDevice, st));
    G_HIP(hipMemcpyAsync(const_cast<float*>(dst.gru_cand_bias[l]), w->gru_cand_bias[l], t->cb[l].n * 4, hipMemcpyDeviceToDevice, st));
  }
  G_HIP(hipMemcpyAsync(const_cast<float*>(dst.head_weights), w->head_weights, t->head_w.n * 4, hipMemcpyDeviceToDevice, st));
  G_HIP(hipMemcpyAsync(const_cast<float*>(dst.head_biases), w->head_biases, t->head_b.n * 4, hipMemcpyDeviceToDevice, st));
  if (t->hE) {
    G_HIP(hipMemcpyAsync(const_cast<float*>(dst.hl_emb_kernel), w->hl_emb_kernel, t->hemb.n * 4, hipMemcpyDeviceToDevice, st));
    G_HIP(hipMemcpyAsync(const_cast<float*>(dst.hl_emb_bias), w->hl_emb_bias, t->hemb_b.n * 4, hipMemcpyDeviceToDevice, st));
    for (int l = 0; l < t->hnl; l++) {
      G_HIP(hipMemcpyAsync(const_cast<float*>(dst.hl_gate_kernel[l]), w->hl_gate_kernel[l], t->hgk[l].n * 4, hipMemcpyDeviceToDevice, st));
      G_HIP(hipMemcpyAsync(const_cast<float*>(dst.hl_gate_bias[l]), w->hl_gate_bias[l], t->hgb[l].n * 4, hipMemcpyDeviceToDevice, st));
      G_HIP(hipMemcpyAsync(const_cast<float*>(dst.hl_cand_kernel[l]), w->hl_cand_kernel[l], t->hck[l].n * 4, hipMemcpyDeviceToDevice, st));
      G_HIP(hipMemcpyAsync(const_cast<float*>(dst.hl_cand_bias[l]), w->hl_cand_bias[l], t->hcb[l].n * 4, hipMemcpyDeviceToDevice, st));
    }
  }
    return HIERTCN_OK;
  };
  if (int rc = fill()) {   // a handle with half-filled masters must not look like one that trains
    gmm_train_destroy(t);
    g->train = nullptr;
    return rc;
  }
  t->step = 0;
  t->stale = true;
  return HIERTCN_OK;
}

size_t hiertcn_gmm_train_workspace_bytes(hiertcn_gmm_handle_t g, int64_t B, int32_t T) {
  if (!g || B < 1 || T < 1) return 0;
  return gt_ws(g, B, T).total;
}

// The step behind hiertcn_gmm_train_step (x_ids: B sequences of T steps, lanes = sequences) and hiertcn_gmm_train_step_hier (rows_in: the dense
// [e, state] rows of B * S sessions of L steps, lanes = users with S * L steps each; d_rows receives the gradient of those rows).  The caller has
// refreshed the compute-layout copies and zeroed the gradient arena.
static int gt_step_core(hiertcn_gmm_ctx* g, const int32_t* x_ids, const float* rows_in, const int32_t* y_ids, int64_t B, int32_t T, int64_t lanes, int32_t Tlane,
                        int32_t mode, const float* users_global, unsigned char* ws, const GtWs& w, double* out2, float* loss_rows, float* d_rows, void* stream) {
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  GmmTrain* t = g->train;
#define F32(o) reinterpret_cast<float*>(ws + (o))
  const int64_t R = B * T;
  const int K = g->K, C = g->Clast, M = g->M, D = g->D, DP = g->DP, N = g->N, C0 = g->C0, nb = g->nb;
  const float* P = t->master();
  float* G = t->grad();
  const int b0 = g->trunk == 2 ? 0 : 1;   // dense input rows: block 0 is an ordinary TemporalBlock on rows_in

  // ================= forward on the masters =================================================================================================
  const float* H = nullptr;
  OneHotGru og;
  memset(&og, 0, sizeof(og));
  const bool gru_trunk = g->trunk == 1 || g->trunk == 3;
  const int in0 = g->trunk == 3 ? g->cfg.input_dim : 0;   // layer 0's dense input width (one-hot rows: none -- the input half is a gathered table row)
  if (gru_trunk) {
    // model_gru_gmm.py:33-48: MultiRNNCell of GRUCells under dynamic_rnn on the one-hot rows (trunk 1) or on the dense [e, state] rows of model_hier
    // (trunk 3, model_hier.py:68-69), from a zero state; every record kept for the backward
    og.N = N; og.H = g->H; og.nl = g->nl;
    og.in_g = P + t->gk[0].off; og.in_c = P + t->ck[0].off;     // (trunk 1) rows [0, N) of the masters' layer-0 kernels: the input half, item-major as it is
    for (int l = 0; l < g->nl; l++) { og.wg[l] = t->wg[l]; og.bg[l] = P + t->gb[l].off; og.wc[l] = t->wc[l]; og.bc[l] = P + t->cb[l].off; }
    if (g->trunk == 1) G_HIP(launch_onehot_gru_records(og, x_ids, B, T, F32(w.hid), reinterpret_cast<int32_t*>(ws + w.seq_len), F32(w.states), F32(w.recs), st));
    else G_HIP(launch_rows_gru_records(og, rows_in, in0, B, T, F32(w.hid), reinterpret_cast<int32_t*>(ws + w.seq_len), F32(w.states), F32(w.recs), st));
    H = F32(w.hid);
  } else {
  if (b0) {
    hipLaunchKernelGGL(k_gt_block0, dim3((unsigned)((R * (C0 >> 2) + 255) / 256)), dim3(256), 0, st, P + t->conv[0].off, P + t->res[0].off, P + t->conv_b[0].off,
                       P + t->res_b[0].off, x_ids, R, T, N, C0, K, F32(w.act[0]), F32(w.cact[0]));
    G_HIP(hipGetLastError());
  }
  for (int i = b0, cin = (b0 ? C0 : g->cfg.input_dim); i < nb; cin = g->cfg.channels[i], i++) {
    const int cout = g->cfg.channels[i];
    const float* a_prev = (i == 0) ? rows_in : F32(w.act[i - 1]);
    const float* res = a_prev;
    int64_t ldres = cin;
    if (cin != cout) {   // customized_tcn_cell.py:102-106: a biased Dense on the residual path
      XGemm r = {};
      r.A = a_prev; r.lda = cin; r.Wt = t->resT[i]; r.taps = 1; r.Cin = cin; r.M = R; r.N = cout; r.bias = P + t->res_b[i].off;
      r.out = F32(w.resb); r.ldo = cout; r.epi = XEPI_LINEAR;
      G_HIP(launch_xgemm(r, st));
      res = F32(w.resb); ldres = cout;
    }
    XGemm x = {};
    x.A = a_prev; x.lda = cin; x.Wt = t->convT[i]; x.taps = K; x.dil = 1 << i; x.L = T; x.Cin = cin; x.M = R; x.N = cout;
    x.bias = P + t->conv_b[i].off; x.res = res; x.ldres = ldres; x.out = F32(w.act[i]); x.out2 = F32(w.cact[i]); x.ldo = cout; x.epi = XEPI_BLOCK;
    G_HIP(launch_xgemm(x, st));
  }
  H = F32(w.act[nb - 1]);
  }
  G_HIP(train_lane_act(y_ids, (int)lanes, Tlane, F32(w.lane_act), F32(w.users), st));
  float* loss = loss_rows ? loss_rows : F32(w.loss);
  const int Tn = g->learned ? 2 * M : M;
  if (g->learned) {
    // ================= head with a learned scale: mu and s materialised, the loss row-wise, their gradients in place, three products per table ====
    float* MUS = F32(w.MUS);
    for (int tb = 0; tb < Tn; tb++) {
      XGemm x = {};
      x.A = H; x.lda = C; x.Wt = t->Wt + (size_t)tb * DP * C; x.taps = 1; x.Cin = C; x.M = R; x.N = DP; x.bias = t->bt + (size_t)tb * DP;
      x.out = MUS + (size_t)tb * R * DP; x.ldo = DP; x.epi = XEPI_LINEAR;
      G_HIP(launch_xgemm(x, st));
    }
    LearnedRows a;
    a.H = H; a.MUS = MUS; a.pw = t->pw; a.pb = t->pb; a.y = y_ids; a.act = F32(w.lane_act); a.users = users_global ? users_global : F32(w.users);
    a.R = R; a.T = Tlane; a.C = C; a.M = M; a.D = D; a.DP = DP; a.mode = mode;
    a.cst_over_D = (-0.5 * (double)D * log(2.0 * M_PI)) / (double)D;
    a.loss = loss; a.gl = F32(w.gv); a.dA = F32(w.dA); a.dH = F32(w.dH);
    const dim3 grid((unsigned)((R + 3) / 4));
    if (C == 64) hipLaunchKernelGGL(k_gtl_rows<1>, grid, dim3(256), 0, st, a);
    else if (C == 128) hipLaunchKernelGGL(k_gtl_rows<2>, grid, dim3(256), 0, st, a);
    else if (C == 256) hipLaunchKernelGGL(k_gtl_rows<4>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_gtl_rows<8>, grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_gtl_grad, dim3((unsigned)(((int64_t)M * R * (DP >> 2) + 255) / 256)), dim3(256), 0, st, MUS, F32(w.gv), y_ids, R, M, D, DP);
    G_HIP(hipGetLastError());
    G_HIP(hipMemsetAsync(t->dbt, 0, (size_t)Tn * DP * 4, st));
    for (int tb = 0; tb < Tn; tb++) {
      const float* dT = MUS + (size_t)tb * R * DP;            // d mu_m (tb < M) or d s_m: [R][DP]
      G_HIP(gemm32_tn(dT, DP, H, C, t->dWt + (size_t)tb * DP * C, C, DP, C, (int)R, 0.f, t->splitk, t->splitk_floats, st));             // d table^T [DP][C] = dT^T . H
      G_HIP(train_colsum_f32(dT, R, DP, DP, t->dbt + (size_t)tb * DP, st));
      G_HIP(gemm32_nn(dT, DP, t->Wt + (size_t)tb * DP * C, C, F32(w.dH), C, (int)R, C, DP, 1.f, st, t->splitk, t->splitk_floats));      // d h += dT . table^T
    }
  } else {
  // ================= head: G_m, v_m, |b_m|^2 from the masters; X_m = H G_m; the row-wise kernel ==============================================
  for (int m = 0; m < M; m++)   // G_m into columns [m C, (m + 1) C) of Gcat [C][M C]
    G_HIP(gemm32_tn(t->Wt + (size_t)m * DP * C, C, t->Wt + (size_t)m * DP * C, C, t->Gm + (size_t)m * C, (int64_t)M * C, C, C, DP, 0.f, t->splitk, t->splitk_floats, st));
  {
    const int dslice = 1024, nz = (DP + dslice - 1) / dslice;   // (the partials live in the split-K scratch: M nz (C + 1) floats, free between the products)
    if ((size_t)M * nz * (C + 1) > t->splitk_floats) return gfail(HIERTCN_ERR_INVALID, "output_dim %d too large for the v_m partials", D);
    hipLaunchKernelGGL(k_gt_vc_part, dim3((unsigned)((C + 63) / 64), (unsigned)M, (unsigned)nz), dim3(256), 0, st, t->Wt, t->bt, DP, C, dslice, t->splitk);
    hipLaunchKernelGGL(k_gt_vc_sum, dim3((unsigned)((C + 1 + 255) / 256), (unsigned)M), dim3(256), 0, st, t->splitk, nz, C, t->vm, t->cm);
    G_HIP(hipGetLastError());
  }
  G_HIP(gemm32_nn(H, C, t->Gm, (int64_t)M * C, F32(w.X), (int64_t)M * C, (int)R, M * C, C, 0.f, st));   // X = H . [G_1 | ... | G_M]
  {
    HeadRows a;
    a.H = H; a.X = F32(w.X); a.Wt = t->Wt; a.bt = t->bt; a.pw = t->pw; a.pb = t->pb; a.vm = t->vm; a.cm = t->cm; a.y = y_ids;
    a.act = F32(w.lane_act); a.users = users_global ? users_global : F32(w.users);
    a.R = R; a.T = Tlane; a.C = C; a.M = M; a.D = D; a.DP = DP; a.mode = mode;
    const double sg = (double)g->cfg.sigma;
    a.inv_s2 = (float)(1.0 / (sg * sg));
    a.cst_over_D = (-(double)D * log(sg) - 0.5 * (double)D * log(2.0 * M_PI)) / (double)D;
    a.loss = loss; a.dH = F32(w.dH); a.GH = F32(w.GH); a.gv = F32(w.gv); a.dA = F32(w.dA);
    const dim3 grid((unsigned)((R + 3) / 4));
    if (C == 64) hipLaunchKernelGGL(k_gt_head_rows<1>, grid, dim3(256), 0, st, a);
    else if (C == 128) hipLaunchKernelGGL(k_gt_head_rows<2>, grid, dim3(256), 0, st, a);
    else if (C == 256) hipLaunchKernelGGL(k_gt_head_rows<4>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(k_gt_head_rows<8>, grid, dim3(256), 0, st, a);
    G_HIP(hipGetLastError());
  }

  // ================= head backward ==============================================================================================================
  G_HIP(hipMemsetAsync(t->sm, 0, (size_t)M * C * 4, st));
  G_HIP(gemm32_tn(F32(w.GH), (int64_t)M * C, H, C, t->Sm, C, M * C, C, (int)R, 0.f, t->splitk, t->splitk_floats, st));   // S [M C][C]: rows [m C, (m + 1) C) = S_m = sum_r g_rm h_r h_r^T
  G_HIP(train_colsum_f32(F32(w.GH), R, (int64_t)M * C, M * C, t->sm, st));                                               // s_m = sum_r g_rm h_r
  for (int m = 0; m < M; m++)
    G_HIP(gemm32_nn(t->Wt + (size_t)m * DP * C, C, t->Sm + (size_t)m * C * C, C, t->dWt + (size_t)m * DP * C, C, DP, C, C, 0.f, st));   // W_m^T S_m
  hipLaunchKernelGGL(k_gt_rowsum, dim3((unsigned)M), dim3(256), 0, st, F32(w.gv), R, t->gs);
  hipLaunchKernelGGL(k_gt_head_fix, dim3((unsigned)(((int64_t)M * D + 3) / 4)), dim3(256), 0, st, t->Wt, t->bt, t->sm, t->gs, M, D, DP, C, t->dWt, t->dbt);
  hipLaunchKernelGGL(k_gt_head_scatter, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, F32(w.GH), F32(w.gv), y_ids, R, M, D, DP, C, t->dWt, t->dbt);
  G_HIP(hipGetLastError());
  }
  if (out2) {
    if (int rc = hiertcn_gmm_lane_loss(g, loss, y_ids, (int32_t)lanes, Tlane, out2, stream)) return rc;
  }
  // prior: d pi_w[m][:] = sum_r dA[r][m] h_r, d pi_b[m] = sum_r dA[r][m]
  G_HIP(gemm32_tn(F32(w.dA), GT_MP, H, C, t->dpw, C, GT_MP, C, (int)R, 0.f, t->splitk, t->splitk_floats, st));
  G_HIP(hipMemsetAsync(t->dpb, 0, GT_MP * 4, st));
  G_HIP(train_colsum_f32(F32(w.dA), R, GT_MP, GT_MP, t->dpb, st));
  // ... into the reference layout [C][M (1 + D)] / [M (1 + D)]
  const int64_t n_out = (int64_t)M + (int64_t)Tn * D;
  float* dW = G + t->head_w.off;
  float* db = G + t->head_b.off;
  for (int m = 0; m < Tn; m++) G_HIP(gt_transpose(t->dWt + (size_t)m * DP * C, C, dW + M + (int64_t)m * D, n_out, D, C, st));
  {
    const int64_t n = (int64_t)Tn * (D > C ? D : C);
    hipLaunchKernelGGL(k_gt_head_small, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, t->dpw, t->dpb, t->dbt, M, Tn, D, DP, C, n_out, dW, db);
    G_HIP(hipGetLastError());
  }

  if (gru_trunk) {
    // ================= backward through time of the GRU stack (customed_gru_cell.py:323-337 differentiated; train.hip's cell chain) =============
    const int Hh = g->H, HT = g->H * g->nl, nl = g->nl, Mb = (int)B, L = T;
    const size_t cell = (size_t)Mb * 5 * Hh;
    const int32_t* seq_len = reinterpret_cast<const int32_t*>(ws + w.seq_len);
    const float* states = F32(w.states);
    const float* recs = F32(w.recs);
    float* dac_all = F32(w.dac_all);
    float* dag_all = F32(w.dag_all);
    float* lcar = F32(w.dcar[0]);
    float* lnext = F32(w.dcar[1]);
    float* dhpart = F32(w.dhpart);
    float* dcat_c = F32(w.dcat_c);
    float* dcat_g = F32(w.dcat_g);
    float* din_up = F32(w.din_up);
    const float* dHid = F32(w.dH);
    G_HIP(hipMemsetAsync(lcar, 0, (size_t)Mb * HT * 4, st));   // the final state is dropped (model_gru_gmm.py:47: outputs only)
    const bool fused = Mb <= gemm32_skinny_max_rows();
    const size_t Kmax_ws = (g->trunk == 3 && (size_t)in0 + Hh > 2 * (size_t)Hh) ? (size_t)in0 + Hh : 2 * (size_t)Hh;   // (gt_ws' pitch of the per-layer scratch)
    if (fused) {   // the (step, layer) wavefront with fused epilogues (train.hip, train_lgru_bptt_wavefront): 3 (L + nl - 1) dependent launches instead of 3 L nl
      LgruBptt bp;
      memset(&bp, 0, sizeof(bp));
      bp.nl = nl; bp.L = L; bp.M = Mb; bp.H = Hh; bp.in0 = in0;
      for (int l = 0; l < nl; l++) { bp.wc[l] = t->wc[l]; bp.wg[l] = t->wg[l]; }
      bp.seq_len = seq_len; bp.states = states; bp.recs = recs;
      bp.d_out = dHid; bp.ld_d_out = (int64_t)L * Hh;
      bp.d_in0 = d_rows; bp.ld_d_in0 = (int64_t)L * in0;
      bp.dac_all = dac_all; bp.dag_all = dag_all;
      bp.car[0] = F32(w.dcar[0]); bp.car[1] = F32(w.dcar[1]); bp.dhpart = dhpart; bp.din_up = din_up;
      bp.dcat_c = dcat_c; bp.dcat_pitch = (size_t)Mb * Kmax_ws;
      G_HIP(train_lgru_bptt_wavefront(bp, st));
    }
    for (int tt = L - 1; tt >= 0 && !fused; tt--) {   // more lanes than the skinny products take: the five-launch cells, one after the other
      for (int l = nl - 1; l >= 0; l--) {
        const int in = (l == 0) ? in0 : Hh;          // one-hot rows: layer 0's input half is the gathered table row, no dense input operand
        const int Kc = in + Hh;
        float* d_in = (l == 0) ? (in0 ? d_rows + (size_t)tt * in0 : din_up) : din_up;     // dense rows: the gradient of step tt's input rows
        const int64_t ld_d_in = (l == 0 && in0) ? (int64_t)L * in0 : Hh;
        const float* rec = recs + ((size_t)tt * nl + l) * cell;
        const float* gates = rec;
        const float* cand = rec + (size_t)Mb * 3 * Hh;
        const float* hold = states + (size_t)tt * Mb * HT + (size_t)l * Hh;   // row stride HT
        const float* dout = (l == nl - 1) ? dHid + (size_t)tt * Hh : din_up;
        const int64_t ld_dout = (l == nl - 1) ? (int64_t)L * Hh : Hh;
        float* dac = dac_all + ((size_t)l * L + tt) * Mb * Hh;
        float* dag = dag_all + ((size_t)l * L + tt) * Mb * 2 * Hh;
        G_HIP(train_lgru_bwd_a(lcar + (size_t)l * Hh, HT, dout, ld_dout, seq_len, tt, hold, HT, gates, cand, Mb, Hh, dac, dag, dhpart, st));
        G_HIP(gemm32_nn(dac, Hh, t->wc[l], Kc, dcat_c, Kc, Mb, Kc, Hh, 0.f, st));            // d [in | r*h] = dac . wc
        G_HIP(train_gru_bwd_b(dcat_c, Kc, in, hold, HT, gates, Mb, Hh, dag, dhpart, st));
        G_HIP(gemm32_nn(dag, 2 * Hh, t->wg[l], Kc, dcat_g, Kc, Mb, Kc, 2 * Hh, 0.f, st));    // d [in | h] = dag . wg
        G_HIP(train_lgru_bwd_c(dcat_c, dcat_g, Kc, in, dhpart, Mb, Hh, d_in, ld_d_in, lnext + (size_t)l * Hh, HT, lcar + (size_t)l * Hh, seq_len, tt, st));
      }
      float* tmp = lcar; lcar = lnext; lnext = tmp;
    }
    // weight gradients of all L * nl cells: one product per layer and operand over the L * M rows (rows (t, m): slab t, session m)
    const int LM = L * Mb;
    for (int l = 0; l < nl; l++) {
      const int in = (l == 0) ? in0 : Hh, Kc = in + Hh;
      const float* dac = dac_all + (size_t)l * L * Mb * Hh;
      const float* dag = dag_all + (size_t)l * L * Mb * 2 * Hh;
      const int64_t rss = (int64_t)((size_t)nl * cell);   // the same record of the next step
      const float* rh = recs + (size_t)l * cell + (size_t)Mb * 2 * Hh;
      const float* hold = states + (size_t)l * Hh;        // states 0 .. L-1: contiguous rows (t, m) of HT floats
      if (l > 0) {
        const float* inp = recs + (size_t)(l - 1) * cell + (size_t)Mb * 4 * Hh;
        G_HIP(gemm32_tn_slabs(dac, Hh, 0, 0, inp, Hh, Mb, rss, t->dwc[l], Kc, Hh, in, LM, 0.f, t->splitk, t->splitk_floats, st));
        G_HIP(gemm32_tn_slabs(dag, 2 * Hh, 0, 0, inp, Hh, Mb, rss, t->dwg[l], Kc, 2 * Hh, in, LM, 0.f, t->splitk, t->splitk_floats, st));
      } else if (in0) {   // dense input rows: row (t, m) of the operand at t * in0 + m * (L * in0)
        G_HIP(gemm32_tn_slabs(dac, Hh, 0, 0, rows_in, (int64_t)L * in0, Mb, in0, t->dwc[l], Kc, Hh, in, LM, 0.f, t->splitk, t->splitk_floats, st));
        G_HIP(gemm32_tn_slabs(dag, 2 * Hh, 0, 0, rows_in, (int64_t)L * in0, Mb, in0, t->dwg[l], Kc, 2 * Hh, in, LM, 0.f, t->splitk, t->splitk_floats, st));
      }
      G_HIP(gemm32_tn_slabs(dac, Hh, 0, 0, rh, Hh, Mb, rss, t->dwc[l] + in, Kc, Hh, Hh, LM, 0.f, t->splitk, t->splitk_floats, st));
      G_HIP(gemm32_tn_slabs(dag, 2 * Hh, 0, 0, hold, HT, 0, 0, t->dwg[l] + in, Kc, 2 * Hh, Hh, LM, 0.f, t->splitk, t->splitk_floats, st));
      G_HIP(train_colsum_f32(dac, LM, Hh, Hh, G + t->cb[l].off, st));
      G_HIP(train_colsum_f32(dag, LM, 2 * Hh, 2 * Hh, G + t->gb[l].off, st));
      // -> the reference layout [in_tf + H][2H | H]: rows behind the input half (layer 0: behind the N table rows)
      const size_t skip = (l == 0 && g->trunk == 1) ? (size_t)N : 0;
      G_HIP(gt_transpose(t->dwg[l], Kc, G + t->gk[l].off + skip * 2 * Hh, 2 * Hh, 2 * Hh, Kc, st));
      G_HIP(gt_transpose(t->dwc[l], Kc, G + t->ck[l].off + skip * Hh, Hh, Hh, Kc, st));
    }
    if (g->trunk == 1) {   // layer 0's input half: the table rows the forward gathered receive the cells' pre-activation gradients
      hipLaunchKernelGGL(k_gt_gru_in_scatter, dim3((unsigned)(((int64_t)LM * (Hh >> 2) + 255) / 256)), dim3(256), 0, st, dac_all, dag_all, x_ids, Mb, L, N, Hh,
                         G + t->gk[0].off, G + t->ck[0].off);
      G_HIP(hipGetLastError());
    }
  } else {
  // ================= TemporalBlocks, last to first (train.hip's fp32 recipe) ===========================================================================
  float* da = F32(w.dH);
  float* dprev = F32(w.bufA);
  float* spare = F32(w.bufB);
  float* ds = F32(w.ds);
  float* dc = F32(w.dc);
  float* sh = F32(w.sh);
  for (int i = nb - 1; i >= b0; i--) {
    const int cin = (i == 0) ? g->cfg.input_dim : g->cfg.channels[i - 1], cout = g->cfg.channels[i];
    const float* a_in = (i == 0) ? rows_in : F32(w.act[i - 1]);
    if (i == 0) dprev = d_rows;   // the gradient of the dense input rows goes straight to the caller's buffer
    const int dil = 1 << i;
    const int64_t nOut = R * cout;
    G_HIP(train_block_bwd_f32(da, F32(w.act[i]), F32(w.cact[i]), nOut, ds, dc, st));
    G_HIP(train_colsum_f32(dc, R, cout, cout, G + t->conv_b[i].off, st));
    if (t->res[i].n) {
      G_HIP(train_colsum_f32(ds, R, cout, cout, G + t->res_b[i].off, st));
      G_HIP(gemm32_tn(ds, cout, a_in, cin, t->dresT[i], cin, cout, cin, (int)R, 0.f, t->splitk, t->splitk_floats, st));   // d resT[Cout][Cin] = ds^T . a_in
      G_HIP(gemm32_nn(ds, cout, t->resT[i], cin, dprev, cin, (int)R, cin, cout, 0.f, st));                                   // d a_in = ds . resT
      G_HIP(gt_transpose(t->dresT[i], cin, G + t->res[i].off, cout, cout, cin, st));                                         // -> [Cin][Cout]
    } else {
      G_HIP(hipMemcpyAsync(dprev, ds, (size_t)nOut * 4, hipMemcpyDeviceToDevice, st));   // identity residual
    }
    for (int j = 0; j < K; j++) {   // tap j reads a_in[t - s_j], s_j = (K-1-j) dil (customized_tcn_cell.py:46-49)
      const int sj = (K - 1 - j) * dil;
      float* dWj = t->dconvT[i] + (size_t)j * cout * cin;
      if (sj >= T) { G_HIP(hipMemsetAsync(dWj, 0, (size_t)cout * cin * 4, st)); continue; }   // the tap only ever saw the zero padding
      const float* a_sh = a_in;
      const float* dc_sh = dc;
      if (sj > 0) { G_HIP(train_shift_rows_f32(a_in, R, cin, T, sj, sh, st)); a_sh = sh; }
      G_HIP(gemm32_tn(dc, cout, a_sh, cin, dWj, cin, cout, cin, (int)R, 0.f, t->splitk, t->splitk_floats, st));   // d W_j[Cout][Cin] = sum_t dc[t]^T . a_in[t - s_j]
      if (sj > 0) { G_HIP(train_shift_rows_f32(dc, R, cout, T, -sj, sh, st)); dc_sh = sh; }
      G_HIP(gemm32_nn(dc_sh, cout, t->convT[i] + (size_t)j * cout * cin, cin, dprev, cin, (int)R, cin, cout, 1.f, st));   // d a_in[tau] += dc[tau + s_j] . W_j
      G_HIP(gt_transpose(dWj, cin, G + t->conv[i].off + (size_t)j * cin * cout, cout, cout, cin, st));                   // -> [Cin][Cout]
    }
    float* tmp = da; da = dprev; dprev = (tmp == F32(w.dH)) ? spare : tmp;
  }
  // block 0 on ids: pointwise part, bias sums, then the row scatter-adds into the gathered tables
  if (b0) {
    const int64_t nOut = R * C0;
    G_HIP(train_block_bwd_f32(da, F32(w.act[0]), F32(w.cact[0]), nOut, ds, dc, st));
    G_HIP(train_colsum_f32(dc, R, C0, C0, G + t->conv_b[0].off, st));
    G_HIP(train_colsum_f32(ds, R, C0, C0, G + t->res_b[0].off, st));
    hipLaunchKernelGGL(k_gt_block0_bwd, dim3((unsigned)((R * (C0 >> 2) + 255) / 256)), dim3(256), 0, st, ds, dc, x_ids, R, T, N, C0, K, G + t->conv[0].off,
                       G + t->res[0].off);
    G_HIP(hipGetLastError());
  }
  }
#undef F32
  return HIERTCN_OK;
}

int hiertcn_gmm_train_step(hiertcn_gmm_handle_t g, const int32_t* x_ids, const int32_t* y_ids, int64_t B, int32_t T, int32_t mode,
                           const float* users_global, void* workspace, size_t workspace_bytes, double* out2, float* loss_rows, void* stream) {
  GT_READY(g);
  if (g->trunk != 0 && g->trunk != 1) return gfail(HIERTCN_ERR_STATE, "this handle reads dense rows (trunk %d): hiertcn_gmm_train_step_hier", g->trunk);
  if (!x_ids || !y_ids || !workspace || B < 1 || T < 1 || T > 4096 || B > (int64_t)1 << 22) return gfail(HIERTCN_ERR_INVALID, "bad argument");
  if (mode != 0 && mode != 1) return gfail(HIERTCN_ERR_INVALID, "mode %d: 0 (gmm_prob) or 1 (gmm_prob_max)", mode);
  const GtWs w = gt_ws(g, B, T);
  if (workspace_bytes < w.total) return gfail(HIERTCN_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, w.total);
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return gfail(HIERTCN_ERR_INVALID, "workspace must be 256-byte aligned");
  if ((int64_t)B * T > (int64_t)1 << 30) return gfail(HIERTCN_ERR_INVALID, "too many rows in one step");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (int rc = gt_refresh(g, st)) return rc;
  G_HIP(hipMemsetAsync(g->train->grad(), 0, g->train->total * sizeof(float), st));
  return gt_step_core(g, x_ids, nullptr, y_ids, B, T, B, T, mode, users_global, reinterpret_cast<unsigned char*>(workspace), w, out2, loss_rows, nullptr, stream);
}

// ---- model_hier with the tcn_gmm low level (model_hier.py:39-93; trunk 2): session mean-pool -> across-session scan (every record kept) -> [e, state] rows
// -> the step above on those rows -> the rows' gradient back into the item table and the states -> backward through the S slots -> the pool's backward.
// Extra workspace behind the trunk's: the high level's buffers.
struct GtHierWs {
  size_t pooled, scan, save, rows, drows, dsb, dpooled, dac_all, dag_all, dcar[2], dhpart, dcat_c, din_up, total;
};
static GtHierWs gt_hier_ws(const GmmTrain* t, int64_t B, int S, int L, size_t base) {
  GtHierWs w;
  memset(&w, 0, sizeof(w));
  size_t off = base;
  auto take = [&](size_t floats) { const size_t o = off; off += ((floats * 4) + 255) & ~(size_t)255; return o; };
  const size_t E = (size_t)t->hE, H = (size_t)t->hH, nl = (size_t)t->hnl, HT = H * nl, W = E + HT, Bz = (size_t)B, R = Bz * S * L;
  w.pooled = take(Bz * S * E);
  w.scan = take(Bz * (S + 1) * HT + nl * Bz * 5 * H);       // launch_gru_scan's own: the state history [B][S + 1][HT] first (pool_gru.hip: gru_workspace_bytes)
  w.save = take((size_t)S * nl * Bz * 5 * H);               // per (slot, layer): gates | r*h | candidate | unmasked output
  w.rows = take(R * W);
  w.drows = take(R * W);
  w.dsb = take(Bz * (S + 1) * HT);
  w.dpooled = take(Bz * S * E);
  w.dac_all = take(nl * S * Bz * H);
  w.dag_all = take(nl * S * Bz * 2 * H);
  w.dcar[0] = take(Bz * HT);
  w.dcar[1] = take(Bz * HT);
  w.dhpart = take(nl * Bz * H);     // (per layer: the cells of one anti-diagonal of the backward run in one launch)
  w.dcat_c = take(nl * Bz * (W > 2 * H ? W : 2 * H));
  w.din_up = take(nl * Bz * H);
  w.total = off;
  return w;
}

size_t hiertcn_gmm_train_hier_workspace_bytes(hiertcn_gmm_handle_t g, int64_t B, int32_t S, int32_t L) {
  if (!g || !g->train || !g->train->hE || B < 1 || S < 1 || L < 1) return 0;
  return gt_hier_ws(g->train, B, S, L, gt_ws(g, B * S, L).total).total;
}

int hiertcn_gmm_train_step_hier(hiertcn_gmm_handle_t g, const int32_t* x_ids, const int32_t* y_ids, const float* mask, const float* state_in, int64_t B, int32_t S,
                                int32_t L, int32_t mode, const float* users_global, void* workspace, size_t workspace_bytes, float* state_out, double* out2,
                                float* loss_rows, void* stream) {
  GT_READY(g);
  GmmTrain* t = g->train;
  if ((g->trunk != 2 && g->trunk != 3) || !t->hE) return gfail(HIERTCN_ERR_STATE, "hiertcn_gmm_train_step_hier is the step of a dense-row trunk (2, 3) enabled with the high level's variables");
  if (!x_ids || !y_ids || !mask || !state_in || !state_out || !workspace || B < 1 || S < 1 || L < 1 || L > 4096) return gfail(HIERTCN_ERR_INVALID, "bad argument");
  if (mode != 0 && mode != 1) return gfail(HIERTCN_ERR_INVALID, "mode %d: 0 (gmm_prob) or 1 (gmm_prob_max)", mode);
  if (B > gemm32_skinny_max_rows()) return gfail(HIERTCN_ERR_INVALID, "%lld lanes: the across-session backward takes up to %d per step", (long long)B, gemm32_skinny_max_rows());
  if ((int64_t)B * S * L > (int64_t)1 << 30) return gfail(HIERTCN_ERR_INVALID, "too many rows in one step");
  const int64_t n_sess = B * S, R = n_sess * L;
  const GtWs w = gt_ws(g, n_sess, L);
  const GtHierWs hw = gt_hier_ws(t, B, S, L, w.total);
  if (workspace_bytes < hw.total) return gfail(HIERTCN_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, hw.total);
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return gfail(HIERTCN_ERR_INVALID, "workspace must be 256-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
#define F32(o) reinterpret_cast<float*>(ws + (o))
  const int E = t->hE, H = t->hH, nl = t->hnl, HT = H * nl, W = E + HT, N = t->hN, Bi = (int)B;
  const float* P = t->master();
  float* G = t->grad();
  if (int rc = gt_refresh(g, st)) return rc;
  G_HIP(hipMemsetAsync(G, 0, t->total * sizeof(float), st));

  // ---- forward of the high level on the masters: the reference's slot loop, restructured as everywhere in this library: the scan consumes the mean
  // TARGET embeddings only, so it runs first for all slots and every (user, slot) is then an independent low-level problem
  ModelDev md;
  memset(static_cast<void*>(&md), 0, sizeof(md));
  tuning_defaults(&md.tune);
  md.tune.gru_scan_persistent = 0;           // (the persistent launch needs the main handle's LDS images: the chain of exact-fp32 cell products here)
  md.hier = 1; md.N = N; md.E = E; md.H = H; md.nl = nl; md.HT = HT;
  md.emb_bias = const_cast<float*>(P + t->hemb_b.off);
  for (int l = 0; l < nl; l++) {
    md.gru_wg[l] = t->hwg[l]; md.gru_bg[l] = const_cast<float*>(P + t->hgb[l].off);
    md.gru_wc[l] = t->hwc[l]; md.gru_bc[l] = const_cast<float*>(P + t->hcb[l].off);
  }
  ExactDev xd;
  memset(&xd, 0, sizeof(xd));
  xd.emb = const_cast<float*>(P + t->hemb.off);
  float* pooled = F32(hw.pooled);
  float* hist = F32(hw.scan);                 // [B][S + 1][HT]
  float* save = F32(hw.save);
  G_HIP(launch_xsession_pool(md, xd, y_ids, n_sess, L, pooled, st));
  G_HIP(launch_gru_scan(md, pooled, mask, state_in, Bi, S, nullptr, state_out, nullptr, hist, st, save));
  hipLaunchKernelGGL(k_gt_hier_rows, dim3((unsigned)((R * (W >> 2) + 255) / 256)), dim3(256), 0, st, xd.emb, x_ids, hist, R, S, L, E, HT, N, F32(hw.rows));
  G_HIP(hipGetLastError());

  // ---- the low level + head, forward and backward, on those rows (lanes = users, S * L steps each)
  if (int rc = gt_step_core(g, nullptr, F32(hw.rows), y_ids, n_sess, L, B, S * L, mode, users_global, ws, w, out2, loss_rows, F32(hw.drows), stream)) return rc;

  // ---- the rows' gradient: item-table rows (scatter-add) and the states the rows read
  float* dsb = F32(hw.dsb);
  G_HIP(hipMemsetAsync(dsb, 0, (size_t)B * (S + 1) * HT * 4, st));
  hipLaunchKernelGGL(k_gt_hier_rows_bwd, dim3((unsigned)((n_sess * (W >> 2) + 255) / 256)), dim3(256), 0, st, F32(hw.drows), x_ids, n_sess, S, L, E, HT, N,
                     G + t->hemb.off, dsb);
  G_HIP(hipGetLastError());

  // ---- backward through the S slots (customed_gru_cell.py:323-337 differentiated, model_hier.py:91-93): three launches per cell
  const size_t cell = (size_t)B * 5 * H;
  const int64_t hs = (int64_t)(S + 1) * HT;      // row stride of the state history / of dsb
  float* dcar = F32(hw.dcar[0]);
  float* dnext = F32(hw.dcar[1]);
  float* dpooled = F32(hw.dpooled);
  float* dhpart = F32(hw.dhpart);
  float* dcat_c = F32(hw.dcat_c);
  float* din_up = F32(hw.din_up);
  float* dac_all = F32(hw.dac_all);
  float* dag_all = F32(hw.dag_all);
  G_HIP(hipMemsetAsync(dcar, 0, (size_t)B * HT * 4, st));     // the state behind the last slot is an output, not part of the loss
  G_HIP(hipMemsetAsync(dnext, 0, (size_t)B * HT * 4, st));
  // wavefront over (slot, layer), last to first, as the low-level stack's (gt_step_core): cell (sl, l) needs (sl + 1, l) and (sl, l + 1) only; the carried
  // gradient alternates between the two buffers by the slot's parity; 3 (S + nl - 1) dependent launches instead of 3 S nl
  float* car[2] = {dcar, dnext};
  const size_t pitch_c = (size_t)B * ((size_t)W > 2 * (size_t)H ? (size_t)W : 2 * (size_t)H);
  for (int tau = 0; tau < S + nl - 1; tau++) {
    GtHlBwdA aa[HIERTCN_MAX_LAYERS];
    GruBwdProd p1[HIERTCN_MAX_LAYERS], p2[HIERTCN_MAX_LAYERS];
    int n = 0;
    for (int l = nl - 1; l >= 0; l--) {
      const int sl = (S - 1) - (tau - (nl - 1 - l));
      if (sl < 0 || sl >= S) continue;
      const int in = (l == 0) ? E : H, Kc = in + H;
      const float* rec = save + ((size_t)sl * nl + l) * cell;
      const float* gates = rec;
      const float* cand = rec + (size_t)B * 3 * H;
      const float* hold = hist + (size_t)sl * HT + (size_t)l * H;        // row stride hs
      float* dac = dac_all + ((size_t)l * S + sl) * B * H;
      float* dag = dag_all + ((size_t)l * S + sl) * B * 2 * H;
      float* dhp = dhpart + (size_t)l * B * H;
      GtHlBwdA& a = aa[n];
      a.dcar = car[(sl + 1) & 1] + (size_t)l * H; a.ldc = HT;
      a.dsb_next = dsb + (size_t)(sl + 1) * HT + (size_t)l * H; a.ld_dsb = hs; a.mask = mask + sl; a.ldm = S;
      a.din_up = (l == nl - 1) ? nullptr : din_up + (size_t)(l + 1) * B * H;   // (written by layer l + 1: its input's gradient)
      a.h = hold; a.ldh = hs; a.gates = gates; a.cand = cand; a.dac = dac; a.dag = dag; a.dhpart = dhp;
      GruBwdEpi ep;
      memset(&ep, 0, sizeof(ep));
      ep.in = in; ep.H = H; ep.K = Kc;
      ep.gates = gates; ep.h = hold; ep.ldh = (int)hs; ep.dag = dag; ep.dhpart = dhp; ep.dcat_c = dcat_c + (size_t)l * pitch_c;
      ep.dst_in = (l == 0) ? dpooled + (size_t)sl * E : din_up + (size_t)l * B * H; ep.ldi = (l == 0) ? S * E : H;
      ep.dst_h = car[sl & 1] + (size_t)l * H; ep.ldo = HT;
      p1[n].A = dac; p1[n].lda = H; p1[n].B = t->hwc[l]; p1[n].ldb = Kc; p1[n].M = Bi; p1[n].N = Kc; p1[n].K = H; p1[n].ep = ep;
      p2[n].A = dag; p2[n].lda = 2 * H; p2[n].B = t->hwg[l]; p2[n].ldb = Kc; p2[n].M = Bi; p2[n].N = Kc; p2[n].K = 2 * H; p2[n].ep = ep;
      n++;
    }
    for (int i = 0; i < n; i += 2) {
      const int k = (n - i) > 1 ? 2 : 1;
      GtHlBwdAPair pa;
      pa.a[0] = aa[i]; pa.a[1] = aa[i + k - 1];
      hipLaunchKernelGGL(k_gt_hl_bwd_a, dim3((unsigned)(((int64_t)B * H + 255) / 256), k), dim3(256), 0, st, pa, Bi, H);
      G_HIP(hipGetLastError());
    }
    for (int i = 0; i < n; i += 2) G_HIP(gemm32_nn_gru_bwd_n(1, p1 + i, (n - i) > 1 ? 2 : 1, st));
    for (int i = 0; i < n; i += 2) G_HIP(gemm32_nn_gru_bwd_n(2, p2 + i, (n - i) > 1 ? 2 : 1, st));
  }
  // weight gradients of all S * nl cells: one product per layer and operand over the S * B rows (rows (slot, user))
  const int SB = S * Bi;
  for (int l = 0; l < nl; l++) {
    const int in = (l == 0) ? E : H, Kc = in + H;
    const float* dac = dac_all + (size_t)l * S * B * H;
    const float* dag = dag_all + (size_t)l * S * B * 2 * H;
    const int64_t rss = (int64_t)((size_t)nl * cell);
    const float* rh = save + (size_t)l * cell + (size_t)B * 2 * H;
    const float* hold = hist + (size_t)l * H;
    if (l == 0) {   // input = pooled[b][slot]: row (slot, b) at slot * E + b * (S * E)
      G_HIP(gemm32_tn_slabs(dac, H, 0, 0, pooled, (int64_t)S * E, Bi, E, t->dhwc[l], Kc, H, in, SB, 0.f, t->splitk, t->splitk_floats, st));
      G_HIP(gemm32_tn_slabs(dag, 2 * H, 0, 0, pooled, (int64_t)S * E, Bi, E, t->dhwg[l], Kc, 2 * H, in, SB, 0.f, t->splitk, t->splitk_floats, st));
    } else {
      const float* inp = save + (size_t)(l - 1) * cell + (size_t)B * 4 * H;
      G_HIP(gemm32_tn_slabs(dac, H, 0, 0, inp, H, Bi, rss, t->dhwc[l], Kc, H, in, SB, 0.f, t->splitk, t->splitk_floats, st));
      G_HIP(gemm32_tn_slabs(dag, 2 * H, 0, 0, inp, H, Bi, rss, t->dhwg[l], Kc, 2 * H, in, SB, 0.f, t->splitk, t->splitk_floats, st));
    }
    G_HIP(gemm32_tn_slabs(dac, H, 0, 0, rh, H, Bi, rss, t->dhwc[l] + in, Kc, H, H, SB, 0.f, t->splitk, t->splitk_floats, st));
    G_HIP(gemm32_tn_slabs(dag, 2 * H, 0, 0, hold, hs, Bi, HT, t->dhwg[l] + in, Kc, 2 * H, H, SB, 0.f, t->splitk, t->splitk_floats, st));
    G_HIP(train_colsum_f32(dac, SB, H, H, G + t->hcb[l].off, st));
    G_HIP(train_colsum_f32(dag, SB, 2 * H, 2 * H, G + t->hgb[l].off, st));
    G_HIP(gt_transpose(t->dhwg[l], Kc, G + t->hgk[l].off, 2 * H, 2 * H, Kc, st));   // -> the reference layout [in + H][2H]
    G_HIP(gt_transpose(t->dhwc[l], Kc, G + t->hck[l].off, H, H, Kc, st));
  }
  // ---- session mean-pool + bias
  hipLaunchKernelGGL(k_gt_pool_bwd, dim3((unsigned)((n_sess * (E >> 2) + 255) / 256)), dim3(256), 0, st, dpooled, y_ids, n_sess, L, E, N, G + t->hemb.off);
  G_HIP(hipGetLastError());
  G_HIP(train_colsum_f32(dpooled, n_sess, E, E, G + t->hemb_b.off, st));
#undef F32
  return HIERTCN_OK;
}

int hiertcn_gmm_adam_step(hiertcn_gmm_handle_t g, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  GT_READY(g);
  GmmTrain* t = g->train;
  G_HIP(train_adam_flat(t->master(), t->grad(), t->mom1(), t->mom2(), t->total, lr, beta1, beta2, eps, grad_scale, t->step + 1, reinterpret_cast<hipStream_t>(stream)));
  t->step += 1;          // (only a step that was enqueued counts)
  t->stale = true;
  return HIERTCN_OK;
}

int hiertcn_gmm_adam_time(hiertcn_gmm_handle_t g, int64_t* get, const int64_t* set) {
  GT_READY(g);
  if (set) {
    if (*set < 0) return gfail(HIERTCN_ERR_INVALID, "the Adam time step cannot be negative");
    g->train->step = *set;
  }
  if (get) *get = g->train->step;
  return HIERTCN_OK;
}

int hiertcn_gmm_train_sync(hiertcn_gmm_handle_t g, void* stream) {
  GT_READY(g);
  GmmTrain* t = g->train;
  if (!t->stale && g->weights_set) return HIERTCN_OK;
  hiertcn_gmm_weights_t w;
  master_ptrs(g, t, t->master(), &w);
  if (int rc = hiertcn_gmm_set_weights(g, &w, stream)) return rc;
  t->stale = false;
  return HIERTCN_OK;
}

int hiertcn_gmm_train_copy(hiertcn_gmm_handle_t g, int32_t which, int32_t to_handle, const hiertcn_gmm_weights_t* bufs, void* stream) {
  GT_READY(g);
  if (!bufs || which < 0 || which > 3) return gfail(HIERTCN_ERR_INVALID, "which: 0 masters, 1 gradients, 2 / 3 the Adam moments");
  GmmTrain* t = g->train;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hiertcn_gmm_weights_t a;
  master_ptrs(g, t, t->arena + (size_t)which * t->total, &a);
  auto cp = [&](const float* arena_ptr, const float* user_ptr, size_t n) -> hipError_t {
    if (!n) return hipSuccess;
    if (!user_ptr) return hipErrorInvalidValue;
    return to_handle ? hipMemcpyAsync(const_cast<float*>(arena_ptr), user_ptr, n * 4, hipMemcpyDeviceToDevice, st)
                     : hipMemcpyAsync(const_cast<float*>(user_ptr), arena_ptr, n * 4, hipMemcpyDeviceToDevice, st);
  };
  for (int i = 0; i < g->nb; i++) {
    G_HIP(cp(a.conv_kernel[i], bufs->conv_kernel[i], t->conv[i].n));
    G_HIP(cp(a.conv_bias[i], bufs->conv_bias[i], t->conv_b[i].n));
    G_HIP(cp(a.res_kernel[i], bufs->res_kernel[i], t->res[i].n));
    G_HIP(cp(a.res_bias[i], bufs->res_bias[i], t->res_b[i].n));
  }
  for (int l = 0; l < g->nl; l++) {
    G_HIP(cp(a.gru_gate_kernel[l], bufs->gru_gate_kernel[l], t->gk[l].n));
    G_HIP(cp(a.gru_gate_bias[l], bufs->gru_gate_bias[l], t->gb[l].n));
    G_HIP(cp(a.gru_cand_kernel[l], bufs->gru_cand_kernel[l], t->ck[l].n));
    G_HIP(cp(a.gru_cand_bias[l], bufs->gru_cand_bias[l], t->cb[l].n));
  }
  G_HIP(cp(a.head_weights, bufs->head_weights, t->head_w.n));
  G_HIP(cp(a.head_biases, bufs->head_biases, t->head_b.n));
  if (t->hE) {
    G_HIP(cp(a.hl_emb_kernel, bufs->hl_emb_kernel, t->hemb.n));
    G_HIP(cp(a.hl_emb_bias, bufs->hl_emb_bias, t->hemb_b.n));
    for (int l = 0; l < t->hnl; l++) {
      G_HIP(cp(a.hl_gate_kernel[l], bufs->hl_gate_kernel[l], t->hgk[l].n));
      G_HIP(cp(a.hl_gate_bias[l], bufs->hl_gate_bias[l], t->hgb[l].n));
      G_HIP(cp(a.hl_cand_kernel[l], bufs->hl_cand_kernel[l], t->hck[l].n));
      G_HIP(cp(a.hl_cand_bias[l], bufs->hl_cand_bias[l], t->hcb[l].n));
    }
  }
  if (to_handle && which == 0) t->stale = true;
  return HIERTCN_OK;
}

int hiertcn_gmm_train_arena(hiertcn_gmm_handle_t g, int32_t which, float** ptr, size_t* count) {
  GT_READY(g);
  if (which < 0 || which > 3 || !ptr || !count) return gfail(HIERTCN_ERR_INVALID, "bad argument");
  *ptr = g->train->arena + (size_t)which * g->train->total;
  *count = g->train->total;
  return HIERTCN_OK;
}

}  // extern "C"
