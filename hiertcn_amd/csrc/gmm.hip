// Mixture-output models (BASELINE.json configs[4]): model_tcn_gmm.py:27-44 (trunk + head; model_gru_gmm.py:49-54
// has the same head) and the two likelihood losses of loss.py:247-257.  C-ABI: hiertcn_gmm_* in include/hiertcn.h.
//
//  * trunk on XING ids: the TemporalConvNet runs on one_hot(id) * sign(id) directly (no 'emb' projection, model_tcn_gmm.py:36), so
//    block 0 has Cin = item_num: a one-hot row times a kernel is a row GATHER -- conv1 = b + sum_j W_j[id[t - (K-1-j)]], Dense residual
//    = Wr[id[t]] + br (customized_tcn_cell.py:46-49,102-106,120-127) -- one HBM-bound kernel over (K + 1) item-major tables; blocks
//    1.. are the layer-at-a-time kernels of the main path (tcn_layers.hip);
//  * head: fully_connected(n_out = M (1 + D)), decoded (the decode is the build's own -- oracle/gmm_oracle.py) as M prior logits +
//    M mean vectors of D.  Neither the [rows, M D] means nor the one-hot y are ever materialised: for a one-hot target
//    sum_d (y_d - mu_d)^2 = |mu_m|^2 - 2 mu_m[target] + 1, so each mixture is ONE pass of the scoring kernel's main loop over its
//    [D, C] table with a sum-of-squares epilogue (score.hip, MODE 3) that also picks the target column by the same MFMA sequence;
//  * k_gmm_finish turns (|mu_m|^2, mu_m[target], prior logits) into the per-step loss.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "gmm.h"

static thread_local char g_gerr[512] = "";
extern "C" const char* hiertcn_gmm_last_error(void) { return g_gerr; }
int gfail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_gerr, sizeof(g_gerr), fmt, ap);
  va_end(ap);
  return code;
}

// ---- block 0 on ids: relu(relu(b + sum_j W_j[id[t-s_j]]) + Wr[id[t]] + br) ------------------------------------------------------
__global__ __launch_bounds__(256) void k_gmm_block0(const uint16_t* __restrict__ tab, const float* __restrict__ b0, const float* __restrict__ br0,
                                                    const int32_t* __restrict__ ids, int64_t R, int T, int N, int C0, int K,
                                                    uint16_t* __restrict__ out) {
  const int ch = C0 >> 3;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * ch) return;
  const int64_t r = i / ch;
  const int c = (int)(i - r * ch) * 8;
  const int t = (int)(r % T);
  float acc[8], res[8];
#pragma unroll
  for (int j = 0; j < 8; j++) { acc[j] = b0[c + j]; res[j] = br0[c + j]; }
  for (int j = 0; j <= K; j++) {
    const int s = (j < K) ? K - 1 - j : 0;   // tap j reads id[t - (K-1-j)] (dilation 1); j == K: the residual reads id[t]
    if (t - s < 0) continue;                 // causal left padding: zeros (customized_tcn_cell.py:47-48)
    int id = ids[r - s];
    if (id <= 0 || id >= N) continue;        // null / out-of-range id: an all-zero one-hot row (model.py:59)
    const uint4 v = *reinterpret_cast<const uint4*>(tab + ((int64_t)j * N + id) * C0 + c);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float* dst = (j < K) ? acc : res;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      dst[2 * q] += bf2f((uint16_t)(w[q] & 0xffff));
      dst[2 * q + 1] += bf2f((uint16_t)(w[q] >> 16));
    }
  }
  uint4 o;
  uint32_t* ow = reinterpret_cast<uint32_t*>(&o);
#pragma unroll
  for (int q = 0; q < 4; q++)
    ow[q] = pack2bf(fmaxf(fmaxf(acc[2 * q], 0.f) + res[2 * q], 0.f), fmaxf(fmaxf(acc[2 * q + 1], 0.f) + res[2 * q + 1], 0.f));
  *reinterpret_cast<uint4*>(out + r * C0 + c) = o;
}

// prior logits: pl[r][m] = hidden[r] . pi_w[m] + pi_b[m]; one wave per row
__global__ __launch_bounds__(256) void k_gmm_prior(const uint16_t* __restrict__ hidden, const float* __restrict__ pi_w, const float* __restrict__ pi_b,
                                                   int64_t R, int C, int M, float* __restrict__ pl) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  for (int m = 0; m < M; m++) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(bf2f(hidden[r * C + c]), pi_w[m * C + c], s);
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) pl[r * M + m] = s + pi_b[m];
  }
}

// loss.py:247-257 from (|mu_m|^2, mu_m[target], prior logits); mode 0 gmm_prob, 1 gmm_prob_max; 0 on rows without a target
// (model.py:111 multiplies the loss by mask_y).  sq / tgt are [M][R].
__global__ void k_gmm_finish(const float* __restrict__ sq, const float* __restrict__ tgt, const float* __restrict__ pl, const int32_t* __restrict__ y,
                             int64_t R, int M, int D, float sigma, int mode, float* __restrict__ loss) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  if (y[r] <= 0) { loss[r] = 0.f; return; }
  float mx = -INFINITY;
  for (int m = 0; m < M; m++) mx = fmaxf(mx, pl[r * M + m]);
  float se = 0.f;
  for (int m = 0; m < M; m++) se += expf(pl[r * M + m] - mx);
  const float lse_pi = mx + logf(se);
  const double cst = -(double)D * log((double)sigma) - 0.5 * (double)D * log(2.0 * M_PI);   // -D log sigma - D/2 log 2pi
  const float inv2s2 = 0.5f / (sigma * sigma);
  // the constant -D log sigma - D/2 log 2pi (2.9e4 at D = 20,778) is common to every mixture: kept out of the fp32 terms
  float best = -INFINITY, zmax = -INFINITY;
  float z[16];
  for (int m = 0; m < M; m++) {
    const float d2 = sq[(int64_t)m * R + r] - 2.f * tgt[(int64_t)m * R + r] + 1.f;
    const float q = -inv2s2 * d2;                                   // mvn_m.log_prob(y) - cst
    const float lpi = pl[r * M + m] - lse_pi;                       // log pi_m
    z[m] = lpi + q;
    zmax = fmaxf(zmax, z[m]);
    best = fmaxf(best, lpi + q / (float)D);
  }
  if (mode == 1) { loss[r] = (float)(-((double)best + cst / (double)D)); return; }   // -log max_m pi_m exp(lp_m / D)
  float s = 0.f;
  for (int m = 0; m < M; m++) s += expf(z[m] - zmax);
  loss[r] = (float)(-((double)(zmax + logf(s)) + cst) / (double)D);                   // -log(exp(log_prob / D))
}

// ---- learned scale: sigma_d = exp(s_d), s = the second block of D outputs per mixture (the decode is the build's own: oracle/gmm_oracle.py)
// log N_m(y) = -1/2 [ Q_m + (1 - 2 mu_m[y]) exp(-2 s_m[y]) ] - S_m - D/2 log 2 pi   for a one-hot y, with
// Q_m = sum_d mu_d^2 exp(-2 s_d) (score.hip PMODE 22: one pass over the interleaved table) and S_m = sum_d s_d (linear in the hidden row).
__global__ void k_gmm_interleave(const uint16_t* __restrict__ muT, const float* __restrict__ mu_bias, int M, int DP, int NPs, int C,
                                 uint16_t* __restrict__ sgT, float* __restrict__ sg_bias) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8 = C / 8;
  if (i >= (int64_t)M * NPs * c8) return;
  const int c = (int)(i % c8) * 8;
  const int64_t mr = i / c8;
  const int row = (int)(mr % NPs), m = (int)(mr / NPs);
  const int d = (row >> 6) * 32 + (row & 31);            // output of this row
  const int tab = (row & 32) ? M + m : m;                 // second half of a stage: the log-scale table
  const bool in = d < DP;
  uint4 v = make_uint4(0, 0, 0, 0);
  if (in) v = *reinterpret_cast<const uint4*>(muT + ((size_t)tab * DP + d) * C + c);   // (cond ? *ptr : zero would go through scratch)
  *reinterpret_cast<uint4*>(sgT + ((size_t)m * NPs + row) * C + c) = v;
  if (c == 0) sg_bias[(size_t)m * NPs + row] = in ? mu_bias[(size_t)tab * DP + d] : 0.f;
}
// svec[m][c] = sum_d table_{M+m}[d][c], s0[m] = sum_d bias_{M+m}[d]   (fp64 sums of the stored values)
__global__ void k_gmm_scale_sums(const uint16_t* __restrict__ muT, const float* __restrict__ mu_bias, int M, int D, int DP, int C,
                                 float* __restrict__ svec, float* __restrict__ s0) {
  const int m = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > C) return;
  double acc = 0.0;
  if (c < C) {
    const uint16_t* tab = muT + (size_t)(M + m) * DP * C;
    for (int d = 0; d < D; d++) acc += (double)bf2f(tab[(size_t)d * C + c]);
    svec[(size_t)m * C + c] = (float)acc;
  } else {
    const float* b = mu_bias + (size_t)(M + m) * DP;
    for (int d = 0; d < D; d++) acc += (double)b[d];
    s0[m] = (float)acc;
  }
}
// 16 lanes per row: prior logits, S_m = h . svec_m + s0_m, mu_m[y] and s_m[y] from the gathered rows of tables m and M + m
__global__ __launch_bounds__(256) void k_gmm_prior_sigma(const uint16_t* __restrict__ hidden, const float* __restrict__ pi_w, const float* __restrict__ pi_b,
                                                          const float* __restrict__ svec, const float* __restrict__ s0, const uint16_t* __restrict__ muT,
                                                          const float* __restrict__ mu_bias, const int32_t* __restrict__ y, int64_t R, int C, int M, int D,
                                                          int DP, float* __restrict__ pl, float* __restrict__ tg, float* __restrict__ sy,
                                                          float* __restrict__ ssum) {
  const int l16 = threadIdx.x & 15;
  const int64_t r = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int64_t rc = r < R ? r : R - 1;
  const int yy = y[rc];
  const bool live = yy > 0 && yy < D;
  for (int m = 0; m < M; m++) {
    const uint16_t* mrow = muT + ((size_t)m * DP + (live ? yy : 0)) * C;
    const uint16_t* srow = muT + ((size_t)(M + m) * DP + (live ? yy : 0)) * C;
    float a[4] = {0.f, 0.f, 0.f, 0.f};   // prior, S, mu[y], s[y]
    for (int c = 8 * l16; c < C; c += 128) {
      const uint4 hv = *reinterpret_cast<const uint4*>(hidden + rc * C + c);
      const uint4 mv = *reinterpret_cast<const uint4*>(mrow + c);
      const uint4 sv = *reinterpret_cast<const uint4*>(srow + c);
      const uint32_t hw[4] = {hv.x, hv.y, hv.z, hv.w}, mw[4] = {mv.x, mv.y, mv.z, mv.w}, sw[4] = {sv.x, sv.y, sv.z, sv.w};
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const float h0 = bf2f((uint16_t)(hw[q] & 0xffff)), h1 = bf2f((uint16_t)(hw[q] >> 16));
        a[0] = fmaf(h0, pi_w[(size_t)m * C + c + 2 * q], fmaf(h1, pi_w[(size_t)m * C + c + 2 * q + 1], a[0]));
        a[1] = fmaf(h0, svec[(size_t)m * C + c + 2 * q], fmaf(h1, svec[(size_t)m * C + c + 2 * q + 1], a[1]));
        a[2] = fmaf(h0, bf2f((uint16_t)(mw[q] & 0xffff)), fmaf(h1, bf2f((uint16_t)(mw[q] >> 16)), a[2]));
        a[3] = fmaf(h0, bf2f((uint16_t)(sw[q] & 0xffff)), fmaf(h1, bf2f((uint16_t)(sw[q] >> 16)), a[3]));
      }
    }
#pragma unroll
    for (int off = 8; off > 0; off >>= 1)
#pragma unroll
      for (int k = 0; k < 4; k++) a[k] += __shfl_xor(a[k], off);
    if (l16 == 0 && r < R) {
      pl[r * M + m] = a[0] + pi_b[m];
      ssum[(size_t)m * R + r] = a[1] + s0[m];
      tg[(size_t)m * R + r] = live ? a[2] + mu_bias[(size_t)m * DP + yy] : 0.f;
      sy[(size_t)m * R + r] = live ? a[3] + mu_bias[(size_t)(M + m) * DP + yy] : 0.f;
    }
  }
}
__global__ void k_gmm_finish_sigma(const float* __restrict__ q, const float* __restrict__ tgt, const float* __restrict__ sy, const float* __restrict__ ssum,
                                   const float* __restrict__ pl, const int32_t* __restrict__ y, int64_t R, int M, int D, int mode, float* __restrict__ loss) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= R) return;
  if (y[r] <= 0 || y[r] >= D) { loss[r] = 0.f; return; }
  float mx = -INFINITY;
  for (int m = 0; m < M; m++) mx = fmaxf(mx, pl[r * M + m]);
  float se = 0.f;
  for (int m = 0; m < M; m++) se += expf(pl[r * M + m] - mx);
  const float lse_pi = mx + logf(se);
  const double cst = -0.5 * (double)D * log(2.0 * M_PI);
  float best = -INFINITY, zmax = -INFINITY;
  float z[16];
  for (int m = 0; m < M; m++) {
    const int64_t k = (int64_t)m * R + r;
    const float lp = -0.5f * (q[k] + (1.f - 2.f * tgt[k]) * expf(-2.f * sy[k])) - ssum[k];   // mvn_m.log_prob(y) - cst
    const float lpi = pl[r * M + m] - lse_pi;
    z[m] = lpi + lp;
    zmax = fmaxf(zmax, z[m]);
    best = fmaxf(best, lpi + lp / (float)D);
  }
  if (mode == 1) { loss[r] = (float)(-((double)best + cst / (double)D)); return; }
  float s = 0.f;
  for (int m = 0; m < M; m++) s += expf(z[m] - zmax);
  loss[r] = (float)(-((double)(zmax + logf(s)) + cst) / (double)D);
}

extern "C" {

int hiertcn_gmm_create(hiertcn_gmm_handle_t* out, const hiertcn_gmm_config_t* cfg) {
  if (!out || !cfg) return gfail(HIERTCN_ERR_INVALID, "null argument");
  if ((cfg->trunk != 2 && cfg->trunk != 3 && cfg->item_num < 2) || cfg->output_dim < 2) return gfail(HIERTCN_ERR_INVALID, "item_num / output_dim must be >= 2");
  const bool gru = cfg->trunk == 1 || cfg->trunk == 3;
  const bool rows = cfg->trunk == 2 || cfg->trunk == 3;   // dense input rows (the low level of model_hier): trunk 2 TemporalConvNet (block 0 an ordinary block), 3 GRU stack
  if (cfg->trunk < 0 || cfg->trunk > 3)
    return gfail(HIERTCN_ERR_INVALID, "trunk %d: 0 (TemporalConvNet on ids), 1 (GRU stack on ids), 2 (TemporalConvNet on rows), 3 (GRU stack on rows)", cfg->trunk);
  if (rows && (cfg->input_dim < 32 || (cfg->input_dim & 31) || cfg->input_dim > 4096))
    return gfail(HIERTCN_ERR_INVALID, "input_dim %d: the width of the input rows is a multiple of 32 up to 4096", cfg->input_dim);
  if (gru) {
    const int H = cfg->hidden_dim;
    if (!(H == 64 || H == 128 || H == 256 || H == 512)) return gfail(HIERTCN_ERR_INVALID, "hidden_dim %d: the head kernel is built for 64, 128, 256, 512", H);
    if (cfg->num_layer < 1 || cfg->num_layer > HIERTCN_MAX_LAYERS) return gfail(HIERTCN_ERR_INVALID, "num_layer out of range");
  }
  if (!gru && (cfg->n_blocks < 1 || cfg->n_blocks > HIERTCN_MAX_BLOCKS)) return gfail(HIERTCN_ERR_INVALID, "n_blocks out of range");
  if (!gru && (cfg->kernel_size < 1 || cfg->kernel_size > 8)) return gfail(HIERTCN_ERR_INVALID, "kernel_size out of range");
  if (cfg->num_mixture < 1 || cfg->num_mixture > 16) return gfail(HIERTCN_ERR_INVALID, "num_mixture %d: 1..16", cfg->num_mixture);
  if (!(cfg->sigma >= 0.f)) return gfail(HIERTCN_ERR_INVALID, "sigma must be > 0, or 0 for a learned scale (args.sigma None: n_out = M (1 + 2 D))");
  for (int i = 0; i < cfg->n_blocks && !gru; i++)
    if (cfg->channels[i] < 32 || (cfg->channels[i] & 31) || cfg->channels[i] > 1024)
      return gfail(HIERTCN_ERR_INVALID, "tcn_channel[%d]=%d: widths are multiples of 32 up to 1024", i, cfg->channels[i]);
  const int cl = gru ? cfg->hidden_dim : cfg->channels[cfg->n_blocks - 1];
  if (!(cl == 64 || cl == 128 || cl == 256 || cl == 512)) return gfail(HIERTCN_ERR_INVALID, "last tcn_channel %d: the head kernel is built for 64, 128, 256, 512", cl);
  if (cfg->sigma == 0.f && cl > 256) return gfail(HIERTCN_ERR_INVALID, "a learned scale is built for a last width of 64, 128 or 256 (got %d)", cl);
  hiertcn_gmm_ctx* c = new hiertcn_gmm_ctx();
  memset(static_cast<void*>(&c->cfg), 0, sizeof(c->cfg));
  c->cfg = *cfg;
  c->weights_set = false;
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; return gfail(HIERTCN_ERR_HIP, "hipGetDevice failed (no GPU?)"); }
  c->N = rows ? 0 : cfg->item_num; c->NP = (c->N + 63) & ~63;
  c->D = cfg->output_dim; c->DP = (c->D + 63) & ~63;
  c->M = cfg->num_mixture; c->K = cfg->kernel_size; c->nb = gru ? 0 : cfg->n_blocks;
  c->C0 = gru ? 0 : cfg->channels[0]; c->Clast = cl;
  c->trunk = cfg->trunk; c->H = gru ? cfg->hidden_dim : 0; c->nl = gru ? cfg->num_layer : 0;
  c->gather = nullptr; c->b0 = c->br0 = nullptr; c->gin_g = c->gin_c = nullptr;
  memset(c->gwg, 0, sizeof(c->gwg)); memset(c->gbg, 0, sizeof(c->gbg)); memset(c->gwc, 0, sizeof(c->gwc)); memset(c->gbc, 0, sizeof(c->gbc));
  memset(c->convT, 0, sizeof(c->convT)); memset(c->conv_bias, 0, sizeof(c->conv_bias));
  memset(c->resT, 0, sizeof(c->resT)); memset(c->res_bias, 0, sizeof(c->res_bias));
#define GA(ptr, count)                                                                              \
  do {                                                                                              \
    hipError_t e_ = galloc(c, &(ptr), (size_t)(count));                                             \
    if (e_ != hipSuccess) { hiertcn_gmm_destroy(c); return gfail(HIERTCN_ERR_HIP, "hipMalloc(%s) failed: %s", #ptr, hipGetErrorString(e_)); } \
  } while (0)
  if (gru) {
    const size_t H = (size_t)c->H;
    if (!rows) {
      GA(c->gin_g, (size_t)c->N * 2 * H);
      GA(c->gin_c, (size_t)c->N * H);
    }
    for (int l = 0; l < c->nl; l++) {
      const size_t kin = (l == 0) ? (rows ? (size_t)cfg->input_dim + H : H) : 2 * H;   // one-hot input: layer 0 keeps the state half only
      GA(c->gwg[l], 2 * H * kin); GA(c->gbg[l], 2 * H);
      GA(c->gwc[l], H * kin);     GA(c->gbc[l], H);
    }
  } else if (!rows) {
    GA(c->gather, (size_t)(c->K + 1) * c->N * c->C0);
    GA(c->b0, c->C0);
    GA(c->br0, c->C0);
  }
  for (int i = rows ? 0 : 1, cin = rows ? cfg->input_dim : c->C0; i < c->nb; cin = cfg->channels[i], i++) {   // (nb == 0 for the GRU trunks)
    GA(c->convT[i], (size_t)c->K * cin * cfg->channels[i]);
    GA(c->conv_bias[i], cfg->channels[i]);
    if (cin != cfg->channels[i]) { GA(c->resT[i], (size_t)cin * cfg->channels[i]); GA(c->res_bias[i], cfg->channels[i]); }
  }
  c->learned = cfg->sigma == 0.f;
  const size_t n_tab = c->learned ? 2 * (size_t)c->M : (size_t)c->M;   // learned scale: tables M .. 2M-1 are the log-scale blocks
  GA(c->muT, n_tab * c->DP * cl);
  GA(c->mu_bias, n_tab * c->DP);
  c->NPs = ((c->D + 31) / 32) * 64;
  c->sgT = nullptr; c->sg_bias = c->svec = c->s0 = nullptr;
  if (c->learned) {
    GA(c->sgT, (size_t)c->M * c->NPs * cl);
    GA(c->sg_bias, (size_t)c->M * c->NPs);
    GA(c->svec, (size_t)c->M * cl);
    GA(c->s0, c->M);
  }
  GA(c->pi_w, (size_t)c->M * cl);
  GA(c->pi_b, c->M);
  c->gram = !c->learned && gmm_gram_applies(cl, c->D) && !cfg->table_pass;
  c->gram64 = nullptr; c->G3 = nullptr; c->gx = c->g0 = nullptr;
  if (c->gram) {
    GA(c->gram64, gmm_gram_scratch_doubles(c->M, cl));
    GA(c->G3, gmm_gram_table_bytes(c->M, cl));
    GA(c->gx, (size_t)c->M * cl);
    GA(c->g0, c->M);
  }
#undef GA
  *out = c;
  return HIERTCN_OK;
}

int hiertcn_gmm_destroy(hiertcn_gmm_handle_t g) {
  if (!g) return HIERTCN_OK;
  gmm_train_destroy(g->train);
  for (void* p : g->allocs) (void)hipFree(p);
  delete g;
  return HIERTCN_OK;
}

// strided 2-D copies used below: dst[r][c] (ld_dst) = src[r][c0 + c] (ld_src), fp32
__global__ void k_copy_cols_f32(const float* __restrict__ src, int64_t ld_src, int64_t c0, float* __restrict__ dst, int64_t ld_dst, int64_t rows,
                                int64_t cols) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  const int64_t r = i / cols, c = i - r * cols;
  dst[r * ld_dst + c] = src[r * ld_src + c0 + c];
}
// muT[m][n][c] = bf16(W[c][M + m D + n]) for n < D, 0 for D <= n < DP, m < T tables (T = M, or 2 M with a learned scale: the log-scale blocks
// follow the mean blocks); one thread per (m, n, c) with c fastest
__global__ void k_gmm_head_tables(const float* __restrict__ W, int64_t n_out, int C, int M, int T, int D, int DP, uint16_t* __restrict__ muT) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)T * DP * C) return;
  const int c = (int)(i % C);
  const int64_t mn = i / C;
  const int n = (int)(mn % DP), m = (int)(mn / DP);
  muT[i] = n < D ? f2bf(W[(int64_t)c * n_out + M + (int64_t)m * D + n]) : (uint16_t)0;
}

int hiertcn_gmm_set_weights(hiertcn_gmm_handle_t g, const hiertcn_gmm_weights_t* w, void* stream) {
  if (!g || !w) return gfail(HIERTCN_ERR_INVALID, "null argument");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (!w->head_weights || !w->head_biases) return gfail(HIERTCN_ERR_INVALID, "the head (fully_connected weights / biases) is required");
  const int N = g->N, C0 = g->C0, K = g->K, M = g->M, D = g->D, DP = g->DP, CL = g->Clast;
  if (g->trunk == 1 || g->trunk == 3) {
    const size_t H = (size_t)g->H;
    for (int l = 0; l < g->nl; l++) {
      if (!w->gru_gate_kernel[l] || !w->gru_gate_bias[l] || !w->gru_cand_kernel[l] || !w->gru_cand_bias[l])
        return gfail(HIERTCN_ERR_INVALID, "GRU layer %d: gate / candidate kernel and bias are required", l);
      const size_t in = (l == 0) ? (g->trunk == 3 ? (size_t)g->cfg.input_dim : (size_t)N) : H;
      if (l == 0 && g->trunk == 3) {   // dense input rows: the whole [in + H][.] kernels, output-major
        G_HIP(launch_transpose_f32(w->gru_gate_kernel[0], g->gwg[0], (int64_t)(in + H), (int64_t)(2 * H), st));
        G_HIP(launch_transpose_f32(w->gru_cand_kernel[0], g->gwc[0], (int64_t)(in + H), (int64_t)H, st));
      } else if (l == 0) {   // rows [0, N) of [N + H][.]: the input half, already item-major; rows [N, N + H): the state half -> output-major
        G_HIP(hipMemcpyAsync(g->gin_g, w->gru_gate_kernel[0], (size_t)N * 2 * H * sizeof(float), hipMemcpyDeviceToDevice, st));
        G_HIP(hipMemcpyAsync(g->gin_c, w->gru_cand_kernel[0], (size_t)N * H * sizeof(float), hipMemcpyDeviceToDevice, st));
        G_HIP(launch_transpose_f32(w->gru_gate_kernel[0] + in * 2 * H, g->gwg[0], (int64_t)H, (int64_t)(2 * H), st));
        G_HIP(launch_transpose_f32(w->gru_cand_kernel[0] + in * H, g->gwc[0], (int64_t)H, (int64_t)H, st));
      } else {
        G_HIP(launch_transpose_f32(w->gru_gate_kernel[l], g->gwg[l], (int64_t)(2 * H), (int64_t)(2 * H), st));
        G_HIP(launch_transpose_f32(w->gru_cand_kernel[l], g->gwc[l], (int64_t)(2 * H), (int64_t)H, st));
      }
      G_HIP(hipMemcpyAsync(g->gbg[l], w->gru_gate_bias[l], 2 * H * sizeof(float), hipMemcpyDeviceToDevice, st));
      G_HIP(hipMemcpyAsync(g->gbc[l], w->gru_cand_bias[l], H * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
  } else {
  const bool rows = g->trunk == 2;
  if (!rows) {
  if (!w->conv_kernel[0] || !w->conv_bias[0] || !w->res_kernel[0] || !w->res_bias[0])
    return gfail(HIERTCN_ERR_INVALID, "block 0 (conv1 + Dense residual: Cin = item_num != Cout) is required");
  // block 0: the K taps [N][C0] and the residual kernel [N][C0] are already item-major in the reference layout: rows -> bf16
  for (int j = 0; j < K; j++) G_HIP(launch_rows_to_bf16(w->conv_kernel[0] + (size_t)j * N * C0, g->gather + (size_t)j * N * C0, N, C0, 0, st));
  G_HIP(launch_rows_to_bf16(w->res_kernel[0], g->gather + (size_t)K * N * C0, N, C0, 0, st));
  G_HIP(launch_copy_pad_f32(w->conv_bias[0], g->b0, C0, C0, 0.f, st));
  G_HIP(launch_copy_pad_f32(w->res_bias[0], g->br0, C0, C0, 0.f, st));
  }
  for (int i = rows ? 0 : 1, cin = rows ? g->cfg.input_dim : C0; i < g->nb; cin = g->cfg.channels[i], i++) {
    const int cout = g->cfg.channels[i];
    if (!w->conv_kernel[i] || !w->conv_bias[i]) return gfail(HIERTCN_ERR_INVALID, "conv_kernel[%d]/conv_bias[%d] missing", i, i);
    if ((cin != cout) != (w->res_kernel[i] != nullptr)) return gfail(HIERTCN_ERR_INVALID, "block %d: a Dense residual exists exactly when Cin != Cout", i);
    for (int j = 0; j < K; j++)
      G_HIP(launch_transpose_to_bf16(w->conv_kernel[i] + (size_t)j * cin * cout, g->convT[i] + (size_t)j * cin * cout, cin, cout, cout, st));
    G_HIP(launch_copy_pad_f32(w->conv_bias[i], g->conv_bias[i], cout, cout, 0.f, st));
    if (w->res_kernel[i]) {
      G_HIP(launch_transpose_to_bf16(w->res_kernel[i], g->resT[i], cin, cout, cout, st));
      G_HIP(launch_copy_pad_f32(w->res_bias[i], g->res_bias[i], cout, cout, 0.f, st));
    }
  }
  }
  // head [C][M (1 + D)]: columns [0, M) -> prior kernel [M][C] fp32; columns M + m D + n -> table m, row n (bf16, item-major)
  const int T = g->learned ? 2 * M : M;   // tables: means, then (learned scale) log-scales
  const int64_t n_out = (int64_t)M + (int64_t)T * D;
  {
    // pi_w[m][c] = W[c][m]
    const int64_t n = (int64_t)M * CL;
    // transpose through the column copy: tmp[c][m] = W[c][m] (ld n_out) then launch_transpose_f32 [C][M] -> [M][C]
    float* tmp = nullptr;
    G_HIP(hipMalloc(reinterpret_cast<void**>(&tmp), (size_t)n * sizeof(float)));
    hipLaunchKernelGGL(k_copy_cols_f32, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w->head_weights, n_out, (int64_t)0, tmp, (int64_t)M,
                       (int64_t)CL, (int64_t)M);
    hipError_t e = launch_transpose_f32(tmp, g->pi_w, CL, M, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    (void)hipFree(tmp);
    G_HIP(e);
  }
  G_HIP(hipMemcpyAsync(g->pi_b, w->head_biases, (size_t)M * sizeof(float), hipMemcpyDeviceToDevice, st));
  {
    const int64_t n = (int64_t)T * DP * CL;
    hipLaunchKernelGGL(k_gmm_head_tables, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, w->head_weights, n_out, CL, M, T, D, DP, g->muT);
    G_HIP(hipGetLastError());
  }
  G_HIP(hipMemsetAsync(g->mu_bias, 0, (size_t)T * DP * sizeof(float), st));
  for (int m = 0; m < T; m++)
    G_HIP(hipMemcpyAsync(g->mu_bias + (size_t)m * DP, w->head_biases + M + (size_t)m * D, (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (g->learned) {
    const int64_t n = (int64_t)M * g->NPs * (CL / 8);
    hipLaunchKernelGGL(k_gmm_interleave, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, g->muT, g->mu_bias, M, DP, g->NPs, CL, g->sgT, g->sg_bias);
    hipLaunchKernelGGL(k_gmm_scale_sums, dim3((unsigned)((CL + 1 + 255) / 256), (unsigned)M), dim3(256), 0, st, g->muT, g->mu_bias, M, D, DP, CL, g->svec, g->s0);
    G_HIP(hipGetLastError());
  }
  if (g->gram) G_HIP(launch_gmm_gram_build(g->muT, g->mu_bias, M, D, DP, CL, g->gram64, g->G3, g->gx, g->g0, st));
  g->weights_set = true;
  return HIERTCN_OK;
}

static int hiertcn_gmm_run_blocks(hiertcn_gmm_ctx* g, const uint16_t* in0, int first, int cin0, int64_t B, int32_t T, uint16_t* hidden, void* workspace,
                                  hipStream_t st);

static int max_w(const hiertcn_gmm_ctx* g) {
  int w = (g->trunk == 1 || g->trunk == 3) ? g->H : 0;
  for (int i = 0; i < g->nb; i++) w = g->cfg.channels[i] > w ? g->cfg.channels[i] : w;
  return w;
}

size_t hiertcn_gmm_workspace_bytes(hiertcn_gmm_handle_t g, int64_t R) {
  if (!g || R < 0) return 0;
  const size_t act = (((size_t)R * max_w(g) * 2) + 255) & ~(size_t)255;
  const size_t per = (((size_t)R * sizeof(float)) + 255) & ~(size_t)255;
  return 2 * act + 4 * g->M * per + ((((size_t)R * g->M * sizeof(float)) + 255) & ~(size_t)255);   // |mu|^2, mu[y], (learned scale) s[y], sum s; prior
}

size_t hiertcn_gmm_forward_workspace_bytes(hiertcn_gmm_handle_t g, int64_t B, int32_t T) {
  if (!g || B < 0 || T < 0) return 0;
  const size_t base = hiertcn_gmm_workspace_bytes(g, B * T);
  const size_t trunk = g->trunk == 1 ? onehot_gru_workspace_bytes(g->H, g->nl, B)
                       : g->trunk == 3 ? rows_gru_workspace_bytes(g->H, g->nl, g->cfg.input_dim, B, T) : 0;
  return base > trunk ? base : trunk;
}

#define G_READY(g)                                                                       \
  if (!(g)) return gfail(HIERTCN_ERR_INVALID, "null handle");                            \
  if (!(g)->weights_set) return gfail(HIERTCN_ERR_STATE, "hiertcn_gmm_set_weights has not been called");   \
  TuneScope tune_(nullptr)   /* the mixture handle has no tuning of its own: the defaults, never another handle's (common.h) */

int hiertcn_gmm_forward(hiertcn_gmm_handle_t g, const int32_t* x_ids, int64_t B, int32_t T, uint16_t* hidden, void* workspace,
                        size_t workspace_bytes, void* stream) {
  G_READY(g);
  if (g->trunk >= 2) return gfail(HIERTCN_ERR_STATE, "this handle reads dense rows (trunk %d): hiertcn_gmm_forward_rows", g->trunk);
  if (!x_ids || !hidden || B < 1 || T < 1 || T > 4096) return gfail(HIERTCN_ERR_INVALID, "bad argument");
  const int64_t R = B * T;
  if (!workspace || workspace_bytes < hiertcn_gmm_forward_workspace_bytes(g, B, T)) return gfail(HIERTCN_ERR_WORKSPACE, "gmm workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (g->trunk == 1) {   // model_gru_gmm.py:33-48: MultiRNNCell of GRUCells under dynamic_rnn on the one-hot rows, from a zero state
    OneHotGru og;
    memset(&og, 0, sizeof(og));
    og.N = g->N; og.H = g->H; og.nl = g->nl; og.in_g = g->gin_g; og.in_c = g->gin_c;
    for (int l = 0; l < g->nl; l++) { og.wg[l] = g->gwg[l]; og.bg[l] = g->gbg[l]; og.wc[l] = g->gwc[l]; og.bc[l] = g->gbc[l]; }
    G_HIP(launch_onehot_gru(og, x_ids, B, T, hidden, workspace, st));
    return HIERTCN_OK;
  }
  const size_t act = (((size_t)R * max_w(g) * 2) + 255) & ~(size_t)255;
  uint16_t* buf[2] = {reinterpret_cast<uint16_t*>(workspace), reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(workspace) + act)};
  uint16_t* out0 = (g->nb == 1) ? hidden : buf[0];
  const int64_t n0 = R * (g->C0 / 8);
  hipLaunchKernelGGL(k_gmm_block0, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, st, g->gather, g->b0, g->br0, x_ids, R, T, g->N, g->C0, g->K, out0);
  G_HIP(hipGetLastError());
  return hiertcn_gmm_run_blocks(g, buf[0], 1, g->C0, B, T, hidden, workspace, st);
}

// blocks first .. nb-1 (dilation 2^i) from `in0` (which may be workspace buffer 0) to `hidden`, ping-ponging through the two workspace buffers
static int hiertcn_gmm_run_blocks(hiertcn_gmm_ctx* g, const uint16_t* in0, int first, int cin0, int64_t B, int32_t T, uint16_t* hidden, void* workspace,
                                  hipStream_t st) {
  const int64_t R = B * T;
  const size_t act = (((size_t)R * max_w(g) * 2) + 255) & ~(size_t)255;
  uint16_t* buf[2] = {reinterpret_cast<uint16_t*>(workspace), reinterpret_cast<uint16_t*>(reinterpret_cast<unsigned char*>(workspace) + act)};
  const uint16_t* cur_in = in0;
  int nxt = (in0 == buf[0]) ? 1 : 0;
  for (int i = first, cin = cin0; i < g->nb; cin = g->cfg.channels[i], i++) {
    LayerArgs a = {};
    a.in = cur_in;
    a.out = (i == g->nb - 1) ? hidden : buf[nxt];
    a.W = g->convT[i]; a.bias = g->conv_bias[i]; a.Wres = g->resT[i]; a.bres = g->res_bias[i];
    a.Cin = cin; a.Cout = g->cfg.channels[i]; a.taps = g->K; a.dil = 1 << i; a.L = T; a.n_seq = B; a.linear = 0;
    G_HIP(launch_tcn_layer(a, st));
    cur_in = a.out;
    nxt ^= 1;
  }
  return HIERTCN_OK;
}

int hiertcn_gmm_forward_rows(hiertcn_gmm_handle_t g, const uint16_t* x_rows, int64_t B, int32_t T, uint16_t* hidden, void* workspace,
                             size_t workspace_bytes, void* stream) {
  G_READY(g);
  if (g->trunk < 2) return gfail(HIERTCN_ERR_STATE, "hiertcn_gmm_forward_rows is for trunks 2 / 3 (dense input rows); this handle's trunk is %d", g->trunk);
  if (!x_rows || !hidden || B < 1 || T < 1 || T > 4096) return gfail(HIERTCN_ERR_INVALID, "bad argument");
  if (!workspace || workspace_bytes < hiertcn_gmm_forward_workspace_bytes(g, B, T)) return gfail(HIERTCN_ERR_WORKSPACE, "gmm workspace too small");
  if (g->trunk == 3) {   // model_hier.py:68-69 -> model_gru_gmm.py:33-48: the GRU stack under dynamic_rnn on the [e_t, state] rows, from a zero state
    OneHotGru og;
    memset(&og, 0, sizeof(og));
    og.N = 0; og.H = g->H; og.nl = g->nl;
    for (int l = 0; l < g->nl; l++) { og.wg[l] = g->gwg[l]; og.bg[l] = g->gbg[l]; og.wc[l] = g->gwc[l]; og.bc[l] = g->gbc[l]; }
    G_HIP(launch_rows_gru(og, x_rows, g->cfg.input_dim, B, T, hidden, workspace, reinterpret_cast<hipStream_t>(stream)));
    return HIERTCN_OK;
  }
  return hiertcn_gmm_run_blocks(g, x_rows, 0, g->cfg.input_dim, B, T, hidden, workspace, reinterpret_cast<hipStream_t>(stream));
}

static void table_model(const hiertcn_gmm_ctx* g, int m, ModelDev* md) {
  memset(md, 0, sizeof(*md));
  tuning_defaults(&md->tune);
  md->N = g->D; md->NP = g->DP; md->Clast = g->Clast;
  md->outT = g->muT + (size_t)m * g->DP * g->Clast;
  md->out_bias = g->mu_bias + (size_t)m * g->DP;
}

int hiertcn_gmm_loss(hiertcn_gmm_handle_t g, const uint16_t* hidden, const int32_t* y_ids, int64_t R, int32_t mode, float* loss, float* sq_out,
                     float* tgt_out, float* prior_out, void* workspace, size_t workspace_bytes, void* stream) {
  G_READY(g);
  if (!hidden || !y_ids || !loss || R < 1 || mode < 0 || mode > 1) return gfail(HIERTCN_ERR_INVALID, "bad argument");
  if (!workspace || workspace_bytes < hiertcn_gmm_workspace_bytes(g, R)) return gfail(HIERTCN_ERR_WORKSPACE, "gmm workspace too small");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const size_t act = (((size_t)R * max_w(g) * 2) + 255) & ~(size_t)255;
  const size_t per = (((size_t)R * sizeof(float)) + 255) & ~(size_t)255;
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace) + 2 * act;
  float* sq = sq_out ? sq_out : reinterpret_cast<float*>(ws);                       // [M][R]
  float* tg = tgt_out ? tgt_out : reinterpret_cast<float*>(ws + g->M * per);        // [M][R]
  float* pl = prior_out ? prior_out : reinterpret_cast<float*>(ws + 4 * g->M * per);   // [R][M]
  if (g->learned) {
    // sq = Q_m = sum_d mu_d^2 exp(-2 s_d): one pass per mixture over the interleaved (means | log-scales) table; everything else row-wise
    float* sy = reinterpret_cast<float*>(ws + 2 * g->M * per);                         // [M][R] s_m[y]
    float* ssum = reinterpret_cast<float*>(ws + 3 * g->M * per);                       // [M][R] sum_d s_d
    for (int m = 0; m < g->M; m++) {
      ModelDev md;
      memset(&md, 0, sizeof(md));
      tuning_defaults(&md.tune);
      md.N = md.NP = g->NPs; md.Clast = g->Clast;
      md.outT = g->sgT + (size_t)m * g->NPs * g->Clast;
      md.out_bias = g->sg_bias + (size_t)m * g->NPs;
      G_HIP(launch_score_sigma(md, hidden, y_ids, R, sq + (size_t)m * R, st));
    }
    hipLaunchKernelGGL(k_gmm_prior_sigma, dim3((unsigned)((R + 15) / 16)), dim3(256), 0, st, hidden, g->pi_w, g->pi_b, g->svec, g->s0, g->muT, g->mu_bias,
                       y_ids, R, g->Clast, g->M, g->D, g->DP, pl, tg, sy, ssum);
    hipLaunchKernelGGL(k_gmm_finish_sigma, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, sq, tg, sy, ssum, pl, y_ids, R, g->M, g->D, mode, loss);
    G_HIP(hipGetLastError());
    return HIERTCN_OK;
  }
  if (g->gram) {
    // |mu_m|^2 = x^T G_m x, x = [h; 1] (gmm_gram.hip): the C x C part on the MFMA, the linear / constant terms, the prior logits and the target
    // mean (a gathered dot product) in one row-wise kernel -- no pass over the [D, C] tables
    G_HIP(launch_gmm_quad(hidden, R, g->Clast, g->M, g->G3, sq, st));
    G_HIP(launch_gmm_prior_gram(hidden, g->pi_w, g->pi_b, g->gx, g->g0, g->muT, g->mu_bias, y_ids, R, g->Clast, g->M, g->D, g->DP, pl, sq, tg, st));
  } else {
    for (int m = 0; m < g->M; m++) {
      ModelDev md;
      table_model(g, m, &md);
      G_HIP(launch_score_sumsq(md, hidden, y_ids, R, sq + (size_t)m * R, tg + (size_t)m * R, st));
    }
    hipLaunchKernelGGL(k_gmm_prior, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, st, hidden, g->pi_w, g->pi_b, R, g->Clast, g->M, pl);
  }
  hipLaunchKernelGGL(k_gmm_finish, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, sq, tg, pl, y_ids, R, g->M, g->D, g->cfg.sigma, mode, loss);
  G_HIP(hipGetLastError());
  return HIERTCN_OK;
}

// model.py:111-117 on a per-step loss: mask by sign(y), per-lane mean over the valid steps (+1e-6), summed over the lanes that have one;
// one workgroup, a lane per thread, fixed-order tree: deterministic.  out2 = [sum of lane means, user count].
__global__ __launch_bounds__(1024) void k_gmm_lane_loss(const float* __restrict__ loss, const int32_t* __restrict__ y, int B, int T,
                                                        double* __restrict__ out2) {
  __shared__ double s_sum[1024];
  __shared__ double s_usr[1024];
  double acc = 0.0, usr = 0.0;
  for (int b = threadIdx.x; b < B; b += 1024) {
    float ls = 0.f;
    int cnt = 0;
    for (int t = 0; t < T; t++) {
      const bool v = y[(int64_t)b * T + t] > 0;
      ls += v ? loss[(int64_t)b * T + t] : 0.f;
      cnt += v ? 1 : 0;
    }
    if (cnt > 0) { acc += (double)(ls / ((float)cnt + 1e-6f)); usr += 1.0; }
  }
  s_sum[threadIdx.x] = acc;
  s_usr[threadIdx.x] = usr;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { s_sum[threadIdx.x] += s_sum[threadIdx.x + o]; s_usr[threadIdx.x] += s_usr[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out2[0] = s_sum[0]; out2[1] = s_usr[0]; }
}

int hiertcn_gmm_lane_loss(hiertcn_gmm_handle_t g, const float* loss, const int32_t* y_ids, int32_t B, int32_t T, double* out2, void* stream) {
  if (!g) return gfail(HIERTCN_ERR_INVALID, "null handle");
  if (!loss || !y_ids || !out2 || B < 1 || T < 1) return gfail(HIERTCN_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(k_gmm_lane_loss, dim3(1), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), loss, y_ids, B, T, out2);
  G_HIP(hipGetLastError());
  return HIERTCN_OK;
}

int hiertcn_gmm_head_means(hiertcn_gmm_handle_t g, const uint16_t* hidden, int64_t R, int32_t m, float* mu_out, void* stream) {
  G_READY(g);
  if (!hidden || !mu_out || R < 1 || m < 0 || m >= (g->learned ? 2 * g->M : g->M)) return gfail(HIERTCN_ERR_INVALID, "bad argument");
  ModelDev md;
  table_model(g, m, &md);
  G_HIP(launch_score(md, hidden, nullptr, R, nullptr, mu_out, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

}  // extern "C"
