// C-ABI of libhiertcn_gfx950.so (include/hiertcn.h).  Host-side glue only: argument checks,
// the per-device context that owns the converted weights, and stream-ordered kernel launches.
// Nothing here allocates or synchronises on the hot path.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "exact.h"
#include "train.h"

static thread_local char g_err[512] = "";

// ---- launch selection + thresholds (include/hiertcn.h: hiertcn_tuning_t).  The library reads no environment variable. -----------------------
void tuning_defaults(hiertcn_tuning_t* t) {
  memset(t, 0, sizeof(*t));
  t->gru_scan_persistent = t->gru_scan_persistent_train = t->gru_bptt_persistent = t->gru_bptt_fused = -1;
  t->lgru_persistent = t->lgru_pipe = t->lgru_tile = t->lgru_tile_bwd = t->lgru_wide = t->lgru_wide_bwd = -1;
  t->lgru_fp32_cells = -1;
  t->tcn_layered = 0;
  t->gru_persist_max_lanes = 96;
  t->gemm_split_max_tiles = 1024;
  t->skinny_max_rows = 4096;
  t->lgru_tile_max_sessions = 131072;
  t->lgru_wide_max_launches = 3;
  t->sgrad_dyn_max_blocks = 1024;
  t->sgrad0_min_stages = 64;
  t->tcn_gs2_max_sessions = 512;
  t->lgru_defer_max_gib = 8;
  t->train_fp32_chunk_mib = 8192;
}
static const hiertcn_tuning_t* default_tuning() {
  static const hiertcn_tuning_t d = [] { hiertcn_tuning_t t; tuning_defaults(&t); return t; }();
  return &d;
}
static thread_local const hiertcn_tuning_t* g_tune = nullptr;   // set only INSIDE an API call (TuneScope); null between calls
const hiertcn_tuning_t& tune_now() { return g_tune ? *g_tune : *default_tuning(); }
TuneScope::TuneScope(const hiertcn_tuning_t* t) : prev(g_tune) { g_tune = t; }
TuneScope::~TuneScope() { g_tune = prev; }

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                                   \
  do {                                                                                                  \
    hipError_t e_ = (expr);                                                                             \
    if (e_ != hipSuccess) return fail(HIERTCN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct hiertcn_ctx {
  hiertcn_config_t cfg;
  ModelDev m;
  int device;
  bool weights_set;
  std::vector<void*> allocs;
  double* metric_scratch;
  TrainCtx train;
  ExactDev x;        // fp32 copies of the 16-bit tensors (hiertcn_set_precision)
  int precision;
  const int32_t* score_ids = nullptr;   // one-shot (hiertcn_set_score_ids): the next evaluation / training step scores against these ids
  // timing hook
  bool timing;
  std::vector<hipEvent_t> ev_begin, ev_end;
  size_t ev_used;
};

template <typename T>
static hipError_t dev_alloc(hiertcn_ctx* c, T** out, size_t count) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, count * sizeof(T) ? count * sizeof(T) : sizeof(T));
  if (e != hipSuccess) return e;
  c->allocs.push_back(p);
  *out = reinterpret_cast<T*>(p);
  return hipSuccess;
}

// Every entry point that touches the device runs on the handle's device, whatever the caller's current device is (a second
// handle on another GPU in the same process must not launch on the first one's device); the caller's device is restored on exit.
struct DeviceGuard {
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    if (prev == dev) prev = -1;
  }
  ~DeviceGuard() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
#define ON_DEVICE(h)                                                                       \
  DeviceGuard guard_((h)->device);                                                         \
  TuneScope tune_(&(h)->m.tune);                                                           \
  if (!guard_.ok) return fail(HIERTCN_ERR_HIP, "cannot switch to the handle's device %d", (h)->device)

extern "C" {

int hiertcn_abi_version(void) { return HIERTCN_ABI_VERSION; }
int hiertcn_storage_dtype(void) {
#ifdef HIERTCN_F16
  return 1;
#else
  return 0;
#endif
}
const char* hiertcn_last_error(void) { return g_err; }
#ifdef HIERTCN_TEST_HOOKS   // libhiertcn_gfx950_testhooks.so only: 1 while this thread is inside a TuneScope (never, seen from a caller between calls)
int hiertcn_test_tune_in_scope(void) { return g_tune != nullptr; }
#endif

int hiertcn_create(hiertcn_handle_t* out, const hiertcn_config_t* cfg_in) {
  if (!out || !cfg_in) return fail(HIERTCN_ERR_INVALID, "null argument");
  hiertcn_config_t cfg_local = *cfg_in;
  const hiertcn_config_t* cfg = &cfg_local;
  const int E = cfg->emb_dim;
  if (cfg->item_num < 2) return fail(HIERTCN_ERR_INVALID, "item_num must be >= 2 (row 0 is the null id)");
  if (E < 32 || (E & 31) || E > 1024)
    return fail(HIERTCN_ERR_INVALID, "emb_dim %d unsupported: widths are multiples of 32 (one MFMA tile), zero-pad other widths on the host", E);
  // low_type 1: the GRU stack of model_gru.py:33-55 as the low level of model_hier (HierGRU, model_hier.py:67-68) or, with hier = 0, as the single-level
  // model_gru of model.py:66-67 (the embedded ids feed the stack directly: model_gru.py:46 on one-hot rows IS the embedding lookup)
  const bool low_gru = cfg->low_type == 1;
  if (cfg->low_type != 0 && !low_gru) return fail(HIERTCN_ERR_INVALID, "low_type %d: 0 (tcn) or 1 (gru)", cfg->low_type);
  if (low_gru) {
    if (!(cfg->hidden_dim == 64 || cfg->hidden_dim == 128 || cfg->hidden_dim == 256 || cfg->hidden_dim == 512))
      return fail(HIERTCN_ERR_INVALID, "HierGRU: hidden_dim %d feeds the scoring kernel, which is built for 64, 128, 256, 512", cfg->hidden_dim);
    if (!cfg->hier) {
      if (!(E == 64 || E == 128 || E == 256 || E == 512))
        return fail(HIERTCN_ERR_INVALID, "model_gru: emb_dim %d: the GRU cells' input width is one of 64, 128, 256, 512", E);
      if (cfg->num_layer < 1 || cfg->num_layer > HIERTCN_MAX_LAYERS) return fail(HIERTCN_ERR_INVALID, "num_layer out of range");
    }
    // the TemporalConvNet fields do not apply: normalise them so that the shared checks / sizes below see one "block" of H channels
    cfg_local.n_blocks = 1; cfg_local.channels[0] = cfg->hidden_dim; cfg_local.kernel_size = 1;
  }
  if (cfg->n_blocks < 1 || cfg->n_blocks > HIERTCN_MAX_BLOCKS) return fail(HIERTCN_ERR_INVALID, "n_blocks out of range");
  bool uniform = (E == 64 || E == 128 || E == 256 || E == 512);
  for (int i = 0; i < cfg->n_blocks; i++) {
    const int ci = cfg->channels[i];
    if (ci < 32 || (ci & 31) || ci > 1024) return fail(HIERTCN_ERR_INVALID, "tcn_channel[%d]=%d: widths are multiples of 32 up to 1024", i, ci);
    uniform = uniform && ci == E;
  }
  const int clast = cfg->channels[cfg->n_blocks - 1];
  if (!(clast == 64 || clast == 128 || clast == 256 || clast == 512))
    return fail(HIERTCN_ERR_INVALID, "last tcn_channel %d unsupported: the scoring kernel is built for 64, 128, 256, 512", clast);
  if (cfg->kernel_size < 1 || cfg->kernel_size > 8) return fail(HIERTCN_ERR_INVALID, "kernel_size out of range");
  if (cfg->hier) {
    if (!(E == 64 || E == 128 || E == 256 || E == 512))
      return fail(HIERTCN_ERR_INVALID, "emb_dim %d: the session mean-pool kernel is built for 64, 128, 256, 512", E);
    if (cfg->num_layer < 1 || cfg->num_layer > HIERTCN_MAX_LAYERS) return fail(HIERTCN_ERR_INVALID, "num_layer out of range");
    if (cfg->hidden_dim < 32 || (cfg->hidden_dim & 31))
      return fail(HIERTCN_ERR_INVALID, "hidden_dim must be a positive multiple of 32 (one MFMA tile of GRU columns)");
  }

  hiertcn_ctx* c = new hiertcn_ctx();
  c->cfg = *cfg;
  c->weights_set = false;
  memset(&c->x, 0, sizeof(c->x));
  c->precision = HIERTCN_PRECISION_STORAGE16;
  c->timing = false;
  c->ev_used = 0;
  c->metric_scratch = nullptr;
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    return fail(HIERTCN_ERR_HIP, "hipGetDevice failed (no GPU?)");
  }
  ModelDev& m = c->m;
  memset(&m, 0, sizeof(m));
  tuning_defaults(&m.tune);
  m.N = cfg->item_num;
  m.NP = (cfg->item_num + 63) & ~63;
  m.E = E;
  m.hier = cfg->hier ? 1 : 0;
  m.H = (m.hier || low_gru) ? cfg->hidden_dim : 0;   // (single-level model_gru: the stack's own width and depth; there is no across-session GRU)
  m.nl = (m.hier || low_gru) ? cfg->num_layer : 0;
  m.HT = m.H * m.nl;
  m.K = cfg->kernel_size;
  m.n_blocks = cfg->n_blocks;
  for (int i = 0; i < m.n_blocks; i++) m.C[i] = cfg->channels[i];
  m.Clast = m.C[m.n_blocks - 1];
  m.low_gru = low_gru ? 1 : 0;
  m.fused_tcn = (uniform && !low_gru) ? 1 : 0;

#define ALLOC_OR_FAIL(ptr, count)                                  \
  do {                                                             \
    hipError_t e_ = dev_alloc(c, &(ptr), (size_t)(count));         \
    if (e_ != hipSuccess) {                                        \
      int rc_ = fail(HIERTCN_ERR_HIP, "hipMalloc(%s) failed: %s", #ptr, hipGetErrorString(e_)); \
      hiertcn_destroy(c);                                          \
      return rc_;                                                  \
    }                                                              \
  } while (0)

  ALLOC_OR_FAIL(m.emb, (size_t)m.N * E);
  ALLOC_OR_FAIL(m.outT, (size_t)m.NP * m.Clast);
  ALLOC_OR_FAIL(m.out_bias, (size_t)m.NP);
  for (int i = 0, cin = E; i < m.n_blocks && !low_gru; cin = m.C[i], i++) {
    ALLOC_OR_FAIL(m.convT[i], (size_t)m.K * cin * m.C[i]);
    ALLOC_OR_FAIL(m.convF[i], (size_t)m.K * cin * m.C[i]);
    ALLOC_OR_FAIL(m.conv_bias[i], (size_t)m.C[i]);
    if (cin != m.C[i]) {  // customized_tcn_cell.py:102-106: a biased Dense down-sample on the residual path
      ALLOC_OR_FAIL(m.resT[i], (size_t)cin * m.C[i]);
      ALLOC_OR_FAIL(m.resF[i], (size_t)cin * m.C[i]);
      ALLOC_OR_FAIL(m.res_bias[i], (size_t)m.C[i]);
    }
  }
  if (m.hier) {
    ALLOC_OR_FAIL(m.emb_bias, (size_t)E);
    ALLOC_OR_FAIL(m.projT, (size_t)E * E);
    ALLOC_OR_FAIL(m.projF, (size_t)E * E);
    ALLOC_OR_FAIL(m.projS, (size_t)m.HT * E);
    for (int l = 0; l < m.nl; l++) {
      const int in = (l == 0) ? E : m.H;
      ALLOC_OR_FAIL(m.gru_wg[l], (size_t)(in + m.H) * 2 * m.H);
      ALLOC_OR_FAIL(m.gru_bg[l], (size_t)2 * m.H);
      ALLOC_OR_FAIL(m.gru_wc[l], (size_t)(in + m.H) * m.H);
      ALLOC_OR_FAIL(m.gru_bc[l], (size_t)m.H);
    }
  }
  if (low_gru) {
    for (int l = 0; l < m.nl; l++) {
      const int in = (l == 0) ? E : m.H;
      {
        ALLOC_OR_FAIL(m.lgru_wg[l], (size_t)(in + m.H) * 2 * m.H);
        ALLOC_OR_FAIL(m.lgru_bg[l], (size_t)2 * m.H);
        ALLOC_OR_FAIL(m.lgru_wc[l], (size_t)(in + m.H) * m.H);
        ALLOC_OR_FAIL(m.lgru_bc[l], (size_t)m.H);
        ALLOC_OR_FAIL(m.lgru_wg_hi[l], (size_t)(in + m.H) * 2 * m.H);
        ALLOC_OR_FAIL(m.lgru_wg_lo[l], (size_t)(in + m.H) * 2 * m.H);
        ALLOC_OR_FAIL(m.lgru_wc_hi[l], (size_t)(in + m.H) * m.H);
        ALLOC_OR_FAIL(m.lgru_wc_lo[l], (size_t)(in + m.H) * m.H);
        ALLOC_OR_FAIL(m.lgru_wgF_hi[l], (size_t)(in + m.H) * 2 * m.H);
        ALLOC_OR_FAIL(m.lgru_wgF_lo[l], (size_t)(in + m.H) * 2 * m.H);
        ALLOC_OR_FAIL(m.lgru_wcF_hi[l], (size_t)(in + m.H) * m.H);
        ALLOC_OR_FAIL(m.lgru_wcF_lo[l], (size_t)(in + m.H) * m.H);
        ALLOC_OR_FAIL(m.lgru_wgTF_hi[l], (size_t)(in + m.H) * 2 * m.H);
        ALLOC_OR_FAIL(m.lgru_wgTF_lo[l], (size_t)(in + m.H) * 2 * m.H);
        ALLOC_OR_FAIL(m.lgru_wcTF_hi[l], (size_t)(in + m.H) * m.H);
        ALLOC_OR_FAIL(m.lgru_wcTF_lo[l], (size_t)(in + m.H) * m.H);
        m.lgru_split = 1;
      }
    }
  }
  ALLOC_OR_FAIL(c->metric_scratch, (size_t)8 * HIERTCN_METRIC_BLOCKS_MAX);
  {  // barrier counters of the persistent GRU scan (gru_persist.hip): 64-bit words on lines of their own, zero once, never reset
    int ncu = 0, dev = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) m.n_cu = ncu;
    unsigned long long* bar = nullptr;
    ALLOC_OR_FAIL(bar, (size_t)512);   // 8 group counters 128 bytes apart + the time-out word; the low-level GRU's 8 counters behind them
    if (hipMemset(bar, 0, 512 * sizeof(unsigned long long)) != hipSuccess) {
      hiertcn_destroy(c);
      return fail(HIERTCN_ERR_HIP, "hipMemset(gru_bar) failed");
    }
    m.gru_bar = bar;
  }
  if (!ABL_ENV("HIERTCN_FIXED_WALK")) {   // the scoring kernel's per-XCD item-walk words live with the handle, i.e. on ITS device
    ALLOC_OR_FAIL(m.walk_pos, (size_t)8 * 32);
    if (hipMemset(m.walk_pos, 0, 8 * 32 * sizeof(int)) != hipSuccess) {
      hiertcn_destroy(c);
      return fail(HIERTCN_ERR_HIP, "hipMemset(walk_pos) failed");
    }
    ALLOC_OR_FAIL(m.walk_pace, (size_t)8 * XW_PACE_INTS);   // slots -1 (= empty), tickets -1 (any start value does)
    if (hipMemset(m.walk_pace, 0xff, 8 * XW_PACE_INTS * sizeof(int)) != hipSuccess) {
      hiertcn_destroy(c);
      return fail(HIERTCN_ERR_HIP, "hipMemset(walk_pace) failed");
    }
  }
#undef ALLOC_OR_FAIL
  *out = c;
  return HIERTCN_OK;
}

int hiertcn_destroy(hiertcn_handle_t h) {
  if (!h) return HIERTCN_OK;
  DeviceGuard guard_(h->device);
  train_destroy(h->train);
  for (void* p : h->allocs) (void)hipFree(p);
  for (hipEvent_t e : h->ev_begin) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_end) (void)hipEventDestroy(e);
  delete h;
  return HIERTCN_OK;
}

int hiertcn_get_config(hiertcn_handle_t h, hiertcn_config_t* out) {
  if (!h || !out) return fail(HIERTCN_ERR_INVALID, "null argument");
  *out = h->cfg;
  return HIERTCN_OK;
}

int hiertcn_set_weights(hiertcn_handle_t h, const hiertcn_weights_t* w, void* stream) {
  if (!h || !w) return fail(HIERTCN_ERR_INVALID, "null argument");
  ON_DEVICE(h);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  ModelDev& m = h->m;
  const int E = m.E;
  if (!w->emb_kernel || !w->out_kernel || !w->out_bias) return fail(HIERTCN_ERR_INVALID, "emb_kernel/out_kernel/out_bias are required");
  // item table: [N][E] fp32 -> bf16, row 0 zeroed (model.py:59)
  HIP_TRY(launch_rows_to_bf16(w->emb_kernel, m.emb, m.N, E, 1, st));
  // scoring kernel: [C][N] -> [NP][C] bf16 (item-major rows = MFMA A operand); bias padded with -inf so
  // that padding items never outrank anything and add exp(-inf) = 0 to the softmax denominator
  HIP_TRY(launch_transpose_to_bf16(w->out_kernel, m.outT, m.Clast, m.N, m.NP, st));
  HIP_TRY(launch_copy_pad_f32(w->out_bias, m.out_bias, m.N, m.NP, -INFINITY, st));
  for (int i = 0, cin = E; i < m.n_blocks && !m.low_gru; cin = m.C[i], i++) {
    const int cout = m.C[i];
    if (!w->conv_kernel[i] || !w->conv_bias[i]) return fail(HIERTCN_ERR_INVALID, "conv_kernel[%d]/conv_bias[%d] missing", i, i);
    if ((cin != cout) != (w->res_kernel[i] != nullptr))
      return fail(HIERTCN_ERR_INVALID, "block %d: a down-sample residual exists exactly when Cin != Cout (customized_tcn_cell.py:102-106)", i);
    for (int j = 0; j < m.K; j++)  // tap j: [Cin][Cout] -> [Cout][Cin]
      HIP_TRY(launch_transpose_to_bf16(w->conv_kernel[i] + (size_t)j * cin * cout, m.convT[i] + (size_t)j * cin * cout, cin, cout, cout, st));
    HIP_TRY(launch_copy_pad_f32(w->conv_bias[i], m.conv_bias[i], cout, cout, 0.f, st));
    if (w->res_kernel[i]) {
      if (!w->res_bias[i]) return fail(HIERTCN_ERR_INVALID, "res_bias[%d] missing", i);
      HIP_TRY(launch_transpose_to_bf16(w->res_kernel[i], m.resT[i], cin, cout, cout, st));
      HIP_TRY(launch_copy_pad_f32(w->res_bias[i], m.res_bias[i], cout, cout, 0.f, st));
    }
  }
  if (m.hier) {
    if (!w->emb_bias || !w->proj_kernel) return fail(HIERTCN_ERR_INVALID, "emb_bias/proj_kernel are required for hier models");
    HIP_TRY(launch_copy_pad_f32(w->emb_bias, m.emb_bias, E, E, 0.f, st));
    HIP_TRY(launch_transpose_to_bf16(w->proj_kernel, m.projT, E, E, E, st));  // rows [0,E): item half, bf16, cout-major
    HIP_TRY(launch_transpose_f32(w->proj_kernel + (size_t)E * E, m.projS, m.HT, E, st));  // state half, fp32, [E][HT]
    for (int l = 0; l < m.nl; l++) {
      if (!w->gru_gate_kernel[l] || !w->gru_gate_bias[l] || !w->gru_cand_kernel[l] || !w->gru_cand_bias[l])
        return fail(HIERTCN_ERR_INVALID, "GRU layer %d weights missing", l);
      const int in = (l == 0) ? E : m.H;
      HIP_TRY(launch_transpose_f32(w->gru_gate_kernel[l], m.gru_wg[l], in + m.H, 2 * m.H, st));
      HIP_TRY(launch_copy_pad_f32(w->gru_gate_bias[l], m.gru_bg[l], 2 * m.H, 2 * m.H, 0.f, st));
      HIP_TRY(launch_transpose_f32(w->gru_cand_kernel[l], m.gru_wc[l], in + m.H, m.H, st));
      HIP_TRY(launch_copy_pad_f32(w->gru_cand_bias[l], m.gru_bc[l], m.H, m.H, 0.f, st));
    }
  }
  if (m.low_gru) {   // the GRU stack of model_gru.py:33-45 (HierGRU's low level, or the single-level model_gru)
    for (int l = 0; l < m.nl; l++) {
      const int in = (l == 0) ? E : m.H;
      {
        if (!w->low_gate_kernel[l] || !w->low_gate_bias[l] || !w->low_cand_kernel[l] || !w->low_cand_bias[l])
          return fail(HIERTCN_ERR_INVALID, "HierGRU: low-level GRU layer %d weights missing", l);
        HIP_TRY(launch_transpose_f32(w->low_gate_kernel[l], m.lgru_wg[l], in + m.H, 2 * m.H, st));
        HIP_TRY(launch_copy_pad_f32(w->low_gate_bias[l], m.lgru_bg[l], 2 * m.H, 2 * m.H, 0.f, st));
        HIP_TRY(launch_transpose_f32(w->low_cand_kernel[l], m.lgru_wc[l], in + m.H, m.H, st));
        HIP_TRY(launch_copy_pad_f32(w->low_cand_bias[l], m.lgru_bc[l], m.H, m.H, 0.f, st));
      }
    }
  }
  HIP_TRY(refresh_frag_major(m, st));
  HIP_TRY(refresh_lgru_split(m, st));
  if (h->train.enabled) HIP_TRY(train_fill_masters(m, h->train, w, st));
  if (h->x.enabled && h->m.train_fp32) {
    h->x.filled = h->x.enabled;   // fp32 training mode: the mode's tensors are the masters train_fill_masters has just written
  } else if (h->x.enabled) {   // exact / split numerics modes: fp32 copies in the same internal layouts (reduction index contiguous)
    ExactDev& x = h->x;
    HIP_TRY(hipMemcpyAsync(x.emb, w->emb_kernel, (size_t)m.N * E * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIP_TRY(hipMemsetAsync(x.emb, 0, (size_t)E * sizeof(float), st));   // row 0 = the null id (model.py:59)
    if (x.enabled & X_FP32) HIP_TRY(launch_transpose_f32(w->out_kernel, x.outT, m.Clast, m.N, st));   // [C][N] -> [N][C]
    if (x.enabled & X_SPLIT) HIP_TRY(launch_transpose_split(w->out_kernel, x.outS, m.Clast, m.N, m.NP, x.outS_scale, st));   // [C][N] -> [NP][hi C | lo C]
    for (int i = 0, cin = E; i < m.n_blocks && !m.low_gru; cin = m.C[i], i++) {
      const int cout = m.C[i];
      for (int j = 0; j < m.K; j++)
        HIP_TRY(launch_transpose_f32(w->conv_kernel[i] + (size_t)j * cin * cout, x.convT[i] + (size_t)j * cin * cout, cin, cout, st));
      if (w->res_kernel[i]) HIP_TRY(launch_transpose_f32(w->res_kernel[i], x.resT[i], cin, cout, st));
    }
    if (m.hier) HIP_TRY(launch_transpose_f32(w->proj_kernel, x.projT, E, E, st));   // rows [0, E) of Wp: the item half
    x.filled = x.enabled;
  }
  h->weights_set = true;
  return HIERTCN_OK;
}

int hiertcn_set_precision(hiertcn_handle_t h, int32_t precision) {
  if (!h) return fail(HIERTCN_ERR_INVALID, "null handle");
  if (precision != HIERTCN_PRECISION_STORAGE16 && precision != HIERTCN_PRECISION_FP32 && precision != HIERTCN_PRECISION_SPLIT16)
    return fail(HIERTCN_ERR_INVALID, "precision %d: HIERTCN_PRECISION_STORAGE16 (0), HIERTCN_PRECISION_FP32 (1) or HIERTCN_PRECISION_SPLIT16 (2)", precision);
  const int need = precision == HIERTCN_PRECISION_FP32 ? (X_LOW | X_FP32) : precision == HIERTCN_PRECISION_SPLIT16 ? (X_LOW | X_SPLIT) : 0;
  if (precision == HIERTCN_PRECISION_SPLIT16 && !score_x3_supported(h->m))
    return fail(HIERTCN_ERR_INVALID, "HIERTCN_PRECISION_SPLIT16: the split-operand scoring kernels are built for a last width of 64, 128, 256 or 512 (this model: %d); "
                "use HIERTCN_PRECISION_FP32", h->m.Clast);
  if (need & ~h->x.enabled) {
    if (h->weights_set)
      return fail(HIERTCN_ERR_STATE, "the first hiertcn_set_precision to a mode must precede hiertcn_set_weights (its weight copies are filled there)");
    if (h->train.enabled) return fail(HIERTCN_ERR_STATE, "a numerics mode must be chosen BEFORE hiertcn_train_enable (fp32: the fp32 training mode; split16: evaluation only)");
    ON_DEVICE(h);
    hiertcn_ctx* c = h;
    ModelDev& m = h->m;
    ExactDev& x = h->x;
    const int E = m.E;
#define XALLOC(ptr, count)                                                                                     \
  do {                                                                                                         \
    hipError_t e_ = dev_alloc(c, &(ptr), (size_t)(count));                                                     \
    if (e_ != hipSuccess) return fail(HIERTCN_ERR_HIP, "hipMalloc(%s) failed: %s", #ptr, hipGetErrorString(e_)); \
  } while (0)
    if (!(x.enabled & X_LOW)) {
      XALLOC(x.emb, (size_t)m.N * E);
      for (int i = 0, cin = E; i < m.n_blocks && !m.low_gru; cin = m.C[i], i++) {
        XALLOC(x.convT[i], (size_t)m.K * cin * m.C[i]);
        if (cin != m.C[i]) XALLOC(x.resT[i], (size_t)cin * m.C[i]);
      }
      if (m.hier) XALLOC(x.projT, (size_t)E * E);
    }
    if ((need & X_FP32) && !(x.enabled & X_FP32)) {
      XALLOC(x.outT, (size_t)m.N * m.Clast);
      x.chunk_rows = exact_chunk_rows(m.N);
      XALLOC(x.logits, (size_t)x.chunk_rows * m.N);
    }
    if ((need & X_SPLIT) && !(x.enabled & X_SPLIT)) {
      XALLOC(x.outS, (size_t)m.NP * 2 * m.Clast);
      XALLOC(x.outS_scale, 4);
    }
#undef XALLOC
    x.enabled |= need;
  }
  if (h->weights_set && (need & ~h->x.filled))
    return fail(HIERTCN_ERR_STATE, "the weight copies of this mode were allocated after hiertcn_set_weights: set the weights again");
  h->precision = precision;
  return HIERTCN_OK;
}

int hiertcn_get_tuning(hiertcn_handle_t h, hiertcn_tuning_t* out) {
  if (!h || !out) return fail(HIERTCN_ERR_INVALID, "null argument");
  *out = h->m.tune;
  return HIERTCN_OK;
}

int hiertcn_set_tuning(hiertcn_handle_t h, const hiertcn_tuning_t* in) {
  if (!h || !in) return fail(HIERTCN_ERR_INVALID, "null argument");
  hiertcn_tuning_t t = *in, d;
  tuning_defaults(&d);
  int32_t* sel[] = {&t.gru_scan_persistent, &t.gru_scan_persistent_train, &t.gru_bptt_persistent, &t.gru_bptt_fused, &t.lgru_persistent, &t.lgru_pipe,
                    &t.lgru_tile, &t.lgru_tile_bwd, &t.lgru_wide, &t.lgru_wide_bwd, &t.lgru_fp32_cells};
  for (int32_t* s : sel)
    if (*s < -1 || *s > 1) return fail(HIERTCN_ERR_INVALID, "hiertcn_set_tuning: a selection field is -1 (automatic), 0 (never) or 1 (wherever the shape allows), not %d", *s);
  if (t.tcn_layered < 0 || t.tcn_layered > 1) return fail(HIERTCN_ERR_INVALID, "hiertcn_set_tuning: tcn_layered is 0 or 1");
#define THRESH(f, lo, hi)                                                                                                  \
  do {                                                                                                                     \
    if (t.f == 0) t.f = d.f;                                                                                               \
    if (t.f < (lo) || t.f > (hi)) return fail(HIERTCN_ERR_INVALID, "hiertcn_set_tuning: " #f " = %d outside [%d, %d]", t.f, (int)(lo), (int)(hi)); \
  } while (0)
  THRESH(gru_persist_max_lanes, 1, 96);            // (96: what the persistent launch's LDS image and its residency check are built for)
  THRESH(gemm_split_max_tiles, 1, 1 << 24);
  THRESH(skinny_max_rows, 1, 1 << 24);
  THRESH(lgru_tile_max_sessions, 1, 1 << 30);
  THRESH(lgru_wide_max_launches, 1, 64);
  THRESH(sgrad_dyn_max_blocks, 1, 1 << 20);
  THRESH(sgrad0_min_stages, 1, 1 << 20);
  THRESH(train_fp32_chunk_mib, 1, 1 << 20);
#undef THRESH
  if (t.lgru_defer_max_gib == 0) t.lgru_defer_max_gib = d.lgru_defer_max_gib;
  if (t.lgru_defer_max_gib < -1 || t.lgru_defer_max_gib > 256)
    return fail(HIERTCN_ERR_INVALID, "hiertcn_set_tuning: lgru_defer_max_gib = %d: -1 (never defer), 0 (the default) or 1 .. 256 GiB", t.lgru_defer_max_gib);
  if (t.tcn_gs2_max_sessions == 0) t.tcn_gs2_max_sessions = d.tcn_gs2_max_sessions;
  if (t.tcn_gs2_max_sessions < -1) return fail(HIERTCN_ERR_INVALID, "hiertcn_set_tuning: tcn_gs2_max_sessions = %d: -1 (never), 0 (the default) or a session count", t.tcn_gs2_max_sessions);
  for (int32_t r : t.reserved)
    if (r != 0) return fail(HIERTCN_ERR_INVALID, "hiertcn_set_tuning: reserved fields must be zero");
  h->m.tune = t;
  return HIERTCN_OK;
}

int hiertcn_get_precision(hiertcn_handle_t h, int32_t* precision) {
  if (!h || !precision) return fail(HIERTCN_ERR_INVALID, "null argument");
  *precision = h->precision;
  return HIERTCN_OK;
}

int hiertcn_struct_sizes(size_t out[HIERTCN_N_STRUCTS]) {
  if (!out) return fail(HIERTCN_ERR_INVALID, "null argument");
  out[0] = sizeof(hiertcn_config_t);
  out[1] = sizeof(hiertcn_weights_t);
  out[2] = sizeof(hiertcn_score_out_t);
  out[3] = sizeof(hiertcn_gmm_config_t);
  out[4] = sizeof(hiertcn_gmm_weights_t);
  out[5] = sizeof(hiertcn_tuning_t);
  return HIERTCN_OK;
}

#define CHECK_READY(h)                                                        \
  do {                                                                        \
    if (!(h)) return fail(HIERTCN_ERR_INVALID, "null handle");                \
    if (!(h)->weights_set) return fail(HIERTCN_ERR_STATE, "hiertcn_set_weights has not been called"); \
  } while (0);                                                                \
  ON_DEVICE(h)
// The one-shot score ids (hiertcn_set_score_ids) are consumed by the step they were armed for WHETHER IT SUCCEEDS OR NOT -- a failed state / device
// check included: take them before anything can return.
static const int32_t* take_score_ids(hiertcn_ctx* h) {
  if (!h) return nullptr;
  const int32_t* ids = h->score_ids;
  h->score_ids = nullptr;
  return ids;
}
#define IS_EXACT(h) ((h)->precision != HIERTCN_PRECISION_STORAGE16)   /* fp32 hidden rows, fp32 low level: the FP32 and the SPLIT16 mode */
#define IS_SPLIT(h) ((h)->precision == HIERTCN_PRECISION_SPLIT16)

int hiertcn_session_pool(hiertcn_handle_t h, const int32_t* y_ids, int64_t n_sessions, int32_t L, float* pooled, void* stream) {
  CHECK_READY(h);
  if (!h->m.hier) return fail(HIERTCN_ERR_STATE, "session_pool is a model_hier operation");
  if (!y_ids || !pooled || n_sessions < 0 || L < 1) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (IS_EXACT(h)) HIP_TRY(launch_xsession_pool(h->m, h->x, y_ids, n_sessions, L, pooled, reinterpret_cast<hipStream_t>(stream)));
  else HIP_TRY(launch_session_pool(h->m, y_ids, n_sessions, L, pooled, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

int hiertcn_embed_concat(hiertcn_handle_t h, const int32_t* x_ids, const float* state, int64_t n_sessions, int32_t L, uint16_t* rows,
                         void* stream) {
  CHECK_READY(h);
  if (!h->m.hier) return fail(HIERTCN_ERR_STATE, "embed_concat is a model_hier operation");
  if (IS_EXACT(h)) return fail(HIERTCN_ERR_STATE, "embed_concat writes storage-type rows: not available in the exact fp32 mode");
  if (!x_ids || !state || !rows || n_sessions < 0 || L < 1) return fail(HIERTCN_ERR_INVALID, "bad argument");
  HIP_TRY(launch_embed_concat(h->m, x_ids, state, n_sessions, L, rows, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

size_t hiertcn_gru_workspace_bytes(hiertcn_handle_t h, int32_t B, int32_t S) {
  if (!h || !h->m.hier || B < 0 || S < 0) return 0;
  return gru_workspace_bytes(h->m, B, S);
}

int hiertcn_gru_scan(hiertcn_handle_t h, const float* pooled, const float* mask, const float* state_in, int32_t B, int32_t S,
                     float* sproj, float* state_out, float* state_before, void* workspace, size_t workspace_bytes, void* stream) {
  CHECK_READY(h);
  if (!h->m.hier) return fail(HIERTCN_ERR_STATE, "gru_scan is a model_hier operation");
  if (!pooled || !mask || !state_in || !sproj || !state_out || !workspace || B < 0 || S < 1) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (workspace_bytes < gru_workspace_bytes(h->m, B, S)) return fail(HIERTCN_ERR_WORKSPACE, "gru workspace too small");
  if ((reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return fail(HIERTCN_ERR_INVALID, "workspace must be 16-byte aligned");
  HIP_TRY(launch_gru_scan(h->m, pooled, mask, state_in, B, S, sproj, state_out, state_before, reinterpret_cast<float*>(workspace),
                          reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

size_t hiertcn_tcn_workspace_bytes(hiertcn_handle_t h, int64_t n_sessions, int32_t L) {
  if (!h || n_sessions < 0 || L < 1) return 0;
  if (IS_EXACT(h)) return exact_tcn_workspace_bytes(h->m, n_sessions, L);
  return tcn_workspace_bytes(h->m, n_sessions, L);
}

int hiertcn_tcn_forward(hiertcn_handle_t h, const int32_t* x_ids, const float* sproj, int64_t n_sessions, int32_t L,
                        void* hidden, void* workspace, size_t workspace_bytes, void* stream) {
  CHECK_READY(h);
  if (!x_ids || !hidden || n_sessions < 0) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (L < 1 || L > 4096) return fail(HIERTCN_ERR_INVALID, "L=%d out of range 1..4096", L);
  if (h->m.hier && !sproj) return fail(HIERTCN_ERR_INVALID, "sproj is required for hier models (run hiertcn_gru_scan first)");
  const size_t need = IS_EXACT(h) ? exact_tcn_workspace_bytes(h->m, n_sessions, L) : tcn_workspace_bytes(h->m, n_sessions, L);
  if (need > 0) {
    if (!workspace || workspace_bytes < need) return fail(HIERTCN_ERR_WORKSPACE, "tcn workspace %zu < required %zu bytes", workspace_bytes, need);
    if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail(HIERTCN_ERR_INVALID, "workspace must be 256-byte aligned");
  }
  if (IS_EXACT(h))
    HIP_TRY(launch_exact_tcn(h->m, h->x, x_ids, h->m.hier ? sproj : nullptr, n_sessions, L, reinterpret_cast<float*>(hidden), workspace,
                             reinterpret_cast<hipStream_t>(stream)));
  else
    HIP_TRY(launch_tcn_forward(h->m, x_ids, h->m.hier ? sproj : nullptr, n_sessions, L, reinterpret_cast<uint16_t*>(hidden), workspace,
                               reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

static int score_timed(hiertcn_handle_t h, const void* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                       hipStream_t st, const int32_t* row_map = nullptr, const int32_t* n_rows_dev = nullptr, float* split_scratch = nullptr) {
  size_t slot = 0;
  const bool timing = h->timing;
  if (timing) {
    if (h->ev_used == h->ev_begin.size()) {
      hipEvent_t a, b;
      HIP_TRY(hipEventCreate(&a));
      HIP_TRY(hipEventCreate(&b));
      h->ev_begin.push_back(a);
      h->ev_end.push_back(b);
    }
    slot = h->ev_used++;
    HIP_TRY(hipEventRecord(h->ev_begin[slot], st));
  }
  if (IS_SPLIT(h)) HIP_TRY(launch_score_x3(h->m, h->x, reinterpret_cast<const float*>(hidden), y, R, out, nullptr, st, row_map, n_rows_dev, split_scratch));
  else if (IS_EXACT(h)) HIP_TRY(launch_exact_score(h->m, h->x, reinterpret_cast<const float*>(hidden), y, R, out, st));
  else HIP_TRY(launch_score(h->m, reinterpret_cast<const uint16_t*>(hidden), y, R, out, nullptr, st, row_map, n_rows_dev, split_scratch));
  if (timing) HIP_TRY(hipEventRecord(h->ev_end[slot], st));
  return HIERTCN_OK;
}

static int check_topk(const hiertcn_ctx* h, const hiertcn_score_out_t* o) {
  if (o->topk < 0 || o->topk > HIERTCN_MAX_TOPK) return fail(HIERTCN_ERR_INVALID, "topk %d out of range 0..%d", o->topk, HIERTCN_MAX_TOPK);
  if (o->topk > 0 && (!o->topk_idx || !o->topk_val)) return fail(HIERTCN_ERR_INVALID, "topk > 0 needs topk_idx and topk_val");
  if (o->topk > h->m.N) return fail(HIERTCN_ERR_INVALID, "topk %d exceeds item_num %d", o->topk, h->m.N);
  if (o->topk > 0 && (!o->rank || !o->ce)) return fail(HIERTCN_ERR_INVALID, "the top-k kernel also writes rank and ce: both are required");
  return HIERTCN_OK;
}

int hiertcn_score_fused(hiertcn_handle_t h, const void* hidden, const int32_t* y_ids, int64_t R, const hiertcn_score_out_t* out,
                        void* stream) {
  CHECK_READY(h);
  if (!hidden || !y_ids || !out || R < 0) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (int rc = check_topk(h, out)) return rc;
  return score_timed(h, hidden, y_ids, R, out, reinterpret_cast<hipStream_t>(stream));
}

int hiertcn_score_logits(hiertcn_handle_t h, const void* hidden, int64_t R, float* logits, void* stream) {
  CHECK_READY(h);
  if (!hidden || !logits || R < 0) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (IS_SPLIT(h)) HIP_TRY(launch_score_x3(h->m, h->x, reinterpret_cast<const float*>(hidden), nullptr, R, nullptr, logits, reinterpret_cast<hipStream_t>(stream), nullptr, nullptr, nullptr));
  else if (IS_EXACT(h)) HIP_TRY(launch_exact_logits(h->m, h->x, reinterpret_cast<const float*>(hidden), R, logits, reinterpret_cast<hipStream_t>(stream)));
  else HIP_TRY(launch_score(h->m, reinterpret_cast<const uint16_t*>(hidden), nullptr, R, nullptr, logits, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

int hiertcn_metrics(hiertcn_handle_t h, const int32_t* rank, const float* ce, const int32_t* y_ids, int32_t B, int32_t T, double* out8,
                    void* stream) {
  if (!h) return fail(HIERTCN_ERR_INVALID, "null handle");
  if (!rank || !ce || !y_ids || !out8 || B < 1 || T < 1) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if ((B + 3) / 4 > HIERTCN_METRIC_BLOCKS_MAX) return fail(HIERTCN_ERR_INVALID, "too many lanes in one call");
  ON_DEVICE(h);
  HIP_TRY(launch_metrics_scratch(h->m, rank, ce, y_ids, B, T, out8, h->metric_scratch, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

// workspace layout of hiertcn_forward_eval (all sub-buffers 256-byte aligned)
struct EvalWs {
  size_t pooled, sproj, hidden, rank, ce, tgt, gru, tcn, map, nvalid, cscratch, split, total;
};
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static EvalWs eval_ws(const ModelDev& m, int64_t B, int64_t S, int64_t L, bool exact = false, bool split = false) {
  EvalWs w;
  size_t off = 0;
  w.pooled = off; off += al256((size_t)B * S * m.E * sizeof(float));
  w.sproj = off;  off += al256((size_t)B * S * m.E * sizeof(float));
  w.hidden = off; off += al256((size_t)B * S * L * m.Clast * (exact ? sizeof(float) : sizeof(uint16_t)));
  w.rank = off;   off += al256((size_t)B * S * L * sizeof(int32_t));
  w.ce = off;     off += al256((size_t)B * S * L * sizeof(float));
  w.gru = off;    off += al256(m.hier ? gru_workspace_bytes(m, (int)B, (int)S) : 0);
  w.tcn = off;    off += al256(exact ? exact_tcn_workspace_bytes(m, B * S, (int)L) : tcn_workspace_bytes(m, B * S, (int)L));
  w.map = off;    off += al256((size_t)B * S * L * sizeof(int32_t));           // ragged batches: rows that carry a target
  w.nvalid = off; off += al256(sizeof(int32_t));
  w.cscratch = off; off += al256(((size_t)B * S * L / 1024 + 2) * sizeof(int32_t));
  w.tgt = off;    off += al256((size_t)B * S * L * sizeof(float));
  w.split = off;  off += al256((size_t)score_split_scratch_parts(B * S * L, m.NP, split ? score_x3_rows_per_wg(m) : 512) * 2 * B * S * L * sizeof(float));   // small batches only (parts > 1)
  w.total = off;
  return w;
}

size_t hiertcn_workspace_bytes(hiertcn_handle_t h, int32_t B, int32_t S, int32_t L) {
  if (!h || B < 0 || S < 0 || L < 0) return 0;
  return eval_ws(h->m, B, S, L, IS_EXACT(h), IS_SPLIT(h)).total;
}

int hiertcn_forward_eval(hiertcn_handle_t h, const int32_t* x_ids, const int32_t* y_ids, const float* mask, const float* state_in,
                         int32_t B, int32_t S, int32_t L, void* workspace, size_t workspace_bytes, float* state_out,
                         double* metrics8, const hiertcn_score_out_t* score_out, void* stream) {
  const int32_t* score_ids = take_score_ids(h);   // one-shot (hiertcn_set_score_ids): consumed by THIS call whether it succeeds or not
  CHECK_READY(h);
  const bool hier = h->m.hier != 0;
  if (!x_ids || !y_ids || !workspace || B < 1 || S < 1) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (hier && (!mask || !state_in || !state_out)) return fail(HIERTCN_ERR_INVALID, "mask / state_in / state_out are required for hier models");
  if (!hier && S != 1) return fail(HIERTCN_ERR_INVALID, "single-level model_tcn: one sequence per lane (S must be 1, L = T)");
  if (L < 1 || L > 4096) return fail(HIERTCN_ERR_INVALID, "L=%d out of range 1..4096", L);
  const bool exact = IS_EXACT(h), split16 = IS_SPLIT(h);
  const EvalWs w = eval_ws(h->m, B, S, L, exact, split16);
  if (workspace_bytes < w.total) return fail(HIERTCN_ERR_WORKSPACE, "workspace %zu < required %zu bytes", workspace_bytes, w.total);
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail(HIERTCN_ERR_INVALID, "workspace must be 256-byte aligned");
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
  float* pooled = reinterpret_cast<float*>(ws + w.pooled);
  float* sproj = reinterpret_cast<float*>(ws + w.sproj);
  void* hidden = ws + w.hidden;
  hiertcn_score_out_t so;
  memset(&so, 0, sizeof(so));
  if (score_out) so = *score_out;
  if (!so.rank) so.rank = reinterpret_cast<int32_t*>(ws + w.rank);
  if (!so.ce) so.ce = reinterpret_cast<float*>(ws + w.ce);
  if (int rc = check_topk(h, &so)) return rc;
  const int64_t n_sessions = (int64_t)B * S;
  const int64_t R = n_sessions * L;
  // warm-start evaluation (model.py:102-103, hiertcn_set_score_ids): targets / masks / loss / metrics from the one-shot score ids, the session mean-pool from y_ids
  const int32_t* y_pool = y_ids;
  if (score_ids) y_ids = score_ids;
  // scan first (it never reads the low level's output: model_hier.py:83-93), then every
  // (user, session) low-level problem is independent.
  // Padded steps (y == 0) have rank 0 / loss 0 by definition (model.py:105): the outputs are zeroed, the rows that do carry a target listed, and
  // only those scored (a ragged XING batch is mostly padding: slot width = longest session of the slot).
  int32_t* row_map = reinterpret_cast<int32_t*>(ws + w.map);
  int32_t* n_valid = reinterpret_cast<int32_t*>(ws + w.nvalid);
  float* split = nullptr;
  if (!(exact && !split16) && so.topk == 0 && score_split_scratch_parts(R, h->m.NP, split16 ? score_x3_rows_per_wg(h->m) : 512) > 1) {
    split = reinterpret_cast<float*>(ws + w.split);
    if (!so.target) so.target = reinterpret_cast<float*>(ws + w.tgt);   // the merge needs the target logit
  }
  bool compacted = false;
  if (hier) {
    if (exact) HIP_TRY(launch_xsession_pool(h->m, h->x, y_pool, n_sessions, L, pooled, st));
    else if (session_pool_compact_applies(R) && y_pool == y_ids) {   // small batches: the row list rides on the pooling launch (it depends on the ids alone)
      HIP_TRY(launch_session_pool_compact(h->m, y_ids, n_sessions, L, pooled, row_map, n_valid, so.rank, so.ce, so.target, st));
      compacted = true;
    } else HIP_TRY(launch_session_pool(h->m, y_pool, n_sessions, L, pooled, st));
    HIP_TRY(launch_gru_scan(h->m, pooled, mask, state_in, B, S, sproj, state_out, nullptr, reinterpret_cast<float*>(ws + w.gru), st));
  }
  if (exact && !split16) {   // fp32 end to end: every row is scored on materialised logits; rows without a target come out as 0 / 0 / 0..k-1 there
    HIP_TRY(launch_exact_tcn(h->m, h->x, x_ids, hier ? sproj : nullptr, n_sessions, L, reinterpret_cast<float*>(hidden), ws + w.tcn, st));
    int rc = score_timed(h, hidden, y_ids, R, &so, st);
    if (rc != HIERTCN_OK) return rc;
    if (metrics8) {
      if ((B + 3) / 4 > HIERTCN_METRIC_BLOCKS_MAX) return fail(HIERTCN_ERR_INVALID, "too many lanes in one call");
      HIP_TRY(launch_metrics_scratch(h->m, so.rank, so.ce, y_ids, B, S * L, metrics8, h->metric_scratch, st));
    }
    return HIERTCN_OK;
  }
  if (split16)   // the fp32 low level (exact.hip), then the fused scoring kernel on split operands (score_x3.hip) exactly as the 16-bit path below
    HIP_TRY(launch_exact_tcn(h->m, h->x, x_ids, hier ? sproj : nullptr, n_sessions, L, reinterpret_cast<float*>(hidden), ws + w.tcn, st));
  else
    HIP_TRY(launch_tcn_forward(h->m, x_ids, hier ? sproj : nullptr, n_sessions, L, reinterpret_cast<uint16_t*>(hidden), ws + w.tcn, st));
  // (the same pass zeroes rank / ce / target of every row: one launch instead of the list's three plus three memsets at XING batch sizes)
  if (!compacted) HIP_TRY(launch_compact_rows(y_ids, R, row_map, n_valid, reinterpret_cast<int32_t*>(ws + w.cscratch), st, so.rank, so.ce, so.target));
  // a masked row is an all-zero logit row (model.py:105): its ranked list is 0..k-1 with values 0 (loss.py:120 on ties)
  if (so.topk > 0) HIP_TRY(launch_topk_fill(so.topk_idx, so.topk_val, R, so.topk, st));
  int rc = score_timed(h, hidden, y_ids, R, &so, st, row_map, n_valid, split);
  if (rc != HIERTCN_OK) return rc;
  if (metrics8) {
    if ((B + 3) / 4 > HIERTCN_METRIC_BLOCKS_MAX) return fail(HIERTCN_ERR_INVALID, "too many lanes in one call");
    HIP_TRY(launch_metrics_scratch(h->m, so.rank, so.ce, y_ids, B, S * L, metrics8, h->metric_scratch, st));
  }
  return HIERTCN_OK;
}

// ---- training ----------------------------------------------------------------------
int hiertcn_train_enable(hiertcn_handle_t h) {
  if (!h) return fail(HIERTCN_ERR_INVALID, "null handle");
  if (h->train.enabled) return HIERTCN_OK;
  if (h->weights_set) return fail(HIERTCN_ERR_STATE, "hiertcn_train_enable must precede hiertcn_set_weights (the fp32 masters are filled there)");
  // fp32 TRAINING mode (ABI 10): hiertcn_set_precision(h, HIERTCN_PRECISION_FP32) before this call -- forward on the fp32 masters, materialised
  // logits, every product of the backward an exact-fp32 GEMM (train.hip); model_hier with the TemporalConvNet low level, the reference's own
  // batch sizes.  The split16 mode stays evaluation only.
  const bool fp32 = h->precision == HIERTCN_PRECISION_FP32 && (h->x.enabled & X_FP32);
  if (h->x.enabled && !fp32)
    return fail(HIERTCN_ERR_STATE, "this handle is in the split16 evaluation mode (hiertcn_set_precision): training runs on 16-bit storage handles, or in the fp32 mode");
  ON_DEVICE(h);
  HIP_TRY(train_enable(h->m, h->train));
  if (fp32) {
    // the fp32 masters have exact.hip's layouts: the mode's weight copies ARE the masters from now on (evaluation between steps reads what Adam wrote)
    h->m.train_fp32 = 1;
    ExactDev& x = h->x;
    float* M = h->train.master();
    x.emb = M + h->train.emb.off;
    x.projT = h->m.hier ? M + h->train.projT.off : nullptr;
    for (int i = 0; i < h->m.n_blocks && !h->m.low_gru; i++) {
      x.convT[i] = M + h->train.convT[i].off;
      x.resT[i] = h->train.resT[i].n ? M + h->train.resT[i].off : nullptr;
    }
    x.outT = M + h->train.outT.off;
  }
  return HIERTCN_OK;
}

size_t hiertcn_train_workspace_bytes(hiertcn_handle_t h, int32_t B, int32_t S, int32_t L) {
  if (!h || B < 1 || S < 1 || L < 1) return 0;
  return train_workspace_bytes(h->m, B, S, L);
}

int hiertcn_train_forward_backward(hiertcn_handle_t h, const int32_t* x_ids, const int32_t* y_ids, const float* mask, const float* state_in,
                                   int32_t B, int32_t S, int32_t L, void* workspace, size_t workspace_bytes, float* state_out,
                                   double* metrics8, void* stream) {
  const int32_t* ys = take_score_ids(h);   // one-shot (hiertcn_set_score_ids): consumed by THIS call whether it succeeds or not
  CHECK_READY(h);
  if (!h->train.enabled) return fail(HIERTCN_ERR_STATE, "training is not enabled on this handle");
  if (!x_ids || !y_ids || !workspace || B < 1 || S < 1 || L < 1) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (h->m.hier && (!mask || !state_in || !state_out)) return fail(HIERTCN_ERR_INVALID, "mask / state_in / state_out are required for hier models");
  if (workspace_bytes < train_workspace_bytes(h->m, B, S, L)) return fail(HIERTCN_ERR_WORKSPACE, "train workspace too small");
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail(HIERTCN_ERR_INVALID, "workspace must be 256-byte aligned");
  if ((B + 3) / 4 > HIERTCN_METRIC_BLOCKS_MAX) return fail(HIERTCN_ERR_INVALID, "too many lanes in one call");
  if (train_forward_backward(h->m, h->train, x_ids, y_ids, mask, state_in, B, S, L, workspace, state_out, metrics8, h->metric_scratch,
                             reinterpret_cast<hipStream_t>(stream), TRAIN_BEGIN | TRAIN_FINISH, nullptr, ys) != 0)
    return fail(HIERTCN_ERR_HIP, "%s", train_last_error());
  return HIERTCN_OK;
}

int hiertcn_train_step_begin(hiertcn_handle_t h, const int32_t* x_ids, const int32_t* y_ids, const float* mask, const float* state_in,
                             int32_t B, int32_t S, int32_t L, void* workspace, size_t workspace_bytes, float* state_out, double* metrics8,
                             const float* users_global, void* stream) {
  // the ids armed for this step stay on the handle for the matching hiertcn_train_step_finish -- but only when this call succeeds: a begin that
  // fails (a state / argument / workspace check or a launch error) drops them, so that they cannot reach an unrelated later step
  struct DropOnFailure {
    hiertcn_ctx* h; const int32_t* ids; bool keep = false;
    ~DropOnFailure() { if (h && keep) h->score_ids = ids; }
  } armed{h, take_score_ids(h)};
  CHECK_READY(h);
  if (!h->train.enabled) return fail(HIERTCN_ERR_STATE, "training is not enabled on this handle");
  if (!x_ids || !y_ids || !workspace || B < 1 || S < 1 || L < 1) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (h->m.hier && (!mask || !state_in || !state_out)) return fail(HIERTCN_ERR_INVALID, "mask / state_in / state_out are required for hier models");
  if (workspace_bytes < train_workspace_bytes(h->m, B, S, L)) return fail(HIERTCN_ERR_WORKSPACE, "train workspace too small");
  if ((reinterpret_cast<uintptr_t>(workspace) & 255) != 0) return fail(HIERTCN_ERR_INVALID, "workspace must be 256-byte aligned");
  if ((B + 3) / 4 > HIERTCN_METRIC_BLOCKS_MAX) return fail(HIERTCN_ERR_INVALID, "too many lanes in one call");
  if (train_forward_backward(h->m, h->train, x_ids, y_ids, mask, state_in, B, S, L, workspace, state_out, metrics8, h->metric_scratch,
                             reinterpret_cast<hipStream_t>(stream), TRAIN_BEGIN, users_global, armed.ids) != 0)
    return fail(HIERTCN_ERR_HIP, "%s", train_last_error());
  armed.keep = true;   // (kept for the matching finish)
  return HIERTCN_OK;
}

int hiertcn_train_step_finish(hiertcn_handle_t h, const int32_t* x_ids, const int32_t* y_ids, const float* mask, int32_t B, int32_t S,
                              int32_t L, void* workspace, size_t workspace_bytes, void* stream) {
  const int32_t* ys = take_score_ids(h);   // (set before the matching begin; consumed here whether the call succeeds or not)
  CHECK_READY(h);
  if (!h->train.enabled) return fail(HIERTCN_ERR_STATE, "training is not enabled on this handle");
  if (!x_ids || !y_ids || !workspace || B < 1 || S < 1 || L < 1) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (h->m.hier && !mask) return fail(HIERTCN_ERR_INVALID, "mask is required for hier models");
  if (workspace_bytes < train_workspace_bytes(h->m, B, S, L)) return fail(HIERTCN_ERR_WORKSPACE, "train workspace too small");
  if (train_forward_backward(h->m, h->train, x_ids, y_ids, mask, nullptr, B, S, L, workspace, nullptr, nullptr, h->metric_scratch,
                             reinterpret_cast<hipStream_t>(stream), TRAIN_FINISH, nullptr, ys) != 0)
    return fail(HIERTCN_ERR_HIP, "%s", train_last_error());
  return HIERTCN_OK;
}

int hiertcn_set_score_ids(hiertcn_handle_t h, const int32_t* y_score_ids) {
  if (!h) return fail(HIERTCN_ERR_INVALID, "null handle");
  h->score_ids = y_score_ids;
  return HIERTCN_OK;
}

int hiertcn_count_users(hiertcn_handle_t h, const int32_t* y_ids, int32_t B, int32_t T, float* users, void* stream) {
  if (!h || !y_ids || !users || B < 1 || T < 1) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (B > 4 * HIERTCN_METRIC_BLOCKS_MAX) return fail(HIERTCN_ERR_INVALID, "too many lanes in one call");
  ON_DEVICE(h);
  // per-lane counts land in the metric scratch (8 * HIERTCN_METRIC_BLOCKS_MAX doubles >= B floats)
  HIP_TRY(train_count_users(y_ids, B, T, reinterpret_cast<float*>(h->metric_scratch), users, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

int hiertcn_train_segment(hiertcn_handle_t h, int32_t which, size_t* offset, size_t* count) {
  if (!h || !h->train.enabled || !offset || !count) return fail(HIERTCN_ERR_STATE, "training is not enabled on this handle / bad argument");
  const TrainCtx& t = h->train;
  const size_t small0 = t.out_bias.off + ((t.out_bias.n + 63) & ~(size_t)63);
  switch (which) {
    case 0: *offset = t.emb.off; *count = t.emb.n; break;                        // item table [N][E]: row-sparse gradient
    case 1: *offset = t.outT.off; *count = small0 - t.outT.off; break;           // scoring kernel [NP][C] + bias [NP]: dense gradient
    case 2: *offset = small0; *count = t.total - small0; break;                  // everything else
    default: return fail(HIERTCN_ERR_INVALID, "segment %d: 0 item table, 1 scoring layer, 2 the rest", which);
  }
  return HIERTCN_OK;
}

size_t hiertcn_grad_rows_workspace_bytes(hiertcn_handle_t h) { return h ? train_rows_workspace_bytes(h->m) : 0; }

int hiertcn_grad_rows_list(hiertcn_handle_t h, const int32_t* x_ids, const int32_t* y_ids, int64_t n_ids, int32_t* ids_out, int64_t cap,
                           int32_t* count_out, void* workspace, size_t workspace_bytes, void* stream) {
  if (!h || !h->train.enabled || !ids_out || !count_out || !workspace || n_ids < 0) return fail(HIERTCN_ERR_INVALID, "bad argument");
  if (workspace_bytes < train_rows_workspace_bytes(h->m)) return fail(HIERTCN_ERR_WORKSPACE, "rows workspace too small");
  const int64_t bound = ((x_ids ? n_ids : 0) + (y_ids ? n_ids : 0)) < (int64_t)h->m.N - 1 ? (x_ids ? n_ids : 0) + (y_ids ? n_ids : 0) : (int64_t)h->m.N - 1;
  if (cap < bound) return fail(HIERTCN_ERR_INVALID, "ids_out holds %lld entries, the batch can touch %lld rows", (long long)cap, (long long)bound);
  ON_DEVICE(h);
  HIP_TRY(train_rows_list(h->m, x_ids, y_ids, n_ids, ids_out, count_out, workspace, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

int hiertcn_grad_rows_op(hiertcn_handle_t h, int32_t op, const int32_t* ids, const int32_t* count, int64_t cap, float* rows, void* stream) {
  if (!h || !h->train.enabled || !ids || !count || cap < 0 || op < 0 || op > 2 || (op != 1 && !rows)) return fail(HIERTCN_ERR_INVALID, "bad argument");
  ON_DEVICE(h);
  HIP_TRY(train_rows_op(h->m, h->train, op, ids, count, cap, rows, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

int hiertcn_grad_arena(hiertcn_handle_t h, float** grad, size_t* count) {
  if (!h || !h->train.enabled || !grad || !count) return fail(HIERTCN_ERR_STATE, "training is not enabled on this handle");
  *grad = h->train.grad();
  *count = h->train.total;
  return HIERTCN_OK;
}

int hiertcn_train_arena(hiertcn_handle_t h, int32_t which, float** ptr, size_t* count) {
  if (!h || !h->train.enabled || !ptr || !count || which < 0 || which > 3) return fail(HIERTCN_ERR_STATE, "training is not enabled on this handle / bad argument");
  *ptr = h->train.arena + (size_t)which * h->train.total;
  *count = h->train.total;
  return HIERTCN_OK;
}

int hiertcn_adam_time(hiertcn_handle_t h, int64_t* get, const int64_t* set) {
  if (!h || !h->train.enabled) return fail(HIERTCN_ERR_STATE, "training is not enabled on this handle");
  if (set) h->train.step = *set;
  if (get) *get = h->train.step;
  return HIERTCN_OK;
}

int hiertcn_adam_step(hiertcn_handle_t h, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream) {
  CHECK_READY(h);
  if (!h->train.enabled) return fail(HIERTCN_ERR_STATE, "training is not enabled on this handle");
  HIP_TRY(train_adam(h->m, h->train, lr, beta1, beta2, eps, grad_scale, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

int hiertcn_export_weights(hiertcn_handle_t h, const hiertcn_weights_t* dst, int32_t which, void* stream) {
  CHECK_READY(h);
  if (!h->train.enabled) return fail(HIERTCN_ERR_STATE, "training is not enabled on this handle");
  if (!dst || which < 0 || which > 1) return fail(HIERTCN_ERR_INVALID, "bad argument");
  HIP_TRY(train_export(h->m, h->train, dst, which, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

int hiertcn_timing_enable(hiertcn_handle_t h, int32_t on) {
  if (!h) return fail(HIERTCN_ERR_INVALID, "null handle");
  h->timing = on != 0;
  return HIERTCN_OK;
}

int hiertcn_timing_read(hiertcn_handle_t h, double* score_ms_avg, int64_t* launches, int32_t reset) {
  if (!h) return fail(HIERTCN_ERR_INVALID, "null handle");
  ON_DEVICE(h);
  double total = 0.0;
  for (size_t i = 0; i < h->ev_used; i++) {
    HIP_TRY(hipEventSynchronize(h->ev_end[i]));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev_begin[i], h->ev_end[i]));
    total += ms;
  }
  if (score_ms_avg) *score_ms_avg = h->ev_used ? total / (double)h->ev_used : 0.0;
  if (launches) *launches = (int64_t)h->ev_used;
  if (reset) h->ev_used = 0;
  return HIERTCN_OK;
}

int hiertcn_persistent_health(hiertcn_handle_t h, int32_t* timed_out) {
  if (!h || !timed_out) return fail(HIERTCN_ERR_INVALID, "null argument");
  ON_DEVICE(h);
  *timed_out = 0;
  if (!h->m.gru_bar) return HIERTCN_OK;
  unsigned long long w = 0;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(&w, h->m.gru_bar + 8 * 16, sizeof(w), hipMemcpyDeviceToHost));   // the word behind the 8 group counters (gru_persist.hip)
  *timed_out = w ? 1 : 0;
  if (w) {
    // a timed-out launch leaves the group counters at a value that is no multiple of its workgroup count, and every later barrier target
    // is derived from them: the handle would mis-synchronise for good.  The device is idle here (synchronised above), so start over --
    // counters and error word back to zero; the caller has been told that the launches since the last check produced wrong states.
    HIP_TRY(hipMemset(h->m.gru_bar, 0, 512 * sizeof(unsigned long long)));
  }
  return HIERTCN_OK;
}

int hiertcn_persistent_health_async(hiertcn_handle_t h, uint64_t* host_word, void* stream) {
  if (!h || !host_word) return fail(HIERTCN_ERR_INVALID, "null argument");
  if (!h->m.gru_bar) return HIERTCN_OK;   // (no persistent launches on this handle: the word stays as the caller left it)
  ON_DEVICE(h);
  HIP_TRY(hipMemcpyAsync(host_word, h->m.gru_bar + 8 * 16, sizeof(uint64_t), hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(stream)));
  return HIERTCN_OK;
}

}  // extern "C"
