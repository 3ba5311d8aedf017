// Shared device/host helpers for the gfx950 HierTCN kernels.  CDNA4 only: wave64, MFMA
// 32x32x16 bf16, 160 KiB LDS.  No portability layer on purpose.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/hiertcn.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define LOG2E_F 1.4426950408889634f
#define LN2_F 0.6931471805599453f

// Storage type of the 16-bit tensors (item tables, conv kernels, activations).  The default build stores bfloat16;
// building with -DHIERTCN_F16 (libhiertcn_gfx950_f16.so, BASELINE.json configs[3]: "fp16 MFMA") stores IEEE half and
// switches the MFMA to its f16 form -- same rate, 3 more mantissa bits, 5 exponent bits.  The helper names keep "bf".
#ifdef HIERTCN_F16
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint16_t f2bf(float f) {
  _Float16 b = (_Float16)f;  // v_cvt_f16_f32: round-to-nearest-even
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t u) { return (float)__builtin_bit_cast(_Float16, u); }
#else
__device__ __forceinline__ uint16_t f2bf(float f) {
  __bf16 b = (__bf16)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even
  return __builtin_bit_cast(uint16_t, b);
}
__device__ __forceinline__ float bf2f(uint16_t u) { return __builtin_bit_cast(float, (uint32_t)u << 16); }
#endif
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return (uint32_t)f2bf(lo) | ((uint32_t)f2bf(hi) << 16);
}

// MFMA 32x32x16 bf16 (cdna_hip_programming.md 3): lane l (r=l&31, h=l>>5) holds
//   A[row r][k=8h+j], B[k=8h+j][col r], j=0..7;  D[row=(reg&3)+8*(reg>>2)+4h][col=r].
__device__ __forceinline__ f32x16 mfma32(uint4 a, uint4 b, f32x16 c) {
#ifdef HIERTCN_F16
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
#endif
}
// accumulator register -> row inside the 32x32 tile (for lane half h)
__device__ __forceinline__ int acc_row(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// Internal (converted) weights + shape, shared by all kernels.
struct ModelDev {
  int N, NP;       // items, items padded to a multiple of 64
  int E;           // embedding width == projection width C0
  int H, nl, HT;   // GRU hidden, layers, H*nl
  int K;           // conv taps
  int n_blocks;
  int C[HIERTCN_MAX_BLOCKS];
  int Clast;
  int hier;
  int low_gru;     // 1: HierGRU -- the low level is model_gru.py's GRU stack instead of the TemporalConvNet (hier only)
  int fused_tcn;   // 1: every width == E in {64,128,256,512} and no down-sample residual -> tcn.hip's fused kernel applies (L <= 32)
  // bf16 tensors (uint16 bit patterns)
  uint16_t* emb;        // [N][E], row 0 zeroed
  uint16_t* projT;      // [E(out)][E(in)]   = transpose(Wp[:E])          (hier only)
  uint16_t* convT[HIERTCN_MAX_BLOCKS];  // [K][Cout][Cin]
  uint16_t* resT[HIERTCN_MAX_BLOCKS];   // [Cout][Cin] or null
  uint16_t* outT;       // [NP][Clast], rows >= N zero
  // fragment-major copies for the LDS-tiled layer kernel (tcn_layers.hip): [Cout/32][taps][Cin/16][64 lanes][8]
  uint16_t* projF;
  uint16_t* convF[HIERTCN_MAX_BLOCKS];
  uint16_t* resF[HIERTCN_MAX_BLOCKS];
  // fp32 tensors
  float* emb_bias;      // [E]
  float* projS;         // [E][HT]  = transpose(Wp[E:]), fp32                  (hier only)
  float* conv_bias[HIERTCN_MAX_BLOCKS];
  float* res_bias[HIERTCN_MAX_BLOCKS];
  float* out_bias;      // [NP], entries >= N are -inf
  float* gru_wg[HIERTCN_MAX_LAYERS];  // [2H][in+H]  (transposed: output-major)
  float* gru_bg[HIERTCN_MAX_LAYERS];
  float* gru_wc[HIERTCN_MAX_LAYERS];  // [H][in+H]
  float* gru_bc[HIERTCN_MAX_LAYERS];
  unsigned long long* gru_bar;        // group-barrier counters + time-out word of the persistent scan (gru_persist.hip), one set per handle, only ever grow
  int n_cu;                           // compute units of the handle's device
  float* lgru_wg[HIERTCN_MAX_LAYERS]; // low-level GRU stack of HierGRU (same layouts), low_gru only
  float* lgru_bg[HIERTCN_MAX_LAYERS];
  float* lgru_wc[HIERTCN_MAX_LAYERS];
  float* lgru_bc[HIERTCN_MAX_LAYERS];
  int lgru_split;                         // 1: the hi / lo 16-bit splits of the low-level GRU kernels below are allocated (gru_x3.hip)
  uint16_t* lgru_wg_hi[HIERTCN_MAX_LAYERS]; uint16_t* lgru_wg_lo[HIERTCN_MAX_LAYERS];   // [2H][in+H]
  uint16_t* lgru_wc_hi[HIERTCN_MAX_LAYERS]; uint16_t* lgru_wc_lo[HIERTCN_MAX_LAYERS];   // [H][in+H]
  // fragment-major copies of the four (lgru_tile.hip): [N/32][(in+H)/16][64 lanes][8]
  uint16_t* lgru_wgF_hi[HIERTCN_MAX_LAYERS]; uint16_t* lgru_wgF_lo[HIERTCN_MAX_LAYERS];
  uint16_t* lgru_wcF_hi[HIERTCN_MAX_LAYERS]; uint16_t* lgru_wcF_lo[HIERTCN_MAX_LAYERS];
  // ... and of their transposes (the backward pass of lgru_tile.hip): [(in+H)/32][N/16][64 lanes][8]
  uint16_t* lgru_wgTF_hi[HIERTCN_MAX_LAYERS]; uint16_t* lgru_wgTF_lo[HIERTCN_MAX_LAYERS];
  uint16_t* lgru_wcTF_hi[HIERTCN_MAX_LAYERS]; uint16_t* lgru_wcTF_lo[HIERTCN_MAX_LAYERS];
  int* walk_pos;        // [8][32] ints, zeroed: per-XCD position words of the scoring kernel's item walk (null: every workgroup starts at stage 0)
  int train_fp32;       // 1: the handle trains in the fp32 mode (hiertcn_set_precision(FP32) + hiertcn_train_enable): fp32 forward on the masters, materialised logits
  int* walk_pace;       // [8][XW_PACE_INTS] ints: per-XCD ticket + position slots of the workgroups that pace their item walk on each other (score_x3w.hip); null: no pacing
  hiertcn_tuning_t tune;   // launch selection + thresholds (include/hiertcn.h: hiertcn_set_tuning); the library reads no environment variable
};
#define XW_PACE_SLOTS 64                    // position slots per XCD class (32 workgroups are resident per XCD at one per CU)
#define XW_PACE_INTS (XW_PACE_SLOTS + 32)   // + a line of its own for the ticket counter
// the tuning of the handle whose API call this thread is in: for the few launch helpers that take no ModelDev (launch_gemm in pool_gru.hip,
// gemm32_skinny_max_rows).  Every entry point that can reach them opens a TuneScope -- hiertcn_* through ON_DEVICE with the handle's tuning,
// hiertcn_gmm_* with nullptr (= the defaults) -- which restores the previous pointer when the call returns: nothing points into a handle
// between calls, so a destroyed handle cannot be read and one handle's thresholds never steer another's launches.
const hiertcn_tuning_t& tune_now();
struct TuneScope {
  const hiertcn_tuning_t* prev;
  explicit TuneScope(const hiertcn_tuning_t* t);
  ~TuneScope();
  TuneScope(const TuneScope&) = delete;
  TuneScope& operator=(const TuneScope&) = delete;
};
void tuning_defaults(hiertcn_tuning_t* t);
// A/B alternatives of superseded kernels exist in the ablation build only (hiertcn_amd/build.py, HIERTCN_ABLATION=1): there, and only there, the
// old environment switches are looked up
#if HIERTCN_ABLATION
#define ABL_ENV(name) (getenv(name) != nullptr)
#else
#define ABL_ENV(name) false
#endif

// launchers implemented in the kernel files (all asynchronous on `st`)
hipError_t launch_rows_to_bf16(const float* src, uint16_t* dst, int64_t rows, int cols, int zero_row0, hipStream_t st);
hipError_t launch_transpose_to_bf16(const float* src, uint16_t* dst, int64_t rows, int64_t cols, int64_t dst_rows_padded, hipStream_t st, int batch = 1);
hipError_t launch_transpose_f32(const float* src, float* dst, int64_t rows, int64_t cols, hipStream_t st);
hipError_t launch_copy_pad_f32(const float* src, float* dst, int64_t n, int64_t npad, float padval, hipStream_t st);
hipError_t launch_session_pool(const ModelDev& m, const int32_t* y, int64_t n_sessions, int L, float* pooled, hipStream_t st);
bool session_pool_compact_applies(int64_t R);
hipError_t launch_session_pool_compact(const ModelDev& m, const int32_t* y, int64_t n_sessions, int L, float* pooled, int32_t* map, int32_t* n_valid,
                                       int32_t* z0, float* z1, float* z2, hipStream_t st);
size_t gru_workspace_bytes(const ModelDev& m, int B, int S);
int gru_persist_groups(const ModelDev& m, int B, int S, bool saving);
// the scan's backward (train.hip) as one persistent launch: da_c / [da_r | da_z] of every (slot, layer) into dac_all / dag_all, d pooled
bool gru_bptt_persist_applies(const ModelDev& m, int B, int S);
hipError_t launch_gru_bptt_persist(const ModelDev& m, const float* save, const float* hist, const float* mask, const float* dsb, int B, int S, float* dac_all,
                                   float* dag_all, float* dpooled, hipStream_t st);
hipError_t launch_gru_scan_persist(const ModelDev& m, const float* pooled, const float* mask, const float* state_in, int B, int S,
                                   float* state_out, float* ws, int G, hipStream_t st, float* save = nullptr);
size_t gru_save_floats(const ModelDev& m, int B, int S);
hipError_t launch_gru_scan(const ModelDev& m, const float* pooled, const float* mask, const float* state_in, int B, int S,
                           float* sproj, float* state_out, float* state_before, float* ws, hipStream_t st, float* save = nullptr);
hipError_t launch_topk_fill(int32_t* idx, float* val, int64_t R, int k, hipStream_t st);
// score_grad.hip: item-stationary d-logits kernel of the training backward (row chunks)
bool dlogits_supported(const ModelDev& m);
int64_t dlogits_ld(const ModelDev& m);
hipError_t launch_dlogits_rowbias(const float* wc, const float* lc, int64_t Rv, int64_t Rpad, float* rb, hipStream_t st);
hipError_t launch_dlogits_rows(const ModelDev& m, const uint16_t* hc, const float* rb, const int32_t* yc, const float* wc, const float* lc,
                               const float* tgt, const int32_t* map, int64_t r0, int64_t r1, uint16_t* gout, float* dbias, hipStream_t st);
// score_bwd.hip: the scoring layer's backward without the d-logits matrix (recompute + second product fused, device-side row count)
bool sgrad_supported(const ModelDev& m);
size_t sgrad_forward_part_floats(const ModelDev& m, int64_t R);
size_t sgrad_dwo_slab_floats(const ModelDev& m, int64_t R);      // scratch of launch_sgrad_dwo's row parts at small catalogues (0: none)   // scratch of launch_sgrad_forward's small-batch split (0: none)
hipError_t launch_sgrad_prepare(const ModelDev& m, const uint16_t* hidden, const int32_t* y, const float* w, const float* ce, const float* tgt, const int32_t* map,
                                const int32_t* n_dev, int64_t R, uint16_t* hc, int32_t* yc, float* wc, float* rb_nat, float* rb_l2, float* bias_l2,
                                hipStream_t st);
hipError_t launch_sgrad_forward(const ModelDev& m, const uint16_t* hidden, const int32_t* y, const float* w, const int32_t* row_map,
                                const int32_t* n_dev, int64_t R, int32_t* o_rank, float* o_ce, float* o_tgt, float* dh, int32_t* ovf,
                                hipStream_t st, float* part = nullptr);
hipError_t launch_sgrad_dh(const ModelDev& m, const uint16_t* hc, const float* rb_l2, const int32_t* yc, const float* wc, const int32_t* row_map,
                           const int32_t* n_dev, int64_t R, float* dh, int32_t* ovf, int vote, hipStream_t st);
hipError_t launch_sgrad_dwo(const ModelDev& m, const uint16_t* hc, const float* rb_nat, const int32_t* yc, const float* wc, const int32_t* n_dev,
                            int64_t R, const float* bias_l2, float* dwo, float* dbias, hipStream_t st, float* slabs = nullptr);
// score.hip: the fix-up flavour of the production scoring kernel alone (row tiles whose ce is non-finite are redone with the per-tile redo)
hipError_t launch_score_fixup_only(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                                   hipStream_t st, const int32_t* row_map, const int32_t* n_rows_dev);
size_t low_gru_workspace_bytes(const ModelDev& m, int64_t n_sess, int L);
// buffers of the HierGRU low level: evaluation keeps two states and one record per layer, training all of them
struct LowGruBufs {
  uint16_t* a_emb;    // [R][E] embedded ids
  uint16_t* u16;      // [R][E] projected [e, state] rows, storage type
  float* uf;          // [R][E] the same as fp32 (the GRU's input)
  int32_t* seq_len;   // [n_sess]
  float* states;      // [n_states][n_sess][HT]; state before step t = buffer t % n_states
  int n_states;       // 2 (ping-pong) or L + 1
  float* rec;         // per layer (keep_all: per step and layer): gates [M][2H] | r*h [M][H] | cand [M][H] | out [M][H]
  int keep_all;
  float* hidden32;    // exact numerics mode (exact.hip): uf is already filled with the fp32 projection and the outputs go here as fp32 rows
  uint16_t* x3;       // scratch of the split-operand cells (gru_x3.hip), low_gru_x3_elems() 16-bit elements; null: the fp32 cells run
  float* persist;     // scratch of the persistent launch (gru_persist.hip), low_gru_persist_floats() floats; null: the launch chain runs
  void* pipe;         // exchange slots + flags of the layer-pipelined launch (lgru_pipe.hip), lgru_pipe_bytes() bytes; null: lgru_tile.hip runs
};
size_t low_gru_x3_elems(const ModelDev& m, int64_t n_sess);
// lgru_tile.hip: the HierGRU low level of small batches, one workgroup per 32 sessions through all steps and layers (no launch chain, no exchange)
bool lgru_tile_applies(const ModelDev& m, int64_t n_sess);
hipError_t launch_lgru_tile_bwd(const ModelDev& m, const float* dH, const int32_t* seq_len, const float* states, const float* rec, int64_t n_sess, int L, float* du,
                                float* dac_all, float* dag_all, hipStream_t st);
hipError_t launch_frag_major_T(const uint16_t* W, uint16_t* WfT, int N, int K, hipStream_t st);
// lgru_pipe.hip: the same cells at H <= 128 with one workgroup per (32-session tile, layer), the layer's weights resident in registers
bool lgru_pipe_applies(const ModelDev& m, int64_t n_sess);
size_t lgru_pipe_bytes(const ModelDev& m, int64_t n_sess, int L);
hipError_t launch_lgru_pipe(const ModelDev& m, const uint16_t* u16, const int32_t* seq_len, int64_t n_sess, int L, uint16_t* hidden, float* states, float* rec,
                            void* scratch, hipStream_t st);
// lgru_wide.hip: H = 256: eight workgroups per (tile, layer), units and reduction split so that the weights stay in registers
bool lgru_wide_applies(const ModelDev& m, int64_t n_sess);
size_t lgru_wide_bytes(const ModelDev& m, int64_t n_sess, int L);
hipError_t launch_lgru_wide(const ModelDev& m, const uint16_t* u16, const int32_t* seq_len, int64_t n_sess, int L, uint16_t* hidden, float* states, float* rec,
                            void* scratch, hipStream_t st);
bool lgru_wide_bwd_applies(const ModelDev& m, int64_t n_sess);
size_t lgru_wide_bwd_bytes(const ModelDev& m, int64_t n_sess, int L);
hipError_t launch_lgru_wide_bwd(const ModelDev& m, const float* dH, const int32_t* seq_len, const float* states, const float* rec, int64_t n_sess, int L, float* du,
                                float* dac_all, float* dag_all, void* scratch, hipStream_t st);
hipError_t launch_lgru_tile(const ModelDev& m, const uint16_t* u16, const int32_t* seq_len, int64_t n_sess, int L, uint16_t* hidden, float* states, float* rec,
                            hipStream_t st);
// gru_persist.hip: the HierGRU low level of small evaluation batches as one persistent launch
bool low_gru_persist_applies(const ModelDev& m, int64_t n_sess);
size_t low_gru_persist_floats(const ModelDev& m, int64_t n_sess, int L);
hipError_t launch_low_gru_persist(const ModelDev& m, const float* u, const int32_t* seq_len, int64_t n_sess, int L, uint16_t* hidden, float* hidden32,
                                  float* ws, hipStream_t st);
// one GRU cell product + epilogue on split 16-bit operands (gru_x3.hip)
enum { X3_GATES = 0, X3_CAND = 1 };
struct GruX3Args {
  const uint16_t* Wh; const uint16_t* Wl; const float* bias;     // [N][K1 + K2] hi / lo, [N]
  const uint16_t* A1h; const uint16_t* A1l; int64_t lda1; int K1;   // first K1 reduction columns: the cell's input (A1l may be null)
  const uint16_t* A2h; const uint16_t* A2l; int64_t lda2; int K2;   // next K2: h (gates) or r * h (candidate)
  int64_t M; int N, H, epi;
  int save_all;                      // 1 (training forward): r, r*h, the candidate and the output are kept in fp32 for the backward pass
  float* gates;                      // [M][2H]: written (GATES) / read (CAND: the update gate)
  const float* h; int64_t ldh;       // the layer's old state slice
  float* rh; uint16_t* rh_hi; uint16_t* rh_lo;                   // GATES: r * h, fp32 [M][H] + its split
  float* cand_out;                   // CAND, optional: tanh(.) [M][H] (kept for the backward pass)
  float* out; uint16_t* out_hi; uint16_t* out_lo;                // CAND: h' [M][H] (+ split for the next layer, optional)
  float* state_new; uint16_t* state_hi; uint16_t* state_lo; int64_t lds_;   // CAND: the carried state slice (h' while the row is active) + split
  const int32_t* seq_len; int tstep;
  uint16_t* hid_out; int64_t ldho;   // CAND, optional: the scoring operand row
};
struct GruX3Batch { GruX3Args g[HIERTCN_MAX_LAYERS]; int n; };   // the cells (layer l, step tau - l) of one wavefront diagonal
hipError_t launch_gru_x3(const GruX3Batch& bt, hipStream_t st);
hipError_t launch_split_f32(const float* v, int64_t n, uint16_t* hi, uint16_t* lo, hipStream_t st);
bool gru_x3_applies(const ModelDev& m);
hipError_t refresh_lgru_split(const ModelDev& m, hipStream_t st);
size_t low_gru_save_floats(const ModelDev& m, int64_t n_sess, int L);
hipError_t launch_low_gru_bufs(const ModelDev& m, const int32_t* x, const float* sproj, int64_t n_sess, int L, uint16_t* hidden,
                               const LowGruBufs& b, hipStream_t st);
// GRU stack on one-hot input (model_gru_gmm's trunk): layer 0's input half as gathered rows of in_g [N][2H] / in_c [N][H]; wg[0] [2H][H] and
// wc[0] [H][H] hold the state half only (output-major), layers >= 1 the full [2H][2H] / [H][2H]
struct OneHotGru {
  int N, H, nl;
  const float* in_g; const float* in_c;
  const float* wg[HIERTCN_MAX_LAYERS]; const float* bg[HIERTCN_MAX_LAYERS];
  const float* wc[HIERTCN_MAX_LAYERS]; const float* bc[HIERTCN_MAX_LAYERS];
};
size_t onehot_gru_workspace_bytes(int H, int nl, int64_t n_sess);
size_t rows_gru_workspace_bytes(int H, int nl, int in_dim, int64_t n_sess, int L);
// the same stack on dense 16-bit rows of width in_dim (wg[0] [2H][in_dim + H], wc[0] [H][in_dim + H]: full kernels, output-major; in_g / in_c unused)
hipError_t launch_rows_gru(const OneHotGru& w, const uint16_t* rows16, int in_dim, int64_t n_sess, int L, uint16_t* hidden, void* workspace, hipStream_t st);
hipError_t launch_onehot_gru(const OneHotGru& w, const int32_t* x, int64_t n_sess, int L, uint16_t* hidden, void* workspace, hipStream_t st);
// ... with every record kept and fp32 output rows (the training step of model_gru_gmm, gmm_train.hip): states [L + 1][M][H nl], recs [L][nl][M 5 H]
hipError_t launch_onehot_gru_records(const OneHotGru& w, const int32_t* x, int64_t n_sess, int L, float* hidden32, int32_t* seq_len, float* states, float* recs,
                                     hipStream_t st);
hipError_t launch_rows_gru_records(const OneHotGru& w, const float* rows, int in_dim, int64_t n_sess, int L, float* hidden32, int32_t* seq_len, float* states,
                                   float* recs, hipStream_t st);   // ... on dense fp32 rows [n_sess * L][in_dim]: wg[0] / wc[0] the full kernels
hipError_t launch_low_gru(const ModelDev& m, const int32_t* x, const float* sproj, int64_t n_sess, int L, uint16_t* hidden, void* workspace,
                          hipStream_t st);
size_t tcn_workspace_bytes(const ModelDev& m, int64_t n_sessions, int L);
// activations the training backward needs, written by the fused low-level kernel on the way (all [n_sessions * L][C], any may be null)
struct TcnSaves {
  uint16_t* emb;
  uint16_t* a[HIERTCN_MAX_BLOCKS + 1];   // a[0]: projection output (hier), a[k + 1]: output of block k
  uint16_t* c[HIERTCN_MAX_BLOCKS];       // relu(conv + b) of block k
};
bool tcn_fused_applies(const ModelDev& m, int L);
hipError_t launch_tcn_forward(const ModelDev& m, const int32_t* x, const float* sproj, int64_t n_sessions, int L,
                              uint16_t* hidden, void* workspace, hipStream_t st, const TcnSaves* sv = nullptr);
// one layer of the layer-at-a-time low level (tcn_layers.hip)
struct LayerArgs {
  const uint16_t* in;      // [n_seq * L][Cin]
  uint16_t* out;           // [n_seq * L][Cout]
  const uint16_t* W;       // [taps][Cout][Cin]
  const float* bias;       // [Cout] or null
  const float* seq_init;   // [n_seq][Cout] fp32 accumulator start per sequence (projection: state . Wp[E:]) or null
  const uint16_t* Wres;    // [Cout][Cin] or null
  const float* bres;       // [Cout] or null
  int Cin, Cout, taps, dil, L;
  int64_t n_seq;
  uint16_t* c_out;         // optional [n_seq * L][Cout]: relu(conv + b) before the residual add (kept for training)
  int linear;              // 1: out = conv (the 'emb' projection, model_tcn.py:35); 0: relu(relu(conv + b) + res)
  const uint16_t* Wf;      // optional fragment-major copies of W / Wres (launch_frag_major): with them, long sequences run the LDS-tiled kernel
  const uint16_t* Wresf;
};

hipError_t launch_tcn_layer(const LayerArgs& a, hipStream_t st);
hipError_t launch_frag_major(const uint16_t* W, uint16_t* Wf, int taps, int Cout, int Cin, hipStream_t st);
hipError_t refresh_frag_major(const ModelDev& m, hipStream_t st);   // all fragment-major copies from projT / convT / resT
hipError_t launch_embed_rows(const ModelDev& m, const int32_t* ids, int64_t R, uint16_t* out, hipStream_t st);
size_t tcn_layers_workspace_bytes(const ModelDev& m, int64_t n_sessions, int L);
hipError_t launch_tcn_layers(const ModelDev& m, const int32_t* x, const float* sproj, int64_t n_sessions, int L, uint16_t* hidden,
                             void* workspace, hipStream_t st);
// row_map / n_rows_dev (optional, production kernel only): score only the rows listed in row_map[0 .. n_rows_dev[0])
hipError_t launch_score(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                        float* logits, hipStream_t st, const int32_t* row_map = nullptr, const int32_t* n_rows_dev = nullptr,
                        float* split_scratch = nullptr);
hipError_t launch_embed_concat(const ModelDev& m, const int32_t* ids, const float* state, int64_t n_sess, int L, uint16_t* out, hipStream_t st);
hipError_t launch_score_sigma(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, float* o_q, hipStream_t st);
hipError_t launch_score_sumsq(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, float* o_sq, float* o_tgt, hipStream_t st);
// Gram form of the mixture losses (gmm_gram.hip): |mu_m|^2 as a C x C quadratic form of the hidden row instead of a pass over the [D, C] table
bool gmm_gram_applies(int C, int D);
size_t gmm_gram_scratch_doubles(int M, int C);
size_t gmm_gram_table_bytes(int M, int C);
hipError_t launch_gmm_gram_build(const uint16_t* muT, const float* mu_bias, int M, int D, int DP, int C, double* gram64, unsigned char* G3, float* gx,
                                 float* g0, hipStream_t st);
hipError_t launch_gmm_quad(const uint16_t* hidden, int64_t R, int C, int M, const unsigned char* G3, float* sq, hipStream_t st);
hipError_t launch_gmm_prior_gram(const uint16_t* hidden, const float* pi_w, const float* pi_b, const float* gx, const float* g0, const uint16_t* muT,
                                 const float* mu_bias, const int32_t* y, int64_t R, int C, int M, int D, int DP, float* pl, float* sq, float* tg,
                                 hipStream_t st);
// small batches: number of catalogue parts the production kernel splits into (1 = none); scratch = parts * 2 * R floats
int score_split_parts(int64_t R, int NP, int rows_per_wg = 512);   // 512: the 16-bit kernels' row tile (SC_THREADS); 256: score_x3.hip's
hipError_t launch_score_merge(const float* partials, int parts, int64_t R, const int32_t* row_map, const int32_t* n_rows_dev, const int32_t* y,
                               const float* tgt, int32_t* o_rank, float* o_ce, hipStream_t st, int dyn_grid = 0, int rows_per_wg = 0, int n_all = 0);
// parts the split scratch of a small batch has to hold (the device-side split may take more parts than score_split_parts: score_geom.h)
int score_split_scratch_parts(int64_t R, int NP, int rows_per_wg = 512);
// ascending list of the rows with y > 0 (device-side, 3 small kernels): map [R], n_valid [1], scratch [(R+1023)/1024 + 1] ints
hipError_t launch_compact_rows(const int32_t* y, int64_t R, int32_t* map, int32_t* n_valid, int32_t* scratch, hipStream_t st, int32_t* z0 = nullptr,
                               float* z1 = nullptr, float* z2 = nullptr);
hipError_t launch_score_grad_chunk(const ModelDev& m, const uint16_t* hidden, const int32_t* y, int64_t R, const float* lse, const float* w,
                                   int stg_begin, int stg_end, uint16_t* gout, hipStream_t st);
hipError_t launch_metrics_scratch(const ModelDev& m, const int32_t* rank, const float* ce, const int32_t* y, int B, int T,
                                  double* out8, double* scratch, hipStream_t st);
#define HIERTCN_METRIC_BLOCKS_MAX 65536  // 4 lanes per block -> up to 262,144 lanes per call
