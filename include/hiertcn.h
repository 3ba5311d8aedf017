/*
 * hiertcn.h -- C-ABI of the MI355X-native HierTCN forward + scoring path.
 *
 * This is the drop-in boundary: every entry point stands where one piece of
 * the reference's TF-1.x graph stood.  Citations are file:line under the reference tree.
 *
 *   reference                                                   this library
 *   ---------------------------------------------------------   ---------------------------
 *   tf variables created by model_hier / model_tcn              hiertcn_set_weights
 *   model_hier.py:83-85  (session mean-pool + 'emb' with bias)  hiertcn_session_pool
 *   model_hier.py:36-37,91,93 + customed_gru_cell.py:309-337,   hiertcn_gru_scan
 *       1050-1073 (MultiRNNCell of GRUCells, mask reset) and the
 *       state half of model_hier.py:54-55 -> model_tcn.py:35
 *   model.py:59-61 + model_hier.py:50 (one-hot "lookup"),        hiertcn_tcn_forward
 *       model_tcn.py:35-36, customized_tcn_cell.py:46-49,109-127,
 *       147-161, customized_convolution_layer.py:171-198
 *   model_tcn.py:41 (scoring dense) + model.py:105 (mask) +      hiertcn_score_fused
 *       loss.py:20-21 (cross entropy) + loss.py:179 (count-greater
 *       rank) + loss.py:120 (top_k, optional)
 *   model_tcn.py:41 alone (materialised logits, parity/debug)    hiertcn_score_logits
 *   model.py:111-117 + loss.py:189-219 (lane means)              hiertcn_metrics
 *   one sess.run of the eval fetch list run_hier_xing.py:145-149 hiertcn_forward_eval
 *
 * Conventions
 *   - plain C, no exceptions, no torch types.  Every pointer is a DEVICE pointer owned by the
 *     caller unless it says "host".  Nothing is allocated or synchronised on the hot path; all
 *     work is enqueued on the caller's hipStream_t (passed as void*).
 *   - returns 0 on success; otherwise a negative hiertcn_status and a thread-local message
 *     retrievable with hiertcn_last_error().
 *   - one handle per device; a handle is not thread-safe, different handles are independent.
 *   - ids are int32, 0 = null/padding (model.py:59: one_hot(id)*sign(id) -> zero embedding, and
 *     model.py:62: mask_y = sign(y_id)).  Batch layout is user-major:
 *       x_ids, y_ids [B][S][L] ; mask [B][S] ; state [B][H*nl] ; rows r = (b*S+s)*L+t.
 *   - "bf16" buffers are uint16_t holding IEEE bfloat16 bit patterns.
 *   - numerics: by default the item tables, conv kernels, scoring kernel and the activations between layers are STORED in 16 bits
 *     (bfloat16; IEEE half in libhiertcn_gfx950_f16.so) and accumulated in fp32.  hiertcn_set_precision(h, HIERTCN_PRECISION_FP32)
 *     selects the exact mode: everything fp32, as the reference computes it (model_tcn.py:35-41); HIERTCN_PRECISION_SPLIT16 the fast
 *     parity-grade mode (split 16-bit operands, three MFMAs per product).  Both meet "fp32 logits within 1e-3" at the logit scale of a
 *     trained model; `hidden` buffers then hold float instead of uint16_t.
 */
#ifndef HIERTCN_H_
#define HIERTCN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIERTCN_ABI_VERSION 12
#define HIERTCN_MAX_BLOCKS 8
#define HIERTCN_MAX_LAYERS 4
#define HIERTCN_MAX_TOPK 16

typedef struct hiertcn_ctx* hiertcn_handle_t;

enum hiertcn_status {
  HIERTCN_OK = 0,
  HIERTCN_ERR_INVALID = -1,     /* bad argument / unsupported shape */
  HIERTCN_ERR_HIP = -2,         /* a HIP runtime call failed */
  HIERTCN_ERR_STATE = -3,       /* weights not set, handle destroyed, ... */
  HIERTCN_ERR_WORKSPACE = -4    /* workspace too small */
};

/* Model shape = the reference's args that size the graph (args.py:47-58) plus emb_dim, which the
 * reference hard-codes to 128 (model_hier.py:50,85; model_tcn.py:35). */
typedef struct {
  int32_t item_num;                      /* N, args.item_num (row 0 = null id)            */
  int32_t emb_dim;                       /* E = C0: 'emb' units                            */
  int32_t hidden_dim;                    /* H, args.hidden_dim (ignored when hier == 0 and low_type == 0) */
  int32_t num_layer;                     /* nl, args.num_layer                             */
  int32_t kernel_size;                   /* K, args.kernel_size                            */
  int32_t n_blocks;                      /* len(args.tcn_channel)                          */
  int32_t channels[HIERTCN_MAX_BLOCKS];  /* args.tcn_channel                               */
  int32_t hier;                          /* 1: model_hier low level; 0: single-level model_tcn / model_gru (model.py:66-73) */
  int32_t low_type;                      /* hier == 1: args.model_low_type -- 0 'tcn' = HierTCN; 1 'gru' = HierGRU, the reference's
                                          * default (args.py:40; model_hier.py:67-68 -> model_gru.py:16-63): n_blocks / channels /
                                          * kernel_size are ignored, the scoring layer reads hidden_dim channels.
                                          * hier == 0: 0 = model_tcn; 1 = the single-level model_gru (args.model_type 'gru', model.py:66-67):
                                          * the embedded ids feed the same GRU stack directly (model_gru.py:46 on one-hot rows is the
                                          * embedding lookup), zero initial state, sequence_length = the count of non-null input steps */
} hiertcn_config_t;

/* Device pointers to fp32 tensors in the REFERENCE's variable layout:
 *   emb_kernel      [N][E]              hier/emb/kernel (or tcn/emb/kernel when hier == 0)
 *   emb_bias        [E]                 hier/emb/bias                (hier only)
 *   proj_kernel     [E+H*nl][E]         hier/tcn/emb/kernel          (hier only)
 *   conv_kernel[i]  [K][Cin][Cout]      .../tblock_i/conv1/kernel
 *   conv_bias[i]    [Cout]              .../tblock_i/conv1/bias
 *   res_kernel[i]   [Cin][Cout] or NULL .../tblock_i/dense/kernel    (only when Cin != Cout)
 *   res_bias[i]     [Cout] or NULL
 *   out_kernel      [C_last][N]         .../tcn/dense/kernel
 *   out_bias        [N]                 .../tcn/dense/bias
 *   gru_*_kernel[l] [in_l+H][2H] / [in_l+H][H], gru_*_bias[l] [2H] / [H]   (hier only)
 * hiertcn_set_weights converts them (device-side) to the internal layouts and
 * keeps its own copies; the caller's buffers may be freed once the stream has drained. */
typedef struct {
  const float* emb_kernel;
  const float* emb_bias;
  const float* proj_kernel;
  const float* conv_kernel[HIERTCN_MAX_BLOCKS];
  const float* conv_bias[HIERTCN_MAX_BLOCKS];
  const float* res_kernel[HIERTCN_MAX_BLOCKS];
  const float* res_bias[HIERTCN_MAX_BLOCKS];
  const float* out_kernel;
  const float* out_bias;
  const float* gru_gate_kernel[HIERTCN_MAX_LAYERS];
  const float* gru_gate_bias[HIERTCN_MAX_LAYERS];
  const float* gru_cand_kernel[HIERTCN_MAX_LAYERS];
  const float* gru_cand_bias[HIERTCN_MAX_LAYERS];
  /* low_type == 1 only: hier/gru/rnn/multi_rnn_cell/cell_{l}/gru_cell/{gates,candidate}/{kernel,bias};
   * proj_kernel is then hier/gru/emb/kernel, out_kernel / out_bias hier/gru/dense/{kernel [H][N], bias}
   * (single-level model_gru: gru/rnn/..., emb_kernel = gru/emb/kernel, out_kernel / out_bias = gru/dense/{kernel, bias}, no proj_kernel) */
  const float* low_gate_kernel[HIERTCN_MAX_LAYERS];
  const float* low_gate_bias[HIERTCN_MAX_LAYERS];
  const float* low_cand_kernel[HIERTCN_MAX_LAYERS];
  const float* low_cand_bias[HIERTCN_MAX_LAYERS];
} hiertcn_weights_t;

/* Outputs of one evaluation step (all optional except where noted; NULL = not wanted). */
typedef struct {
  int32_t* rank;       /* [R]  loss.py:179 count of candidates strictly greater than the target; 0 on masked rows */
  float* ce;           /* [R]  loss.py:21 cross entropy  (LSE - target logit); 0 on masked rows      */
  float* target;       /* [R]  the target logit z[y]                                                  */
  int32_t* topk_idx;   /* [R][topk]  loss.py:120 order: value descending, ties -> lower index        */
  float* topk_val;     /* [R][topk]                                                                   */
  int32_t topk;        /* 0..HIERTCN_MAX_TOPK                                                         */
} hiertcn_score_out_t;

int hiertcn_abi_version(void);
/* 0: the 16-bit tensors are bfloat16 (libhiertcn_gfx950.so); 1: IEEE half (libhiertcn_gfx950_f16.so, inference only) */
int hiertcn_storage_dtype(void);
const char* hiertcn_last_error(void);

/* create on the CURRENT hip device; allocates the internal weight copies (sized from cfg). */
int hiertcn_create(hiertcn_handle_t* out, const hiertcn_config_t* cfg);
int hiertcn_destroy(hiertcn_handle_t h);
int hiertcn_get_config(hiertcn_handle_t h, hiertcn_config_t* out);

int hiertcn_set_weights(hiertcn_handle_t h, const hiertcn_weights_t* w, void* stream);

/* Numerics mode of the evaluation path (session_pool, gru_scan, tcn_forward, score_fused, score_logits, forward_eval).
 *   HIERTCN_PRECISION_STORAGE16 (default)  16-bit storage / fp32 accumulate: the throughput kernels (8 mantissa bits per factor of a product).
 *   HIERTCN_PRECISION_FP32                 fp32 end to end on v_mfma_f32_32x32x2_f32, logits materialised per row chunk; rank / cross
 *                                          entropy / ranked top-k taken on the materialised fp32 values (loss.py:21,120,179).  For parity
 *                                          runs, not for throughput (~20x slower than STORAGE16).
 *   HIERTCN_PRECISION_SPLIT16 (ABI 9)      the parity-grade FAST mode: the low level in fp32 (as FP32), the scoring layer -- 99 % of the work --
 *                                          FUSED as in STORAGE16 (no logits written; rank / cross entropy / ranked top-k in registers) on the
 *                                          16-bit MFMA with every factor carried as hi + lo IEEE-half parts (22 mantissa bits, in either library
 *                                          build): h.w ~= h_hi w_hi + h_lo w_hi + h_hi w_lo in one fp32 accumulator (csrc/score_x3.hip).  Logit
 *                                          error of fp32 order (~1e-6 of the logit scale): inside the north star's "fp32 logits within 1e-3" at
 *                                          the scale of a trained model (|z| ~ 10-30), ranks and ranked top-k equal to the fp32 reference's
 *                                          wherever its own gaps exceed that error.  A third of STORAGE16's rate.  Every last width (64, 128, 256;
 *                                          512 since ABI 10: csrc/score_x3w.hip).
 *                                          (The table and every hidden row are scaled by powers of two before the split: no range limit
 *                                          short of fp32's own.)
 * In FP32 and SPLIT16 every `hidden` argument is float [R][C_last].  SPLIT16 is evaluation only.  FP32 also TRAINS (ABI 10): hiertcn_set_precision(h,
 * HIERTCN_PRECISION_FP32) followed by hiertcn_train_enable puts the handle into the fp32 training mode -- the step of model.py:135-141 as the reference takes
 * it: fp32 forward on the masters, logits [R][NP] materialised in row chunks of at most tuning.train_fp32_chunk_mib MiB (the reference's own batch: one chunk; sized for such batches), every
 * product of the backward an exact-fp32 GEMM; gradients agree with fp64 autograd on the unrounded weights to ~1e-6.  model_hier with low_type 0 (TemporalConvNet) or 1 (the GRU stack) and single-level model_tcn.
 * FP32 keeps ONE logits scratch per handle: calls of a handle in that mode must not overlap on different streams (SPLIT16 and STORAGE16 have no such state).
 * The first switch to a mode must precede hiertcn_set_weights (it allocates the weight copies that set_weights fills -- for FP32 also a logits
 * scratch of <= 2 GiB owned by the handle); after that the modes may be switched back and forth between calls. */
#define HIERTCN_PRECISION_STORAGE16 0
#define HIERTCN_PRECISION_FP32 1
#define HIERTCN_PRECISION_SPLIT16 2
int hiertcn_set_precision(hiertcn_handle_t h, int32_t precision);
int hiertcn_get_precision(hiertcn_handle_t h, int32_t* precision);

/* Launch selection and thresholds of a handle (ABI 10).  The library reads NO environment variable: what used to be ambient process state
 * (HIERTCN_GRU_CHAIN, HIERTCN_LGRU_PIPE, ..., looked up inside the launch path) is a field of this struct, defaulted by hiertcn_create, read
 * from the handle by every launch, changed only through hiertcn_set_tuning.  hiertcn_amd/_lib.py maps the old variable names onto the fields
 * ONCE, when an Engine is created (tuning_from_env), so scripts that export them keep working.  The A/B alternatives of superseded kernels
 * are not here: they exist only in the ablation build (hiertcn_amd/build.py, HIERTCN_ABLATION=1).
 * Selection fields: -1 automatic (the library decides from the shape: the default), 0 never, 1 wherever the shape allows.
 * Thresholds: hiertcn_get_tuning reports the value in force; 0 in hiertcn_set_tuning = "the default".
 * Changing a field changes which kernels run and the scratch some of them need: set the tuning BEFORE sizing workspaces
 * (hiertcn_workspace_bytes / hiertcn_train_workspace_bytes are functions of it). */
typedef struct {
  int32_t gru_scan_persistent;        /* the across-session GRU scan as ONE persistent launch (csrc/gru_persist.hip) instead of 2 (S + 1) GEMM
                                       * launches; 0 on a GPU shared with another process / rank (its workgroups must be co-resident)      */
  int32_t gru_scan_persistent_train;  /* ... in the training forward, which also writes the backward's records                              */
  int32_t gru_bptt_persistent;        /* the scan's backward through time as one persistent launch                                          */
  int32_t gru_bptt_fused;             /* fused epilogues of the chained BPTT GEMMs (0: separate element-wise launches)                      */
  int32_t lgru_persistent;            /* HierGRU low level: the launches that wait on other workgroups (persistent / layer-pipelined / wide);
                                       * 0 on a shared GPU; 1 also takes the persistent launch at H = 256, where the chain is as fast       */
  int32_t lgru_pipe;                  /* csrc/lgru_pipe.hip (H <= 128: one workgroup per (32-session tile, layer))                          */
  int32_t lgru_tile;                  /* csrc/lgru_tile.hip (one workgroup per 32 sessions through all steps and layers)                    */
  int32_t lgru_tile_bwd;              /* ... its backward                                                                                   */
  int32_t lgru_wide;                  /* csrc/lgru_wide.hip (H = 256: eight workgroups per (tile, layer))                                   */
  int32_t lgru_wide_bwd;              /* ... its backward                                                                                   */
  int32_t lgru_fp32_cells;            /* 1: the fp32 GRU cells instead of the split-operand ones (default -1 = 0)                           */
  int32_t tcn_layered;                /* 1: the layer-at-a-time TemporalConvNet kernels even where the fused kernel applies (default 0)     */
  int32_t gru_persist_max_lanes;      /* lanes up to which the persistent scan runs (default 96: the measured cross-over with the chain)    */
  int32_t gemm_split_max_tiles;       /* output tiles up to which the fp32 GEMMs of the scan split their reduction (default 1024)           */
  int32_t skinny_max_rows;            /* rows up to which the backward's fp32 GEMMs take the skinny kernel (default 4096)                   */
  int32_t lgru_tile_max_sessions;     /* sessions up to which the session-tile kernels run (default 131072)                                 */
  int32_t lgru_wide_max_launches;     /* launches a batch may be cut into for lgru_wide at H = 256 (default 3)                              */
  int32_t sgrad_dyn_max_blocks;       /* row blocks up to which the scoring backward deals its grid on the device (default 1024)            */
  int32_t sgrad0_min_stages;          /* catalogue stages per part of the scoring backward's forward split (default 64)                     */
  int32_t tcn_gs2_max_sessions;       /* sessions up to which the fused TCN kernel runs two sessions per workgroup (default 512; -1 never)  */
  int32_t lgru_defer_max_gib;         /* GiB of records up to which HierGRU's weight gradients are deferred to one GEMM (default 8; -1 never) */
  int32_t train_fp32_chunk_mib;       /* fp32 training mode: MiB of materialised fp32 logits per row chunk of the scoring layer (default 8192)   */
  int32_t reserved[10];               /* zero                                                                                               */
} hiertcn_tuning_t;
int hiertcn_get_tuning(hiertcn_handle_t h, hiertcn_tuning_t* out);
int hiertcn_set_tuning(hiertcn_handle_t h, const hiertcn_tuning_t* in);

/* sizeof() of the six structs of this header as the LIBRARY was compiled, in this order: hiertcn_config_t, hiertcn_weights_t,
 * hiertcn_score_out_t, hiertcn_gmm_config_t, hiertcn_gmm_weights_t, hiertcn_tuning_t -- lets a binding written in another language verify its
 * own struct definitions at load time (tests/test_cabi_load.py does, for hiertcn_amd/_lib.py and for the stub in INTEGRATION.md). */
#define HIERTCN_N_STRUCTS 6
int hiertcn_struct_sizes(size_t out[HIERTCN_N_STRUCTS]);

/* model_hier.py:83-85.  y_ids [n_sessions][L] -> pooled [n_sessions][E] fp32. */
int hiertcn_session_pool(hiertcn_handle_t h, const int32_t* y_ids, int64_t n_sessions, int32_t L,
                         float* pooled, void* stream);

/* model_hier.py:50,54-55 as rows: x_ids [n_sessions][L], state [n_sessions][H*nl] fp32 (the state each slot's low level sees:
 * hiertcn_gru_scan's state_before) -> rows [n_sessions*L][E + H*nl] in the storage type: E[x_t] (zeros for the null id) | state.
 * The input of a low level that has no 'emb' projection of its own (model_low_type tcn_gmm: hiertcn_gmm_forward_rows). */
int hiertcn_embed_concat(hiertcn_handle_t h, const int32_t* x_ids, const float* state, int64_t n_sessions, int32_t L, uint16_t* rows,
                         void* stream);

/* model_hier.py:91-93 scanned over the S slots for every lane, plus the state half of the low
 * level's input projection (model_hier.py:54-55 -> model_tcn.py:35: [e, state].Wp = e.Wp[:E] +
 * state.Wp[E:]).  pooled [B][S][E], mask [B][S], state_in [B][H*nl] ->
 *   sproj        [B][S][E]     state_before_slot_s . Wp[E:]        (consumed by hiertcn_tcn_forward)
 *   state_out    [B][H*nl]     state after the last slot
 *   state_before [B][S][H*nl]  optional (NULL): the state each slot's low level saw
 *   workspace    hiertcn_gru_workspace_bytes(h, B, S) bytes of scratch, 256-byte aligned (state history) */
size_t hiertcn_gru_workspace_bytes(hiertcn_handle_t h, int32_t B, int32_t S);
int hiertcn_gru_scan(hiertcn_handle_t h, const float* pooled, const float* mask, const float* state_in,
                     int32_t B, int32_t S, float* sproj, float* state_out, float* state_before,
                     void* workspace, size_t workspace_bytes, void* stream);

/* model.py:59-61 + model_hier.py:50 + model_tcn.py:35-36: ids -> last TCN activations.
 * x_ids [n_sessions][L]; sproj [n_sessions][E] or NULL (single-level) ->
 * hidden [n_sessions*L][C_last]: uint16_t bit patterns of the storage type (float in HIERTCN_PRECISION_FP32).
 * (widths: emb_dim and tcn_channel are multiples of 32; the last channel count, which the scoring kernels are built
 *  for, is one of 64, 128, 256, 512.)
 * Two device paths, same results: the fused kernel (every width == emb_dim in {64,128,256}, L <= 32:
 * gather + projection + all blocks in LDS; needs no workspace) and the layer-at-a-time path (any
 * L <= 4096, per-block widths in multiples of 32, Dense down-sample residual where Cin != Cout --
 * the single-level model_tcn of model.py:71-73 with T <= 500 and channels [128,128,128,128,256,256]),
 * which needs hiertcn_tcn_workspace_bytes() bytes of 256-byte aligned scratch. */
size_t hiertcn_tcn_workspace_bytes(hiertcn_handle_t h, int64_t n_sessions, int32_t L);
int hiertcn_tcn_forward(hiertcn_handle_t h, const int32_t* x_ids, const float* sproj, int64_t n_sessions,
                        int32_t L, void* hidden, void* workspace, size_t workspace_bytes, void* stream);

/* model_tcn.py:41 + model.py:105 + loss.py:21,179(,120) fused: logits are never written. */
int hiertcn_score_fused(hiertcn_handle_t h, const void* hidden, const int32_t* y_ids, int64_t R,
                        const hiertcn_score_out_t* out, void* stream);

/* parity/debug: materialise logits [R][N] fp32 (pre-mask, i.e. model_tcn.py:41's output). */
int hiertcn_score_logits(hiertcn_handle_t h, const void* hidden, int64_t R, float* logits, void* stream);

/* model.py:111-117 + loss.py:189-219.  rank, ce, y_ids are [B][T] (T = S*L); out8 (device, 8
 * doubles) = SUMS over lanes of the per-lane means, ready for a cross-rank all-reduce:
 *   [0] loss  [1] recall@1  [2] recall@5  [3] recall@10  [4] reciprocal rank  [5] rank/N
 *   [6] user_count (lanes with >= 1 valid step)  [7] valid steps
 * batch metric = out8[k] / out8[6]. */
int hiertcn_metrics(hiertcn_handle_t h, const int32_t* rank, const float* ce, const int32_t* y_ids,
                    int32_t B, int32_t T, double* out8, void* stream);

/* Bytes of scratch hiertcn_forward_eval needs for a [B][S][L] batch. */
size_t hiertcn_workspace_bytes(hiertcn_handle_t h, int32_t B, int32_t S, int32_t L);

/* One evaluation step = the fetch list of run_hier_xing.py:145-149 without `pred`:
 * pool -> gru scan -> tcn -> fused scoring -> lane metrics, all enqueued on `stream`.
 * score_out may be NULL (then rank/ce live in the workspace only).
 * Single-level handles (hier == 0; run_nn_xing.py:43-47's fetch list): S must be 1, L = T, x_ids/y_ids
 * [B][1][T]; mask, state_in, state_out are ignored and may be NULL. */
int hiertcn_forward_eval(hiertcn_handle_t h, const int32_t* x_ids, const int32_t* y_ids, const float* mask,
                         const float* state_in, int32_t B, int32_t S, int32_t L, void* workspace,
                         size_t workspace_bytes, float* state_out, double* metrics8,
                         const hiertcn_score_out_t* score_out, void* stream);

/* ---- training step (model.py:135-141: Adam on the masked lane-mean cross entropy of model.py:102-117) ----------
 * hiertcn_train_enable   before hiertcn_set_weights: allocates fp32 masters / gradients / Adam moments for every
 *                        variable (4 x parameter count floats).
 * hiertcn_train_forward_backward   one `sess.run([train_op minus the apply, loss, state])` of run_hier_xing.py:301:
 *                        forward with activations kept in `workspace`, loss into metrics8 (as hiertcn_metrics), gradients
 *                        of every variable into the gradient arena.  No gradient crosses the batch boundary (state_in is data).
 * hiertcn_grad_arena     the flat fp32 gradient buffer (device) -- the RCCL all-reduce target of data-parallel runs.
 * hiertcn_adam_step      TF-1.x AdamOptimizer update (lr_t = lr*sqrt(1-b2^t)/(1-b1^t); theta -= lr_t*m/(sqrt(v)+eps)) of the
 *                        masters from grad * grad_scale, then refreshes the bf16 device copies.
 * hiertcn_export_weights which = 0: masters, 1: gradients -> fp32 device tensors in the REFERENCE layouts (`dst` holds
 *                        destination pointers; NULL members are skipped) -- checkpoints and tests.
 * Both model kinds (hier == 1: x/y [B][S][L]; hier == 0: [B][1][T], mask/state NULL), per-block widths, Dense residual. */
int hiertcn_train_enable(hiertcn_handle_t h);
size_t hiertcn_train_workspace_bytes(hiertcn_handle_t h, int32_t B, int32_t S, int32_t L);
int hiertcn_train_forward_backward(hiertcn_handle_t h, const int32_t* x_ids, const int32_t* y_ids, const float* mask,
                                   const float* state_in, int32_t B, int32_t S, int32_t L, void* workspace,
                                   size_t workspace_bytes, float* state_out, double* metrics8, void* stream);
int hiertcn_grad_arena(hiertcn_handle_t h, float** grad, size_t* count);

/* ---- data-parallel training (users sharded over ranks, a full replica per GPU; model.py:135-141 applied to the
 * GLOBAL batch).  The step in two halves, so that the gradient collective of the scoring layer -- the large dense one -- runs
 * while the rest of the backward pass computes:
 *   hiertcn_train_step_begin    forward + loss + backward of the scoring layer (model_tcn.py:41): the gradients of out_kernel /
 *                               out_bias are FINAL behind it on `stream`.  users_global (device, 1 float, may be NULL): the
 *                               number of lanes with a target over ALL ranks (hiertcn_count_users, summed by the caller); it
 *                               replaces this batch's own count in the loss normalisation of model.py:113-117, which makes the
 *                               plain SUM of the ranks' gradients the gradient of the global batch's loss, ragged shards included.
 *   hiertcn_train_step_finish   the rest of the backward (blocks, projection, item embedding, GRU, session pool); same arguments.
 *   hiertcn_train_forward_backward == begin (users_global = NULL) + finish.
 *   hiertcn_count_users         users[0] = number of lanes of y [B][T] with at least one target (model.py:113).
 *   hiertcn_train_segment       which = 0 item table, 1 scoring kernel + bias (contiguous), 2 every other variable: offset / count
 *                               (floats) of that part inside each flat arena (hiertcn_grad_arena, hiertcn_train_arena).
 * Row-sparse exchange of the item table's gradient (only ids that occur in the batch have a non-zero row):
 *   hiertcn_grad_rows_list      ascending list of the distinct ids in x_ids / y_ids (either may be NULL; n_ids entries each) into
 *                               ids_out (capacity cap >= min(N - 1, ids given)), their number into count_out[0]; workspace of
 *                               hiertcn_grad_rows_workspace_bytes() bytes.
 *   hiertcn_grad_rows_op        for i < min(count[0], cap): op 0 rows[i][:] = grad_E[ids[i]][:]; 1 grad_E[ids[i]][:] = 0;
 *                               2 grad_E[ids[i]][:] += rows[i][:]   (rows [cap][E] fp32).                                        */
int hiertcn_train_step_begin(hiertcn_handle_t h, const int32_t* x_ids, const int32_t* y_ids, const float* mask, const float* state_in,
                             int32_t B, int32_t S, int32_t L, void* workspace, size_t workspace_bytes, float* state_out, double* metrics8,
                             const float* users_global, void* stream);
int hiertcn_train_step_finish(hiertcn_handle_t h, const int32_t* x_ids, const int32_t* y_ids, const float* mask, int32_t B, int32_t S,
                              int32_t L, void* workspace, size_t workspace_bytes, void* stream);
int hiertcn_count_users(hiertcn_handle_t h, const int32_t* y_ids, int32_t B, int32_t T, float* users, void* stream);
/* Warm-start runs (ABI 11; args.warm_start: model.py:40,102-103 `mask_y *= mask_warmstart`, fed at run_hier_xing.py:125-129,286-289 and
 * run_nn_xing.py:38-41,148-151 from the events' timestamps).  y_score_ids [B][S][L] = y_ids with the steps that do not count set to 0.
 * ONE-SHOT: the next hiertcn_forward_eval, hiertcn_train_forward_backward or hiertcn_train_step_begin + _finish pair of this handle takes
 * its targets, masks, loss, ranks, metrics and row weights from y_score_ids, while the session mean-pool of model_hier.py:83-85 (and its
 * backward) keeps reading the y_ids passed to that call; the pointer must stay valid until that call's work has run.  NULL clears it.
 * The ids are consumed by that call WHETHER IT SUCCEEDS OR NOT (a failed state check included); a hiertcn_train_step_begin that returns an
 * error drops them instead of keeping them for a finish that will not come: no error path leaves ids armed for an unrelated later step.
 * (hiertcn_count_users for such a step takes y_score_ids too: a lane counts when it has a SCORED target.) */
int hiertcn_set_score_ids(hiertcn_handle_t h, const int32_t* y_score_ids);
int hiertcn_train_segment(hiertcn_handle_t h, int32_t which, size_t* offset, size_t* count);
size_t hiertcn_grad_rows_workspace_bytes(hiertcn_handle_t h);
int hiertcn_grad_rows_list(hiertcn_handle_t h, const int32_t* x_ids, const int32_t* y_ids, int64_t n_ids, int32_t* ids_out, int64_t cap,
                           int32_t* count_out, void* workspace, size_t workspace_bytes, void* stream);
int hiertcn_grad_rows_op(hiertcn_handle_t h, int32_t op, const int32_t* ids, const int32_t* count, int64_t cap, float* rows, void* stream);
/* which = 0 masters, 1 gradients, 2 Adam m, 3 Adam v: the flat fp32 arenas (internal layouts; valid for this config
 * only) -- lets a checkpoint carry the optimizer slots as tf.train.Saver does; hiertcn_adam_time reads / sets t. */
int hiertcn_train_arena(hiertcn_handle_t h, int32_t which, float** ptr, size_t* count);
int hiertcn_adam_time(hiertcn_handle_t h, int64_t* get, const int64_t* set);
int hiertcn_adam_step(hiertcn_handle_t h, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
int hiertcn_export_weights(hiertcn_handle_t h, const hiertcn_weights_t* dst, int32_t which, void* stream);

/* ---- mixture-output models (BASELINE.json configs[4]: model_tcn_gmm.py:12-44; the head of model_gru_gmm.py:49-54; the likelihood
 * losses of loss.py:247-257).  A handle of its own: the trunk differs from model_tcn's (no 'emb' projection -- on XING the
 * TemporalConvNet reads one_hot(id)*sign(id) directly, so block 0 has Cin = item_num and a Dense residual) and the head has
 * n_out = num_mixture * (1 + output_dim) outputs with args.sigma set, num_mixture * (1 + 2 output_dim) with a learned scale (cfg.sigma = 0:
 * model_tcn_gmm.py:39-40; head columns: M priors, M*D means mixture-major, M*D LOG-scales mixture-major, sigma = exp(s); last width <= 256).
 * The DECODE of the head output is not in the reference (calc_loss_gmm has no caller): this library uses pi = softmax of the first
 * M outputs, mu = the next M*D outputs mixture-major, sigma = args.sigma -- "parity unpinned" for the decode; the head GEMM and the
 * formulas are checked against oracle/gmm_oracle.py.
 *   hiertcn_gmm_set_weights  fp32 device tensors in the reference layouts: conv_kernel[i] [K][Cin][Cout] (Cin = item_num for i = 0),
 *                            conv_bias[i], res_kernel[i] [Cin][Cout] / res_bias[i] where Cin != Cout (always for i = 0),
 *                            head_weights [C_last][M(1+D)], head_biases [M(1+D)]  (<scope>/fully_connected/{weights,biases}).
 *   hiertcn_gmm_forward      x_ids [B][T] -> hidden [B*T][C_last] (storage type), model_tcn_gmm.py:27-36; trunk 1: the GRU stack of
 *                            model_gru_gmm.py:33-48 on one-hot rows -- layer 0's input half is a row gather, sequence_length = the count
 *                            of ids > 0 (model_gru_gmm.py:57-67), C_last = hidden_dim.
 *   hiertcn_gmm_loss         hidden [R][C_last], y_ids [R] -> loss [R] (mode 0: loss.py:248 gmm_prob, 1: loss.py:249-257
 *                            gmm_prob_max; 0 where y == 0, as model.py:111 masks it).  Neither the [R][M*D] means nor the one-hot
 *                            targets are materialised: sum_d (y_d - mu_d)^2 = |mu_m|^2 - 2 mu_m[y] + 1.  Optional outputs (NULL =
 *                            not wanted): sq [M][R] = |mu_m|^2, tgt [M][R] = mu_m[y], prior [R][M] = the prior logits.
 *                            |mu_m|^2 is linear algebra on the hidden row -- x^T G_m x, x = [h; 1], G_m = [W_m^T | b_m]^T [W_m^T | b_m]
 *                            formed in fp64 by hiertcn_gmm_set_weights -- so where 3 C_last < D (and C_last <= 256, bf16 library)
 *                            no pass over the [D][C_last] mean tables is made at all (csrc/gmm_gram.hip); the environment variable
 *                            HIERTCN_GMM_TABLE_PASS=1, read by hiertcn_gmm_create, keeps the table pass (A/B, checks).
 *   hiertcn_gmm_head_means   parity/debug: materialise mu_m [R][D] fp32 for one mixture m.                                        */
typedef struct hiertcn_gmm_ctx* hiertcn_gmm_handle_t;
typedef struct {
  int32_t item_num;                      /* width of the one-hot input (args.item_num)            */
  int32_t output_dim;                    /* D, args.output_dim (= item_num on XING)                 */
  int32_t num_mixture;                   /* M, args.num_mixture                                     */
  int32_t kernel_size;
  int32_t n_blocks;
  int32_t channels[HIERTCN_MAX_BLOCKS];
  float sigma;                           /* args.sigma (> 0); 0 = learned scale (args.sigma None)   */
  int32_t trunk;                         /* 0: TemporalConvNet (model_tcn_gmm.py:27-36); 1: GRU stack run by dynamic_rnn on the one-hot
                                          *    input (model_gru_gmm.py:33-48) -- n_blocks / channels / kernel_size are then ignored;
                                          *    2 / 3 (input_dim below): the same two trunks on DENSE rows -- the low levels of model_hier    */
  int32_t hidden_dim;                    /* trunk 1: args.hidden_dim in {64, 128, 256, 512}         */
  int32_t num_layer;                     /* trunk 1: args.num_layer                                 */
  int32_t input_dim;                     /* trunk 2 (ABI 8): the TemporalConvNet on DENSE rows the caller provides -- model_hier.py:64-65, the
                                          *    mixture model as the LOW level of model_hier: rows [e_t, state] of width emb + hidden_dim*num_layer
                                          *    (hiertcn_embed_concat) -- width of those rows, a multiple of 32; block 0 is then an ordinary
                                          *    TemporalBlock with Cin = input_dim (conv_kernel[0] [K][input_dim][C0], Dense residual where they
                                          *    differ); item_num is ignored; hiertcn_gmm_forward_rows instead of hiertcn_gmm_forward.
                                          *    trunk 3 (ABI 9): the GRU stack on the same rows -- model_hier.py:68-69, model_low_type 'gru_gmm':
                                          *    gru_gate_kernel[0] [input_dim + H][2H], gru_cand_kernel[0] [input_dim + H][H], sequence_length =
                                          *    the count of rows that are not all-zero (model_gru_gmm.py:57-67); hidden_dim / num_layer as trunk 1 */
  int32_t table_pass;                    /* (ABI 10) 1: hiertcn_gmm_loss always as M passes of the scoring kernel over the [D, C] mean tables, also where
                                          *    the Gram form (3 C < D, C <= 256: csrc/gmm_gram.hip) applies -- A/B runs and tests; 0: the library decides */
} hiertcn_gmm_config_t;
typedef struct {
  const float* conv_kernel[HIERTCN_MAX_BLOCKS];
  const float* conv_bias[HIERTCN_MAX_BLOCKS];
  const float* res_kernel[HIERTCN_MAX_BLOCKS];
  const float* res_bias[HIERTCN_MAX_BLOCKS];
  const float* head_weights;
  const float* head_biases;
  /* trunk 1 (rnn.GRUCell layouts, customed_gru_cell.py:309-337): gate kernel [in + H][2H] with in = item_num for layer 0 and H above,
   * gate bias [2H], candidate kernel [in + H][H], candidate bias [H]  (<scope>/rnn/multi_rnn_cell/cell_l/gru_cell/{gates,candidate}/...) */
  const float* gru_gate_kernel[HIERTCN_MAX_LAYERS];
  const float* gru_gate_bias[HIERTCN_MAX_LAYERS];
  const float* gru_cand_kernel[HIERTCN_MAX_LAYERS];
  const float* gru_cand_bias[HIERTCN_MAX_LAYERS];
  /* TRAINING of trunk 2 / 3 as the low level of model_hier (hiertcn_gmm_train_step_hier, ABI 12) only -- the HIGH level's variables, which the step
   * differentiates too: hier/emb/kernel [item_num][E] and hier/emb/bias [E] (E = input_dim - hidden_dim * num_layer), and the across-session cells
   * hier/multi_rnn_cell/cell_l/gru_cell/{gates,candidate}/{kernel,bias} in the rnn.GRUCell layouts (in_0 = E, in_l = hidden_dim).  Ignored (may be
   * NULL) everywhere else: evaluation takes the high level from the main handle (hiertcn_session_pool, hiertcn_gru_scan, hiertcn_embed_concat). */
  const float* hl_emb_kernel;
  const float* hl_emb_bias;
  const float* hl_gate_kernel[HIERTCN_MAX_LAYERS];
  const float* hl_gate_bias[HIERTCN_MAX_LAYERS];
  const float* hl_cand_kernel[HIERTCN_MAX_LAYERS];
  const float* hl_cand_bias[HIERTCN_MAX_LAYERS];
} hiertcn_gmm_weights_t;
const char* hiertcn_gmm_last_error(void);
int hiertcn_gmm_create(hiertcn_gmm_handle_t* out, const hiertcn_gmm_config_t* cfg);
int hiertcn_gmm_destroy(hiertcn_gmm_handle_t g);
int hiertcn_gmm_set_weights(hiertcn_gmm_handle_t g, const hiertcn_gmm_weights_t* w, void* stream);
size_t hiertcn_gmm_workspace_bytes(hiertcn_gmm_handle_t g, int64_t R);
/* workspace of hiertcn_gmm_forward for B sequences of T steps (>= hiertcn_gmm_workspace_bytes(g, B*T): one buffer serves both calls) */
size_t hiertcn_gmm_forward_workspace_bytes(hiertcn_gmm_handle_t g, int64_t B, int32_t T);
int hiertcn_gmm_forward(hiertcn_gmm_handle_t g, const int32_t* x_ids, int64_t B, int32_t T, uint16_t* hidden, void* workspace,
                        size_t workspace_bytes, void* stream);
/* trunks 2, 3: x_rows [B*T][input_dim] (storage type) -> hidden [B*T][C_last]; workspace: hiertcn_gmm_forward_workspace_bytes */
int hiertcn_gmm_forward_rows(hiertcn_gmm_handle_t g, const uint16_t* x_rows, int64_t B, int32_t T, uint16_t* hidden, void* workspace,
                             size_t workspace_bytes, void* stream);
int hiertcn_gmm_loss(hiertcn_gmm_handle_t g, const uint16_t* hidden, const int32_t* y_ids, int64_t R, int32_t mode, float* loss,
                     float* sq_out, float* tgt_out, float* prior_out, void* workspace, size_t workspace_bytes, void* stream);
int hiertcn_gmm_head_means(hiertcn_gmm_handle_t g, const uint16_t* hidden, int64_t R, int32_t m, float* mu_out, void* stream);
/* model.py:111-117 applied to the per-step mixture loss [B][T] (mask = sign(y), per-lane mean over valid steps + 1e-6): out2 (device, 2
 * doubles) = [sum over lanes of the lane means, number of lanes with a target] -- ready for a cross-rank sum; batch loss = out2[0] / out2[1]. */
int hiertcn_gmm_lane_loss(hiertcn_gmm_handle_t g, const float* loss, const int32_t* y_ids, int32_t B, int32_t T, double* out2, void* stream);

/* ---- training step of the mixture-output model (ABI 12; model.py:135-141 on the lane-mean mixture loss of model.py:111-117 + loss.py:247-257).
 * The reference itself cannot train these models on XING (its graph does not build): the step differentiates the decode
 * documented above, and is checked against fp64 autograd of that decode.  Built for the models on ids -- trunk 0 (model_tcn_gmm: the
 * TemporalConvNet) and trunk 1 (model_gru_gmm: the GRU stack, backward through time over its T steps) --, a fixed scale (cfg.sigma > 0) or a
 * learned one (cfg.sigma = 0), loss modes 0 / 1 -- and for the dense-row trunks 2 / 3 as the low level of model_hier, together with the high level
 * (below).
 * Everything is fp32 on master copies kept in the REFERENCE's layouts (a checkpoint is a copy; Adam is elementwise).  With a fixed scale no
 * [rows][M D] matrix is formed in either direction -- the head's backward is linear algebra on the hidden rows through G_m = W_m W_m^T
 * (csrc/gmm_train.hip); a learned scale materialises the means and log-scales ([2 M][rows][D] fp32 in the workspace: size it with
 * hiertcn_gmm_train_workspace_bytes) and differentiates them in place.
 *   hiertcn_gmm_train_enable          allocates masters / gradients / the two Adam moments, fills the masters from `w` (hiertcn_gmm_set_weights'
 *                                     tensors), zeroes the moments and the time step.
 *   hiertcn_gmm_train_step            x_ids, y_ids [B][T] -> gradients of every variable of  loss = (1 / users) sum_b (sum_t l_bt [y_bt > 0]) /
 *                                     (act_b + 1e-6)  (users_global: device float [1], the lanes-with-a-target count of the GLOBAL batch in a
 *                                     data-parallel run, or NULL: this batch's own); out2 as hiertcn_gmm_lane_loss; loss_rows [B*T] optional.
 *   hiertcn_gmm_adam_step             TF-form Adam on the masters (grad_scale multiplies the gradients first); hiertcn_gmm_adam_time reads / sets t.
 *   hiertcn_gmm_train_sync            re-derives the evaluation tables of hiertcn_gmm_forward / hiertcn_gmm_loss from the masters (what
 *                                     hiertcn_gmm_set_weights does; synchronises the stream): call it before evaluating a handle that has stepped.
 *   hiertcn_gmm_train_copy            which = 0 masters, 1 gradients, 2 / 3 Adam moments; to_handle = 0: into the caller's fp32 device buffers
 *                                     `bufs` (reference layouts -- they are WRITTEN despite the const), 1: from them (checkpoint restore).
 *   hiertcn_gmm_train_arena           the flat arena `which` (device pointer, floats): a data-parallel run all-reduces arena 1 between
 *                                     hiertcn_gmm_train_step and hiertcn_gmm_adam_step.                                                        */
int hiertcn_gmm_train_enable(hiertcn_gmm_handle_t g, const hiertcn_gmm_weights_t* w, void* stream);
size_t hiertcn_gmm_train_workspace_bytes(hiertcn_gmm_handle_t g, int64_t B, int32_t T);
int hiertcn_gmm_train_step(hiertcn_gmm_handle_t g, const int32_t* x_ids, const int32_t* y_ids, int64_t B, int32_t T, int32_t mode,
                           const float* users_global, void* workspace, size_t workspace_bytes, double* out2, float* loss_rows, void* stream);
/* model_hier with model_low_type 'tcn_gmm' / 'gru_gmm' (model_hier.py:39-93 around trunk 2 / 3) trains as ONE step on the mixture handle: enabled with the high level's
 * variables (hiertcn_gmm_weights_t.hl_*; cfg.item_num / hidden_dim / num_layer describe them, emb width = input_dim - hidden_dim * num_layer), the step runs
 * session mean-pool -> across-session scan -> [e, state] rows -> TemporalConvNet or GRU stack -> head -> loss and all of its backward (the rows' gradient into the item
 * table and the states, backward through the S slots with the user-boundary mask, the pool), x_ids / y_ids [B][S][L], mask [B][S], state_in / state_out
 * [B][hidden_dim * num_layer]; lanes = users (B <= 4096 per step).  Every session holds at least one id > 0 (the loader's do): an empty one is the
 * reference's 0 / 0 (model_hier.py:84) -- a NaN state, and NaN gradients behind it.  Evaluation keeps using the main handle for the high level: after training, export
 * the hl_* masters (hiertcn_gmm_train_copy) into it (hiertcn_set_weights). */
size_t hiertcn_gmm_train_hier_workspace_bytes(hiertcn_gmm_handle_t g, int64_t B, int32_t S, int32_t L);
int hiertcn_gmm_train_step_hier(hiertcn_gmm_handle_t g, const int32_t* x_ids, const int32_t* y_ids, const float* mask, const float* state_in, int64_t B, int32_t S,
                                int32_t L, int32_t mode, const float* users_global, void* workspace, size_t workspace_bytes, float* state_out, double* out2,
                                float* loss_rows, void* stream);
int hiertcn_gmm_adam_step(hiertcn_gmm_handle_t g, float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);
int hiertcn_gmm_adam_time(hiertcn_gmm_handle_t g, int64_t* get, const int64_t* set);
int hiertcn_gmm_train_sync(hiertcn_gmm_handle_t g, void* stream);
int hiertcn_gmm_train_copy(hiertcn_gmm_handle_t g, int32_t which, int32_t to_handle, const hiertcn_gmm_weights_t* bufs, void* stream);
int hiertcn_gmm_train_arena(hiertcn_gmm_handle_t g, int32_t which, float** ptr, size_t* count);

/* Measurement hook for bench.py: average duration in milliseconds of the scoring kernel over the
 * launches made since the last reset, measured with HIP events recorded on the launch stream.
 * Disabled (0 cost) unless enabled. */
int hiertcn_timing_enable(hiertcn_handle_t h, int32_t on);
int hiertcn_timing_read(hiertcn_handle_t h, double* score_ms_avg, int64_t* launches, int32_t reset);

/* Health word of the handle's persistent launches (the across-session GRU scan of small batches, csrc/gru_persist.hip): every wait of
 * one workgroup for another is bounded, and a wait that ran out sets this word instead of hanging the device -- the results of that
 * launch are then wrong.  *timed_out = 1 if that happened on this handle since the last call (0 in a healthy run).  Synchronises the
 * device; a reported time-out also resets the handle's barrier counters, so that later launches synchronise correctly again.  The run
 * loops read it once per evaluation epoch and raise (hiertcn_amd/run_hier_xing.py). */
int hiertcn_persistent_health(hiertcn_handle_t h, int32_t* timed_out);
/* The same word without a synchronisation (ABI 10): enqueues, on `stream`, a copy of the 8-byte health word into *host_word -- pinned host
 * memory owned by the caller -- and returns.  Once work recorded behind the call on that stream has completed (an event), *host_word != 0
 * says that a bounded wait ran out in a launch enqueued before the call.  Nothing is reset: a run loop that sees a non-zero word stops and
 * calls hiertcn_persistent_health.  (hiertcn_amd/run_hier_xing.py polls it where the reference writes results.txt, every 20 batches.)
 * *host_word is left untouched on a handle without persistent launches (single-level models). */
int hiertcn_persistent_health_async(hiertcn_handle_t h, uint64_t* host_word, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* HIERTCN_H_ */
