// Training state of one handle: fp32 master weights, gradients and the two Adam
// moments of every variable, kept in the INTERNAL layouts of ModelDev (Adam is elementwise, so the layout
// is free; exporting to the reference's TF layouts is a transpose per tensor).
#pragma once
#include "common.h"

struct TrainSeg {
  size_t off = 0, n = 0;  // in floats, inside each of the four arenas
};

struct TrainCtx {
  bool enabled = false;
  float* arena = nullptr;  // [4][total]: master | grad | m | v
  size_t total = 0;
  TrainSeg emb, emb_bias, projT, projS, outT, out_bias;
  TrainSeg convT[HIERTCN_MAX_BLOCKS], conv_bias[HIERTCN_MAX_BLOCKS], resT[HIERTCN_MAX_BLOCKS], res_bias[HIERTCN_MAX_BLOCKS];
  TrainSeg wg[HIERTCN_MAX_LAYERS], bg[HIERTCN_MAX_LAYERS], wc[HIERTCN_MAX_LAYERS], bc[HIERTCN_MAX_LAYERS];
  TrainSeg lwg[HIERTCN_MAX_LAYERS], lbg[HIERTCN_MAX_LAYERS], lwc[HIERTCN_MAX_LAYERS], lbc[HIERTCN_MAX_LAYERS];   // HierGRU low level
  uint16_t* dlay = nullptr;   // one allocation behind projD / convD / resD: the 16-bit kernels with the OUTPUT channel contiguous
  uint16_t* projD = nullptr;  // [Ein][Eout]
  uint16_t* convD[HIERTCN_MAX_BLOCKS] = {};   // [tap][Cin][Cout]
  uint16_t* resD[HIERTCN_MAX_BLOCKS] = {};    // [Cin][Cout]
  int64_t step = 0;      // Adam time step t
  float* master() const { return arena; }
  float* grad() const { return arena + total; }
  float* mom1() const { return arena + 2 * total; }
  float* mom2() const { return arena + 3 * total; }
};

const char* train_last_error();
// lays out the segments (no allocation); returns total floats per arena
size_t train_layout(const ModelDev& m, TrainCtx& t);
// allocates the arenas and points ModelDev's fp32 tensors INTO the master arena
hipError_t train_enable(ModelDev& m, TrainCtx& t);
void train_destroy(TrainCtx& t);
// fp32 masters of the bf16-stored tensors, from the reference-layout sources (called by hiertcn_set_weights)
hipError_t train_fill_masters(const ModelDev& m, TrainCtx& t, const hiertcn_weights_t* w, hipStream_t st);
// bf16 device copies (item table, projection, conv kernels, scoring kernel) re-derived from the masters
hipError_t train_refresh_bf16(const ModelDev& m, const TrainCtx& t, hipStream_t st);
size_t train_workspace_bytes(const ModelDev& m, int B, int S, int L);
// forward (training mode: activations kept) + loss + backward of every variable into t.grad(); returns 0 or -1 (train_last_error)
#define TRAIN_BEGIN 1    // forward + loss + backward of the scoring layer (its gradients are final behind it)
#define TRAIN_FINISH 2   // the rest of the backward pass
int train_forward_backward(const ModelDev& m, TrainCtx& t, const int32_t* x, const int32_t* y, const float* mask, const float* state_in, int B,
                           int S, int L, void* ws, float* state_out, double* metrics8, double* metric_scratch, hipStream_t st,
                           int phases = TRAIN_BEGIN | TRAIN_FINISH, const float* users_global = nullptr, const int32_t* y_score = nullptr);
// y_score (optional, hiertcn_set_score_ids): the ids the LOSS sees -- y with the steps that do not count set to 0 (model.py:102-103, mask_y *= mask_warmstart) --
// while the session mean-pool (model_hier.py:83-85) and its backward keep reading y
// data-parallel helpers: lanes with a target -> users[0]; the item table's row-sparse gradient exchange
hipError_t train_count_users(const int32_t* y, int B, int T, float* act_scratch, float* users, hipStream_t st);
size_t train_rows_workspace_bytes(const ModelDev& m);
hipError_t train_rows_list(const ModelDev& m, const int32_t* x, const int32_t* y, int64_t n_ids, int32_t* ids_out, int32_t* count_out, void* ws,
                           hipStream_t st);
// op: 0 rows_out[i] = grad_emb[ids[i]]; 1 grad_emb[ids[i]] = 0; 2 grad_emb[ids[i]] += rows[i]   (i < count[0], count <= cap)
hipError_t train_rows_op(const ModelDev& m, const TrainCtx& t, int op, const int32_t* ids, const int32_t* count, int64_t cap, float* rows,
                         hipStream_t st);
// TF-1.x AdamOptimizer (model.py:139): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v updates; theta -= lr_t*m/(sqrt(v)+eps)
hipError_t train_adam(const ModelDev& m, TrainCtx& t, float lr, float beta1, float beta2, float eps, float grad_scale, hipStream_t st);
// which = 0: masters, 1: gradients -> fp32 tensors in the reference layouts (device destinations)
hipError_t train_export(const ModelDev& m, const TrainCtx& t, const hiertcn_weights_t* dst, int which, hipStream_t st);

// pieces of the fp32 training step shared with the mixture model's (gmm_train.hip): TemporalBlock pointwise backward (ds = da [out > 0], dc = ds [c > 0]),
// row shifts inside sequences of L rows, column sums (ACCUMULATED into `out`: zero it first), lane activity + user count (model.py:112-113),
// TF-form Adam on flat arenas at time step `step` (>= 1)
hipError_t train_block_bwd_f32(const float* da, const float* aout, const float* cact, int64_t n, float* ds, float* dc, hipStream_t st);
hipError_t train_shift_rows_f32(const float* src, int64_t R, int C, int L, int shift, float* dst, hipStream_t st);
hipError_t train_colsum_f32(const float* src, int64_t rows, int64_t ld, int cols, float* out, hipStream_t st);
hipError_t train_lane_act(const int32_t* y, int B, int T, float* act, float* users, hipStream_t st);
hipError_t train_adam_flat(float* p, const float* g, float* m1, float* m2, size_t n, float lr, float beta1, float beta2, float eps, float grad_scale,
                           int64_t step, hipStream_t st);
// pointwise pieces of a GRU cell's backward under dynamic_rnn's sequence_length (model_gru.py:47-53; customed_gru_cell.py:323-337 differentiated):
//   a: d = active ? dcar + dout : 0;  dac = d (1-u) (1-c^2);  dag[:, H:] = d (h - c) u (1-u);  dhpart = d u
//   b: drh = dcat_c[:, in:];  dag[:, :H] = drh h r (1-r);  dhpart += drh r
//   c: d input = dcat_c[:, :in] + dcat_g[:, :in];  d old state = active ? dhpart + dcat_g[:, in:] : dcar
hipError_t train_lgru_bwd_a(const float* dcar, int ldd, const float* dout, int64_t ld_dout, const int32_t* seq_len, int tstep, const float* h, int ldh,
                            const float* gates, const float* cand, int M, int H, float* dac, float* dag, float* dhpart, hipStream_t st);
hipError_t train_gru_bwd_b(const float* dcat_c, int K, int in, const float* h, int ldh, const float* gates, int M, int H, float* dag, float* dhpart,
                           hipStream_t st);
hipError_t train_lgru_bwd_c(const float* dcat_c, const float* dcat_g, int K, int in, const float* dhpart, int M, int H, float* dst_in, int64_t ldi, float* dst_h,
                            int ldo, const float* dcar, const int32_t* seq_len, int tstep, hipStream_t st);

// Backward through time of a GRU stack under dynamic_rnn's sequence_length as a (step, layer) WAVEFRONT (train.hip): cell (t, l) needs (t + 1, l) [the carried
// state gradient] and (t, l + 1) [its output's gradient] only, so the cells with t + l = const share their launches, two cells each -- one pointwise step and the
// two skinny fp32 products whose epilogues carry the other two pointwise steps (gemm32_nn_gru_bwd_n): 3 (L + nl - 1) dependent launches instead of 5 L nl.
// M <= gemm32_skinny_max_rows().  Every cell's pre-activation gradients are kept (dac_all [nl][L][M][H], dag_all [nl][L][M][2H]) for the weight gradients behind.
struct LgruBptt {
  int nl, L, M, H;
  int in0;                                   // layer 0's dense input width (0: a one-hot input -- no dense input half)
  const float* wc[HIERTCN_MAX_LAYERS];       // [H][in_l + H]  (output-major candidate kernels)
  const float* wg[HIERTCN_MAX_LAYERS];       // [2H][in_l + H]
  const int32_t* seq_len;
  const float* states;                       // [L + 1][M][H nl]: the state before step t
  const float* recs;                         // [L][nl][gates 2H | r*h H | cand H | out H per row]
  const float* d_out; int64_t ld_d_out;      // the top layer's output gradient: row m of step t at d_out + t * H + m * ld_d_out
  float* d_in0; int64_t ld_d_in0;            // gradient of layer 0's dense input rows (row m of step t at d_in0 + t * in0 + m * ld_d_in0); unused when in0 == 0
  float* dac_all; float* dag_all;
  float* car[2];                             // [M][H nl] each: the carried state gradient by the step's parity (both are overwritten)
  float* dhpart; float* din_up;              // [nl][M][H] each
  float* dcat_c; size_t dcat_pitch;          // [nl][dcat_pitch] floats, dcat_pitch >= M * max_l (in_l + H)
};
hipError_t train_lgru_bptt_wavefront(const LgruBptt& b, hipStream_t st);
