// The exact (fp32 end-to-end) numerics mode: fp32 copies of the tensors the production path stores in 16 bits, and its launchers (exact.hip).
#pragma once
#include "common.h"

enum { X_LOW = 1, X_FP32 = 2, X_SPLIT = 4 };   // fp32 copies of the low level | fp32 scoring table + logits scratch | split 16-bit scoring table

struct ExactDev {
  int enabled;          // X_* bits: what is allocated (hiertcn_set_precision before hiertcn_set_weights)
  int filled;           // ... and filled by hiertcn_set_weights
  float* emb;           // [N][E], row 0 zeroed
  float* projT;         // [E out][E in] = transpose(Wp[:E])                  (hier only)
  float* convT[HIERTCN_MAX_BLOCKS];   // [K][Cout][Cin]
  float* resT[HIERTCN_MAX_BLOCKS];    // [Cout][Cin] or null
  float* outT;          // [N][Clast]                                                                                   (X_FP32)
  uint16_t* outS;       // [NP][2 Clast] IEEE half: row n = [hi(Wo[:, n]) | lo(Wo[:, n])], rows >= N zero (score_x3.hip)          (X_SPLIT)
  float* outS_scale;    // 4 words: [0] 2^kw, the power of two outS was scaled by; [1] 2^-kw; [2] max |Wo| (bits)                     (X_SPLIT)
  float* logits;        // [chunk_rows][N] scratch of the scoring layer (owned by the handle: hiertcn_score_fused takes no workspace)
  int64_t chunk_rows;
};

enum { XEPI_LINEAR = 0, XEPI_BLOCK = 1 };

struct XGemm {
  const float* A; int64_t lda;        // activations [M][lda], or (ids != null) an item table gathered by ids[m]
  const int32_t* ids; int n_tab;      // gather: ids outside (0, n_tab) are zero rows
  const float* Wt;                    // [taps][N][Cin]  (output-major)
  int taps, dil, L;                   // causal taps inside sequences of L rows (L == 0: plain rows)
  int Cin;                            // multiple of 32
  int64_t M; int N;
  const float* bias;                  // [N] or null
  const float* seq_init; int Lseq;    // [M / Lseq][N] or null: added per sequence (the state half of the input projection)
  const float* res; int64_t ldres;    // XEPI_BLOCK: residual rows
  float* out; int64_t ldo;
  int epi;
  float* out2;                        // XEPI_BLOCK, optional: relu(conv + bias) before the residual is added (ldo): the fp32 training step keeps it
};

hipError_t launch_xgemm(const XGemm& g, hipStream_t st);
hipError_t launch_xsession_pool(const ModelDev& m, const ExactDev& x, const int32_t* y, int64_t n_sessions, int L, float* pooled, hipStream_t st);
size_t exact_tcn_workspace_bytes(const ModelDev& m, int64_t n_sessions, int L);
hipError_t launch_exact_tcn(const ModelDev& m, const ExactDev& x, const int32_t* ids, const float* sproj, int64_t n_sessions, int L, float* hidden,
                            void* workspace, hipStream_t st);
int64_t exact_chunk_rows(int N);
// pieces of the fp32 path the fp32 TRAINING step composes itself (train.hip): embedded rows; rank / ce / target of materialised logits (row stride ld)
hipError_t launch_xembed_rows(const float* emb, const int32_t* ids, int64_t R, int E, int N, float* out, hipStream_t st);
hipError_t launch_xrow_stats(const float* logits, int64_t ld, const int32_t* y, int64_t R, int N, int32_t* o_rank, float* o_ce, float* o_tgt, hipStream_t st);
hipError_t launch_exact_logits(const ModelDev& m, const ExactDev& x, const float* hidden, int64_t R, float* logits, hipStream_t st);
// score_x3.hip: the fused scoring kernel on split 16-bit operands (HIERTCN_PRECISION_SPLIT16); hidden fp32 [R][Clast]
bool score_x3_supported(const ModelDev& m);
int score_x3_rows_per_wg(const ModelDev& m);
hipError_t launch_score_x3w(const ModelDev& m, const ExactDev& x, const float* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                            float* logits, hipStream_t st, const int32_t* row_map, const int32_t* n_rows_dev, float* split_scratch);   // score_x3w.hip: last width 512
hipError_t launch_transpose_split(const float* src, uint16_t* dst, int C, int64_t N, int64_t NP, float* scale, hipStream_t st);
hipError_t launch_score_x3(const ModelDev& m, const ExactDev& x, const float* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                           float* logits, hipStream_t st, const int32_t* row_map, const int32_t* n_rows_dev, float* split_scratch);
hipError_t launch_exact_score(const ModelDev& m, const ExactDev& x, const float* hidden, const int32_t* y, int64_t R, const hiertcn_score_out_t* out,
                              hipStream_t st);
