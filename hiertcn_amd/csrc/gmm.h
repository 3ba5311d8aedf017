// The mixture-output model's handle (gmm.hip: create / set_weights / forward / loss; gmm_train.hip: the training step).
#pragma once
#include <vector>

#include "common.h"

int gfail(int code, const char* fmt, ...);   // sets hiertcn_gmm_last_error's thread-local message, returns `code`
#define G_HIP(expr)                                                                                      \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return gfail(HIERTCN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

struct GmmTrain;   // gmm_train.hip: fp32 masters / gradients / Adam slots of a handle that trains (null until hiertcn_gmm_train_enable)
void gmm_train_destroy(GmmTrain* t);

struct hiertcn_gmm_ctx {
  hiertcn_gmm_config_t cfg;
  int device;
  int N, NP;        // trunk input width (item_num)
  int D, DP;        // output_dim, padded to 64
  int M, C0, Clast, K, nb;
  bool weights_set;
  std::vector<void*> allocs;
  uint16_t* gather;                      // [K + 1][N][C0]: taps 0..K-1 of block 0's conv1, then the Dense residual kernel
  float* b0;                             // [C0] conv1 bias of block 0
  float* br0;                            // [C0] residual bias of block 0
  uint16_t* convT[HIERTCN_MAX_BLOCKS];   // blocks 1..: [K][Cout][Cin]
  float* conv_bias[HIERTCN_MAX_BLOCKS];
  uint16_t* resT[HIERTCN_MAX_BLOCKS];
  float* res_bias[HIERTCN_MAX_BLOCKS];
  uint16_t* muT;                         // [M][DP][Clast]: item-major tables of the M mean blocks of the head, pad rows zero
  float* mu_bias;                        // [M][DP], pad entries ZERO (a pad output must add 0 to |mu|^2)
  float* pi_w;                           // [M][Clast] fp32
  float* pi_b;                           // [M]
  // learned scale (cfg.sigma == 0; model_tcn_gmm.py:39-40: n_out = M (1 + 2 D)): tables M .. 2M-1 of muT / mu_bias hold the log-scale blocks
  bool learned;
  int NPs;                               // rows of an interleaved table: 64 per stage of 32 outputs
  uint16_t* sgT;                         // [M][NPs][Clast]: per stage the 32 mean rows, then the 32 log-scale rows of the same outputs
  float* sg_bias;                        // [M][NPs]
  float* svec;                           // [M][Clast]: sum_d of the log-scale table's rows (sum_d s_d is linear in the hidden row)
  float* s0;                             // [M]: sum_d of the log-scale biases
  // Gram form of |mu_m|^2 (gmm_gram.hip): used by hiertcn_gmm_loss when it applies and HIERTCN_GMM_TABLE_PASS is not set
  bool gram;
  double* gram64;                        // [M][C + 1][C + 1] (scratch of set_weights)
  unsigned char* G3;                     // per (m, 32-row tile): the LDS image of k_gmm_quad (hi | mid | lo parts of G_m[:C][:C])
  float* gx;                             // [M][C]: 2 G_m[c][C]
  float* g0;                             // [M]: G_m[C][C]
  // trunk 1 (model_gru_gmm): GRU stack on one-hot rows
  int trunk, H, nl;
  float* gin_g;                          // [N][2H]: input half of layer 0's gate kernel (a one-hot row times it = its row)
  float* gin_c;                          // [N][H]:  input half of layer 0's candidate kernel
  float* gwg[HIERTCN_MAX_LAYERS];        // output-major: layer 0 [2H][H] (state half), layers >= 1 [2H][2H]
  float* gbg[HIERTCN_MAX_LAYERS];
  float* gwc[HIERTCN_MAX_LAYERS];        // layer 0 [H][H], layers >= 1 [H][2H]
  float* gbc[HIERTCN_MAX_LAYERS];
  GmmTrain* train = nullptr;             // hiertcn_gmm_train_enable
};

template <typename T>
static hipError_t galloc(hiertcn_gmm_ctx* c, T** out, size_t count) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, (count ? count : 1) * sizeof(T));
  if (e != hipSuccess) return e;
  c->allocs.push_back(p);
  *out = reinterpret_cast<T*>(p);
  return hipSuccess;
}

