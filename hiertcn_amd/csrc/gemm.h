// Plain GEMMs of the training backward pass (gemm.hip): row-major operands, fp32 results, everything asynchronous on `st`.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// fp32, bit-exact multiply-add chains on v_mfma_f32_32x32x2_f32
//   NN: C[M][N] (ldc) = beta C + A[M][K] (lda) . B[K][N] (ldb)
// epilogue operands of gemm32_nn_gru_bwd: the elementwise pieces of a GRU cell's backward (train.hip, k_gru_bwd_b / k_gru_bwd_c) folded into its two products
struct GruBwdEpi {
  int in, H, K;                                  // K = in + H
  const float* gates; const float* h; int ldh;   // mode 1: (r, u) [M][2H], the old state slice (row stride ldh)
  float* dag; float* dhpart;                     // mode 1: d gates[:, :H] written, d h (partial) updated; mode 2: dhpart read
  const float* dcat_c;                           // [M][K]: mode 1 writes its input half, mode 2 reads it
  float* dst_in; int ldi; float* dst_h; int ldo; // mode 2: d input, d old state
  // mode 2, optional (na_dac != null; in == H): the cell BELOW is next in the BPTT and its first elementwise step (k_gru_bwd_a) rides on this epilogue --
  // its output gradient is (carried + state-projection path) * mask (k_gru_out_grad) + the d input just formed, which is then not stored
  const float* na_dcar; const float* na_dsb; int64_t na_ld_dsb; const float* na_mask; int na_ldm; int na_ldc;
  const float* na_gates; const float* na_cand; const float* na_h; int na_ldh;
  float* na_dac; float* na_dag; float* na_dhpart;
};
hipError_t gemm32_nn_gru_bwd(int mode, const float* A, int64_t lda, const float* B, int64_t ldb, int M, int N, int K, const GruBwdEpi& ep, hipStream_t st);
struct GruBwdProd { const float* A; int64_t lda; const float* B; int64_t ldb; int M, N, K; GruBwdEpi ep; };
hipError_t gemm32_nn_gru_bwd_n(int mode, const GruBwdProd* pr, int n, hipStream_t st);   // n = 1 or 2 independent products as one launch
int gemm32_skinny_max_rows();
// scratch (optional): partial tiles of a reduction sliced over blockIdx.z where the output has few tiles and K is long -- the fp32 training mode's
// d hidden [R][C] = d logits [R][NP] . Wo^T [NP][C] is 50-100 tiles behind a reduction of 20,800 (gemm32_tn's slicing, summed in slice order)
hipError_t gemm32_nn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int M, int N, int K, float beta, hipStream_t st,
                     float* scratch = nullptr, size_t scratch_floats = 0);
//   TN: C[N1][N2] (ldc) = beta C + sum_{m < Mred} A[m][N1] (lda) . B[m][N2] (ldb); scratch (optional): partial tiles of the row slices
hipError_t gemm32_tn(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc, int N1, int N2, int Mred, float beta,
                     float* scratch, size_t scratch_floats, hipStream_t st);

//   TN over slabs: reduction row m of A (B) lies at (m / a_div) * a_ss + (m % a_div) * lda when a_div > 0 (else m * lda): ONE product over the rows of
//   many records that are a fixed distance apart -- the deferred weight gradients of a scan (train.hip)
// one product of gemm32_tn_slabs_batch: the arguments of gemm32_tn_slabs
struct Gemm32TnDesc {
  const float* A; int64_t lda; int a_div; int64_t a_ss; const float* B; int64_t ldb; int b_div; int64_t b_ss; float* C; int64_t ldc; int N1, N2, Mred; float beta;
};
bool gemm32_tn_slabs_batch_applies(const Gemm32TnDesc* d, int n);
hipError_t gemm32_tn_slabs_batch(const Gemm32TnDesc* d, int n, float* scratch, size_t scratch_floats, hipStream_t st);
hipError_t gemm32_tn_slabs(const float* A, int64_t lda, int a_div, int64_t a_ss, const float* B, int64_t ldb, int b_div, int64_t b_ss, float* C,
                           int64_t ldc, int N1, int N2, int Mred, float beta, float* scratch, size_t scratch_floats, hipStream_t st);

// 16-bit storage operands (the library's storage type), fp32 results; row shifts per tap inside sequences of L rows (L = 0: none)
//   NT: C[M][N] (ldc) = beta C + sum_taps A[r + shift_tap][K] (lda) . Bt[tap][N][K]^T (ldb)
hipError_t gemm16_nt(const uint16_t* A, int64_t lda, const uint16_t* Bt, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, int taps,
                     const int* shift, int L, float beta, hipStream_t st);
//   TN: C[tap][N1][N2] (tap stride c_tap) = beta C + sum_r A[r][N1]^T . B[r + shift_tap][N2]; scratch holds the row slices' partials
hipError_t gemm16_tn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t c_tap, int N1, int N2, int64_t R, int taps,
                     const int* shift, int L, float beta, float* scratch, size_t scratch_floats, hipStream_t st);
//   NN: C[M][N] (ldc) = beta C + A[M][K] (lda) . B[K][N] (ldb)
hipError_t gemm16_nn(const uint16_t* A, int64_t lda, const uint16_t* B, int64_t ldb, float* C, int64_t ldc, int64_t M, int N, int K, float beta,
                     hipStream_t st);
