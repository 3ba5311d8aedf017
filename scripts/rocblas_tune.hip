// Are there faster rocBLAS solutions than the default for the two big products of the training backward?
//   d hidden [rows x C] = G [rows x N] . Wo [N x C]      -> gemm(N, N, m = C, n = rows, k = N, A = outT (lda C), B = G (ldb ldg))
//   d Wo     [N x C]   += G^T . hidden [rows x C]        -> gemm(N, T, m = C, n = N, k = rows, A = hidden (lda C), B = G (ldb ldg))
// with rows = 28,608 (one 16 GiB chunk), N = 300,032, C = 256, bf16 in / fp32 out.  Prints the default's time and every solution
// that beats it (rocblas_gemm_ex_get_solutions, beta API).
#define ROCBLAS_BETA_FEATURES_API
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdio.h>
#include <vector>
#include <algorithm>

static float time_it(rocblas_handle h, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const void* A, int lda, const void* B,
                     int ldb, float* C, int ldc, rocblas_gemm_algo algo, int sol, int reps) {
  float one = 1.f, zero = 0.f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  rocblas_status s = rocblas_gemm_ex(h, ta, tb, m, n, k, &one, A, rocblas_datatype_bf16_r, lda, B, rocblas_datatype_bf16_r, ldb, &zero, C,
                                     rocblas_datatype_f32_r, ldc, C, rocblas_datatype_f32_r, ldc, rocblas_datatype_f32_r, algo, sol, 0);
  if (s != rocblas_status_success) return -1.f;
  hipEventRecord(e0);
  for (int i = 0; i < reps; i++)
    rocblas_gemm_ex(h, ta, tb, m, n, k, &one, A, rocblas_datatype_bf16_r, lda, B, rocblas_datatype_bf16_r, ldb, &zero, C, rocblas_datatype_f32_r, ldc,
                    C, rocblas_datatype_f32_r, ldc, rocblas_datatype_f32_r, algo, sol, 0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms / reps;
}

static void tune(rocblas_handle h, const char* name, rocblas_operation ta, rocblas_operation tb, int m, int n, int k, const void* A, int lda,
                 const void* B, int ldb, float* C, int ldc) {
  float one = 1.f, zero = 0.f;
  const float base = time_it(h, ta, tb, m, n, k, A, lda, B, ldb, C, ldc, rocblas_gemm_algo_standard, 0, 3);
  printf("%s: default %.3f ms (%.0f TFLOP/s)\n", name, base, 2.0 * m * n * k / base / 1e9);
  rocblas_int cnt = 0;
  rocblas_gemm_ex_get_solutions(h, ta, tb, m, n, k, &one, A, rocblas_datatype_bf16_r, lda, B, rocblas_datatype_bf16_r, ldb, &zero, C,
                                rocblas_datatype_f32_r, ldc, C, rocblas_datatype_f32_r, ldc, rocblas_datatype_f32_r, rocblas_gemm_algo_solution_index,
                                0, nullptr, &cnt);
  std::vector<rocblas_int> sols(cnt);
  rocblas_gemm_ex_get_solutions(h, ta, tb, m, n, k, &one, A, rocblas_datatype_bf16_r, lda, B, rocblas_datatype_bf16_r, ldb, &zero, C,
                                rocblas_datatype_f32_r, ldc, C, rocblas_datatype_f32_r, ldc, rocblas_datatype_f32_r, rocblas_gemm_algo_solution_index,
                                0, sols.data(), &cnt);
  printf("  %d solutions\n", (int)cnt);
  std::vector<std::pair<float, int>> res;
  for (int i = 0; i < cnt; i++) {
    const float ms = time_it(h, ta, tb, m, n, k, A, lda, B, ldb, C, ldc, rocblas_gemm_algo_solution_index, sols[i], 1);
    if (ms > 0) res.push_back({ms, sols[i]});
  }
  std::sort(res.begin(), res.end());
  for (size_t i = 0; i < res.size() && i < 6; i++) {
    const float ms = time_it(h, ta, tb, m, n, k, A, lda, B, ldb, C, ldc, rocblas_gemm_algo_solution_index, res[i].second, 3);
    printf("  solution %d: %.3f ms (%.0f TFLOP/s)%s\n", res[i].second, ms, 2.0 * m * n * k / ms / 1e9, ms < base ? "  <-- faster" : "");
  }
}

int main() {
  rocblas_handle h; rocblas_create_handle(&h);
  const int rows = 28608, N = 300032, C = 256;
  uint16_t *G, *W, *H; float *dH, *dW;
  hipMalloc(&G, (size_t)rows * N * 2); hipMalloc(&W, (size_t)N * C * 2); hipMalloc(&H, (size_t)rows * C * 2);
  hipMalloc(&dH, (size_t)rows * C * 4); hipMalloc(&dW, (size_t)N * C * 4);
  hipMemset(G, 0x3c, (size_t)rows * N * 2); hipMemset(W, 0x3c, (size_t)N * C * 2); hipMemset(H, 0x3c, (size_t)rows * C * 2);
  tune(h, "d hidden", rocblas_operation_none, rocblas_operation_none, C, rows, N, W, C, G, N, dH, C);
  tune(h, "d Wo", rocblas_operation_none, rocblas_operation_transpose, C, N, rows, H, C, G, N, dW, C);
  return 0;
}
