// which formulation of d Wo = hidden^T . G does rocBLAS run faster at the training step's real sizes?
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdio.h>
int main() {
  rocblas_handle h; rocblas_create_handle(&h);
  const int R = 614400, C = 256, NC = 3456;
  void *dH, *dG; float* dW;
  hipMalloc(&dH, (size_t)R * C * 2); hipMalloc(&dG, (size_t)R * NC * 2); hipMalloc(&dW, (size_t)NC * C * 4);
  hipMemset(dH, 0x3c, (size_t)R * C * 2); hipMemset(dG, 0x3c, (size_t)R * NC * 2);
  float one = 1.f, zero = 0.f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int form = 0; form < 2; form++)
    for (int it = 0; it < 3; it++) {
      hipEventRecord(e0);
      rocblas_status s;
      if (form == 0)   // col-major [C x NC] = H_cm[C x R] . G_cm^T
        s = rocblas_gemm_ex(h, rocblas_operation_none, rocblas_operation_transpose, C, NC, R, &one, dH, rocblas_datatype_bf16_r, C, dG,
                            rocblas_datatype_bf16_r, NC, &zero, dW, rocblas_datatype_f32_r, C, dW, rocblas_datatype_f32_r, C,
                            rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
      else             // col-major [NC x C] = G_cm[NC x R] . H_cm^T
        s = rocblas_gemm_ex(h, rocblas_operation_none, rocblas_operation_transpose, NC, C, R, &one, dG, rocblas_datatype_bf16_r, NC, dH,
                            rocblas_datatype_bf16_r, C, &zero, dW, rocblas_datatype_f32_r, NC, dW, rocblas_datatype_f32_r, NC,
                            rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("form %d: status %d, %.3f ms, %.1f TF/s\n", form, (int)s, ms, 2.0 * R * C * NC / ms / 1e9);
    }
  // d hidden: [C x R] += outT_cm[C x NC] . G_cm[NC x R]
  float* dDH; hipMalloc(&dDH, (size_t)R * C * 4); void* dWt; hipMalloc(&dWt, (size_t)NC * C * 2);
  for (int it = 0; it < 3; it++) {
    hipEventRecord(e0);
    rocblas_status s = rocblas_gemm_ex(h, rocblas_operation_none, rocblas_operation_none, C, R, NC, &one, dWt, rocblas_datatype_bf16_r, C, dG,
                                       rocblas_datatype_bf16_r, NC, &one, dDH, rocblas_datatype_f32_r, C, dDH, rocblas_datatype_f32_r, C,
                                       rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dH: status %d, %.3f ms, %.1f TF/s\n", (int)s, ms, 2.0 * R * C * NC / ms / 1e9);
  }
  return 0;
}
