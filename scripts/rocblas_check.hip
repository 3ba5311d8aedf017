// Does rocblas_gemm_ex (bf16 in, f32 out/accumulate) do what the training backward needs on gfx950?
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <cmath>
#include <string.h>
static uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fff + ((u >> 16) & 1); return u >> 16; }
static float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  rocblas_handle h; rocblas_create_handle(&h);
  // case 1: dW[NC][C] (row-major) = G^T[NC x R] H[R x C]  => gemm(N,T,m=C,n=NC,k=R,A=H(lda=C),B=G(ldb=NC))
  const int R = 200000, C = 256, NC = 512;
  std::vector<uint16_t> H((size_t)R * C), G((size_t)R * NC);
  srand(1);
  for (auto& v : H) v = f2bf((rand() % 2001 - 1000) / 1000.f);
  for (auto& v : G) v = f2bf((rand() % 2001 - 1000) / 100000.f);
  uint16_t *dH, *dG; float* dW;
  hipMalloc(&dH, H.size() * 2); hipMalloc(&dG, G.size() * 2); hipMalloc(&dW, (size_t)NC * C * 4);
  hipMemcpy(dH, H.data(), H.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dG, G.data(), G.size() * 2, hipMemcpyHostToDevice);
  float one = 1.f, zero = 0.f;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int it = 0; it < 2; it++) {
    hipEventRecord(e0);
    rocblas_status s = rocblas_gemm_ex(h, rocblas_operation_none, rocblas_operation_transpose, C, NC, R, &one, dH, rocblas_datatype_bf16_r, C, dG,
                                       rocblas_datatype_bf16_r, NC, &zero, dW, rocblas_datatype_f32_r, C, dW, rocblas_datatype_f32_r, C,
                                       rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dW gemm status %d, %.3f ms, %.1f TF/s\n", (int)s, ms, 2.0 * R * C * NC / ms / 1e9);
  }
  std::vector<float> W((size_t)NC * C);
  hipMemcpy(W.data(), dW, W.size() * 4, hipMemcpyDeviceToHost);
  double maxerr = 0;
  for (int n = 0; n < 3; n++) for (int c = 0; c < 3; c++) {
    double ref = 0; for (int r = 0; r < R; r++) ref += (double)bf2f(G[(size_t)r * NC + n]) * bf2f(H[(size_t)r * C + c]);
    maxerr = fmax(maxerr, fabs(ref - W[(size_t)n * C + c]));
    printf("  dW[%d][%d] = %.6f ref %.6f\n", n, c, W[(size_t)n * C + c], ref);
  }
  // case 2: dH[R][C] (row-major) += G[R x NC] outT[NC x C] => gemm(N,N,m=C,n=R,k=NC,A=outT(lda=C),B=G(ldb=NC)), beta=1
  std::vector<uint16_t> Wt((size_t)NC * C); for (auto& v : Wt) v = f2bf((rand() % 2001 - 1000) / 1000.f);
  uint16_t* dWt; float* dDH; hipMalloc(&dWt, Wt.size() * 2); hipMalloc(&dDH, (size_t)R * C * 4);
  hipMemcpy(dWt, Wt.data(), Wt.size() * 2, hipMemcpyHostToDevice); hipMemset(dDH, 0, (size_t)R * C * 4);
  for (int it = 0; it < 2; it++) {
    hipEventRecord(e0);
    rocblas_status s = rocblas_gemm_ex(h, rocblas_operation_none, rocblas_operation_none, C, R, NC, &one, dWt, rocblas_datatype_bf16_r, C, dG,
                                       rocblas_datatype_bf16_r, NC, &one, dDH, rocblas_datatype_f32_r, C, dDH, rocblas_datatype_f32_r, C,
                                       rocblas_datatype_f32_r, rocblas_gemm_algo_standard, 0, 0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("dH gemm status %d, %.3f ms, %.1f TF/s\n", (int)s, ms, 2.0 * R * C * NC / ms / 1e9);
  }
  std::vector<float> DH(4 * C); hipMemcpy(DH.data(), dDH, DH.size() * 4, hipMemcpyDeviceToHost);
  for (int r = 0; r < 2; r++) for (int c = 0; c < 2; c++) {
    double ref = 0; for (int n = 0; n < NC; n++) ref += (double)bf2f(G[(size_t)r * NC + n]) * bf2f(Wt[(size_t)n * C + c]);
    printf("  dH[%d][%d] = %.6f ref(x2 accumulations) %.6f\n", r, c, DH[r * C + c], 2 * ref);
  }
  // case 3: f32 gemm small (GRU): out[B][K] = dac[B][H] W[H][K] => gemm(N,N,m=K,n=B,k=H)
  const int B = 2048, Hd = 256, K = 512;
  std::vector<float> A((size_t)B * Hd, 0.5f), Wg((size_t)Hd * K, 0.25f), O((size_t)B * K);
  float *dA, *dWg, *dO; hipMalloc(&dA, A.size() * 4); hipMalloc(&dWg, Wg.size() * 4); hipMalloc(&dO, O.size() * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dWg, Wg.data(), Wg.size() * 4, hipMemcpyHostToDevice);
  for (int it = 0; it < 3; it++) {
    hipEventRecord(e0);
    rocblas_status s = rocblas_sgemm(h, rocblas_operation_none, rocblas_operation_none, K, B, Hd, &one, dWg, K, dA, Hd, &zero, dO, K);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("sgemm status %d, %.3f ms\n", (int)s, ms);
  }
  hipMemcpy(O.data(), dO, O.size() * 4, hipMemcpyDeviceToHost);
  printf("  sgemm out[0] = %f (expect %f)\n", O[0], 0.5 * 0.25 * Hd);
  printf("maxerr %g\n", maxerr);
  return 0;
}
