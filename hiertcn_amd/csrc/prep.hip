// Weight conversion kernels: reference (TF) fp32 layouts -> internal gfx950 layouts.
// Run once per hiertcn_set_weights; HBM-bound streaming, 16-byte accesses where the layout allows.
#include "common.h"

// [rows][cols] fp32 -> [rows][cols] bf16, optionally zeroing row 0 (the null id: model.py:59
// multiplies the one-hot by sign(id), so E[0] is never read -- store it as zeros so gathers
// need no branch).
__global__ void k_rows_to_bf16(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t n, int cols, int zero_row0) {
  int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x * 4;
  for (; i < n; i += stride) {
    float v[4];
#pragma unroll
    for (int j = 0; j < 4; j++) v[j] = (i + j < n) ? src[i + j] : 0.f;
    if (zero_row0 && i < cols) {
#pragma unroll
      for (int j = 0; j < 4; j++)
        if (i + j < cols) v[j] = 0.f;
    }
    if (i + 3 < n) {
      uint2 p;
      p.x = pack2bf(v[0], v[1]);
      p.y = pack2bf(v[2], v[3]);
      *reinterpret_cast<uint2*>(dst + i) = p;
    } else {
      for (int j = 0; j < 4 && i + j < n; j++) dst[i + j] = f2bf(v[j]);
    }
  }
}

hipError_t launch_rows_to_bf16(const float* src, uint16_t* dst, int64_t rows, int cols, int zero_row0, hipStream_t st) {
  int64_t n = rows * (int64_t)cols;
  int blocks = (int)((n / 4 + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_rows_to_bf16, dim3(blocks), dim3(256), 0, st, src, dst, n, cols, zero_row0);
  return hipGetLastError();
}

// src [rows][cols] fp32 -> dst [cols (padded to dst_rows)][rows] bf16, rows beyond `cols` zero.
// 32x32 tiles through LDS (+1 pad); reads coalesced along cols, writes coalesced along rows.
__global__ void k_transpose_to_bf16(const float* __restrict__ src, uint16_t* __restrict__ dst, int64_t rows, int64_t cols,
                                    int64_t dst_rows) {
  __shared__ float tile[32][33];
  src += (int64_t)blockIdx.z * rows * cols;     // blockIdx.z: one of several equally shaped, contiguous matrices (the taps of a conv kernel)
  dst += (int64_t)blockIdx.z * rows * cols;     // (batched launches have dst_rows == cols: no padding between the outputs)
  const int64_t c0 = (int64_t)blockIdx.x * 32;  // along src cols / dst rows
  const int64_t r0 = (int64_t)blockIdx.y * 32;  // along src rows / dst cols
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 8 rows per pass
  for (int i = ty; i < 32; i += 8) {
    int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int64_t c = c0 + i, r = r0 + tx;  // dst[c][r]
    if (c < dst_rows && r < rows) dst[c * rows + r] = f2bf(tile[tx][i]);
  }
}

hipError_t launch_transpose_to_bf16(const float* src, uint16_t* dst, int64_t rows, int64_t cols, int64_t dst_rows_padded,
                                    hipStream_t st, int batch) {
  if (batch > 1 && dst_rows_padded != cols) return hipErrorInvalidValue;
  dim3 grid((unsigned)((dst_rows_padded + 31) / 32), (unsigned)((rows + 31) / 32), (unsigned)(batch > 1 ? batch : 1));
  hipLaunchKernelGGL(k_transpose_to_bf16, grid, dim3(256), 0, st, src, dst, rows, cols, dst_rows_padded);
  return hipGetLastError();
}

__global__ void k_copy_pad_f32(const float* __restrict__ src, float* __restrict__ dst, int64_t n, int64_t npad, float padval) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < npad; i += stride) dst[i] = (i < n) ? src[i] : padval;
}

hipError_t launch_copy_pad_f32(const float* src, float* dst, int64_t n, int64_t npad, float padval, hipStream_t st) {
  int blocks = (int)((npad + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(k_copy_pad_f32, dim3(blocks), dim3(256), 0, st, src, dst, n, npad, padval);
  return hipGetLastError();
}

// src [rows][cols] fp32 -> dst [cols][rows] fp32 (GRU-side weights: [in][out] -> [out][in])
__global__ void k_transpose_f32(const float* __restrict__ src, float* __restrict__ dst, int64_t rows, int64_t cols) {
  __shared__ float tile[32][33];
  const int64_t c0 = (int64_t)blockIdx.x * 32, r0 = (int64_t)blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int i = ty; i < 32; i += 8) {
    int64_t r = r0 + i, c = c0 + tx;
    tile[i][tx] = (r < rows && c < cols) ? src[r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    int64_t c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[c * rows + r] = tile[tx][i];
  }
}

hipError_t launch_transpose_f32(const float* src, float* dst, int64_t rows, int64_t cols, hipStream_t st) {
  dim3 grid((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32));
  hipLaunchKernelGGL(k_transpose_f32, grid, dim3(256), 0, st, src, dst, rows, cols);
  return hipGetLastError();
}
