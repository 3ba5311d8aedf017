// Lane (user) aggregation of the per-step scoring outputs: model.py:111-117 (loss) and
// loss.py:189-219 (recall@1/5/10, reciprocal rank, rank fraction): per-lane mean over the lane's
// valid steps (denominator activity_count + 1e-6, model.py:114), summed over lanes.  The division
// by user_count (model.py:117, loss.py:215-219) is left to the caller so that partial sums from
// several GPUs can be all-reduced first.
//
// Deterministic: every lane's sum is accumulated in step order by one thread; lanes are combined
// by a fixed-shape tree.  HBM-bound: 12 bytes per step.
#include "common.h"

#define MT_THREADS 256

// One wave per lane (user): the wave's 64 threads stride over the lane's T steps (coalesced 12 B per step), per-thread
// partial sums are combined by a fixed shuffle tree, the 4 lanes of a workgroup by a fixed-order loop.
__global__ __launch_bounds__(MT_THREADS) void k_lane_metrics(const int32_t* __restrict__ rank, const float* __restrict__ ce,
                                                             const int32_t* __restrict__ y, int B, int T, int N,
                                                             double* __restrict__ partial /*[gridDim.x][8]*/) {
  __shared__ double red[MT_THREADS / 64][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * (MT_THREADS / 64) + wave;
  float loss = 0.f, r1 = 0.f, r5 = 0.f, r10 = 0.f, rr = 0.f, rf = 0.f;
  int act = 0;
  if (b < B) {
    const int64_t base = (int64_t)b * T;
    const float invN = 1.0f / (float)N;
    for (int t = lane; t < T; t += 64) {
      if (y[base + t] > 0) {
        const float rk = (float)rank[base + t];
        act++;
        loss += ce[base + t];
        r1 += (rk <= 0.f) ? 1.f : 0.f;
        r5 += (rk <= 4.f) ? 1.f : 0.f;
        r10 += (rk <= 9.f) ? 1.f : 0.f;
        rr += 1.0f / (1.0f + rk);
        rf += rk * invN;
      }
    }
  }
  for (int off = 32; off > 0; off >>= 1) {
    loss += __shfl_down(loss, off); r1 += __shfl_down(r1, off); r5 += __shfl_down(r5, off); r10 += __shfl_down(r10, off);
    rr += __shfl_down(rr, off); rf += __shfl_down(rf, off); act += __shfl_down(act, off);
  }
  if (lane == 0) {
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (act > 0) {
      const float den = (float)act + 1e-6f;  // model.py:114 (fp32 in the reference graph)
      v[0] = loss / den; v[1] = r1 / den; v[2] = r5 / den; v[3] = r10 / den; v[4] = rr / den; v[5] = rf / den;
      v[6] = 1.0;
      v[7] = (double)act;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) red[wave][k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    double s = 0.0;
    for (int w = 0; w < MT_THREADS / 64; w++) s += red[w][threadIdx.x];
    partial[(int64_t)blockIdx.x * 8 + threadIdx.x] = s;
  }
}

// 8 columns x 32 threads: strided partial sums, then a fixed shuffle tree per column
__global__ __launch_bounds__(256) void k_metrics_final(const double* __restrict__ partial, int nblocks, double* __restrict__ out8) {
  const int col = threadIdx.x >> 5, j = threadIdx.x & 31;
  double s = 0.0;
  for (int i = j; i < nblocks; i += 32) s += partial[(int64_t)i * 8 + col];
  for (int off = 16; off > 0; off >>= 1) s += __shfl_down(s, off, 32);
  if (j == 0) out8[col] = s;
}

// `partial` scratch lives behind out8: the caller passes a buffer of 8 * (1 + ceil(B/256)) doubles
// through the handle's metric scratch (see api.hip).
// Small batches (the reference's 32 lanes): ONE workgroup of 16 waves does both steps -- wave w takes lanes w, w + 16, ... in order and
// adds their quotients up in fp64, the sixteen partial rows are combined in a fixed order.  Same arithmetic per lane as k_lane_metrics.
__global__ __launch_bounds__(1024) void k_metrics_small(const int32_t* __restrict__ rank, const float* __restrict__ ce, const int32_t* __restrict__ y,
                                                        int B, int T, int N, double* __restrict__ out8) {
  __shared__ double red[16][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const float invN = 1.0f / (float)N;
  for (int b = wave; b < B; b += 16) {
    float loss = 0.f, r1 = 0.f, r5 = 0.f, r10 = 0.f, rr = 0.f, rf = 0.f;
    int act = 0;
    const int64_t base = (int64_t)b * T;
    for (int t = lane; t < T; t += 64) {
      if (y[base + t] > 0) {
        const float rk = (float)rank[base + t];
        act++;
        loss += ce[base + t];
        r1 += (rk <= 0.f) ? 1.f : 0.f;
        r5 += (rk <= 4.f) ? 1.f : 0.f;
        r10 += (rk <= 9.f) ? 1.f : 0.f;
        rr += 1.0f / (1.0f + rk);
        rf += rk * invN;
      }
    }
    for (int off = 32; off > 0; off >>= 1) {
      loss += __shfl_down(loss, off); r1 += __shfl_down(r1, off); r5 += __shfl_down(r5, off); r10 += __shfl_down(r10, off);
      rr += __shfl_down(rr, off); rf += __shfl_down(rf, off); act += __shfl_down(act, off);
    }
    if (lane == 0 && act > 0) {
      const float den = (float)act + 1e-6f;  // model.py:114 (fp32 in the reference graph)
      acc[0] += loss / den; acc[1] += r1 / den; acc[2] += r5 / den; acc[3] += r10 / den; acc[4] += rr / den; acc[5] += rf / den;
      acc[6] += 1.0;
      acc[7] += (double)act;
    }
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 8; k++) red[wave][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    double s_ = 0.0;
    for (int w = 0; w < 16; w++) s_ += red[w][threadIdx.x];
    out8[threadIdx.x] = s_;
  }
}

hipError_t launch_metrics_scratch(const ModelDev& m, const int32_t* rank, const float* ce, const int32_t* y, int B, int T,
                                  double* out8, double* scratch, hipStream_t st) {
  if (B <= 32) {   // (measured: at 128 lanes the two-launch form is faster again)
    hipLaunchKernelGGL(k_metrics_small, dim3(1), dim3(1024), 0, st, rank, ce, y, B, T, m.N, out8);
    return hipGetLastError();
  }
  const int nblocks = (B + MT_THREADS / 64 - 1) / (MT_THREADS / 64);   // 4 lanes per workgroup
  hipLaunchKernelGGL(k_lane_metrics, dim3(nblocks), dim3(MT_THREADS), 0, st, rank, ce, y, B, T, m.N, scratch);
  hipLaunchKernelGGL(k_metrics_final, dim3(1), dim3(256), 0, st, scratch, nblocks, out8);
  return hipGetLastError();
}


// ---------------------------------------------------------------------------------------------
// Row compaction for ragged batches: the ascending list of rows that carry a target (y > 0).
// Three tiny launches (per-1024-row counts, a single-workgroup scan, the write pass); order is
// deterministic, so is everything computed through the list.
// ---------------------------------------------------------------------------------------------
#define CP_ROWS 1024
__global__ __launch_bounds__(256) void k_valid_counts(const int32_t* __restrict__ y, int64_t R, int32_t* __restrict__ counts) {
  __shared__ int red[4];
  const int64_t r0 = (int64_t)blockIdx.x * CP_ROWS + threadIdx.x * 4;
  int c = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) c += (r0 + j < R && y[r0 + j] > 0) ? 1 : 0;
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = c;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// exclusive scan of counts[0..n) in place (one workgroup; n <= a few thousand), total -> n_valid
__global__ __launch_bounds__(1024) void k_scan_counts(int32_t* __restrict__ counts, int n, int32_t* __restrict__ n_valid) {
  __shared__ int part[1024];
  const int per = (n + 1023) / 1024;
  const int lo = threadIdx.x * per, hi = (lo + per < n) ? lo + per : n;
  int s = 0;
  for (int i = lo; i < hi; i++) s += counts[i];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {   // Hillis-Steele inclusive scan of the 1024 partials
    const int v = (threadIdx.x >= off) ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - s;   // exclusive prefix of this thread's slice
  for (int i = lo; i < hi; i++) {
    const int c = counts[i];
    counts[i] = run;
    run += c;
  }
  if (threadIdx.x == 1023) n_valid[0] = part[1023];
}
__global__ __launch_bounds__(256) void k_write_map(const int32_t* __restrict__ y, int64_t R, const int32_t* __restrict__ offsets,
                                                  int32_t* __restrict__ map, int32_t* __restrict__ z0, float* __restrict__ z1, float* __restrict__ z2) {
  __shared__ int wsum[4];
  const int64_t r0 = (int64_t)blockIdx.x * CP_ROWS + threadIdx.x * 4;
  int f[4], c = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) { f[j] = (r0 + j < R && y[r0 + j] > 0) ? 1 : 0; c += f[j]; }
#pragma unroll
  for (int j = 0; j < 4; j++) {   // per-row outputs of the scoring launch start from zero: the rows it skips (y == 0) must read 0 (model.py:105)
    if (r0 + j < R) {
      if (z0) z0[r0 + j] = 0;
      if (z1) z1[r0 + j] = 0.f;
      if (z2) z2[r0 + j] = 0.f;
    }
  }
  // exclusive scan of c over the 256 threads: inside each wave by shuffles, then across the 4 waves
  int incl = c;
  const int lane = threadIdx.x & 63;
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(incl, off);
    if (lane >= off) incl += v;
  }
  if (lane == 63) wsum[threadIdx.x >> 6] = incl;
  __syncthreads();
  int base = offsets[blockIdx.x];
  for (int w = 0; w < (int)(threadIdx.x >> 6); w++) base += wsum[w];
  int pos = base + incl - c;
#pragma unroll
  for (int j = 0; j < 4; j++)
    if (f[j]) map[pos++] = (int32_t)(r0 + j);
}

// up to 8,192 rows (a 32-lane XING batch has 6,400): count, scan and write in ONE workgroup, 1024 rows at a time (thread t takes row
// chunk * 1024 + t: coalesced; positions from a wave ballot + the sixteen wave totals + the running base)
__global__ __launch_bounds__(1024) void k_compact_small(const int32_t* __restrict__ y, int R, int32_t* __restrict__ map, int32_t* __restrict__ n_valid,
                                                        int32_t* __restrict__ z0, float* __restrict__ z1, float* __restrict__ z2) {
  __shared__ int wtot[16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int base = 0;
  for (int r0 = 0; r0 < R; r0 += 1024) {
    const int r = r0 + threadIdx.x;
    const bool f = r < R && y[r] > 0;
    if (r < R) {
      if (z0) z0[r] = 0;
      if (z1) z1[r] = 0.f;
      if (z2) z2[r] = 0.f;
    }
    const unsigned long long bal = __ballot(f);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wave] = __popcll(bal);
    __syncthreads();
    int off = base, tot = 0;
    for (int w = 0; w < 16; w++) {
      const int c = wtot[w];
      off += (w < wave) ? c : 0;
      tot += c;
    }
    if (f) map[off + before] = r;
    base += tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) n_valid[0] = base;
}

// z0 / z1 / z2 (optional): per-row int32 / fp32 / fp32 arrays of R entries that the same pass sets to zero (the scoring outputs)
hipError_t launch_compact_rows(const int32_t* y, int64_t R, int32_t* map, int32_t* n_valid, int32_t* scratch, hipStream_t st, int32_t* z0, float* z1,
                               float* z2) {
  if (R <= 8192) {
    hipLaunchKernelGGL(k_compact_small, dim3(1), dim3(1024), 0, st, y, (int)R, map, n_valid, z0, z1, z2);
    return hipGetLastError();
  }
  const int nblk = (int)((R + CP_ROWS - 1) / CP_ROWS);
  hipLaunchKernelGGL(k_valid_counts, dim3(nblk), dim3(256), 0, st, y, R, scratch);
  hipLaunchKernelGGL(k_scan_counts, dim3(1), dim3(1024), 0, st, scratch, nblk, n_valid);
  hipLaunchKernelGGL(k_write_map, dim3(nblk), dim3(256), 0, st, y, R, scratch, map, z0, z1, z2);
  return hipGetLastError();
}

// ranked lists of masked rows: indices 0..k-1, values 0 (the scoring kernel then overwrites the rows that carry a target)
__global__ void k_topk_fill(int32_t* __restrict__ idx, float* __restrict__ val, int64_t n, int k) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  idx[i] = (int32_t)(i % k);
  val[i] = 0.f;
}
hipError_t launch_topk_fill(int32_t* idx, float* val, int64_t R, int k, hipStream_t st) {
  const int64_t n = R * k;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(k_topk_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, idx, val, n, k);
  return hipGetLastError();
}
