// The across-session GRU scan of small batches as ONE persistent launch (model_hier.py:91-93 -> customed_gru_cell.py:323-337).
//
// launch_gru_scan (pool_gru.hip) runs the scan as a chain of 2 (S + nl - 1) dependent GEMM launches; at the reference's own batch size
// (32 lanes) each of them is a ~11 us launch for ~1 us of work and the chain is 60 % of the batch.  Here the whole scan is one kernel:
//
//  * lanes do not interact in the scan, so the batch is cut into GP_GROUPS = 8 lane groups and each group is scanned by its own set of
//    P = H / 8 workgroups (group = blockIdx % 8: the blocks the dispatcher is observed to deal to one XCD -- a speed hint only, nothing
//    below relies on it).  Workgroup t of a group OWNS hidden units [8 t, 8 t + 8) of every layer: the three weight rows of a unit (reset
//    gate, update gate, candidate: gru_wg rows j and H + j, gru_wc row j) are read ONCE into LDS and stay there for all S slots, so a cell
//    costs no weight traffic at all -- only the group's cell inputs ([in | h], 2 KB per lane at H = 256) are exchanged;
//  * thread (lane b of 4, unit u of 8, k-slice ks of 8): one eighth of the k range of one lane for the three columns of one unit; the
//    eight partial sums are folded with wavefront shuffles (xor 1, 2, 4), fixed order, fp32 throughout (the carried state is never
//    rounded); the cell inputs are staged once per workgroup in LDS;
//  * cells (s, l) with s + l = tau are independent (the same wavefront as launch_gru_scan); a diagonal is two phases: A = gates, r * h,
//    and the input half of the candidate; B = the (r * h) half, tanh, blend, masked next state;
//  * the exchange between the workgroups of a group (r * h, a layer's output, the next state: 4 bytes per lane and unit) goes through
//    global memory as `sc1` stores and `sc1` loads around ONE agent-scope counter per group (MI355X_MICROARCH.md, inter-workgroup
//    visibility: one lane per workgroup adds after every storing wave's vmcnt(0) and a workgroup barrier; the poll is an sc1 load; the
//    other waves load behind a workgroup barrier) -- no L2 write-back / invalidate fences, correct wherever the workgroups run.  A
//    barrier of the group's P workgroups sits behind each phase.  The counters only grow (64 bit): a barrier waits for the next multiple
//    of P above the value its own add returned, so nothing is ever reset and a hipGraph replay is as good as the first launch.
//    Every spin is bounded: a barrier that times out sets the handle's error word and the launch finishes with wrong numbers instead of
//    hanging the device -- all 8 P <= 256 workgroups are resident by construction (one per CU).
//    (Measured and dropped: 8-byte {value, launch tag} elements polled by the consumers instead of a barrier -- the same fabric round
//    trips, twice the bytes: 0.319 vs 0.298 ms per 32-lane batch.)
//
// Results differ from the GEMM chain in summation order only (both are fp32 dot products): tests/test_gpu_gru_persist.py compares the two
// directly and soaks the hand-off.
#include "common.h"
#include <cstring>
#include <cstdlib>

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int GP_THREADS = 256;
constexpr int GP_GROUPS = 8;    // lane groups = independent scans
constexpr int GP_U = 8;         // hidden units per workgroup
constexpr int GP_KS = 8;        // k slices per (lane, unit)
constexpr int GP_LB = GP_THREADS / (GP_U * GP_KS);   // 4 lanes per pass
constexpr int GP_NLD = 4;       // staging loads in flight per thread and wait
constexpr unsigned GP_SPIN_MAX = 1u << 19;           // ~ 0.2 s of polling; a healthy hand-off takes microseconds
constexpr unsigned GP_SPIN_FIRST = 1u << 24;         // the launch's FIRST barrier also waits for residency: a long kernel running beside this
                                                     // one (another stream) can hold CUs for hundreds of milliseconds before the last workgroup starts

struct GruPersistArgs {
  const float* pooled; const float* mask; const float* state_in;
  float* hist; float* rh; float* hcur; float* state_out;
  const float* wg[HIERTCN_MAX_LAYERS]; const float* bg[HIERTCN_MAX_LAYERS];
  const float* wc[HIERTCN_MAX_LAYERS]; const float* bc[HIERTCN_MAX_LAYERS];
  unsigned long long* bar;   // [GP_GROUPS] counters, 128 bytes apart, then the time-out word
  int B, S, H, HT, nl, P;
  float* save;               // training (optional): per (slot, layer) the record [gates B x 2H | r*h B x H | candidate B x H | output B x H] the BPTT reads
};

__device__ __forceinline__ f4 ld_sc1(const void* p) {
  f4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// every wave's stores are complete (vmcnt(0)) before the workgroup's one add; the poll and everything behind it sit behind workgroup barriers.
// `dead` (LDS): once a wait of this workgroup has run out, its later barriers add and go on without waiting -- a launch that cannot
// complete correctly ends in about a second instead of twenty-two time-outs.
// (Measured and dropped, round 4: later barriers of a launch know their target -- the first one's + k P -- so their add need not return before the polling
// starts; with a non-returning add and polls from the moment of arrival the 32-lane batch took 0.226 ms against 0.2225: P workgroups polling a line that
// the same P workgroups are still adding to costs more than the return trip saves.)
__device__ __forceinline__ void group_barrier(unsigned long long* bar, unsigned P, unsigned long long* err, int* dead, unsigned spin_max) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned long long old = __hip_atomic_fetch_add(bar, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long target = (old / P + 1ull) * P;
    unsigned spins = 0;
    // (the workgroup whose add completes the epoch -- old + 1 == target -- has seen every arrival in the add's own return value: no poll; 0.2265 -> 0.2225 ms)
    while (old + 1ull < target && !*dead && __hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_max) {   // never in a healthy run: leave a mark and go on rather than hang the device
        __hip_atomic_store(err, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *dead = 1;
      }
    }
  }
  __syncthreads();
}

__device__ __forceinline__ float dot4(const f4 a, const f4 b, float acc) {
  acc = fmaf(a.x, b.x, acc);
  acc = fmaf(a.y, b.y, acc);
  acc = fmaf(a.z, b.z, acc);
  return fmaf(a.w, b.w, acc);
}

// LDS image of a layer's weights: row (kind, u) = 2 Q 8 chunks of 16 bytes in the order [operand half][i][ks], rows 8 chunks further apart
// than they are long, so that the 16 lanes a ds_read_b128 serves together (u, ks mixed) fall into 16 different bank groups
template <int Q>
__device__ __forceinline__ int wchunk(int kind, int u, int part, int i, int ks) {
  return (kind * GP_U + u) * (2 * Q * GP_KS + 8) + (part * Q + i) * GP_KS + ks;
}

struct SegDesc { const float* base; unsigned long long lane_floats; int dst; };   // one operand half of one cell: n lanes x H floats

// Stage `n_seg` segments (H floats for each of n_lanes lanes) into LDS, 16 bytes per load, GP_NLD loads in flight per thread and wait.
// No branch sits between a load and its wait (an index past the end is clamped and its load repeated, only the LDS write is guarded).
__device__ __forceinline__ void stage_segments(const SegDesc* segs, int n_seg, int n_lanes, int H, int b_lo, int b_hi, float* xs, int lane_floats) {
  const int per_seg = n_lanes * (H / 4), total = n_seg * per_seg;
  for (int base = 0; base < total; base += GP_NLD * GP_THREADS) {
    f4 v[GP_NLD];
    int dst[GP_NLD];
#pragma unroll
    for (int it = 0; it < GP_NLD; it++) {
      const int w = min(base + it * GP_THREADS + (int)threadIdx.x, total - 1);
      const int k = w / per_seg, r = w - k * per_seg, pl = r / (H / 4), q = r - pl * (H / 4);
      const SegDesc sd = segs[k];
      const int b = min(b_lo + pl, b_hi - 1);
      v[it] = ld_sc1(sd.base + (size_t)b * sd.lane_floats + 4 * q);
      dst[it] = sd.dst + pl * lane_floats + 4 * q;
    }
    static_assert(GP_NLD == 4, "the wait below names four registers");
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3])::"memory");
#pragma unroll
    for (int it = 0; it < GP_NLD; it++)
      if (base + it * GP_THREADS + (int)threadIdx.x < total) *reinterpret_cast<f4*>(xs + dst[it]) = v[it];
  }
}

template <int Q>   // Q = float4 per thread and operand half = H / 32 (the persistent path needs E == H)
__global__ __launch_bounds__(GP_THREADS) void k_gru_scan_persist(const GruPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ SegDesc segs[2 * HIERTCN_MAX_LAYERS];
  __shared__ int s_dead;
  const int tid = threadIdx.x;
  const int ks = tid & (GP_KS - 1), u = (tid >> 3) & (GP_U - 1), bl = tid >> 6;
  const int H = a.H, HT = a.HT, nl = a.nl, B = a.B, S = a.S, K = 2 * H;
  const int grp = (int)blockIdx.x % GP_GROUPS, t = (int)blockIdx.x / GP_GROUPS;
  const int Bc = (B + GP_GROUPS - 1) / GP_GROUPS;
  const int b_lo = grp * Bc, b_hi = min(B, b_lo + Bc);
  if (b_lo >= b_hi) return;   // a group without lanes
  unsigned long long* bar = a.bar + (size_t)grp * 16;
  unsigned long long* err = a.bar + (size_t)GP_GROUPS * 16;
  if (tid == 0) s_dead = 0;
  const int n_pass = (b_hi - b_lo + GP_LB - 1) / GP_LB, n_lanes = n_pass * GP_LB;
  const int j = t * GP_U + u;   // this thread's hidden unit
  constexpr int WL = 3 * GP_U * (2 * Q * GP_KS + 8);   // 16-byte chunks per layer
  f4* wl = reinterpret_cast<f4*>(lds);
  float* xs = reinterpret_cast<float*>(wl + (size_t)nl * WL);     // [cell of the diagonal][lane of the group][in | h][H]: the staged cell inputs
  const int lane_floats = 2 * H, cell_floats = n_lanes * lane_floats;
  float* keep = xs + (size_t)nl * cell_floats;                     // [cell][lane][unit] x (z, old h, candidate part): phase A -> phase B
  for (int l = 0; l < nl; l++) {
    for (int c = tid; c < 3 * GP_U * 2 * Q * GP_KS; c += GP_THREADS) {
      const int row = c / (2 * Q * GP_KS), r = c - row * (2 * Q * GP_KS);
      const int kind = row / GP_U, uu = row - kind * GP_U;
      const int part = r / (Q * GP_KS), r2 = r - part * (Q * GP_KS), i = r2 / GP_KS, kk = r2 - i * GP_KS;
      const int jj = t * GP_U + uu;
      const float* src = (kind == 2) ? a.wc[l] + (size_t)jj * K : a.wg[l] + (size_t)(kind * H + jj) * K;
      wl[(size_t)l * WL + wchunk<Q>(kind, uu, part, i, kk)] = *reinterpret_cast<const f4*>(src + part * H + kk * (H >> 3) + 4 * i);
    }
  }
  // slot 0 of the history is the incoming state (the state half of the low level's projection reads it: pool_gru.hip)
  for (int i = tid; i < (b_hi - b_lo) * GP_U * nl; i += GP_THREADS) {
    const int b = b_lo + i / (GP_U * nl), r = i % (GP_U * nl), l = r / GP_U, uu = r % GP_U;
    a.hist[(size_t)b * (S + 1) * HT + (size_t)l * H + t * GP_U + uu] = a.state_in[(size_t)b * HT + (size_t)l * H + t * GP_U + uu];
  }
  __syncthreads();

  for (int tau = 0; tau < S + nl - 1; tau++) {
    int l_first = 0, n_cells = 0;                  // the cells of this diagonal: layers l_first .. l_first + n_cells - 1, slot = tau - layer
    for (int l = nl - 1; l >= 0; l--)
      if (tau - l >= 0 && tau - l < S) { l_first = l; n_cells++; }
    // ---------------- phase A: gates, r * h, input half of the candidate
    __syncthreads();
    if (tid < 2 * n_cells) {
      const int ci = tid >> 1, part = tid & 1, l = l_first + ci, s = tau - l;
      SegDesc sd;
      if (part == 0) {
        if (l == 0) { sd.base = a.pooled + (size_t)s * H; sd.lane_floats = (unsigned long long)S * H; }
        else { sd.base = a.hcur + ((size_t)(s & 1) * nl + (l - 1)) * B * H; sd.lane_floats = (unsigned long long)H; }
      } else {
        if (s == 0) { sd.base = a.state_in + (size_t)l * H; sd.lane_floats = (unsigned long long)HT; }
        else { sd.base = a.hist + (size_t)s * HT + (size_t)l * H; sd.lane_floats = (unsigned long long)(S + 1) * HT; }
      }
      sd.dst = ci * cell_floats + part * H;
      segs[tid] = sd;
    }
    __syncthreads();
    stage_segments(segs, 2 * n_cells, n_lanes, H, b_lo, b_hi, xs, lane_floats);
    __syncthreads();
    for (int ci = 0; ci < n_cells; ci++) {
      const int l = l_first + ci;
      const f4* w = wl + (size_t)l * WL;
      for (int ps = 0; ps < n_pass; ps++) {
        const f4* xp = reinterpret_cast<const f4*>(xs + (size_t)ci * cell_floats + (size_t)(ps * GP_LB + bl) * lane_floats);
        float ar = 0.f, au = 0.f, ac = 0.f;
#pragma unroll
        for (int i = 0; i < Q; i++) {
          const f4 xv = xp[ks * Q + i];
          ar = dot4(xv, w[wchunk<Q>(0, u, 0, i, ks)], ar);
          au = dot4(xv, w[wchunk<Q>(1, u, 0, i, ks)], au);
          ac = dot4(xv, w[wchunk<Q>(2, u, 0, i, ks)], ac);
        }
#pragma unroll
        for (int i = 0; i < Q; i++) {
          const f4 hv = xp[(H / 4) + ks * Q + i];
          ar = dot4(hv, w[wchunk<Q>(0, u, 1, i, ks)], ar);
          au = dot4(hv, w[wchunk<Q>(1, u, 1, i, ks)], au);
        }
#pragma unroll
        for (int d = 1; d < GP_KS; d <<= 1) {
          ar += __shfl_xor(ar, d);
          au += __shfl_xor(au, d);
          ac += __shfl_xor(ac, d);
        }
        const int b = b_lo + ps * GP_LB + bl;
        if (ks == 0 && b < b_hi) {
          const float hj = reinterpret_cast<const float*>(xp)[H + j];
          const float r = 1.0f / (1.0f + expf(-(ar + a.bg[l][j])));        // customed_gru_cell.py:323-324: r is the first half
          const float z = 1.0f / (1.0f + expf(-(au + a.bg[l][H + j])));
          st_sc1(a.rh + ((size_t)l * B + b) * H + j, r * hj);              // :326
          if (a.save) {
            float* rec = a.save + ((size_t)(tau - l) * nl + l) * (size_t)B * 5 * H;
            rec[(size_t)b * 2 * H + j] = r;
            rec[(size_t)b * 2 * H + H + j] = z;
            rec[(size_t)B * 2 * H + (size_t)b * H + j] = r * hj;
          }
          float* kp = keep + (((size_t)ci * n_lanes + ps * GP_LB + bl) * GP_U + u) * 3;
          kp[0] = z; kp[1] = hj; kp[2] = ac;
        }
      }
    }
    group_barrier(bar, (unsigned)a.P, err, &s_dead, tau == 0 ? GP_SPIN_FIRST : GP_SPIN_MAX);   // (its closing workgroup barrier also frees the staging buffer and the descriptors)
    // ---------------- phase B: (r * h) half of the candidate, tanh, blend, masked next state
    if (tid < n_cells) {
      SegDesc sd;
      sd.base = a.rh + (size_t)(l_first + tid) * B * H; sd.lane_floats = (unsigned long long)H;
      sd.dst = tid * cell_floats;
      segs[tid] = sd;
    }
    __syncthreads();
    stage_segments(segs, n_cells, n_lanes, H, b_lo, b_hi, xs, lane_floats);
    __syncthreads();
    for (int ci = 0; ci < n_cells; ci++) {
      const int l = l_first + ci, s = tau - l;
      const f4* w = wl + (size_t)l * WL;
      for (int ps = 0; ps < n_pass; ps++) {
        const f4* xp = reinterpret_cast<const f4*>(xs + (size_t)ci * cell_floats + (size_t)(ps * GP_LB + bl) * lane_floats);
        float ac = 0.f;
#pragma unroll
        for (int i = 0; i < Q; i++) ac = dot4(xp[ks * Q + i], w[wchunk<Q>(2, u, 1, i, ks)], ac);
#pragma unroll
        for (int d = 1; d < GP_KS; d <<= 1) ac += __shfl_xor(ac, d);
        const int b = b_lo + ps * GP_LB + bl;
        if (ks == 0 && b < b_hi) {
          const float mk = a.mask[(size_t)b * S + s];
          const float* kp = keep + (((size_t)ci * n_lanes + ps * GP_LB + bl) * GP_U + u) * 3;
          const float z = kp[0], hj = kp[1];
          const float c = tanhf(kp[2] + ac + a.bc[l][j]);                       // :333-335
          const float hn = z * hj + (1.0f - z) * c;                             // :336
          if (a.save) {
            float* rec = a.save + ((size_t)s * nl + l) * (size_t)B * 5 * H;
            rec[(size_t)B * 3 * H + (size_t)b * H + j] = c;
            rec[(size_t)B * 4 * H + (size_t)b * H + j] = hn;
          }
          if (l + 1 < nl) st_sc1(a.hcur + (((size_t)(s & 1) * nl + l) * B + b) * H + j, hn);   // the next layer's input (not masked)
          st_sc1(a.hist + ((size_t)b * (S + 1) + s + 1) * HT + (size_t)l * H + j, hn * mk);     // model_hier.py:93
          if (s == S - 1) a.state_out[(size_t)b * HT + (size_t)l * H + j] = hn * mk;
        }
      }
    }
    if (tau + 1 < S + nl - 1) group_barrier(bar, (unsigned)a.P, err, &s_dead, GP_SPIN_MAX);
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The HierGRU LOW level (model_hier.py:67-68 -> model_gru.py:47-55: dynamic_rnn of a MultiRNNCell of GRUCells from a zero state, with
// sequence_length) of small batches as ONE persistent launch -- the same machinery as the scan above: sessions do not interact, so they
// are cut into 8 groups, each scanned by H / 8 workgroups that keep their units' weight rows in LDS for all L steps; cells (t, l) with
// t + l = tau form a diagonal of two phases with one group barrier behind each.  At the reference's batch (32 lanes x 10 slots = 320
// sessions, L = 20, two layers) the launch chain it replaces is 2 (L + nl - 1) = 42 dependent launches of the split-operand cell kernel
// (gru_x3.hip), ~25 us each for ~2 us of work.  Differences from the scan: the cell input of layer 0 is the projected row u[session][t]
// (fp32, launch_low_gru_bufs); a step t >= sequence_length[session] leaves every layer's state as it was and emits a zero output
// (dynamic_rnn: state copied through, output zeroed); the top layer's outputs are the scoring operand rows [session * L + t][H], in the
// storage type or (exact / split16 modes) fp32; a group's sessions are staged in CHUNKS of lanes, so any number of sessions fits the LDS.
// Exact fp32 dot products (the launch chain's split-operand cells carry 2^-16 per product): tests/test_gpu_hiergru.py compares both with
// the oracle and with each other.
// stage_segments for the low-level kernel: the width is a template constant (index arithmetic in shifts), up to 8 loads in flight per thread
// and wait -- a group of 40 sessions stages 80 KB per phase, and every wait is a round trip to the L2 (sc1)
template <int Q>
__device__ __forceinline__ void stage_segments_wide(const SegDesc* segs, int n_seg, int n_lanes, int b_lo, int b_hi, float* xs, int lane_floats) {
  constexpr int NLD = 8, HQ = 8 * Q;   // float4 per lane and segment (H / 4)
  const int per_seg = n_lanes * HQ, total = n_seg * per_seg;
  for (int base = 0; base < total; base += NLD * GP_THREADS) {
    f4 v[NLD];
    int dst[NLD];
#pragma unroll
    for (int it = 0; it < NLD; it++) {
      const int w = min(base + it * GP_THREADS + (int)threadIdx.x, total - 1);
      const int k = (w >= per_seg ? 1 : 0) + (w >= 2 * per_seg ? 1 : 0) + (w >= 3 * per_seg ? 1 : 0) + (w >= 4 * per_seg ? 1 : 0) +
                    (w >= 5 * per_seg ? 1 : 0) + (w >= 6 * per_seg ? 1 : 0) + (w >= 7 * per_seg ? 1 : 0);
      const int r = w - k * per_seg, pl = r / HQ, q = r % HQ;
      const SegDesc sd = segs[k];
      const int b = min(b_lo + pl, b_hi - 1);
      v[it] = ld_sc1(sd.base + (size_t)b * sd.lane_floats + 4 * q);
      dst[it] = sd.dst + pl * lane_floats + 4 * q;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int it = 0; it < NLD; it++) {
      asm volatile("" : "+v"(v[it]));   // (not before the wait above)
      if (base + it * GP_THREADS + (int)threadIdx.x < total) *reinterpret_cast<f4*>(xs + dst[it]) = v[it];
    }
  }
}

struct LowPersistArgs {
  const float* u; const int32_t* seq_len;
  float* hist;       // [M][L + 1][HT]: every layer's state before step t (slot 0 = the zero state, written here)
  float* rh;         // [nl][M][H]
  float* hcur;       // [2 step parities][nl][M][H]: a layer's output, the next layer's input
  uint16_t* hidden; float* hidden32;
  const float* wg[HIERTCN_MAX_LAYERS]; const float* bg[HIERTCN_MAX_LAYERS];
  const float* wc[HIERTCN_MAX_LAYERS]; const float* bc[HIERTCN_MAX_LAYERS];
  unsigned long long* bar;   // this kernel's own 8 group counters (behind the scan's) + the shared time-out word
  unsigned long long* err;
  int M, L, H, HT, nl, P, CH;   // CH: lanes staged at a time (a multiple of GP_LB)
};

template <int Q>
__global__ __launch_bounds__(GP_THREADS) void k_low_gru_persist(const LowPersistArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ SegDesc segs[2 * HIERTCN_MAX_LAYERS];
  __shared__ int s_dead;
  const int tid = threadIdx.x;
  const int ks = tid & (GP_KS - 1), u = (tid >> 3) & (GP_U - 1), bl = tid >> 6;
  const int H = a.H, HT = a.HT, nl = a.nl, M = a.M, L = a.L, K = 2 * H, CH = a.CH;
  const int grp = (int)blockIdx.x % GP_GROUPS, t = (int)blockIdx.x / GP_GROUPS;
  const int Bc = (M + GP_GROUPS - 1) / GP_GROUPS;
  const int b_lo = grp * Bc, b_hi = min(M, b_lo + Bc);
  if (b_lo >= b_hi) return;   // a group without sessions
  unsigned long long* bar = a.bar + (size_t)grp * 16;
  if (tid == 0) s_dead = 0;
  const int n_grp = b_hi - b_lo;                       // sessions of this group
  const int n_grp_pad = (n_grp + GP_LB - 1) / GP_LB * GP_LB;
  const int j = t * GP_U + u;   // this thread's hidden unit
  constexpr int WL = 3 * GP_U * (2 * Q * GP_KS + 8);   // 16-byte chunks per layer
  f4* wl = reinterpret_cast<f4*>(lds);
  float* xs = reinterpret_cast<float*>(wl + (size_t)nl * WL);     // [cell of the diagonal][lane of the chunk][in | h][H]
  const int lane_floats = 2 * H, cell_floats = CH * lane_floats;
  float* keep = xs + (size_t)nl * cell_floats;                     // [cell][session of the group][unit] x (z, old h, candidate part): phase A -> phase B
  for (int l = 0; l < nl; l++) {
    for (int c = tid; c < 3 * GP_U * 2 * Q * GP_KS; c += GP_THREADS) {
      const int row = c / (2 * Q * GP_KS), r = c - row * (2 * Q * GP_KS);
      const int kind = row / GP_U, uu = row - kind * GP_U;
      const int part = r / (Q * GP_KS), r2 = r - part * (Q * GP_KS), i = r2 / GP_KS, kk = r2 - i * GP_KS;
      const int jj = t * GP_U + uu;
      const float* src = (kind == 2) ? a.wc[l] + (size_t)jj * K : a.wg[l] + (size_t)(kind * H + jj) * K;
      wl[(size_t)l * WL + wchunk<Q>(kind, uu, part, i, kk)] = *reinterpret_cast<const f4*>(src + part * H + kk * (H >> 3) + 4 * i);
    }
  }
  // zero_state (model_gru.py:52, initial_state=None): slot 0 of the history, this workgroup's units
  for (int i = tid; i < n_grp * GP_U * nl; i += GP_THREADS) {
    const int b = b_lo + i / (GP_U * nl), r = i % (GP_U * nl), l = r / GP_U, uu = r % GP_U;
    st_sc1(a.hist + (size_t)b * (L + 1) * HT + (size_t)l * H + t * GP_U + uu, 0.f);
  }
  // the first phase reads slot 0 of EVERY unit: the group's workgroups meet once before it (also the residency wait of the launch)
  group_barrier(bar, (unsigned)a.P, a.err, &s_dead, GP_SPIN_FIRST);

  for (int tau = 0; tau < L + nl - 1; tau++) {
    int l_first = 0, n_cells = 0;                  // the cells of this diagonal: layers l_first .. l_first + n_cells - 1, step = tau - layer
    for (int l = nl - 1; l >= 0; l--)
      if (tau - l >= 0 && tau - l < L) { l_first = l; n_cells++; }
    // ---------------- phase A: gates, r * h, input half of the candidate
    for (int c_lo = 0; c_lo < n_grp; c_lo += CH) {
      const int c_n = min(CH, n_grp - c_lo), c_pad = (c_n + GP_LB - 1) / GP_LB * GP_LB;
      __syncthreads();
      if (tid < 2 * n_cells) {
        const int ci = tid >> 1, part = tid & 1, l = l_first + ci, s = tau - l;
        SegDesc sd;
        if (part == 0) {
          if (l == 0) { sd.base = a.u + (size_t)s * H; sd.lane_floats = (unsigned long long)L * H; }
          else { sd.base = a.hcur + ((size_t)(s & 1) * nl + (l - 1)) * M * H; sd.lane_floats = (unsigned long long)H; }
        } else {
          sd.base = a.hist + (size_t)s * HT + (size_t)l * H; sd.lane_floats = (unsigned long long)(L + 1) * HT;
        }
        sd.dst = ci * cell_floats + part * H;
        segs[tid] = sd;
      }
      __syncthreads();
      stage_segments_wide<Q>(segs, 2 * n_cells, c_pad, b_lo + c_lo, b_lo + c_lo + c_n, xs, lane_floats);
      __syncthreads();
      // Passes (4 lanes each) in blocks of eight: the dot products of a pass leave their three sums on all eight k-slice lanes of a (lane, unit)
      // after the shuffle fold, and k-slice lane p keeps those of the block's p-th pass -- so the epilogue (two sigmoids, three stores) runs
      // ONCE per block with every lane busy, instead of once per pass on one lane in eight.
      for (int ci = 0; ci < n_cells; ci++) {
        const int l = l_first + ci;
        const f4* w = wl + (size_t)l * WL;
        for (int pb = 0; pb < c_pad / GP_LB; pb += GP_KS) {
          float my_r = 0.f, my_u = 0.f, my_c = 0.f;
#pragma unroll
          for (int p8 = 0; p8 < GP_KS; p8++) {
            const int ps = pb + p8;
            if (ps >= c_pad / GP_LB) break;   // (uniform)
            const f4* xp = reinterpret_cast<const f4*>(xs + (size_t)ci * cell_floats + (size_t)(ps * GP_LB + bl) * lane_floats);
            float ar = 0.f, au = 0.f, ac = 0.f;
#pragma unroll
            for (int i = 0; i < Q; i++) {
              const f4 xv = xp[ks * Q + i];
              ar = dot4(xv, w[wchunk<Q>(0, u, 0, i, ks)], ar);
              au = dot4(xv, w[wchunk<Q>(1, u, 0, i, ks)], au);
              ac = dot4(xv, w[wchunk<Q>(2, u, 0, i, ks)], ac);
            }
#pragma unroll
            for (int i = 0; i < Q; i++) {
              const f4 hv = xp[(H / 4) + ks * Q + i];
              ar = dot4(hv, w[wchunk<Q>(0, u, 1, i, ks)], ar);
              au = dot4(hv, w[wchunk<Q>(1, u, 1, i, ks)], au);
            }
#pragma unroll
            for (int d = 1; d < GP_KS; d <<= 1) {
              ar += __shfl_xor(ar, d);
              au += __shfl_xor(au, d);
              ac += __shfl_xor(ac, d);
            }
            my_r = (ks == p8) ? ar : my_r;
            my_u = (ks == p8) ? au : my_u;
            my_c = (ks == p8) ? ac : my_c;
          }
          const int ps = pb + ks;
          const int gl = c_lo + ps * GP_LB + bl;   // session inside the group
          if (ps < c_pad / GP_LB && gl < n_grp) {
            const int b = b_lo + gl;
            const float hj = xs[(size_t)ci * cell_floats + (size_t)(ps * GP_LB + bl) * lane_floats + H + j];
            const float r = 1.0f / (1.0f + expf(-(my_r + a.bg[l][j])));        // customed_gru_cell.py:323-324: r is the first half
            const float z = 1.0f / (1.0f + expf(-(my_u + a.bg[l][H + j])));
            st_sc1(a.rh + ((size_t)l * M + b) * H + j, r * hj);                // :326
            float* kp = keep + (((size_t)ci * n_grp_pad + gl) * GP_U + u) * 3;
            kp[0] = z; kp[1] = hj; kp[2] = my_c;
          }
        }
      }
    }
    group_barrier(bar, (unsigned)a.P, a.err, &s_dead, GP_SPIN_MAX);
    // ---------------- phase B: (r * h) half of the candidate, tanh, blend; sequence_length: a finished session keeps its state, emits zeros
    for (int c_lo = 0; c_lo < n_grp; c_lo += CH) {
      const int c_n = min(CH, n_grp - c_lo), c_pad = (c_n + GP_LB - 1) / GP_LB * GP_LB;
      __syncthreads();
      if (tid < n_cells) {
        SegDesc sd;
        sd.base = a.rh + (size_t)(l_first + tid) * M * H; sd.lane_floats = (unsigned long long)H;
        sd.dst = tid * cell_floats;
        segs[tid] = sd;
      }
      __syncthreads();
      stage_segments_wide<Q>(segs, n_cells, c_pad, b_lo + c_lo, b_lo + c_lo + c_n, xs, lane_floats);
      __syncthreads();
      for (int ci = 0; ci < n_cells; ci++) {
        const int l = l_first + ci, s = tau - l;
        const f4* w = wl + (size_t)l * WL;
        for (int pb = 0; pb < c_pad / GP_LB; pb += GP_KS) {
          float my_c = 0.f;
#pragma unroll
          for (int p8 = 0; p8 < GP_KS; p8++) {
            const int ps = pb + p8;
            if (ps >= c_pad / GP_LB) break;   // (uniform)
            const f4* xp = reinterpret_cast<const f4*>(xs + (size_t)ci * cell_floats + (size_t)(ps * GP_LB + bl) * lane_floats);
            float ac = 0.f;
#pragma unroll
            for (int i = 0; i < Q; i++) ac = dot4(xp[ks * Q + i], w[wchunk<Q>(2, u, 1, i, ks)], ac);
#pragma unroll
            for (int d = 1; d < GP_KS; d <<= 1) ac += __shfl_xor(ac, d);
            my_c = (ks == p8) ? ac : my_c;
          }
          const int ps = pb + ks;
          const int gl = c_lo + ps * GP_LB + bl;
          if (ps < c_pad / GP_LB && gl < n_grp) {
            const int b = b_lo + gl;
            const bool active = s < a.seq_len[b];
            const float* kp = keep + (((size_t)ci * n_grp_pad + gl) * GP_U + u) * 3;
            const float z = kp[0], hj = kp[1];
            const float c = tanhf(kp[2] + my_c + a.bc[l][j]);                     // :333-335
            const float hn = z * hj + (1.0f - z) * c;                             // :336
            const float out = active ? hn : 0.f;
            if (l + 1 < nl) st_sc1(a.hcur + (((size_t)(s & 1) * nl + l) * M + b) * H + j, out);   // the next layer's input
            st_sc1(a.hist + ((size_t)b * (L + 1) + s + 1) * HT + (size_t)l * H + j, active ? hn : hj);
            if (l == nl - 1) {
              if (a.hidden32) a.hidden32[((size_t)b * L + s) * H + j] = out;
              else a.hidden[((size_t)b * L + s) * H + j] = f2bf(out);
            }
          }
        }
      }
    }
    if (tau + 1 < L + nl - 1) group_barrier(bar, (unsigned)a.P, a.err, &s_dead, GP_SPIN_MAX);
  }
}

template <int Q>
hipError_t launch_low_q(const LowPersistArgs& a, size_t lds_bytes, hipStream_t st) {
  auto kern = k_low_gru_persist<Q>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(GP_GROUPS * (a.H / GP_U))), dim3(GP_THREADS), lds_bytes, st, a);
  return hipGetLastError();
}

// lanes staged at a time for M sessions (0: the weights alone do not leave room for one pass of 4 lanes)
int low_persist_chunk(int H, int nl, int64_t M) {
  const int Q = H >> 5;
  const int Bc = (int)((M + GP_GROUPS - 1) / GP_GROUPS), pad = (Bc + GP_LB - 1) / GP_LB * GP_LB;
  const size_t fixed = (size_t)nl * 3 * GP_U * (2 * Q * GP_KS + 8) * 16 + (size_t)nl * pad * GP_U * 3 * 4;
  const size_t budget = 150 * 1024;
  if (fixed >= budget) return 0;
  int ch = (int)((budget - fixed) / ((size_t)nl * 2 * H * 4)) / GP_LB * GP_LB;
  return ch > pad ? pad : ch;
}

template <int Q>
hipError_t launch_q(const GruPersistArgs& a, size_t lds_bytes, hipStream_t st) {
  auto kern = k_gru_scan_persist<Q>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(GP_GROUPS * (a.H / GP_U))), dim3(GP_THREADS), lds_bytes, st, a);
  return hipGetLastError();
}

size_t persist_lds_bytes(int H, int nl, int B) {
  const int Q = H >> 5;
  const int Bc = (B + GP_GROUPS - 1) / GP_GROUPS, n_pass = (Bc + GP_LB - 1) / GP_LB;
  // weights + the staged inputs of a whole diagonal (every layer x every lane of the group) + the cross-phase scratch
  return (size_t)nl * 3 * GP_U * (2 * Q * GP_KS + 8) * 16 + (size_t)nl * n_pass * GP_LB * 2 * H * 4 + (size_t)nl * n_pass * GP_LB * GP_U * 3 * 4;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward through the across-session scan (customed_gru_cell.py:323-337 / model_hier.py:91-93 differentiated by model.py:139-141) of small
// batches as ONE persistent launch (round 4) -- the mirror image of k_gru_scan_persist.  train.hip's chain is 2 nl + 1 dependent launches per
// slot (a first elementwise step and two skinny products with fused epilogues per cell): ~50 launches of 5-8 us for ~1 us of work each.
// Here the same 8 lane groups x P = H / 8 workgroups; workgroup t OWNS, of every layer, units [8 t, 8 t + 8) -- as elements of the cell's
// output, and as COLUMNS 8 t + u and in + 8 t + u of the two products' outputs
//     d[in | r*h] = da_c . Wc        (da_c [lanes][H],  Wc [H][K],  K = in + H, in = E = H)
//     d[in | h]   = [da_r | da_z] . Wg      ([lanes][2H], Wg [2H][K])
// whose weight columns (Wc[:, col], Wg[:, col]: 48 H floats per layer) it keeps in LDS for all slots.  With that ownership everything
// elementwise is local -- the gradient of a cell's output (carried + state-projection path, masked, + the layer above's d input), da_c,
// da_z, the partial d h, d(r*h) -> da_r, d input, the new carried gradient -- and exactly TWO vectors per cell cross workgroups: da_c (all
// units, before the first product) and [da_r | da_z] (before the second); they go through their slots of dac_all / dag_all, which the
// weight-gradient products behind the scan read anyway (sc1 stores, one group barrier, sc1 loads: gru_persist's forward protocol).
// Cells (s, l) with (S - 1 - s) + (nl - 1 - l) = tau are independent: S + nl - 1 diagonals of two phases.
// fp32 dot products in a fixed order (k slices folded by shuffles): equal to the chain up to summation order.
struct GruBpttArgs {
  const float* save;     // forward records, per (s, l): gates [B][2H] | r*h [B][H] | candidate [B][H] | output [B][H]
  const float* hist;     // [B][S + 1][HT]: the state before slot s (the cell's old state)
  const float* mask;     // [B][S]
  const float* dsb;      // [B][S + 1][HT]: d state_before from the state projection; cell s takes slot s + 1
  const float* wc[HIERTCN_MAX_LAYERS]; const float* wg[HIERTCN_MAX_LAYERS];
  float* dac_all;        // [nl][S][B][H]
  float* dag_all;        // [nl][S][B][2H]
  float* dpooled;        // [B][S][E]: d input of layer 0
  unsigned long long* bar;
  int B, S, H, HT, nl, P;
};

// LDS image of a layer's weight COLUMNS: row (kind, u) with kind 0 / 1 = Wc[:, 8t+u] / Wc[:, in+8t+u] (H floats), 2 / 3 = Wg[:, 8t+u] / Wg[:, in+8t+u]
// (2H floats); inside a row the 16-byte chunks in the order [i][ks] (thread ks takes reduction indices ks * len/8 + 4 i ..), rows 8 chunks apart
template <int Q>
__device__ __forceinline__ int bchunk(int kind, int u, int i, int ks) {
  // chunks per row: kind < 2: Q * 8, else 2 Q * 8 (+ 8 of padding each)
  const int base = kind < 2 ? (kind * GP_U + u) * (Q * GP_KS + 8) : 2 * GP_U * (Q * GP_KS + 8) + ((kind - 2) * GP_U + u) * (2 * Q * GP_KS + 8);
  return base + i * GP_KS + ks;
}

template <int Q>
__global__ __launch_bounds__(GP_THREADS) void k_gru_bptt_persist(const GruBpttArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  __shared__ SegDesc segs[HIERTCN_MAX_LAYERS];
  __shared__ int s_dead;
  const int tid = threadIdx.x;
  const int ks = tid & (GP_KS - 1), u = (tid >> 3) & (GP_U - 1), bl = tid >> 6;
  const int H = a.H, HT = a.HT, nl = a.nl, B = a.B, S = a.S, K = 2 * H;
  const int grp = (int)blockIdx.x % GP_GROUPS, t = (int)blockIdx.x / GP_GROUPS;
  const int Bc = (B + GP_GROUPS - 1) / GP_GROUPS;
  const int b_lo = grp * Bc, b_hi = min(B, b_lo + Bc);
  if (b_lo >= b_hi) return;   // a group without lanes
  unsigned long long* bar = a.bar + (size_t)grp * 16;
  unsigned long long* err = a.bar + (size_t)GP_GROUPS * 16;
  if (tid == 0) s_dead = 0;
  const int n_pass = (b_hi - b_lo + GP_LB - 1) / GP_LB, n_lanes = n_pass * GP_LB;
  const int j = t * GP_U + u;   // this thread's unit
  constexpr int WL = 2 * GP_U * (Q * GP_KS + 8) + 2 * GP_U * (2 * Q * GP_KS + 8);   // 16-byte chunks per layer
  f4* wl = reinterpret_cast<f4*>(lds);
  float* xs = reinterpret_cast<float*>(wl + (size_t)nl * WL);          // [cell][lane][2H]: the staged da_c (H) or [da_r | da_z] (2H) of the diagonal's cells
  const int lane_floats = 2 * H, cell_floats = n_lanes * lane_floats;
  float* keep = xs + (size_t)nl * cell_floats;                          // per (cell, lane, unit): d h partial, d input part of the first product
  float* dcar = keep + (size_t)nl * n_lanes * GP_U * 2;                 // [layer][lane][unit]: carried state gradient (own units)
  float* dabove = dcar + (size_t)nl * n_lanes * GP_U;                   // [layer][lane][unit]: d input handed down by the layer above (own units)
  // ---- the weight columns, once: for every reduction row the 8 + 8 floats at columns 8 t .. and in + 8 t .. (32-byte pieces of the rows)
  for (int l = 0; l < nl; l++) {
    float* wf = reinterpret_cast<float*>(wl + (size_t)l * WL);
    for (int c = tid; c < 3 * H * 2; c += GP_THREADS) {   // (row of Wc: H, of Wg: 2H) x (input part, state part)
      const int row = c >> 1, part = c & 1;
      const bool isg = row >= H;
      const int jj = isg ? row - H : row;
      const float* src = (isg ? a.wg[l] : a.wc[l]) + (size_t)jj * K + part * H + t * GP_U;
      const f4 v0 = *reinterpret_cast<const f4*>(src), v1 = *reinterpret_cast<const f4*>(src + 4);
      const int kind = (isg ? 2 : 0) + part, len = isg ? 2 * H : H;
      const int kk = jj / (len >> 3), r = jj - kk * (len >> 3), i = r >> 2, e = r & 3;   // reduction index jj -> (k slice, chunk, element)
      const float vv[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
      for (int uu = 0; uu < GP_U; uu++) wf[(size_t)bchunk<Q>(kind, uu, i, kk) * 4 + e] = vv[uu];
    }
  }
  for (int i = tid; i < 2 * nl * n_lanes * GP_U; i += GP_THREADS) dcar[i] = 0.f;   // dcar and dabove (contiguous): no gradient enters behind the last slot / above the top layer
  __syncthreads();

  const size_t cellf = (size_t)B * 5 * H;
  const int64_t hs = (int64_t)(S + 1) * HT;
  for (int tau = 0; tau < S + nl - 1; tau++) {
    // the cells of this diagonal: layers l_first .. (descending slot): cell ci has layer l = l_first + ci, slot s = S - 1 - (tau - (nl - 1 - l))
    int l_first = 0, n_cells = 0;
    for (int l = nl - 1; l >= 0; l--) {
      const int s = S - 1 - (tau - (nl - 1 - l));
      if (s >= 0 && s < S) { l_first = l; n_cells++; }
    }
    // ---------------- phase A (local): d = (carried + state-projection path) * mask + d input from above;  da_c, da_z, partial d h
    if (ks == 0) {
      for (int ci = 0; ci < n_cells; ci++) {
        const int l = l_first + ci, s = S - 1 - (tau - (nl - 1 - l));
        const float* rec = a.save + ((size_t)s * nl + l) * cellf;
        for (int ps = 0; ps < n_pass; ps++) {
          const int pl = ps * GP_LB + bl, b = b_lo + pl;
          if (b >= b_hi) continue;
          const int li = (l * n_lanes + pl) * GP_U + u;
          const float d = (dcar[li] + a.dsb[(size_t)b * hs + (size_t)(s + 1) * HT + (size_t)l * H + j]) * a.mask[(size_t)b * S + s] + dabove[li];
          const float uz = rec[(size_t)b * 2 * H + H + j], c = rec[(size_t)B * 3 * H + (size_t)b * H + j];
          const float hv = a.hist[(size_t)b * hs + (size_t)s * HT + (size_t)l * H + j];
          const float dac = d * (1.f - uz) * (1.f - c * c);
          st_sc1(a.dac_all + (((size_t)l * S + s) * B + b) * H + j, dac);
          st_sc1(a.dag_all + (((size_t)l * S + s) * B + b) * 2 * H + H + j, d * (hv - c) * uz * (1.f - uz));
          keep[((ci * n_lanes + pl) * GP_U + u) * 2] = d * uz;
        }
      }
    }
    group_barrier(bar, (unsigned)a.P, err, &s_dead, tau == 0 ? GP_SPIN_FIRST : GP_SPIN_MAX);
    // ---------------- phase B: stage da_c of every unit, first product (own columns), d(r*h) -> da_r, partial d h; publish da_r
    if (tid < n_cells) {
      const int l = l_first + tid, s = S - 1 - (tau - (nl - 1 - l));
      SegDesc sd;
      sd.base = a.dac_all + ((size_t)l * S + s) * B * H; sd.lane_floats = (unsigned long long)H; sd.dst = tid * cell_floats;
      segs[tid] = sd;
    }
    __syncthreads();
    stage_segments(segs, n_cells, n_lanes, H, b_lo, b_hi, xs, lane_floats);
    __syncthreads();
    for (int ci = 0; ci < n_cells; ci++) {
      const int l = l_first + ci, s = S - 1 - (tau - (nl - 1 - l));
      const f4* w = wl + (size_t)l * WL;
      const float* rec = a.save + ((size_t)s * nl + l) * cellf;
      for (int ps = 0; ps < n_pass; ps++) {
        const int pl = ps * GP_LB + bl, b = b_lo + pl;
        const f4* xp = reinterpret_cast<const f4*>(xs + (size_t)ci * cell_floats + (size_t)pl * lane_floats);
        float ai = 0.f, ar = 0.f;
#pragma unroll
        for (int i = 0; i < Q; i++) {
          const f4 xv = xp[ks * Q + i];
          ai = dot4(xv, w[bchunk<Q>(0, u, i, ks)], ai);
          ar = dot4(xv, w[bchunk<Q>(1, u, i, ks)], ar);
        }
#pragma unroll
        for (int d = 1; d < GP_KS; d <<= 1) {
          ai += __shfl_xor(ai, d);
          ar += __shfl_xor(ar, d);
        }
        if (ks == 0 && b < b_hi) {
          const float r = rec[(size_t)b * 2 * H + j];
          const float hv = a.hist[(size_t)b * hs + (size_t)s * HT + (size_t)l * H + j];
          st_sc1(a.dag_all + (((size_t)l * S + s) * B + b) * 2 * H + j, ar * hv * r * (1.f - r));   // da_r = d(r*h) h r (1 - r)
          float* kp = keep + ((ci * n_lanes + pl) * GP_U + u) * 2;
          kp[0] += ar * r;     // d h += d(r*h) r
          kp[1] = ai;          // the candidate's share of d input
        }
      }
    }
    group_barrier(bar, (unsigned)a.P, err, &s_dead, GP_SPIN_MAX);
    // ---------------- phase C: stage [da_r | da_z] of every unit, second product (own columns); d input -> the layer below / d pooled; carried gradient
    if (tid < n_cells) {
      const int l = l_first + tid, s = S - 1 - (tau - (nl - 1 - l));
      SegDesc sd;
      sd.base = a.dag_all + ((size_t)l * S + s) * B * 2 * H; sd.lane_floats = (unsigned long long)2 * H; sd.dst = tid * cell_floats;
      segs[tid] = sd;
    }
    __syncthreads();
    stage_segments(segs, n_cells, n_lanes, 2 * H, b_lo, b_hi, xs, lane_floats);
    __syncthreads();
    for (int ci = 0; ci < n_cells; ci++) {
      const int l = l_first + ci, s = S - 1 - (tau - (nl - 1 - l));
      const f4* w = wl + (size_t)l * WL;
      for (int ps = 0; ps < n_pass; ps++) {
        const int pl = ps * GP_LB + bl, b = b_lo + pl;
        const f4* xp = reinterpret_cast<const f4*>(xs + (size_t)ci * cell_floats + (size_t)pl * lane_floats);
        float gi = 0.f, gh = 0.f;
#pragma unroll
        for (int i = 0; i < 2 * Q; i++) {
          const f4 xv = xp[ks * 2 * Q + i];
          gi = dot4(xv, w[bchunk<Q>(2, u, i, ks)], gi);
          gh = dot4(xv, w[bchunk<Q>(3, u, i, ks)], gh);
        }
#pragma unroll
        for (int d = 1; d < GP_KS; d <<= 1) {
          gi += __shfl_xor(gi, d);
          gh += __shfl_xor(gh, d);
        }
        if (ks == 0 && b < b_hi) {
          const float* kp = keep + ((ci * n_lanes + pl) * GP_U + u) * 2;
          const float din = kp[1] + gi;
          if (l > 0) dabove[((l - 1) * n_lanes + pl) * GP_U + u] = din;   // the cell below at this slot is on the next diagonal
          else a.dpooled[((size_t)b * S + s) * H + j] = din;
          dcar[(l * n_lanes + pl) * GP_U + u] = kp[0] + gh;               // cell (s - 1, l): next diagonal
        }
      }
    }
    __syncthreads();   // (keep / dcar / dabove of this diagonal are complete before the next phase A reads them)
  }
}

size_t bptt_lds_bytes(int H, int nl, int B) {
  const int Q = H >> 5;
  const int Bc = (B + GP_GROUPS - 1) / GP_GROUPS, n_pass = (Bc + GP_LB - 1) / GP_LB, n_lanes = n_pass * GP_LB;
  const size_t WL = 2 * GP_U * (Q * GP_KS + 8) + 2 * GP_U * (2 * Q * GP_KS + 8);
  return (size_t)nl * WL * 16 + (size_t)nl * n_lanes * 2 * H * 4 + (size_t)nl * n_lanes * GP_U * 4 * 4;   // weights | staging | keep (2) + dcar + dabove
}

template <int Q>
hipError_t launch_bptt_q(const GruBpttArgs& a, size_t lds_bytes, hipStream_t st) {
  auto kern = k_gru_bptt_persist<Q>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3((unsigned)(GP_GROUPS * (a.H / GP_U))), dim3(GP_THREADS), lds_bytes, st, a);
  return hipGetLastError();
}

}  // namespace

bool gru_persist_shape_ok(const ModelDev& m, int B, int S);
// the scan's backward as one persistent launch: the forward's shapes (gru_persist_shape_ok) where its own LDS image fits
bool gru_bptt_persist_applies(const ModelDev& m, int B, int S) {
  if (m.tune.gru_bptt_persistent == 0 || m.tune.gru_scan_persistent == 0) return false;   // (hiertcn_set_tuning: train.hip's launch chain)
  return gru_persist_shape_ok(m, B, S) && bptt_lds_bytes(m.H, m.nl, B) <= 150 * 1024;
}

hipError_t launch_gru_bptt_persist(const ModelDev& m, const float* save, const float* hist, const float* mask, const float* dsb, int B, int S, float* dac_all,
                                   float* dag_all, float* dpooled, hipStream_t st) {
  GruBpttArgs a;
  memset(&a, 0, sizeof(a));
  a.save = save; a.hist = hist; a.mask = mask; a.dsb = dsb; a.dac_all = dac_all; a.dag_all = dag_all; a.dpooled = dpooled;
  for (int l = 0; l < m.nl; l++) { a.wc[l] = m.gru_wc[l]; a.wg[l] = m.gru_wg[l]; }
  a.bar = m.gru_bar;
  a.B = B; a.S = S; a.H = m.H; a.HT = m.HT; a.nl = m.nl; a.P = m.H / GP_U;
  const size_t lds_bytes = bptt_lds_bytes(m.H, m.nl, B);
  switch (m.H >> 5) {
    case 2: return launch_bptt_q<2>(a, lds_bytes, st);
    case 4: return launch_bptt_q<4>(a, lds_bytes, st);
    case 8: return launch_bptt_q<8>(a, lds_bytes, st);
    default: return hipErrorInvalidValue;
  }
}

// shapes the persistent launch handles (no environment switch looked at: the workspace is sized by this)
bool gru_persist_shape_ok(const ModelDev& m, int B, int S) {
  if (!m.gru_bar || m.n_cu <= 0 || B <= 0 || S <= 0 || B > 96) return false;   // measured cross-over with the GEMM chain: ~100 lanes
  if (m.E != m.H || (m.H != 64 && m.H != 128 && m.H != 256)) return false;      // (E != H, H = 512 or odd widths: the GEMM chain)
  if (GP_GROUPS * (m.H / GP_U) > m.n_cu) return false;                           // every workgroup resident, one per CU
  return persist_lds_bytes(m.H, m.nl, B) <= 150 * 1024;   // (H = 256, two layers: up to 64 lanes)
}

// 0 = not applicable (the caller runs the GEMM chain), else the number of workgroups per lane group
int gru_persist_groups(const ModelDev& m, int B, int S, bool saving) {
  if (m.tune.gru_scan_persistent == 0) return 0;   // always the GEMM chain (a shared GPU, A/B runs, tests)
  if ((saving && m.tune.gru_scan_persistent_train == 0) || !gru_persist_shape_ok(m, B, S)) return 0;   // (training: the launch also writes the BPTT's records)
  if (B > m.tune.gru_persist_max_lanes) return 0;
  return m.H / GP_U;
}

hipError_t launch_gru_scan_persist(const ModelDev& m, const float* pooled, const float* mask, const float* state_in, int B, int S,
                                   float* state_out, float* ws, int P, hipStream_t st, float* save) {
  const int H = m.H, HT = m.HT;
  GruPersistArgs a;
  memset(&a, 0, sizeof(a));
  a.pooled = pooled; a.mask = mask; a.state_in = state_in; a.state_out = state_out; a.save = save;
  a.hist = ws;                                                   // the workspace layout of launch_gru_scan
  float* gates0 = a.hist + (size_t)B * (S + 1) * HT;
  a.rh = gates0 + (size_t)m.nl * B * 2 * H;
  a.hcur = a.rh + (size_t)m.nl * B * H;
  for (int l = 0; l < m.nl; l++) { a.wg[l] = m.gru_wg[l]; a.bg[l] = m.gru_bg[l]; a.wc[l] = m.gru_wc[l]; a.bc[l] = m.gru_bc[l]; }
  a.bar = m.gru_bar;
  a.B = B; a.S = S; a.H = H; a.HT = HT; a.nl = m.nl; a.P = P;
#ifdef HIERTCN_TEST_HOOKS   // libhiertcn_gfx950_testhooks.so only (hiertcn_amd/build.py): the shipped libraries read no test switch on this path
  if (getenv("HIERTCN_GRU_TEST_MISSING_WG")) a.P = P + 1;   // every barrier waits for a workgroup that does not exist -> the bounded waits run out
#endif
  const size_t lds_bytes = persist_lds_bytes(H, m.nl, B);
  switch (H >> 5) {
    case 2: return launch_q<2>(a, lds_bytes, st);
    case 4: return launch_q<4>(a, lds_bytes, st);
    case 8: return launch_q<8>(a, lds_bytes, st);
    default: return hipErrorInvalidValue;
  }
}

// ---- HierGRU low level as one persistent launch (k_low_gru_persist) ------------------------------------------------------------------
// applies to evaluation batches of up to 1024 sessions (beyond that the split-operand MFMA cells of gru_x3.hip win), E == H in {64, 128}; at
// H = 256 two layers of weight rows leave LDS room for 8 staged sessions at a time and the launch is no faster than the chain it replaces
// (1.53 vs 1.41 ms per 32-lane batch; H = 128: 0.81 vs 1.04, H = 64: 0.63 vs 0.91) -- tune.lgru_persistent = 1 forces it there too (A/B, tests)
bool low_gru_persist_applies(const ModelDev& m, int64_t n_sess) {
  if (m.tune.lgru_persistent == 0) return false;   // always the launch chain; 1: the persistent launch wherever it fits
  if (!m.low_gru || !m.gru_bar || m.n_cu <= 0 || n_sess <= 0 || n_sess > 1024) return false;
  if (m.E != m.H || (m.H != 64 && m.H != 128 && m.H != 256)) return false;
  if (m.H == 256 && m.tune.lgru_persistent != 1) return false;
  if (GP_GROUPS * (m.H / GP_U) > m.n_cu) return false;
  return low_persist_chunk(m.H, m.nl, n_sess) >= GP_LB;
}

size_t low_gru_persist_floats(const ModelDev& m, int64_t n_sess, int L) {
  return (size_t)n_sess * (L + 1) * m.HT + (size_t)m.nl * n_sess * m.H + 2 * (size_t)m.nl * n_sess * m.H;   // hist | rh | hcur
}

hipError_t launch_low_gru_persist(const ModelDev& m, const float* u, const int32_t* seq_len, int64_t n_sess, int L, uint16_t* hidden, float* hidden32,
                                  float* ws, hipStream_t st) {
  const int H = m.H;
  LowPersistArgs a;
  memset(&a, 0, sizeof(a));
  a.u = u; a.seq_len = seq_len; a.hidden = hidden; a.hidden32 = hidden32;
  a.hist = ws;
  a.rh = a.hist + (size_t)n_sess * (L + 1) * m.HT;
  a.hcur = a.rh + (size_t)m.nl * n_sess * H;
  for (int l = 0; l < m.nl; l++) { a.wg[l] = m.lgru_wg[l]; a.bg[l] = m.lgru_bg[l]; a.wc[l] = m.lgru_wc[l]; a.bc[l] = m.lgru_bc[l]; }
  a.bar = m.gru_bar + 9 * 16;                 // the scan's eight counters and the time-out word come first
  a.err = m.gru_bar + (size_t)GP_GROUPS * 16;
  a.M = (int)n_sess; a.L = L; a.H = H; a.HT = m.HT; a.nl = m.nl; a.P = H / GP_U;
  a.CH = low_persist_chunk(H, m.nl, n_sess);
  const int Q = H >> 5;
  const int Bc = (int)((n_sess + GP_GROUPS - 1) / GP_GROUPS), pad = (Bc + GP_LB - 1) / GP_LB * GP_LB;
  const size_t lds_bytes = (size_t)m.nl * 3 * GP_U * (2 * Q * GP_KS + 8) * 16 + (size_t)m.nl * a.CH * 2 * H * 4 + (size_t)m.nl * pad * GP_U * 3 * 4;
  switch (Q) {
    case 2: return launch_low_q<2>(a, lds_bytes, st);
    case 4: return launch_low_q<4>(a, lds_bytes, st);
    case 8: return launch_low_q<8>(a, lds_bytes, st);
    default: return hipErrorInvalidValue;
  }
}
