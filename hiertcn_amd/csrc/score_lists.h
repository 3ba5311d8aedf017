// Per-lane ranked top-k lists of the scoring kernels (score.hip, score_x3.hip): loss.py:120's order -- values descending, equal values in
// index order (tf.nn.top_k).
#pragma once
#include "common.h"

// Running top-KT list of one lane (one step, half of the items): values descending, equal values in arrival
// (= index) order, which is the order tf.nn.top_k returns (loss.py:120: ties -> lower index first).
// A candidate does not go into the sorted list at once: it is parked in a two-entry buffer, and the buffers of all 64 lanes are
// flushed into their lists (two insertions) only when a lane that already holds two needs a third.  A wave pays for a list insertion
// whenever ONE of its lanes has a candidate (~6,000 times per pass over 300,000 items), a buffered candidate costs a handful of selects
// and the flushes are ~40 x rarer; the list's last entry -- the admission threshold -- is merely a little stale in between, which admits
// a few candidates more and changes nothing in the result (top_insert re-checks; arrival order is kept: buffer order = arrival order).
template <int KT>
struct TopList {
  float v[KT];
  int i[KT];
  float pv0, pv1;   // parked candidates
  int pi0, pi1;
  int pc;           // how many are parked (0..2)
};

// insert (val, idx) below the last entry and bubble it up; strict '>' keeps equal values in arrival order
template <int KT>
__device__ __forceinline__ void top_insert(TopList<KT>& L, float val, int idx) {
  const bool take = val > L.v[KT - 1];
  L.v[KT - 1] = take ? val : L.v[KT - 1];
  L.i[KT - 1] = take ? idx : L.i[KT - 1];
#pragma unroll
  for (int k = KT - 1; k > 0; k--) {
    const bool up = L.v[k] > L.v[k - 1];
    const float hv = up ? L.v[k] : L.v[k - 1], lv = up ? L.v[k - 1] : L.v[k];
    const int hi = up ? L.i[k] : L.i[k - 1], li = up ? L.i[k - 1] : L.i[k];
    L.v[k - 1] = hv; L.v[k] = lv;
    L.i[k - 1] = hi; L.i[k] = li;
  }
}

template <int KT>
__device__ __forceinline__ void top_init(TopList<KT>& L) {
#pragma unroll
  for (int k = 0; k < KT; k++) { L.v[k] = -INFINITY; L.i[k] = 0x7fffffff; }
  L.pv0 = L.pv1 = -INFINITY;
  L.pi0 = L.pi1 = 0x7fffffff;
  L.pc = 0;
}

// parked candidates -> the sorted list, in arrival order (an empty slot holds -inf: its insertion is a no-op)
template <int KT>
__device__ __forceinline__ void top_flush(TopList<KT>& L) {
  top_insert<KT>(L, L.pv0, L.pi0);
  top_insert<KT>(L, L.pv1, L.pi1);
  L.pv0 = L.pv1 = -INFINITY;
  L.pc = 0;
}

// one 32-item x 32-step accumulator tile against the lane's list; item index of register r is item0 + 8*(r>>2) + (r&3)
template <int KT>
__device__ __forceinline__ void top_tile(TopList<KT>& L, const f32x16& v, int item0) {
  float tm = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
  tm = fmaxf(tm, fmaxf(fmaxf(fmaxf(v[8], v[9]), fmaxf(v[10], v[11])), fmaxf(fmaxf(v[12], v[13]), fmaxf(v[14], v[15]))));
  if (!__any(tm > L.v[KT - 1])) return;  // wave-uniform: no lane of this wave has a candidate in this tile
#pragma unroll
  for (int r = 0; r < 16; r++) {
    bool take = v[r] > L.v[KT - 1];
    if (__any(take)) {
      if (__any(take && L.pc == 2)) {   // a lane with a full buffer needs a third slot: everybody flushes (rare)
        top_flush<KT>(L);
        take = v[r] > L.v[KT - 1];      // the threshold may have risen
      }
      const int idx = item0 + 8 * (r >> 2) + (r & 3);
      const bool t0 = take && L.pc == 0, t1 = take && L.pc == 1;
      L.pv0 = t0 ? v[r] : L.pv0; L.pi0 = t0 ? idx : L.pi0;
      L.pv1 = t1 ? v[r] : L.pv1; L.pi1 = t1 ? idx : L.pi1;
      L.pc += take ? 1 : 0;
    }
  }
}

// the slow half of top_tile alone (the caller has already voted that some lane of the wave has a candidate in this tile)
template <int KT>
__device__ __forceinline__ void top_tile_slow(TopList<KT>& L, const f32x16& v, int item0) {
#ifdef HIERTCN_TOPK_NOSLOW   // timing experiment (wrong lists): what the list maintenance itself costs
  L.v[KT - 1] = fmaxf(L.v[KT - 1], v[0]);
  return;
#endif
#pragma unroll
  for (int r = 0; r < 16; r++) {
    bool take = v[r] > L.v[KT - 1];
    if (__any(take)) {
      if (__any(take && L.pc == 2)) {   // a lane with a full buffer needs a third slot: everybody flushes (rare)
        top_flush<KT>(L);
        take = v[r] > L.v[KT - 1];      // the threshold may have risen
      }
      const int idx = item0 + 8 * (r >> 2) + (r & 3);
      const bool t0 = take && L.pc == 0, t1 = take && L.pc == 1;
      L.pv0 = t0 ? v[r] : L.pv0; L.pi0 = t0 ? idx : L.pi0;
      L.pv1 = t1 ? v[r] : L.pv1; L.pi1 = t1 ? idx : L.pi1;
      L.pc += take ? 1 : 0;
    }
  }
}

