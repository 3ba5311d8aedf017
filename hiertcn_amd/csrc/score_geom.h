// Shared by the scoring kernels (score.hip) and the training backward's d-logits kernel (score_grad.hip).
#pragma once
#include "common.h"

#define SC_THREADS 512
#define SC_STAGE 64  // items per LDS stage (two 32-item MFMA row blocks)

template <int C>
struct ScGeom {
  static constexpr int KS = C / 16;
  static constexpr int CH = C / 8;
  static constexpr int SWZ = (CH < 16 ? CH : 16) - 1;
  static constexpr int STAGE_BYTES = SC_STAGE * 2 * C;
  static constexpr int CHUNKS_PER_THREAD = (SC_STAGE * CH) / SC_THREADS;  // 16-byte chunks each thread stages
  static_assert((SC_STAGE * CH) % SC_THREADS == 0, "stage must split evenly");
};


// Small batches (score.hip, k_score_pipe with part_stages < 0): the split of the catalogue is decided ON THE DEVICE from the count of rows that carry
// a target -- a ragged XING batch has about half the rows of its host-side bound, and a split sized for the bound leaves half the workgroups without
// work.  `grid` workgroups are dealt as (row tile, catalogue part) = (id % tiles, id / tiles); the merge kernel derives the same numbers.
struct ScDynSplit { int tiles, parts, part_stages; };
__host__ __device__ inline ScDynSplit score_dyn_split(int64_t rows, int rows_per_wg, int n_all, int grid, int max_parts) {
  ScDynSplit d;
  d.tiles = (int)((rows + rows_per_wg - 1) / rows_per_wg);
  if (d.tiles < 1) d.tiles = 1;
  int parts = grid / d.tiles;
  const int by_len = n_all / 16;   // a part is at least 16 stages: below that the prologue (hidden rows, target logits) outweighs the stages
  if (parts > by_len) parts = by_len;
  if (parts > max_parts) parts = max_parts;
  if (parts < 1) parts = 1;
  if (grid > 256 && d.tiles * parts > 256) {
    // a grid of several rounds of the 256 CUs (k_sgrad mode 2 past 128 row blocks): the part count that makes rounds x (stages per part + a workgroup's fixed
    // cost, ~16 stages: hidden rows, target logits, the slab it stores) smallest -- 138 live blocks: 5 parts = 3 rounds x 81 against 1 round x 341
    int best = 1, best_cost = 0x7fffffff;
    for (int p = 1; p <= parts; p++) {
      const int cost = ((d.tiles * p + 255) / 256) * ((n_all + p - 1) / p + 16);
      if (cost < best_cost) { best_cost = cost; best = p; }
    }
    parts = best;
  }
  d.parts = parts;
  d.part_stages = (n_all + parts - 1) / parts;
  return d;
}
#define SC_DYN_MAX_PARTS 32
#define SC_DYN_GRID 256
