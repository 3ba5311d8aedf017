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

