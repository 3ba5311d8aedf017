"""model_tcn with the reference's signature (model_tcn.py:13), backed by the HIP kernels.

    model_tcn(args, x, x_gap=None, x_impression=None, name='tcn', reuse=True,
              training=False, mask=None) -> pred

Single-level use (model.py:71-73): x = ids [B, T]; the first dense of the reference
(model_tcn.py:35 on a one-hot input) is the embedding lookup with row 0 -> 0.
The reference's single-level configuration (args.py:310-311: six blocks, channels
[128,128,128,128,256,256], T <= 500, a Dense down-sample on the residual of block 4) runs on the
layer-at-a-time kernels (csrc/tcn_layers.hip); short equal-width stacks take the fused kernel.
"""
from __future__ import annotations

import numpy as np
import torch

from .model import LazyLogits, _to_ids, get_scope


def model_tcn(args, x, x_gap=None, x_impression=None, name="tcn", reuse=True, training=False, mask=None):
    from .engine import check_supported
    check_supported(args)
    ids = _to_ids(x)
    if ids.ndim != 2:
        raise NotImplementedError("model_tcn takes ids [B, T] (the one-hot of model.py:56 is undone); "
                                  "the float-input form is reached through model_hier")
    model = get_scope(args, name)
    eng = model.engine
    B, T = ids.shape
    xd = torch.from_numpy(np.ascontiguousarray(ids)).to(eng.device)
    hidden = eng.tcn_forward(xd, None)
    return LazyLogits(eng, hidden, B, 1, T, [T])
