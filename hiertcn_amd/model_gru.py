"""model_gru with the reference's signature (model_gru.py:16), backed by the HIP kernels.

    model_gru(args, x, x_gap=None, x_impression=None, initial_state=None, return_state=False, name='gru', reuse=True) -> pred

Single-level use (model.py:66-67): x = ids [B, T]; the 'emb' dense of the reference (model_gru.py:46 on a one-hot input) is the embedding
lookup with row 0 -> 0; the MultiRNNCell of GRUCells runs under dynamic_rnn from the zero state with sequence_length = the COUNT of non-null
input steps used as a prefix length (:47-53, :109-118) -- the loader's leading null step (data_loader.py:78-98: x is y shifted right) cuts every
sequence one step short, and the cut step outputs zeros (its logits are the dense bias).  The scoring dense (:57) is applied lazily (LazyLogits).
As the low level of model_hier (model_hier.py:67-68) the same stack is reached through hiertcn_amd.model_hier with args.model_low_type = 'gru'.
"""
from __future__ import annotations

import numpy as np
import torch

from .model import LazyLogits, _to_ids, get_scope


def model_gru(args, x, x_gap=None, x_impression=None, initial_state=None, return_state=False, name="gru", reuse=True):
    from .engine import check_supported
    check_supported(args)
    if initial_state is not None or return_state:
        raise NotImplementedError("model_gru: initial_state / return_state (model_gru.py:22-23,59-62) have no caller on the XING path (model.py:67)")
    ids = _to_ids(x)
    if ids.ndim != 2:
        raise NotImplementedError("model_gru takes ids [B, T] (the one-hot of model.py:56 is undone); the float-input form is reached through model_hier")
    if not name.startswith("gru"):
        raise ValueError("model_gru: scope names start with 'gru' (hiertcn_amd.model.get_scope picks the model kind from the scope)")
    model = get_scope(args, name)
    eng = model.engine
    B, T = ids.shape
    xd = torch.from_numpy(np.ascontiguousarray(ids)).to(eng.device)
    hidden = eng.tcn_forward(xd, None)
    return LazyLogits(eng, hidden, B, 1, T, [T])
