"""model_hier with the reference's signature (model_hier.py:21), backed by the HIP kernels.

    model_hier(args, x, y, mask, state, x_gap=None, x_impression=None, name='hier',
               reuse=True, training=False) -> (pred_all, state)

x, y   : lists of S id arrays [B, L_i]  (the reference feeds one_hot(ids)*sign(ids), model.py:59-61;
         the one-hot is an artefact of using a dense layer as lookup -- ids are taken directly, the
         row-0-is-zero semantics kept)
mask   : list of S arrays [B, 1]                       (data_loader.py:186-187,254)
state  : [B, hidden_dim*num_layer]
returns: pred_all  a LazyLogits standing for the [B, sum L_i, N] logits of model_hier.py:79
         state     torch.float32 [B, hidden_dim*num_layer] on the GPU (model_hier.py:93-94)

Only the default-graph branches are implemented: model_low_type 'tcn', 'gru', 'tcn_gmm' or 'gru_gmm' (model_hier.py:62-69), no gap decay, no
impressions, no noise, no batch/weight/layer norm; anything else raises.
"""
from __future__ import annotations

import numpy as np
import torch

from .model import LazyLogits, get_scope, pack_slots


def _check_args(args):
    from .engine import check_supported
    if getattr(args, "model_low_type", "tcn") not in ("tcn", "gru"):
        raise NotImplementedError(f"model_low_type={args.model_low_type!r}: 'tcn' (HierTCN), 'gru' (HierGRU), 'tcn_gmm' and 'gru_gmm' are built")
    check_supported(args)   # (Engine() runs the same check: the run loops and the CLI reach it too)


def model_hier(args, x, y, mask, state, x_gap=None, x_impression=None, name="hier", reuse=True, training=False):
    if getattr(args, "model_low_type", "tcn") in ("tcn_gmm", "gru_gmm"):   # model_hier.py:64-65,68-69: pred is a mixture head (model_hier_gmm.py)
        from .model_hier_gmm import model_hier_tcn_gmm
        return model_hier_tcn_gmm(args, x, y, mask, state, name)
    _check_args(args)
    model = get_scope(args, name)
    eng = model.engine
    xs, ys, ms, lens = pack_slots(x, y, mask)
    B, S, Lmax = xs.shape
    dev = eng.device
    xd, yd, md = (torch.from_numpy(v).to(dev) for v in (xs, ys, ms))
    if isinstance(state, torch.Tensor):
        st = state.to(device=dev, dtype=torch.float32).contiguous()
    else:
        st = torch.as_tensor(np.asarray(state, dtype=np.float32)).to(dev).contiguous()
    # scan first: the high level never consumes the low level's output (model_hier.py:83-93)
    pooled = eng.session_pool(yd.view(B * S, Lmax)).view(B, S, -1)
    sproj, state_out, _ = eng.gru_scan(pooled, md, st)
    hidden = eng.tcn_forward(xd.view(B * S, Lmax), sproj.view(B * S, -1))
    return LazyLogits(eng, hidden, B, S, Lmax, lens), state_out
