"""loss.py surface of the reference (calc_loss :5, calc_metric_fast :163, calc_metric :108) over the
fused scoring kernel.  `pred` / `score` is the LazyLogits returned by model_hier / model_tcn; `y` is
the id tensor [B, T] (= concat(y_list), run_hier_xing.py:116) instead of its one-hot.

The model.py:105 masking (pred *= mask_y) is part of the fused kernel's contract: masked rows
behave as all-zero logit rows (rank 0, loss 0), so callers do not multiply anything.
"""
from __future__ import annotations

import torch

from .model import LazyLogits


def _need_lazy(pred):
    if not isinstance(pred, LazyLogits):
        raise TypeError("expected the pred handle returned by model_hier / model_tcn")


def calc_loss(pred, y, y_impression=None, loss="cross_entropy"):
    """loss.py:5-73, args.loss == 'cross_entropy' branch (:20-21): per-step softmax cross entropy
    [B, T]; 0 on masked steps.  The impression-based losses need y_impression, which is never fed on
    XING (model.py:108) -- out of scope."""
    _need_lazy(pred)
    if loss != "cross_entropy" or y_impression is not None:
        raise NotImplementedError("only the cross_entropy loss of the XING path is built")
    return pred.unpad(pred.scored(y)["ce"])


def activity_and_user_count(y):
    """model.py:62,112-114: mask_y, activity_count (+1e-6 AFTER user_count is taken), user_count."""
    mask_y = (torch.as_tensor(y) > 0).to(torch.float32)
    activity_count = mask_y.sum(1)
    user_count = torch.sign(activity_count).sum()
    return mask_y, activity_count + 1e-6, user_count


def calc_metric_fast(score, mask_y, activity_count, user_count, y):
    """loss.py:163-221 -> (recall1, recall5, recall10, mrr, rank_fraction, ranks_float [B,T], ranks [B,T]),
    the order of loss.py:221.  ranks = #{n : z_n > z_target} (strict, :179), from the fused kernel."""
    _need_lazy(score)
    sc = score.scored(y)
    dev = sc["rank"].device
    ranks = score.unpad(sc["rank"]).to(torch.float32)
    mask_y = torch.as_tensor(mask_y, dtype=torch.float32, device=dev)
    activity_count = torch.as_tensor(activity_count, dtype=torch.float32, device=dev)
    user_count = torch.as_tensor(user_count, dtype=torch.float32, device=dev)
    ranks_float = ranks / float(score.engine.N)              # :189-190
    rr = 1.0 / (1.0 + ranks)                                   # :191
    r1, r5, r10 = (ranks <= 0).float(), (ranks <= 4).float(), (ranks <= 9).float()  # :194-196
    ranks = ranks * mask_y                                     # :199-205
    ranks_float = ranks_float * mask_y
    rr, r1, r5, r10 = rr * mask_y, r1 * mask_y, r5 * mask_y, r10 * mask_y

    def user_mean(v):                                          # :208-219
        return (v.sum(1) / activity_count).sum() / user_count

    return user_mean(r1), user_mean(r5), user_mean(r10), user_mean(rr), user_mean(ranks_float), ranks_float, ranks


def calc_metric_topk(score, y_id, k: int = 10):
    """The ranked candidate list of loss.py:120 cut at k <= 16, from the fused kernel (no logits):
    -> (ranks_indices [B, T, k] int32, ranks_values [B, T, k] fp32).  Value descending, ties -> lower
    index first ([TF-sem] tf.nn.top_k); masked rows are all-zero rows (model.py:105) -> indices 0..k-1."""
    _need_lazy(score)
    sc = score.scored(y_id, topk=int(k))
    return score.unpad(sc["topk_idx"][:, :k]), score.unpad(sc["topk_val"][:, :k])


def calc_metric(score, mask_y, activity_count, user_count, y_id=None):
    """loss.py:108-160, the slow variant (commented out at model.py:128-129): a FULL top_k(k=N)
    over materialised logits -- parity/debug only, O(N log N) per step and 4N bytes per step of HBM
    (the ranked top-k for k <= 16 comes from the fused kernel: calc_metric_topk).
    Ties -> lower index first ([TF-sem] tf.nn.top_k), masked rows are all-zero rows (model.py:105).
    Returns the tuple of loss.py:160."""
    _need_lazy(score)
    logits = score.materialize()
    dev = logits.device
    y_id = torch.as_tensor(y_id, device=dev).to(torch.int64)
    mask_y = torch.as_tensor(mask_y, dtype=torch.float32, device=dev)
    activity_count = torch.as_tensor(activity_count, dtype=torch.float32, device=dev)
    user_count = torch.as_tensor(user_count, dtype=torch.float32, device=dev)
    logits = logits * mask_y[..., None]
    ranks_values, ranks_indices = torch.sort(logits, dim=-1, descending=True, stable=True)     # :120
    ranks = (ranks_indices == y_id[..., None]).to(torch.int32).argmax(-1).to(torch.float32)     # :124
    ranks_float = ranks / float(score.engine.N)
    rr = 1.0 / (1.0 + ranks)
    r1, r5, r10 = (ranks <= 0).float(), (ranks <= 4).float(), (ranks <= 9).float()
    ranks = ranks * mask_y
    ranks_float = ranks_float * mask_y
    rr, r1, r5, r10 = rr * mask_y, r1 * mask_y, r5 * mask_y, r10 * mask_y

    def user_mean(v):
        return (v.sum(1) / activity_count).sum() / user_count

    return (user_mean(r1), user_mean(r5), user_mean(r10), user_mean(rr), user_mean(ranks_float), ranks_float,
            ranks_indices, ranks_values, ranks)
