"""Shared helpers for the -m gpu parity tests (HIP path vs the CPU oracle, same seeded inputs)."""
import numpy as np

from hiertcn_amd.args import hier_tcn_args
from oracle import hiertcn_oracle as O


def small_args(N=1000, d=64, H=32, nl=2, blocks=2, K=5, L=20, S=10, B=32):
    return hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=H, num_layer=nl, tcn_channel=[d] * blocks, kernel_size=K,
                         max_activity_len=L, max_session_num=S, batch_size=B)


def make_params(args, seed=2, bias_noise=0.05):
    """glorot weights (oracle.init_params) with non-zero biases so that a dropped bias shows."""
    p = O.init_params(args, seed=seed)
    rng = np.random.default_rng(seed + 100)
    for k in p:
        if k.endswith("/bias") and "gates" not in k:
            p[k] = (p[k] + bias_noise * rng.standard_normal(p[k].shape)).astype(np.float32)
    return p


def make_batch(rng, B, S, L, N, ragged=True, zipf=False):
    """ids [B,S,L]: x[:,:,0]=0, x[t]=ids[t-1], y=ids (data_loader.py:246-261); mask [B,S]."""
    if zipf:
        w = 1.0 / np.arange(1, N) ** 1.1
        cdf = np.cumsum(w / w.sum())
        ids = 1 + np.searchsorted(cdf, rng.random((B, S, L)))
        ids = np.minimum(ids, N - 1)
    else:
        ids = rng.integers(1, N, (B, S, L))
    if ragged:
        lens = rng.integers(1, L + 1, (B, S))
        ids = ids * (np.arange(L)[None, None, :] < lens[..., None])
    x = np.zeros_like(ids)
    x[:, :, 1:] = ids[:, :, :-1]
    y = ids
    mask = (rng.random((B, S)) > 0.15).astype(np.float32)
    return x.astype(np.int32), y.astype(np.int32), mask


def as_lists(x, y, mask):
    S = x.shape[1]
    return [x[:, s] for s in range(S)], [y[:, s] for s in range(S)], [mask[:, s:s + 1] for s in range(S)]


class OracleSession:
    """A model handle for the run loops backed by the CPU oracle (test infrastructure only): the
    same `eval_step` fetch list as hiertcn_amd.model.HierTCNModel, computed by oracle.model_hier +
    oracle.masked_loss_and_metrics."""

    def __init__(self, args, params, num=None):
        self.args, self.params, self.num = args, params, num or O.EXACT64

    def eval_step(self, x_list, y_list, mask_list, state_np, mask_warmstart=None):
        pred, state = O.model_hier(self.args, x_list, y_list, mask_list, np.asarray(state_np, np.float64), self.params, self.num)
        y_id = np.concatenate([np.asarray(v).astype(np.int64) for v in y_list], axis=1)
        if mask_warmstart is not None:   # model.py:102-103: mask_y = sign(y_id) * mask_warmstart -- the model above still pooled the unmasked y_list
            y_id = y_id * (np.asarray(mask_warmstart) > 0)
        m = O.masked_loss_and_metrics(pred, y_id, int(self.args.item_num))
        return dict(loss=float(m["loss"]), state=state, ranks_float_mean_user=float(m["rank_frac"]),
                    ranks_float=m["ranks_float"], mask_y=(y_id > 0).astype(np.float64),
                    reciprocal_ranks_float_mean_user=float(m["mrr"]), recall1_mean_user=float(m["recall1"]),
                    recall5_mean_user=float(m["recall5"]), recall10_mean_user=float(m["recall10"]), ranks=m["ranks"])
