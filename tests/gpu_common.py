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
