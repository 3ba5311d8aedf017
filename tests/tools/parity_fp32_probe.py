"""How far are the device's fp32 logits from the reference arithmetic on UNROUNDED fp32 weights (what a TF checkpoint holds)?
Prints max / quantiles of |device - O.model_hier(..., p_unrounded, EXACT64)| for the shipped bf16 path (and, when present,
the exact-fp32 parity mode) at BASELINE.json configs[1]'s shape and at the north star's catalogue width.

    python tests/tools/parity_fp32_probe.py            (on the GPU box)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hiertcn_amd.engine import Engine  # noqa: E402
from oracle import hiertcn_oracle as O  # noqa: E402
from tests.gpu_common import as_lists, make_batch, make_params, small_args  # noqa: E402


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")


def probe(name, N, d, B, S, L, **eng_kw):
    args = small_args(N=N, d=d, H=d, nl=2, blocks=2, K=5, L=L, S=S, B=B)
    p = make_params(args)
    eng = Engine(args, hier=True, **eng_kw)
    eng.set_weights(p)
    rng = np.random.default_rng(N)
    x, y, mask = make_batch(rng, B, S, L, N, ragged=True, zipf=True)
    st0 = (0.2 * rng.standard_normal((B, 2 * d))).astype(np.float32)
    pooled = eng.session_pool(dev(y.reshape(B * S, L))).view(B, S, -1)
    sproj, _, _ = eng.gru_scan(pooled, dev(mask), dev(st0))
    hid = eng.tcn_forward(dev(x.reshape(B * S, L)), sproj.view(B * S, -1))
    logits = eng.score_logits(hid).cpu().numpy().reshape(B, S * L, N)
    xl, yl, ml = as_lists(x, y, mask)
    exact, _ = O.model_hier(args, xl, yl, ml, st0, p, O.EXACT64)
    mirror, _ = O.model_hier(args, xl, yl, ml, st0, O.round_params_bf16(p, E=d), O.MIRROR)
    valid = y.reshape(B, S * L) > 0
    e_exact = np.abs(logits - exact)[valid]
    e_mirror = np.abs(logits - mirror)[valid]
    out = dict(shape=name, N=N, d=d, rows=int(valid.sum()), logit_abs_max=float(np.abs(exact[valid]).max()),
               logit_std=float(exact[valid].std()),
               vs_exact_unrounded=dict(max=float(e_exact.max()), q999=float(np.quantile(e_exact, 0.999)), mean=float(e_exact.mean())),
               vs_mirror=dict(max=float(e_mirror.max()), q999=float(np.quantile(e_mirror, 0.999)), mean=float(e_mirror.mean())))
    print(json.dumps(out), flush=True)
    eng.close()
    return out


if __name__ == "__main__":
    res = [probe("cfg2-xing", 20_778, 256, 32, 10, 20), probe("north-star-width", 300_000, 256, 4, 5, 30)]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "parity_fp32_probe.json"), "w") as f:
        json.dump(res, f, indent=1)
