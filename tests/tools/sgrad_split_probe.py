"""Run by tests/test_gpu_train.py twice in processes of their own (the library reads HIERTCN_SGRAD_DYN_MAX_BLOCKS once): one training forward + backward
of a 260-lane batch (52,000 padded rows, ~152 live row blocks of 256, XING-sized catalogue) -- with the default the scoring backward's first kernel deals
(row block, table part) on the device over a grid of several rounds of the chip (score_bwd.hip, sg_dyn_grid: 5 parts here), with the variable at 127 every
row block walks the whole table.  Prints the loss, the carried state's and every gradient's sum of squares and a strided sample of entries as JSON."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine, metrics_dict
from hiertcn_amd.params import init_params

N, d, B, S, L = 20778, 256, 260, 10, 20
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, batch_size=B)
eng = Engine(args, hier=True, train=True)
eng.set_weights(init_params(args, seed=3))
rng = np.random.default_rng(11)
lens = rng.integers(L // 2, L + 1, (B, S))                                  # ~75 % of the padded rows carry a target
y = (rng.integers(1, N, (B, S, L)) * (np.arange(L)[None, None, :] < lens[..., None])).astype(np.int32)
x = np.zeros_like(y)
x[:, :, 1:] = y[:, :, :-1]
mask = (rng.random((B, S)) > 0.15).astype(np.float32)
st0 = (0.3 * rng.standard_normal((B, eng.HT))).astype(np.float32)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda().contiguous()  # noqa: E731
out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
torch.cuda.synchronize()
g = eng.export("grads")
res = {"loss": metrics_dict(out["metrics8"])["loss"], "live_rows": int((y > 0).sum()),
       "state": float((out["state"].double() ** 2).sum()),
       "grads": {k: [float((v.astype(np.float64) ** 2).sum())] + [float(t) for t in v.reshape(-1)[:: max(1, v.size // 64)][:64]] for k, v in sorted(g.items())}}
eng.close()
print("SGRAD_SPLIT_PROBE " + json.dumps(res))
