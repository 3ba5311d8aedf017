"""The fp32 TRAINING mode at the reference's own batch (32 lanes x 10 slots x <= 20 steps, XING catalogue): ms per step for model_hier with the
TemporalConvNet / GRU low level; under `rocprofv3 --kernel-trace --stats` the per-kernel split.  usage: [LOW=tcn|gru] [D=256] python scripts/fp32_train_bench.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
low, d = os.environ.get("LOW", "tcn"), int(os.environ.get("D", "256"))
N, B, S, L = 20_778, 32, 10, 20
rng = np.random.default_rng(0)
ids = rng.integers(1, N, (B, S, L)) * (np.arange(L)[None, None, :] < rng.integers(1, L + 1, (B, S))[..., None])
x = np.zeros_like(ids); x[..., 1:] = ids[..., :-1]
mask = (rng.random((B, S)) > 0.15).astype(np.float32)
xs, ys, ms = (torch.as_tensor(v).cuda().contiguous() for v in (x.astype(np.int32), ids.astype(np.int32), mask))
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, batch_size=B, model_low_type=low)
eng = Engine(args, hier=True, train=True, dtype="fp32")
eng.set_weights(init_params(args, seed=2))
st = torch.zeros((B, eng.HT), dtype=torch.float32, device="cuda")
for _ in range(5):
    eng.train_forward_backward(xs, ys, ms, st); eng.adam_step(1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    eng.train_forward_backward(xs, ys, ms, st); eng.adam_step(1e-3)
torch.cuda.synchronize()
print("fp32 training step, %s d=%d, 32 x 10 x 20, N = 20,778: %.3f ms per step" % (low, d, (time.perf_counter() - t0) / steps * 1e3))
