import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params
N, B, S, L = 20778, 32, 10, 20
for low, d in (("tcn", 256), ("gru", 128)):
    rng = np.random.default_rng(0)
    ids = rng.integers(1, N, (B, S, L)) * (np.arange(L)[None, None, :] < rng.integers(1, L + 1, (B, S))[..., None])
    x = np.zeros_like(ids); x[..., 1:] = ids[..., :-1]
    mask = (rng.random((B, S)) > 0.15).astype(np.float32)
    xs, ys, ms = (torch.as_tensor(v).cuda().contiguous() for v in (x.astype(np.int32), ids.astype(np.int32), mask))
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, batch_size=B, model_low_type=low)
    eng = Engine(args, hier=True, train=True)
    eng.set_weights(init_params(args, seed=2))
    st = torch.zeros((B, eng.HT), dtype=torch.float32, device="cuda")
    for i in range(5):
        eng.train_forward_backward(xs, ys, ms, st); eng.adam_step(1e-3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30):
        eng.train_forward_backward(xs, ys, ms, st); eng.adam_step(1e-3)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s d=%d: enqueue %.3f ms/step, total %.3f ms/step" % (low, d, (t1 - t0) / 30 * 1e3, (t2 - t0) / 30 * 1e3), flush=True)
    eng.close()
