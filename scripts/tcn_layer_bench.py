"""Layer-at-a-time low level (k_tcn_layer): the single-level model_tcn shape (6 blocks 128x4 + 256x2, T = 500) and the d = 512 hier shape.
usage: python scripts/tcn_layer_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

for name, kw, hier, B, T in (("single-level 6 blocks, T=500, 256 sequences", dict(item_num=20778, emb_dim=128, hidden_dim=128, tcn_channel=[128] * 4 + [256] * 2), False, 256, 500),
                             ("hier d=512, 20480 sessions x 30", dict(item_num=50000, emb_dim=512, hidden_dim=512, tcn_channel=[512, 512]), True, 20480, 30)):
    args = hier_tcn_args(**kw)
    eng = Engine(args, hier=hier)
    eng.set_weights(init_params(args, 2, model="hier" if hier else "tcn") if not hier else init_params(args, 2))
    x = torch.randint(1, kw["item_num"], (B, T), device="cuda", dtype=torch.int32)
    sproj = torch.randn(B, kw["emb_dim"], device="cuda") * 0.1 if hier else None
    eng.tcn_forward(x, sproj); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        eng.tcn_forward(x, sproj)
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms", flush=True)
    eng.close()
