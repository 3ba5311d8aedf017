"""Ranked top-k flavour of the scoring kernel against loss + rank alone.  usage: [D=512] python scripts/topk_bench.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params
d = int(os.environ.get("D", "256"))
rows, N = (614400, 300000) if d <= 256 else (307200, 300000)
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, tcn_channel=[d, d])
eng = Engine(args); eng.set_weights(init_params(args, 2))
hid = (torch.randn(rows, d, device="cuda") * 0.5).to(eng.act_dtype)
y = torch.randint(1, N, (rows,), device="cuda", dtype=torch.int32)
for k in (0, 10, 16):
    eng.score_fused(hid, y, topk=k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.score_fused(hid, y, topk=k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"topk={k}: {dt*1e3:.1f} ms, {2.0*rows*d*N/dt/1e12:.0f} TF/s", flush=True)
