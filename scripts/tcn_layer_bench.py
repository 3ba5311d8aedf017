"""Layer-at-a-time low level (k_tcn_layer_lds / k_tcn_layer): the single-level model_tcn shape (6 blocks 128x4 + 256x2, T = 500, args.py:310-311)
and the d = 512 hier shape; with `eval` also the whole single-level evaluation step (embedding + blocks + scoring at the XING catalogue).
usage: python scripts/tcn_layer_bench.py [single|hier512|eval ...]   (HIERTCN_TCN_LAYER_V1=1: the un-staged kernel, for A/B)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

which = sys.argv[1:] or ["single", "hier512"]
SHAPES = {"single": ("single-level 6 blocks, T=500, 256 sequences", dict(item_num=20778, emb_dim=128, hidden_dim=128, tcn_channel=[128] * 4 + [256] * 2), False, 256, 500),
          "hier512": ("hier d=512, 20480 sessions x 30", dict(item_num=50000, emb_dim=512, hidden_dim=512, tcn_channel=[512, 512]), True, 20480, 30)}
for key in which:
    if key == "eval":
        continue
    name, kw, hier, B, T = SHAPES[key]
    args = hier_tcn_args(**kw)
    eng = Engine(args, hier=hier)
    eng.set_weights(init_params(args, 2, model="hier" if hier else "tcn") if not hier else init_params(args, 2))
    x = torch.randint(1, kw["item_num"], (B, T), device="cuda", dtype=torch.int32)
    sproj = torch.randn(B, kw["emb_dim"], device="cuda") * 0.1 if hier else None
    eng.tcn_forward(x, sproj); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        eng.tcn_forward(x, sproj)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    ch, K, cin, flop = kw["tcn_channel"], 5, kw["emb_dim"], 0.0
    for c in ch:
        flop += 2.0 * B * T * (K * cin * c + (cin * c if cin != c else 0))
        cin = c
    if hier:
        flop += 2.0 * B * T * kw["emb_dim"] ** 2
    print(f"{name}: {ms:.3f} ms  ({flop / ms / 1e9:.0f} TFLOP/s of bf16 MFMA work)", flush=True)
    eng.close()
if "eval" in which:   # BASELINE.json configs[0]'s model at the XING catalogue, wide batch: one evaluation step end to end
    kw = dict(item_num=20778, emb_dim=128, hidden_dim=128, tcn_channel=[128] * 4 + [256] * 2)
    args = hier_tcn_args(**kw)
    eng = Engine(args, hier=False)
    eng.set_weights(init_params(args, 2, model="tcn"))
    B, T = 256, 500
    ids = torch.randint(1, 20778, (B, 1, T + 1), device="cuda", dtype=torch.int32)
    x, y = ids[:, :, :-1].contiguous(), ids[:, :, 1:].contiguous()
    eng.forward_eval(x, y, None, None); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.forward_eval(x, y, None, None)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"single-level evaluation step, 256 users x 500 steps, N = 20,778: {ms:.3f} ms = {B * T / ms * 1e3 / 1e6:.2f} M steps/s", flush=True)
    eng.close()
