"""Diagnostic: time hiertcn_score_fused alone for several catalogue sizes (random hidden rows).
usage: python scripts/score_diag.py [rows] [N ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2048 * 600
Ns = [int(v) for v in sys.argv[2:]] or [300000, 65536, 8192]
d = int(os.environ.get("D", "256"))
for N in Ns:
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, tcn_channel=[d, d])
    p = init_params(args, 2)
    if os.environ.get("ZERO_TABLE"):   # data-dependence probe: same instruction stream, no operand toggling
        p["hier/tcn/dense/kernel"] = np.zeros_like(p["hier/tcn/dense/kernel"])
    eng = Engine(args, dtype=os.environ.get("DTYPE", "bf16"))
    eng.set_weights(p)
    hid = (torch.randn(rows, d, device="cuda") * float(os.environ.get("HID_SCALE", "0.5"))).to(eng.act_dtype)
    y = torch.randint(1, N, (rows,), device="cuda", dtype=torch.int32)
    eng.score_fused(hid, y)
    torch.cuda.synchronize()
    eng.timing_enable(True); eng.timing_read(True)
    for _ in range(3):
        eng.score_fused(hid, y)
    torch.cuda.synchronize()
    ms, n = eng.timing_read(True)
    NP = (N + 63) // 64 * 64
    tf = 2.0 * rows * d * N / (ms * 1e-3) / 1e12
    stages = NP // 64
    wgs = (rows + 511) // 512
    rounds = -(-wgs // 256)
    print(f"N={N} rows={rows} ms={ms:.3f} TF={tf:.1f} ns/stage/round={ms*1e6/(stages*rounds):.1f}", flush=True)
    eng.close()
