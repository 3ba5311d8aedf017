"""k_xgemm (csrc/exact.hip) on logits-shaped products: time against K at the bench's chunk shape (768 rows x 300,000 items), and the same FLOP and
output bytes with ten times the rows and a tenth of the items (rows of the output 120 KB apart instead of 1.2 MB).  usage: python scripts/xgemm_k_sweep.py"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params
for N, R, d in ((300000, 768, 128), (300000, 768, 256), (300000, 768, 512), (30000, 7680, 256), (3000, 76800, 256)):
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, tcn_channel=[d, d])
    eng = Engine(args, dtype="fp32"); eng.set_weights(init_params(args, 2))
    hid = torch.randn(R, d, device="cuda")
    for _ in range(3): z = eng.score_logits(hid)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): z = eng.score_logits(hid)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print("k_xgemm logits %d x %d x %d: %.3f ms = %.1f TFLOP/s (%.2f of 157)" % (R, N, d, dt * 1e3, 2.0 * R * N * d / dt / 1e12, 2.0 * R * N * d / dt / 157e12), flush=True)
    eng.close(); del z
