import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params
d = 256
for N, R in ((300000, 128), (300000, 256), (300000, 384), (300000, 768), (300000, 1536), (75000, 768), (150000, 768), (600000, 768)):
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, tcn_channel=[d, d])
    eng = Engine(args, dtype="fp32"); eng.set_weights(init_params(args, 2))
    hid = torch.randn(R, d, device="cuda")
    for _ in range(3): z = eng.score_logits(hid)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): z = eng.score_logits(hid)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    tiles = ((N + 127) // 128) * ((R + 127) // 128)
    print("R %5d N %6d tiles %6d (%.1f rounds of 768): %.3f ms = %.1f us per round, %.1f TFLOP/s" % (R, N, tiles, tiles / 768, dt * 1e3, dt * 1e6 / (tiles / 768), 2.0 * R * N * d / dt / 1e12), flush=True)
    eng.close(); del z
