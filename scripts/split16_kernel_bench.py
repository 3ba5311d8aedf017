"""The split-operand scoring kernels alone (Engine(dtype='split16').score_fused on resident fp32 hidden rows): ms per launch and TFLOP/s
(algorithmic = one product per logit; executed = three MFMAs per product) at a given width / catalogue / row count.
usage: python scripts/split16_kernel_bench.py d N rows [topk] [reps]      e.g. 512 1000000 307200   (BASELINE.json configs[3]) or 256 300000 614400"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

d, N, R = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
topk = int(sys.argv[4]) if len(sys.argv) > 4 else 0
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, max_activity_len=30, max_session_num=20, batch_size=64)
eng = Engine(args, hier=True, dtype="split16")
p = init_params(args, seed=2)
p["hier/tcn/dense/kernel"] = (p["hier/tcn/dense/kernel"] * 30).astype(np.float32)     # a trained model's logit range rather than glorot's
eng.set_weights(p)
g = torch.Generator("cuda:0").manual_seed(1)
hid = torch.relu(torch.randn((R, d), device="cuda:0", generator=g))
y = torch.randint(1, N, (R,), device="cuda:0", dtype=torch.int32, generator=g)
eng.score_fused(hid, y, topk=topk); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ms = []
for _ in range(reps):
    e0.record(); eng.score_fused(hid, y, topk=topk); e1.record(); torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
ms = float(np.median(ms))
flop = 2.0 * R * d * N
print(f"split16 d={d} N={N} rows={R} topk={topk}: {ms:.2f} ms per launch (median of {reps}) = {flop / ms / 1e9:.0f} TFLOP/s algorithmic = {flop / ms / 1e9 / 2500:.3f} of 2.5 PF, "
      f"{3 * flop / ms / 1e9:.0f} TFLOP/s executed = {3 * flop / ms / 1e9 / 2500:.3f}", flush=True)
eng.close()
