"""The split-operand scoring kernel (HIERTCN_PRECISION_SPLIT16, csrc/score_x3.hip) at the cfg-3 width: kernel time of loss + rank and of the
ranked top-10 flavour on 614,400 rows x 300,000 items (as scripts/topk_bench.py for the 16-bit kernels), then the whole evaluation step
next to the 16-bit mode's.  usage: python scripts/split16_bench.py [users]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N, d, S, L = bench.N_ITEMS, bench.DIM, bench.S_SESS, bench.L_STEPS
rows = 614400
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, max_activity_len=L, max_session_num=S, batch_size=B)
eng = Engine(args, hier=True, dtype="split16")
eng.set_weights(init_params(args, seed=2))
hid = torch.relu(torch.randn(rows, d, device="cuda")) * 0.5
y = torch.randint(1, N, (rows,), device="cuda", dtype=torch.int32)
for k in (0, 10):
    eng.score_fused(hid, y, topk=k); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): eng.score_fused(hid, y, topk=k)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
    print(f"split16 topk={k}: {dt*1e3:.1f} ms per {rows} x {N} launch = {3 * 2.0*rows*d*N/dt/1e12:.0f} TF/s executed (3 MFMAs per product), "
          f"{2.0*rows*d*N/dt/1e12:.0f} TF/s algorithmic", flush=True)
del hid, y
x, yb, mask = bench.make_users(bench.zipf_sampler(N), 0, B, False)
xd, yd, md = (torch.from_numpy(v).cuda() for v in (x, yb, mask))
st0 = torch.zeros((B, eng.HT), device="cuda")
eng.forward_eval(xd, yd, md, st0); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3): eng.forward_eval(xd, yd, md, st0)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
print(f"split16 evaluation step: {B} users x {S} x {L}, N = {N}: {dt*1e3:.1f} ms = {B*S/dt:,.0f} sessions/s", flush=True)
eng.close()
