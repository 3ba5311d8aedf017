"""Mixture head throughput (BASELINE.json configs[4]): hiertcn_gmm_loss on random hidden rows -- the Gram form (csrc/gmm_gram.hip, default) or,
with HIERTCN_GMM_TABLE_PASS=1, M passes of the scoring kernel with the sum-of-squares epilogue over the [D, C] mean tables; + prior + finish.
usage: [HIERTCN_GMM_TABLE_PASS=1 | SIGMA=none] python scripts/gmm_bench.py [rows] [N] [M]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import make_args
from hiertcn_amd.model_tcn_gmm import TcnGmmModel, init_params

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 614400
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20778
M = int(sys.argv[3]) if len(sys.argv) > 3 else 5
d = 256
a = make_args([], model_type="tcn_gmm", item_num=N, output_dim=N, num_mixture=M, sigma=0.1, loss="gmm_prob", tcn_channel=[d, d], kernel_size=5, adjust=False)
if os.environ.get("SIGMA", "").lower() == "none":   # a learned scale: n_out = M (1 + 2 D), one paired pass per mixture over the (means | log-scales) table
    a.sigma = None
model = TcnGmmModel(a, init_params(a, 3))
hid = (torch.randn(rows, d, device="cuda") * 0.5).to(torch.bfloat16)
y = torch.randint(1, N, (rows,), device="cuda", dtype=torch.int32)
model.loss_rows(hid, y); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    model.loss_rows(hid, y)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
if a.sigma is None:
    print(f"gmm loss (learned scale): rows {rows} N {N} M {M}: {dt*1e3:.2f} ms, {2.0*rows*d*((N+31)//32*64)*M/dt/1e12:.0f} TFLOP/s over the M interleaved tables", flush=True)
elif os.environ.get("HIERTCN_GMM_TABLE_PASS"):
    print(f"gmm loss (table pass): rows {rows} N {N} M {M}: {dt*1e3:.2f} ms, {2.0*rows*d*((N+63)//64*64)*M/dt/1e12:.0f} TFLOP/s over the M mean tables", flush=True)
else:
    print(f"gmm loss (Gram form): rows {rows} N {N} M {M}: {dt*1e3:.2f} ms = {rows/dt/1e6:.0f} M rows/s ({2.0*rows*3*d*d*M/dt/1e12:.0f} TFLOP/s on the three-part quadratic form)", flush=True)
model.close()
