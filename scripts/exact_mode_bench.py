"""Throughput of the exact fp32 evaluation mode (Engine(dtype='fp32'), csrc/exact.hip) at the cfg-3 shape, next to the production path on the
same handle.  It is a parity mode, not a fast path; this script documents what it costs.  usage: python scripts/exact_mode_bench.py [users]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N, d, S, L = bench.N_ITEMS, bench.DIM, bench.S_SESS, bench.L_STEPS
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, max_activity_len=L, max_session_num=S, batch_size=B)
eng = Engine(args, hier=True, dtype="fp32")
eng.set_weights(init_params(args, seed=2))
x, y, mask = bench.make_users(bench.zipf_sampler(N), 0, B, False)
xd, yd, md = (torch.from_numpy(v).cuda() for v in (x, y, mask))
st0 = torch.zeros((B, eng.HT), device="cuda")
for mode in ("fp32", "bf16"):
    eng.set_precision(mode)
    eng.forward_eval(xd, yd, md, st0); torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 2 if mode == "fp32" else 10
    for _ in range(n):
        eng.forward_eval(xd, yd, md, st0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    flop = 2.0 * B * S * L * d * N
    print(f"{mode}: {B} users x {S} x {L}, N = {N}: {dt * 1e3:.1f} ms per evaluation step = {B * S / dt:,.0f} sessions/s "
          f"(scoring product alone: {flop / dt / 1e12:.0f} TFLOP/s-equivalent)", flush=True)
eng.close()
