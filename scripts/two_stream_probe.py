"""Experiment: evaluation steps alternating between two HIP streams (two handles, two workspaces), so that the small kernels of step i + 1
(pool, GRU scan, low level) can fill the tail round of step i's scoring launch.  usage: python scripts/two_stream_probe.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = 4096
N, d, S, L = bench.N_ITEMS, bench.DIM, bench.S_SESS, bench.L_STEPS
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, max_activity_len=L, max_session_num=S, batch_size=B)
params = init_params(args, seed=2)
draw = bench.zipf_sampler(N)
batches = []
for i in range(4):
    x, y, mask = bench.make_users(draw, i * B, B, False)
    batches.append(tuple(torch.from_numpy(v).cuda() for v in (x, y, mask)))
for n_streams in (1, 2, 1, 2):
    engs = [Engine(args, hier=True) for _ in range(n_streams)]
    for e in engs:
        e.set_weights(params)
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    st0 = torch.zeros((B, engs[0].HT), device="cuda")
    outs = [torch.empty_like(st0) for _ in range(n_streams)]
    m8 = [torch.zeros(8, dtype=torch.float64, device="cuda") for _ in range(steps + 2)]
    torch.cuda.synchronize()

    def run(i0, i1):
        for i in range(i0, i1):
            k = i % n_streams
            with torch.cuda.stream(streams[k]):
                xb, yb, mb = batches[i % 4]
                engs[k].forward_eval(xb, yb, mb, st0, state_out=outs[k], metrics8=m8[i])
    run(0, 2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(2, steps + 2)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(f"{n_streams} stream(s): {dt * 1e3:.2f} ms per step, {B * S / dt:,.0f} sessions/s", flush=True)
    for e in engs:
        e.close()
