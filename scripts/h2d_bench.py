"""One-off: the bench step with the ids starting in HOST memory (pageable and pinned), to quote the PCIe-inclusive rate in DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
N, d, S, L = bench.N_ITEMS, bench.DIM, bench.S_SESS, bench.L_STEPS
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, max_activity_len=L,
                     max_session_num=S, batch_size=B)
eng = Engine(args, hier=True)
eng.set_weights(init_params(args, seed=2))
draw = bench.zipf_sampler(N)
host = [bench.make_users(draw, i * B, B, False) for i in range(4)]
pinned = [tuple(torch.from_numpy(v).pin_memory() for v in h) for h in host]
st0 = torch.zeros((B, eng.HT), dtype=torch.float32, device="cuda")
dx, dy, dm = (torch.from_numpy(v).cuda() for v in host[0])
for mode in ("resident", "pageable", "pinned"):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(4):
            if mode == "resident":
                x, y, m = dx, dy, dm
            elif mode == "pageable":
                x, y, m = (torch.from_numpy(v).cuda() for v in host[i])
            else:
                x, y, m = (v.cuda(non_blocking=True) for v in pinned[i])
            eng.forward_eval(x, y, m, st0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
    print(f"{mode}: {dt*1e3:.2f} ms/step, {B*S/dt:.0f} sessions/s", flush=True)
