"""Full-size check of everything in front of the scoring kernel (gather, scan, low level) against the oracle; lives under tests/ because it
uses oracle/ (checker code).  usage (GPU box): python tests/tools/soak_frontend.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params
B = 4096
N, d, S, L = 60000, bench.DIM, bench.S_SESS, bench.L_STEPS
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, max_activity_len=L, max_session_num=S, batch_size=B)
eng = Engine(args, hier=True); eng.set_weights(init_params(args, seed=2))
rng = np.random.default_rng(0)
ids = rng.integers(1, N, (B, S, L)).astype(np.int32)
x = np.zeros_like(ids); x[:, :, 1:] = ids[:, :, :-1]
mask = (rng.random((B, S)) > 0.1).astype(np.float32)
xd, yd, md = (torch.from_numpy(v).cuda() for v in (x, ids, mask))
st0 = (0.1 * torch.randn((B, eng.HT), device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))).float()
pooled = eng.session_pool(yd.view(B * S, L)).view(B, S, -1)
sproj, st_out, _ = eng.gru_scan(pooled, md, st0)
hid = eng.tcn_forward(xd.view(B * S, L), sproj.view(B * S, -1))
h = hid.view(torch.int16).to(torch.int64)
print("pooled", float(pooled.double().sum()), "state", float(st_out.double().sum()), "hidden-bits", int(h.sum()), int((h * (torch.arange(h.numel(), device="cuda").view_as(h) % 1009)).sum()))

# the scan at this size against the CPU oracle (fp64): pooled rows and the carried state
from oracle import hiertcn_oracle as O   # (a script, not the product: the oracle is the checker here as in tests/)
p = init_params(args, seed=2)
pr = O.round_params_bf16(p, E=d)
stn = st0.cpu().numpy().astype(np.float64)
worst_pool = 0.0
for s in range(S):
    pl = O.session_mean_pool(ids[:, s], pr)
    worst_pool = max(worst_pool, float(np.abs(pl - pooled[:, s].cpu().numpy()).max()))
    _, stn = O.multi_rnn_cell(pl, stn, pr, args.hidden_dim, args.num_layer)
    stn = stn * mask[:, s:s + 1]
print("vs oracle at B=%d: pooled max abs diff %.3e, final state max abs diff %.3e" % (B, worst_pool, float(np.abs(stn - st_out.cpu().numpy()).max())))
