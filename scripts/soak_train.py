"""One-off full-coverage check of the scoring layer's backward at the north-star catalogue size: EVERY entry of d Wo [256 x 300,000]
and d bo [300,000] of one training step against a direct evaluation from the HIP path's own hidden rows, logits and log-sum-exp
(the committed test samples 17 columns).  A stale LDS stage or a lost store in k_dlogits would show up as a wrong column here.
usage: python scripts/soak_train.py [users]   (48 users = 28,800 rows = two d-logits chunks; ~1 minute)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 48
N, d, S, L = 300_000, 256, 20, 30
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, max_activity_len=L, max_session_num=S,
                     batch_size=B)
p = init_params(args, seed=2)
rng = np.random.default_rng(11)
p["hier/tcn/dense/bias"] = (0.5 * rng.standard_normal(N)).astype(np.float32)
eng = Engine(args, hier=True, train=True)
eng.set_weights(p)
ids = np.minimum(1 + (rng.pareto(1.1, (B, S, L)) * 50).astype(np.int64), N - 1)
ids[:, :, 20:] *= (rng.random((B, S, 1)) > 0.3)
x = np.zeros_like(ids); x[:, :, 1:] = ids[:, :, :-1]
dev = lambda a, t=None: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0", dtype=t)
xd, yd = dev(x, torch.int32), dev(ids, torch.int32)
mask = dev((rng.random((B, S)) > 0.1).astype(np.float32))
st0 = torch.zeros((B, eng.HT), dtype=torch.float32, device="cuda:0")
eng.train_forward_backward(xd, yd, mask, st0)
g = eng.export("grads")
dW = torch.from_numpy(g["hier/tcn/dense/kernel"]).cuda()     # [d, N]
db = torch.from_numpy(g["hier/tcn/dense/bias"]).cuda().double()
pooled = eng.session_pool(yd.view(B * S, L)).view(B, S, -1)
sproj, _, _ = eng.gru_scan(pooled, mask, st0)
hid = eng.tcn_forward(xd.view(B * S, L), sproj.view(B * S, -1))
yv = yd.view(-1).long()
sc = eng.score_fused(hid, yd.view(-1), want_target=True)
valid = yv > 0
lse = (sc["ce"] + sc["target"])
act = valid.view(B, -1).sum(1).double()
users = (act > 0).sum().double()
w = ((1.0 / ((act + 1e-6) * users)).repeat_interleave(S * L) * valid).float()
dW_ref = torch.zeros((N, d), dtype=torch.float32, device="cuda")
db_ref = torch.zeros(N, dtype=torch.float64, device="cuda")
rows = torch.nonzero(valid).flatten()
for i in range(0, rows.numel(), 4096):
    r = rows[i:i + 4096]
    z = eng.score_logits(hid[r].contiguous())                        # [n, N] fp32, the HIP path's own logits
    gq = torch.exp(z - lse[r][:, None])
    gq[torch.arange(r.numel(), device="cuda"), yv[r]] -= 1.0
    gq *= w[r][:, None]
    db_ref += gq.double().sum(0)
    dW_ref += gq.bfloat16().float().T @ hid[r].float()              # d-logits rounded to bf16 as the device rounds them
    del z, gq
dW_ref = dW_ref.T
eW = (dW - dW_ref).abs()
colnorm = dW_ref.norm(dim=0) + 1e-12
rel_col = (dW - dW_ref).norm(dim=0) / colnorm
print(f"rows scored {rows.numel()}; d Wo: global rel L2 {float((dW - dW_ref).norm() / dW_ref.norm()):.3e}, worst column rel L2 {float(rel_col.max()):.3e} "
      f"(column {int(rel_col.argmax())}), max abs {float(eW.max()):.3e} of |ref| max {float(dW_ref.abs().max()):.3e}")
print(f"d bo: max abs diff {float((db - db_ref).abs().max()):.3e} of |ref| max {float(db_ref.abs().max()):.3e}; sum {float(db.sum()):.3e}")
bad = int((rel_col > 2e-2).sum())
print("columns with rel L2 > 2e-2:", bad)
sys.exit(1 if bad else 0)
