"""Soak of the persistent GRU scan's inter-workgroup hand-off: thousands of launches over several batch shapes on one handle, interleaved with
other kernels, every result compared bit for bit with the first launch of its shape; then the handle's time-out word.
usage: python scripts/soak_gru_persist.py [launches]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
d, nl = 256, 2
args = hier_tcn_args(item_num=500, emb_dim=d, hidden_dim=d, tcn_channel=[d, d], num_layer=nl)
eng = Engine(args)
eng.set_weights(init_params(args, 5))
rng = np.random.default_rng(1)
shapes = [(32, 10), (5, 3), (64, 7), (96, 4), (1, 1), (33, 20)]
data = []
for B, S in shapes:
    pooled = torch.from_numpy(rng.standard_normal((B, S, d)).astype(np.float32) * 0.3).cuda()
    mask = torch.from_numpy((rng.random((B, S)) > 0.25).astype(np.float32)).cuda()
    st0 = torch.from_numpy(rng.standard_normal((B, d * nl)).astype(np.float32) * 0.2).cuda()
    data.append((pooled, mask, st0))
first = [[t.clone() for t in eng.gru_scan(*dat, want_state_before=True)] for dat in data]
junk = torch.randn(4096, 4096, device="cuda")
bad = 0
for it in range(n):
    i = int(rng.integers(len(shapes)))
    if it % 7 == 0:
        junk = junk @ junk * 1e-4          # another kernel in the stream between two scans
    out = eng.gru_scan(*data[i], want_state_before=True)
    for g, f in zip(out, first[i]):
        if not torch.equal(g, f):
            bad += 1
torch.cuda.synchronize()
print(f"{n} launches over {len(shapes)} shapes: {bad} mismatching outputs; timed out: {eng.persistent_timed_out()}")
eng.close()
