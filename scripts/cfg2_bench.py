"""Evaluation at the reference's own batching (BASELINE.json configs[1]: model_hier d=256, XING catalogue N=20,778,
B=32 lanes x S=10 slots x L<=20 steps per batch): batches/s and sessions/s, ids resident on the device.
usage: [LOW=gru] python scripts/cfg2_bench.py [lanes] [batches]     (LOW=gru: HierGRU, the reference's default low level)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params



def make_batch(rng, B, S, L, N):
    """ragged ids [B,S,L] (x = y shifted by one step, data_loader.py:246-261) and the last-session mask"""
    ids = rng.integers(1, N, (B, S, L)) * (np.arange(L)[None, None, :] < rng.integers(1, L + 1, (B, S))[..., None])
    x = np.zeros_like(ids)
    x[:, :, 1:] = ids[:, :, :-1]
    return x.astype(np.int32), ids.astype(np.int32), (rng.random((B, S)) > 0.15).astype(np.float32)


B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
N, d, S, L = 20778, int(os.environ.get("D", "256")), 10, 20
args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, batch_size=B,
                     model_low_type=os.environ.get("LOW", "tcn"))
eng = Engine(args, hier=True, **({"dtype": os.environ["DTYPE"]} if os.environ.get("DTYPE") else {}))   # DTYPE=split16|fp32: the parity-grade modes
eng.set_weights(init_params(args, seed=2))
rng = np.random.default_rng(0)
batches = []
for _ in range(8):
    x, y, mask = make_batch(rng, B, S, L, N)
    batches.append(tuple(torch.as_tensor(v).cuda().contiguous() for v in (x, y, mask)))
st = torch.zeros((B, eng.HT), dtype=torch.float32, device="cuda")
st2 = torch.empty_like(st)
m8 = torch.zeros(8, dtype=torch.float64, device="cuda")
for i in range(10):
    eng.forward_eval(*batches[i % 8], st, state_out=st2, metrics8=m8)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n):
    eng.forward_eval(*batches[i % 8], st, state_out=st2, metrics8=m8)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("cfg2: lanes %d: %.3f ms per batch, %.0f sessions/s" % (B, dt * 1e3, B * S / dt), flush=True)
# the same step replayed from a hipGraph, carried state chained on the device
g = eng.capture_eval_graph(B, S, L)
for i in range(10):
    g.x.copy_(batches[i % 8][0]); g.y.copy_(batches[i % 8][1]); g.mask.copy_(batches[i % 8][2]); g.replay(); g.chain_state()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(n):
    xb, yb, mb = batches[i % 8]
    g.x.copy_(xb, non_blocking=True); g.y.copy_(yb, non_blocking=True); g.mask.copy_(mb, non_blocking=True)
    g.replay(); g.chain_state()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("cfg2: lanes %d, hipGraph replay: %.3f ms per batch, %.0f sessions/s" % (B, dt * 1e3, B * S / dt), flush=True)
