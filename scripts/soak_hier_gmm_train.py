"""Soak of model_hier's training step with a mixture low level (hiertcn_gmm_train_step_hier) at the reference's batch (32 lanes x 10 sessions x <= 20 steps,
XING catalogue, emb 128, 2 x 128 across-session cells, 5 mixtures):
 * the forward is repeatable bit for bit (loss rows and the carried state of `n` passes over one batch), the gradients within 1e-5 relative (fp32 atomics
   in the item table's scatter-adds reorder sums);
 * `steps` optimizer steps over 8 rotating ragged batches with the state carried from step to step: finite everywhere, the loss falls, and a second handle
   fed the same batches ends on the same weights and the same carried state;
 * the evaluation path (both handles, after sync) agrees with the training forward: loss 2e-3 (16-bit tables vs fp32 masters), state 1e-3.
usage: python scripts/soak_hier_gmm_train.py [tcn_gmm|gru_gmm] [n] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.model_hier_gmm import HierGmmModel, init_params

low = sys.argv[1] if len(sys.argv) > 1 else "tcn_gmm"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
N, M, B, S, L = 20_778, 5, 32, 10, 20
rng = np.random.default_rng(9)


def batch():
    lens = rng.integers(1, L + 1, (B, S))                                  # every session holds an event (an empty one is the reference's 0 / 0)
    y = rng.integers(1, N, (B, S, L)) * (np.arange(L)[None, None, :] < lens[..., None])
    x = np.zeros_like(y)
    x[..., 1:] = y[..., :-1]
    mask = (rng.random((B, S)) > 0.1).astype(np.float32)                   # user boundaries
    dev = lambda v, t: torch.as_tensor(np.ascontiguousarray(v.astype(t))).cuda()  # noqa: E731
    return dev(x, np.int32), dev(y, np.int32), dev(mask, np.float32)


a = hier_tcn_args(item_num=N, output_dim=N, emb_dim=128, hidden_dim=128, num_layer=2, tcn_channel=[128, 128], kernel_size=5, num_mixture=M,
                  model_low_type=low, loss="gmm_prob", sigma=0.1)
p = init_params(a, 3)
m1, m2 = HierGmmModel(a, p, "hier"), HierGmmModel(a, p, "hier")
m1.enable_training()
m2.enable_training()
x, y, mask = batch()
st0 = (0.2 * torch.randn((B, 256), generator=torch.Generator().manual_seed(1))).cuda()
first = m1.train_forward_backward_hier(x, y, mask, st0, want_rows=True)
rows0, state0, g0 = first["loss_rows"].clone(), first["state"].clone(), m1.arena(1).clone()
worst = 0.0
for i in range(n):
    out = m1.train_forward_backward_hier(x, y, mask, st0, want_rows=True)
    assert torch.equal(out["loss_rows"], rows0) and torch.equal(out["state"], state0), "pass %d: the forward is not repeatable" % i
    worst = max(worst, float((m1.arena(1) - g0).abs().max() / g0.abs().max()))
assert worst < 1e-5, worst
print(f"model_hier + {low}: {n} passes: loss rows and carried state bitwise, gradients within {worst:.1e} of the first pass (relative to the largest)")
batches = [batch() for _ in range(8)]
s1 = s2 = torch.zeros((B, 256), dtype=torch.float32, device="cuda")
losses = []
for i in range(steps):
    xb, yb, mb = batches[i % 8]
    o1 = m1.train_forward_backward_hier(xb, yb, mb, s1)
    m1.adam_step(1e-3)
    o2 = m2.train_forward_backward_hier(xb, yb, mb, s2)
    m2.adam_step(1e-3)
    s1, s2 = o1["state"], o2["state"]
    losses.append(o1["sums"])
losses = torch.stack(losses).cpu().numpy()
losses = losses[:, 0] / losses[:, 1]
assert np.isfinite(losses).all() and bool(torch.isfinite(m1.arena(0)).all()) and bool(torch.isfinite(s1).all())
assert losses[-8:].mean() < losses[:8].mean(), (losses[:8].mean(), losses[-8:].mean())
dw, ds = float((m1.arena(0) - m2.arena(0)).abs().max()), float((s1 - s2).abs().max())
assert dw < 1e-4 and ds < 1e-4, (dw, ds)
print(f"model_hier + {low}: {steps} Adam steps, state carried: loss {losses[:8].mean():.7f} -> {losses[-8:].mean():.7f}; two handles end {dw:.1e} apart "
      f"(max abs weight difference), carried states {ds:.1e}")
xb, yb, mb = batches[0]
tr = m1.train_forward_backward_hier(xb, yb, mb, st0, want_rows=True)
ev = m1.eval_step_hier(xb, yb, mb, st0)
t_loss = float(tr["sums"][0] / tr["sums"][1])
e_loss = float(ev["sums"][0] / ev["sums"][1])
es = float((tr["state"] - ev["state"]).abs().max())
assert abs(t_loss - e_loss) < 2e-3 * max(1.0, abs(t_loss)) and es < 1e-3, (t_loss, e_loss, es)
print(f"model_hier + {low}: evaluation (both handles, after sync) vs the training forward: loss {e_loss:.6f} vs {t_loss:.6f}, carried state within {es:.1e}")
m1.close()
m2.close()
print("soak_hier_gmm_train: ok")
