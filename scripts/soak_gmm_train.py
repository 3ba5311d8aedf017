"""Soak of the mixture models' training step (csrc/gmm_train.hip) at the reference's batch (32 lanes x 200 steps, XING catalogue, d = 256, 5 mixtures):
 * the forward is repeatable bit for bit (loss rows of `n` passes over the same batch on the same weights: the v_m / |b_m|^2 reductions run in a fixed order,
   every product is split in a fixed order), the gradients within 1e-5 relative (fp32 atomics in the scatter-adds and column sums reorder sums);
 * `steps` optimizer steps over 8 rotating batches: finite everywhere, the loss falls, and a second handle fed the same batches ends on the same weights;
 * the evaluation path after hiertcn_gmm_train_sync agrees with the training forward's loss rows (16-bit tables against fp32 masters: 2e-3).
usage: python scripts/soak_gmm_train.py [tcn|gru] [n] [steps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hiertcn_amd.args import make_args

which = sys.argv[1] if len(sys.argv) > 1 else "tcn"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 200
N, M, d, B, T = 20_778, 5, 256, 32, 200
rng = np.random.default_rng(5)


def batch():
    ids = rng.integers(1, N, (B, T + 1))
    ids = ids * (np.arange(T + 1)[None, :] <= rng.integers(2, T + 1, B)[:, None])
    dev = lambda v: torch.as_tensor(np.ascontiguousarray(v.astype(np.int32))).cuda()  # noqa: E731
    return dev(ids[:, :-1]), dev(ids[:, 1:])


if which == "tcn":
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel as Model, init_params
    a = make_args([], model_type="tcn_gmm", item_num=N, output_dim=N, num_mixture=M, sigma=0.1, loss="gmm_prob", tcn_channel=[d, d], kernel_size=5, adjust=False)
else:
    from hiertcn_amd.model_gru_gmm import GruGmmModel as Model, init_params
    a = make_args([], model_type="gru_gmm", item_num=N, output_dim=N, num_mixture=M, sigma=0.1, loss="gmm_prob", hidden_dim=d, num_layer=2, adjust=False)
p = init_params(a, 3)
m1, m2 = Model(a, p), Model(a, p)
m1.enable_training()
m2.enable_training()
x, y = batch()
first = m1.train_forward_backward(x, y, want_rows=True)
rows0 = first["loss_rows"].clone()
g0 = m1.arena(1).clone()
worst = 0.0
for i in range(n):
    out = m1.train_forward_backward(x, y, want_rows=True)
    assert torch.equal(out["loss_rows"], rows0), "pass %d: the forward is not repeatable" % i
    g = m1.arena(1)
    worst = max(worst, float((g - g0).abs().max() / g0.abs().max()))
assert worst < 1e-5, worst
print(f"model_{which}_gmm: {n} passes: loss rows bitwise, gradients within {worst:.1e} of the first pass (relative to the largest)")
batches = [batch() for _ in range(8)]
losses = []
for i in range(steps):
    xb, yb = batches[i % 8]
    s1 = m1.train_forward_backward(xb, yb)["sums"]
    m1.adam_step(1e-3)
    m2.train_forward_backward(xb, yb)
    m2.adam_step(1e-3)
    losses.append(s1)
losses = torch.stack(losses).cpu().numpy()
losses = losses[:, 0] / losses[:, 1]
assert np.isfinite(losses).all() and bool(torch.isfinite(m1.arena(0)).all())
assert losses[-8:].mean() < losses[:8].mean(), (losses[:8].mean(), losses[-8:].mean())   # (-log_prob / D with D = 20,778: the values move in the sixth digit)
dw = float((m1.arena(0) - m2.arena(0)).abs().max())
assert dw < 1e-4, dw
print(f"model_{which}_gmm: {steps} Adam steps: loss {losses[:8].mean():.7f} -> {losses[-8:].mean():.7f}, two handles end {dw:.1e} apart (max abs weight difference)")
xb, yb = batches[0]
tr = m1.train_forward_backward(xb, yb, want_rows=True)["loss_rows"].cpu().numpy()
ev = m1.eval_step(xb.cpu().numpy(), yb.cpu().numpy())["loss_rows"].cpu().numpy()
live = yb.cpu().numpy() > 0
err = np.abs(tr - ev)[live].max()
assert err < 2e-3 * max(1.0, np.abs(tr[live]).max()), err
print(f"model_{which}_gmm: evaluation after sync vs the training forward: max |loss row difference| {err:.2e} (16-bit tables vs fp32 masters)")
m1.close()
m2.close()
print("soak_gmm_train: ok")
