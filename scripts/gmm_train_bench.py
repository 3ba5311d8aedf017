"""Training step of the mixture-output models at the reference's batch (32 lanes x 200 steps of the single-level loader, XING catalogue, d = 256,
5 mixtures): ms per optimizer step of model_tcn_gmm / model_gru_gmm (forward + backward of every variable + TF-form Adam), ids resident.
    python scripts/gmm_train_bench.py [tcn|gru] [steps] [learned]    (under rocprofv3 --kernel-trace --stats: where the step's time goes;
                                                                      learned: args.sigma None, M (1 + 2 D) head outputs, mu / s materialised)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from hiertcn_amd.args import make_args

which = sys.argv[1] if len(sys.argv) > 1 else "tcn"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
learned = len(sys.argv) > 3 and sys.argv[3] == "learned"
N, M, d, B, T = 20_778, 5, 256, 32, 200
rng = np.random.default_rng(5)
ids = rng.integers(1, N, (B, T + 1))
ids = ids * (np.arange(T + 1)[None, :] <= rng.integers(2, T + 1, B)[:, None])
x = torch.as_tensor(np.ascontiguousarray(ids[:, :-1].astype(np.int32))).cuda()
y = torch.as_tensor(np.ascontiguousarray(ids[:, 1:].astype(np.int32))).cuda()
if which == "tcn":
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel, init_params
    a = make_args([], model_type="tcn_gmm", item_num=N, output_dim=N, num_mixture=M, sigma=0.1, loss="gmm_prob", tcn_channel=[d, d], kernel_size=5, adjust=False)
    if learned:
        a.sigma = None
    model = TcnGmmModel(a, init_params(a, 3))
else:
    from hiertcn_amd.model_gru_gmm import GruGmmModel, init_params
    a = make_args([], model_type="gru_gmm", item_num=N, output_dim=N, num_mixture=M, sigma=0.1, loss="gmm_prob", hidden_dim=d, num_layer=2, adjust=False)
    if learned:
        a.sigma = None
    model = GruGmmModel(a, init_params(a, 3))
model.enable_training()
for _ in range(3):
    model.train_forward_backward(x, y)
    model.adam_step(1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = model.train_forward_backward(x, y)
    model.adam_step(1e-3)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
s = out["sums"].cpu().numpy()
print(f"model_{which}_gmm{' (learned scale)' if learned else ''}: {dt * 1e3:.3f} ms per step ({B * T} rows, {B * T / dt:.0f} rows/s), loss {s[0] / s[1]:.5f}")
