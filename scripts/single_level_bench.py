import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from hiertcn_amd.args import make_args
from hiertcn_amd.model import GruModel, TcnModel
N = 20778
for kind in ("gru", "tcn"):
    a = make_args([], model_type=kind, item_num=N)     # reference defaults: gru hidden 200 (-> 256), tcn six blocks
    a.batch_size = 32
    rng = np.random.default_rng(0)
    T = 200
    lens = rng.integers(10, T + 1, 32)
    y = (rng.integers(1, N, (32, T)) * (np.arange(T)[None, :] < lens[:, None])).astype(np.int32)
    x = np.zeros_like(y); x[:, 1:] = y[:, :-1]
    m = (GruModel if kind == "gru" else TcnModel)(a, train=True)
    for _ in range(3): m.eval_step(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): m.eval_step(x, y)
    torch.cuda.synchronize(); te = (time.perf_counter() - t0) / 20
    for _ in range(3): m.train_step(x, y, 1e-3)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): r = m.train_step(x, y, 1e-3)
    torch.cuda.synchronize(); tt = (time.perf_counter() - t0) / 20
    print("single-level %s (reference defaults), 32 users x T = 200, N = 20,778: eval %.2f ms, train %.2f ms per batch; loss %.4f" % (kind, te * 1e3, tt * 1e3, r["loss"]), flush=True)
    m.engine.close()
