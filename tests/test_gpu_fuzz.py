"""-m gpu: seeded random shapes through the whole evaluation path (hiertcn_forward_eval) against the CPU oracle -- widths, catalogue
sizes that are not multiples of any tile, one-row batches, L on both sides of the fused low level's limit, 1..3 blocks, K = 1..6,
HierTCN and HierGRU.  Catches shape-dependent slips the fixed fixtures cannot (tail workgroups, padding stages, peeled loops).
The committed counts keep the suite short; one-off runs of 400 evaluation and 150 training shapes (other seeds) passed on MI355X."""
import copy

import numpy as np
import pytest

from oracle import hiertcn_oracle as O
from tests.gpu_common import as_lists, make_batch, make_params, small_args

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _cases(n=24, seed=20261002):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        d = int(rng.choice([64, 128, 256]))
        low = "gru" if i % 4 == 3 else "tcn"
        H = int(rng.choice([64, 128])) if low == "gru" else int(rng.choice([32, 64]))
        out.append(dict(d=d, H=H, low=low, N=int(rng.integers(2, 3000)), B=int(rng.integers(1, 41)), S=int(rng.integers(1, 5)),
                        L=int(rng.integers(1, 34)), K=int(rng.integers(1, 7)), blocks=int(rng.integers(1, 4)), nl=int(rng.integers(1, 3)),
                        seed=int(rng.integers(1 << 30))))
    # 66,560 padded rows, ~half of them live: past the one-block scoring kernel's device-side catalogue split (65,024 rows), inside the two-block
    # kernel's (130,048); 2,100 items = 33 stages = two parts
    out.append(dict(d=64, H=32, low="tcn", N=2100, B=260, S=8, L=32, K=3, blocks=1, nl=1, seed=77))
    return out


@pytest.mark.parametrize("c", _cases(), ids=lambda c: "{low}-d{d}-N{N}-B{B}-S{S}-L{L}-K{K}-b{blocks}".format(**c))
def test_random_shape_forward_eval_matches_oracle(c):
    from hiertcn_amd.engine import Engine, metrics_dict
    args = copy.copy(small_args(N=c["N"], d=c["d"], H=c["H"], nl=c["nl"], blocks=c["blocks"], K=c["K"], L=c["L"], S=c["S"], B=c["B"]))
    args.model_low_type = c["low"]
    p = make_params(args, seed=c["seed"] % 1000)
    eng = Engine(args, hier=True)
    eng.set_weights(p)
    pr = O.round_params_bf16(p, E=c["d"])
    rng = np.random.default_rng(c["seed"])
    B, S, L, N = c["B"], c["S"], c["L"], c["N"]
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.25 * rng.standard_normal((B, c["H"] * c["nl"]))).astype(np.float32)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")  # noqa: E731
    out = eng.forward_eval(dev(x), dev(y), dev(mask), dev(st0), want_rows=True)
    m = metrics_dict(out["metrics8"])
    xl, yl, ml = as_lists(x, y, mask)
    pred, st = O.model_hier(args, xl, yl, ml, st0, pr, O.MIRROR)
    yy = y.reshape(B, S * L)
    ref = O.masked_loss_and_metrics(pred, yy, N)
    assert np.abs(out["state"].cpu().numpy() - st).max() < 1e-5
    assert abs(m["loss"] - ref["loss"]) < 1e-3, (m["loss"], ref["loss"])
    assert m["user_count"] == float((yy > 0).any(1).sum())
    z = pred * (yy[..., None] > 0)
    zt = np.take_along_axis(z, yy[..., None].astype(np.int64), -1)
    lo, hi = (z > zt + 2e-3).sum(-1), (z > zt - 2e-3).sum(-1) - 1
    r = out["rank"].cpu().numpy().reshape(B, S * L)
    v = yy > 0
    assert np.all(lo[v] <= r[v]) and np.all(r[v] <= np.maximum(hi, lo)[v]) and not r[~v].any()
    ce = out["ce"].cpu().numpy().reshape(B, S * L)
    assert np.abs(ce - np.asarray(ref["ce"]))[v].max() < 2e-3 if v.any() else True
    eng.close()


def _train_cases(n=10, seed=4242):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        low = "gru" if i % 3 == 2 else "tcn"
        d = int(rng.choice([64, 128, 256]))
        out.append(dict(d=d, H=int(rng.choice([64, 128])) if low == "gru" else int(rng.choice([32, 64])), low=low, N=int(rng.integers(40, 1500)),
                        B=int(rng.integers(1, 9)), S=int(rng.integers(1, 4)), L=int(rng.integers(2, 13)), K=int(rng.integers(1, 6)),
                        blocks=int(rng.integers(1, 4)), nl=int(rng.integers(1, 3)), seed=int(rng.integers(1 << 30))))
    return out


@pytest.mark.parametrize("c", _train_cases(), ids=lambda c: "{low}-d{d}-N{N}-B{B}-S{S}-L{L}-K{K}-b{blocks}-nl{nl}".format(**c))
def test_random_shape_training_step_matches_autograd(c):
    """Every gradient of one training step against autograd on the torch fp64 restatements (tests/test_gpu_train.py,
    tests/test_gpu_hiergru.py), same tolerances: relative L2 < 3e-2, cosine > 0.999."""
    from hiertcn_amd.engine import Engine, metrics_dict
    from tests.test_gpu_hiergru import torch_loss_gru
    from tests.test_gpu_train import torch_loss
    args = copy.copy(small_args(N=c["N"], d=c["d"], H=c["H"], nl=c["nl"], blocks=c["blocks"], K=c["K"], L=c["L"], S=c["S"], B=c["B"]))
    args.model_low_type = c["low"]
    p = make_params(args, seed=c["seed"] % 1000)
    eng = Engine(args, hier=True, train=True)
    eng.set_weights(p)
    rng = np.random.default_rng(c["seed"])
    B, S, L, N = c["B"], c["S"], c["L"], c["N"]
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.25 * rng.standard_normal((B, c["H"] * c["nl"]))).astype(np.float32)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")  # noqa: E731
    out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    grads = eng.export("grads")
    pr = O.round_params_bf16(p, E=c["d"])
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in pr.items()}
    ref_loss, _ = (torch_loss_gru if c["low"] == "gru" else torch_loss)(args, P, x, y, mask, st0)
    ref_loss.backward()
    assert abs(metrics_dict(out["metrics8"])["loss"] - float(ref_loss.detach())) < 2e-3
    assert set(grads) == set(P)
    bad, report = [], []
    for k in sorted(P):
        g, r = grads[k].astype(np.float64), (P[k].grad.numpy() if P[k].grad is not None else np.zeros(P[k].shape))
        nr = np.linalg.norm(r)
        rel = np.linalg.norm(g - r) / max(nr, 1e-30)
        cos = float((g * r).sum() / max(np.linalg.norm(g) * nr, 1e-30))
        report.append(f"{k}: rel {rel:.3e} cos {cos:.5f} |ref| {nr:.3e} |got| {np.linalg.norm(g):.3e}")
        if (nr > 1e-9 and (rel > 3e-2 or cos < 0.999)) or (nr <= 1e-9 and np.linalg.norm(g) > 1e-7):
            bad.append(k)
    assert not bad, "\n".join(report)
    eng.close()
