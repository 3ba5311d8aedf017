"""-m gpu: the training step (model.py:102-117 loss, model.py:139-141 Adam) of the HIP path against an
independent PyTorch-CPU fp64 formulation differentiated by autograd (a floating-point kernel: the torch
reference is the checker here, as for the forward pass in tests/test_oracle_vs_torch.py).

Tolerances: the device multiplies bf16 operands (weights, activations, d-logits) with fp32 accumulation
and passes gradients straight through the forward roundings, the reference is fp64 on the bf16-rounded
weights: per-tensor relative L2 error < 2e-2 and cosine > 0.999; Adam arithmetic itself (given the
device's own gradients) to 1e-6."""
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.nn.functional as F  # noqa: E402

from oracle import hiertcn_oracle as O  # noqa: E402
from tests.gpu_common import make_batch, make_params, small_args  # noqa: E402

pytestmark = pytest.mark.gpu


def ste_bf16(a):
    """Round to bf16 in the forward pass, identity in the backward pass: the device keeps its activations in
    bf16 and passes gradients straight through the roundings."""
    return a + (a.detach().to(torch.float32).to(torch.bfloat16).to(torch.float64) - a.detach())


def torch_loss(args, P, x, y, mask, state):
    """model_hier.py:39-93 + model.py:102-117 on torch tensors (P: name -> tensor), activations rounded where
    the device rounds them (after the projection and after every block)."""
    B, S, L = x.shape
    H, nl, N = args.hidden_dim, args.num_layer, args.item_num
    st = torch.as_tensor(state, dtype=torch.float64)
    E = P["hier/emb/kernel"]
    xt, yt = torch.as_tensor(x, dtype=torch.int64), torch.as_tensor(y, dtype=torch.int64)
    mk = torch.as_tensor(mask, dtype=torch.float64)
    logits = []
    for s in range(S):
        e = F.embedding(xt[:, s], E) * (xt[:, s] > 0)[..., None]
        u = torch.cat([e, st[:, None, :].expand(-1, L, -1)], -1) @ P["hier/tcn/emb/kernel"]
        a = ste_bf16(u).transpose(1, 2)
        for i in range(len(args.tcn_channel)):
            pre = f"hier/tcn/temporal_conv_net/tblock_{i}"
            W = P[f"{pre}/conv1/kernel"].permute(2, 1, 0)
            d, K = 2 ** i, W.shape[-1]
            c = torch.relu(F.conv1d(F.pad(a, ((K - 1) * d, 0)), W, P[f"{pre}/conv1/bias"], dilation=d))
            a = ste_bf16(torch.relu(c + a))
        logits.append(a.transpose(1, 2) @ P["hier/tcn/dense/kernel"] + P["hier/tcn/dense/bias"])
        ey = F.embedding(yt[:, s], E) * (yt[:, s] > 0)[..., None]
        cur = ey.sum(1) / (yt[:, s] > 0).sum(1, keepdim=True) + P["hier/emb/bias"]
        new = []
        for l in range(nl):
            pre = f"hier/multi_rnn_cell/cell_{l}/gru_cell"
            h = st[:, l * H:(l + 1) * H]
            g = torch.sigmoid(torch.cat([cur, h], 1) @ P[f"{pre}/gates/kernel"] + P[f"{pre}/gates/bias"])
            r, uu = g.chunk(2, 1)
            c = torch.tanh(torch.cat([cur, r * h], 1) @ P[f"{pre}/candidate/kernel"] + P[f"{pre}/candidate/bias"])
            cur = uu * h + (1 - uu) * c
            new.append(cur)
        st = torch.cat(new, 1) * mk[:, s:s + 1]
    z = torch.cat(logits, 1)
    yy = yt.reshape(B, S * L)
    m = (yy > 0).double()
    z = z * m[..., None]
    ce = F.cross_entropy(z.reshape(-1, N), yy.reshape(-1), reduction="none").reshape(B, -1) * m
    act = m.sum(1)
    users = (act > 0).double().sum()
    return ((ce.sum(1) / (act + 1e-6)).sum() / users), st


# the last shape has 11,520 rows: the weight-gradient GEMMs of csrc/gemm.hip then cut their row reduction into slices (partial tiles summed in
# slice order), which the 252-row shapes never do
# d128_h128: E == H -- the shape whose across-session scan AND its backward run as persistent launches (csrc/gru_persist.hip, k_gru_scan_persist with the
# training records, k_gru_bptt_persist); 33 lanes: two passes of lanes per group
# d64_rows33280: 130 padded row blocks -- the training forward's device-side table split on a grid of several rounds (score_bwd.hip, sg_dyn_grid), the scan's
# backward through the fused skinny products at 130 lanes
@pytest.fixture(scope="module", params=[(64, 700, 32, (7, 4, 9)), (128, 1500, 64, (7, 4, 9)), (512, 900, 64, (7, 4, 9)), (128, 1500, 64, (96, 4, 30)),
                                        (128, 900, 128, (33, 5, 9)), (64, 2100, 32, (130, 8, 32))],
                ids=["d64", "d128", "d512", "d128_rows11520", "d128_h128", "d64_rows33280"])
def trained(request):
    from hiertcn_amd.engine import Engine, metrics_dict
    d, N, H, (B, S, L) = request.param
    args = small_args(N=N, d=d, H=H)
    p = make_params(args)
    eng = Engine(args, hier=True, train=True)
    eng.set_weights(p)
    rng = np.random.default_rng(d)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, H * args.num_layer))).astype(np.float32)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda().contiguous()  # noqa: E731
    out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    torch.cuda.synchronize()
    grads = eng.export("grads")
    loss = metrics_dict(out["metrics8"])["loss"]
    pr = O.round_params_bf16(p, E=d)
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in pr.items()}
    ref_loss, ref_state = torch_loss(args, P, x, y, mask, st0)
    ref_loss.backward()
    ref = {k: v.grad.numpy() for k, v in P.items()}
    yield dict(args=args, p=p, eng=eng, grads=grads, ref=ref, loss=loss, ref_loss=float(ref_loss.detach()), state=out["state"].cpu().numpy(),
               ref_state=ref_state.detach().numpy(), batch=(x, y, mask, st0), dev=dev)
    eng.close()


def test_loss_and_state_match(trained):
    t = trained
    assert abs(t["loss"] - t["ref_loss"]) < 2e-3
    assert np.abs(t["state"] - t["ref_state"]).max() < 1e-3


def test_every_gradient_matches_autograd(trained):
    t = trained
    assert set(t["grads"]) == set(t["ref"])
    report = []
    for k in sorted(t["ref"]):
        g, r = t["grads"][k].astype(np.float64), t["ref"][k]
        assert g.shape == r.shape, k
        nr = np.linalg.norm(r)
        rel = np.linalg.norm(g - r) / max(nr, 1e-30)
        cos = float((g * r).sum() / max(np.linalg.norm(g) * nr, 1e-30))
        report.append((k, rel, cos, nr))
    bad = [(k, rel, cos) for k, rel, cos, nr in report if nr > 1e-9 and (rel > 2e-2 or cos < 0.999)]
    assert not bad, "\n".join(f"{k}: rel {rel:.3e} cos {cos:.5f}" for k, rel, cos in bad) + "\nall:\n" + "\n".join(
        f"{k}: rel {rel:.3e} cos {cos:.5f} |ref| {nr:.3e}" for k, rel, cos, nr in report)
    # the null id never receives a gradient (model.py:59)
    assert not t["grads"]["hier/emb/kernel"][0].any()


def test_adam_is_tf_adam(trained):
    """Two steps on the same batch; the reference update is computed in NumPy from the device's own gradients:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; theta -= lr_t m/(sqrt(v)+eps)."""
    t = trained
    eng, dev = t["eng"], t["dev"]
    x, y, mask, st0 = t["batch"]
    lr, b1, b2, eps = 1e-2, 0.9, 0.999, 1e-8
    w = {k: v.astype(np.float64) for k, v in eng.export("weights").items()}
    for k in t["p"]:   # the masters hold the fp32 parameters as given (not their bf16 roundings)
        assert np.array_equal(w[k].astype(np.float32), np.asarray(t["p"][k], np.float32)), k
    m1 = {k: np.zeros_like(v) for k, v in w.items()}
    m2 = {k: np.zeros_like(v) for k, v in w.items()}
    losses = []
    from hiertcn_amd.engine import metrics_dict
    for step in (1, 2, 3):
        out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
        losses.append(metrics_dict(out["metrics8"])["loss"])
        g = {k: v.astype(np.float64) for k, v in eng.export("grads").items()}
        eng.adam_step(lr, b1, b2, eps)
        got = eng.export("weights")
        lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        for k in w:
            m1[k] = b1 * m1[k] + (1 - b1) * g[k]
            m2[k] = b2 * m2[k] + (1 - b2) * g[k] ** 2
            w[k] = w[k] - lr_t * m1[k] / (np.sqrt(m2[k]) + eps)
            assert np.abs(got[k] - w[k]).max() < 2e-6 * max(1.0, np.abs(w[k]).max()), (k, step)
            w[k] = got[k].astype(np.float64)   # follow the device's fp32 state so that rounding does not accumulate
    assert losses[2] < losses[0], losses   # three Adam steps on one batch lower its loss


def test_gradient_all_reduce_view_is_the_arena(trained):
    eng = trained["eng"]
    g = eng.grad_arena()
    assert g.dtype == torch.float32 and g.is_cuda and g.numel() > 0
    total = sum(v.size for v in trained["grads"].values())
    assert g.numel() >= total    # segments are padded to 64 floats
    s = float(g.double().abs().sum())
    assert np.isfinite(s) and s > 0


def test_chunked_dlogits_equal_single_chunk():
    """The d-logits matrix is produced in chunks of rows (64-row stages; C_last = 512: chunks of items); forcing one-stage
    chunks -- 64 rows, then a ragged tail -- must give the same gradients as one chunk (only fp32 summation orders change)."""
    import os
    import subprocess
    import sys
    import tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, numpy as np, torch
sys.path.insert(0, %r)
from tests.gpu_common import make_batch, make_params, small_args
from hiertcn_amd.engine import Engine
args = small_args(N=700, d=64, H=32)
eng = Engine(args, hier=True, train=True); eng.set_weights(make_params(args))
x, y, mask = make_batch(np.random.default_rng(3), 5, 3, 8, 700)
dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda().contiguous()
eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(np.zeros((5, 64), np.float32)))
g = eng.export("grads")
np.savez(sys.argv[1], **{k.replace("/", "."): v for k, v in g.items()})
''' % root
    with tempfile.TemporaryDirectory() as tmp:
        outs = []
        # (third run: the layer-at-a-time low level instead of the fused kernel that writes the kept activations on its way)
        for i, env in enumerate(({}, {"HIERTCN_SGRAD0_MIN_STAGES": "1"}, {"HIERTCN_FORCE_LAYERED": "1"})):   # (mapped onto hiertcn_tuning_t when the Engine is created)
            f = os.path.join(tmp, f"g{i}.npz")
            subprocess.run([sys.executable, "-c", code, f], check=True, env={**os.environ, **env})
            outs.append(dict(np.load(f)))
    for other in outs[1:]:
        assert set(outs[0]) == set(other)
        for k in outs[0]:
            a, b = outs[0][k].astype(np.float64), other[k].astype(np.float64)
            assert np.linalg.norm(a - b) <= 1e-5 * max(np.linalg.norm(a), 1e-12), k


@pytest.mark.parametrize("low,H", [("tcn", 32), ("gru", 64)])
def test_run_hier_trains_saves_and_restores(tmp_path, low, H):
    """run_hier with args.train on the synthetic XING CSV (HierTCN and the reference's default HierGRU): epochs 0..2 of 6
    batches, lr 1e-2, checkpoint at epoch 2, evaluation at epoch 2; the loss falls, the checkpoint restores into a fresh
    model that evaluates identically."""
    import os
    from hiertcn_amd import checkpoint
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.data_loader import Dataloader_hier_model_xing
    from hiertcn_amd.data_prepare import load_xing
    from hiertcn_amd.model import HierTCNModel
    from hiertcn_amd.run_hier_xing import evaluate_hier, run_hier
    data = load_xing(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "interactions_synth.csv"))
    n_items = int(data[1][:, 1].max()) + 1
    args = hier_tcn_args(item_num=n_items, emb_dim=64, hidden_dim=H, num_layer=2, tcn_channel=[64, 64], kernel_size=5,
                         batch_size=4, max_session_num=3, max_activity_len=6)
    args.model_low_type = low
    args.log_path, args.save_path = str(tmp_path) + "/log/", str(tmp_path) + "/model/"
    args.train, args.validate, args.test, args.save = True, False, True, True
    args.epoch_max, args.epoch_batches_train, args.save_epoch, args.test_epoch, args.learning_rate = 2, 6, 2, 2, 1e-2
    mk = lambda split: Dataloader_hier_model_xing(args, split, data=data)  # noqa: E731
    model = HierTCNModel(args)
    st0 = np.zeros((4, 2 * H))
    os.makedirs(args.log_path + "before/")
    before, *_ = evaluate_hier(args, mk("test"), st0, model, prefix="before")
    run_hier(args, mk("train"), mk("validate"), mk("test"), sess=model)
    txt = open(args.log_path + "test/results.txt").read()
    after_loss = float(txt.split("final\n")[1].split()[1].rstrip(","))
    assert after_loss < before["loss_epoch"] - 0.05, (before["loss_epoch"], after_loss)
    import json
    tr = [json.loads(l) for l in open(args.log_path + "train/scalars.jsonl")]
    assert [r["step"] for r in tr] == [0, 1, 2] and tr[2]["value"] < tr[0]["value"]
    ck = args.save_path + "2.ckpt"
    params, extra = checkpoint.load(ck, with_extra=True)
    assert "state_np" not in extra          # the carried state is re-zeroed on restart, as in the reference (run_hier_xing.py:241)
    assert ("hier/gru/dense/kernel" in params) == (low == "gru")
    fresh = HierTCNModel(args, params, train=False)
    os.makedirs(args.log_path + "a/")
    os.makedirs(args.log_path + "b/")
    ra, *_ = evaluate_hier(args, mk("test"), st0, model, prefix="a")
    rb, *_ = evaluate_hier(args, mk("test"), st0, fresh, prefix="b")
    assert ra == rb
    model.engine.close()
    fresh.engine.close()


def torch_tcn_loss(args, P, x, y):
    """Single-level model (model.py:56-57,71-73,102-117): ids -> 'emb' -> TemporalConvNet (Dense down-sample where
    Cin != Cout, customized_tcn_cell.py:102-106) -> scoring -> masked lane-mean cross entropy."""
    B, T = x.shape
    N = args.item_num
    xt, yt = torch.as_tensor(x, dtype=torch.int64), torch.as_tensor(y, dtype=torch.int64)
    a = (F.embedding(xt, P["tcn/emb/kernel"]) * (xt > 0)[..., None]).transpose(1, 2)   # bf16-stored table: already exact
    for i in range(len(args.tcn_channel)):
        pre = f"tcn/temporal_conv_net/tblock_{i}"
        W = P[f"{pre}/conv1/kernel"].permute(2, 1, 0)
        d, K = 2 ** i, W.shape[-1]
        c = torch.relu(F.conv1d(F.pad(a, ((K - 1) * d, 0)), W, P[f"{pre}/conv1/bias"], dilation=d))
        if f"{pre}/dense/kernel" in P:
            r = (a.transpose(1, 2) @ P[f"{pre}/dense/kernel"] + P[f"{pre}/dense/bias"]).transpose(1, 2)
        else:
            r = a
        a = ste_bf16(torch.relu(c + r))
    z = a.transpose(1, 2) @ P["tcn/dense/kernel"] + P["tcn/dense/bias"]
    m = (yt > 0).double()
    z = z * m[..., None]
    ce = F.cross_entropy(z.reshape(-1, N), yt.reshape(-1), reduction="none").reshape(B, -1) * m
    act = m.sum(1)
    return (ce.sum(1) / (act + 1e-6)).sum() / (act > 0).double().sum()


def test_single_level_training_with_downsample_residual():
    """model_tcn with unequal widths ([64, 64, 128, 128]: a Dense residual on block 2), T = 70: gradients vs autograd,
    then TcnModel.train_step lowers the loss."""
    from hiertcn_amd.args import make_args
    from hiertcn_amd.engine import Engine, metrics_dict
    from hiertcn_amd.model import TcnModel
    from hiertcn_amd.params import init_params
    N = 600
    a = make_args([], model_type="hier", item_num=N, emb_dim=64, tcn_channel=[64, 64, 128, 128], kernel_size=3)
    a.model_type = "tcn"
    rng = np.random.default_rng(5)
    p = init_params(a, 2, "tcn")
    p = {k: (v + 0.05 * rng.standard_normal(v.shape).astype(np.float32)) if k.endswith("/bias") else v for k, v in p.items()}
    B, T = 5, 70
    lens = rng.integers(5, T + 1, B)
    y = (rng.integers(1, N, (B, T)) * (np.arange(T)[None, :] < lens[:, None])).astype(np.int32)
    x = np.zeros_like(y)
    x[:, 1:] = y[:, :-1]
    eng = Engine(a, hier=False, train=True)
    eng.set_weights(p)
    dev = lambda v: torch.as_tensor(np.ascontiguousarray(v)).cuda().contiguous()  # noqa: E731
    out = eng.train_forward_backward(dev(x).view(B, 1, T), dev(y).view(B, 1, T))
    grads = eng.export("grads")
    pr = O.round_params_bf16(p, E=64)
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in pr.items()}
    ref_loss = torch_tcn_loss(a, P, x, y)
    ref_loss.backward()
    assert abs(metrics_dict(out["metrics8"])["loss"] - float(ref_loss.detach())) < 2e-3
    assert set(grads) == set(P) and "tcn/temporal_conv_net/tblock_2/dense/kernel" in grads
    for k in sorted(P):
        g, r = grads[k].astype(np.float64), P[k].grad.numpy()
        rel = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)
        cos = float((g * r).sum() / max(np.linalg.norm(g) * np.linalg.norm(r), 1e-30))
        assert rel < 2e-2 and cos > 0.999, (k, rel, cos)
    eng.close()
    model = TcnModel(a, p, train=True)
    losses = [model.train_step(x, y, 1e-2)["loss"] for _ in range(4)]
    assert losses[-1] < losses[0], losses
    assert set(model.params) == set(p)
    model.engine.close()


def test_optimizer_state_round_trip(trained):
    """Adam slots + time step survive save/restore: a restored model's next update equals the original's."""
    from hiertcn_amd.engine import Engine
    t = trained
    eng, dev = t["eng"], t["dev"]
    x, y, mask, st0 = t["batch"]
    eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    eng.adam_step(1e-2)
    w_mid, opt = eng.export("weights"), eng.optimizer_state()
    assert int(opt["adam_t"][0]) >= 1 and np.abs(opt["adam_v"]).sum() > 0
    eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    eng.adam_step(1e-2)
    want = eng.export("weights")
    other = Engine(t["args"], hier=True, train=True)
    other.set_weights(w_mid)
    other.load_optimizer_state(opt)
    other.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    other.adam_step(1e-2)
    got = other.export("weights")
    for k in want:
        # (not bitwise: fp32 atomics in the scatter-adds and the item-walk start of the scoring kernel reorder sums)
        # without the restored slots the two updates differ by ~lr = 1e-2 per element
        assert np.abs(got[k] - want[k]).max() <= 1e-4 * max(1.0, np.abs(want[k]).max()), (k, np.abs(got[k] - want[k]).max())
    other.close()


def test_scoring_backward_properties_at_north_star_size():
    """N = 300,000, d = 256 (BASELINE.json configs[2] catalogue), 33,600 rows (about 30,200 scored) = two row chunks of the 16 GiB d-logits budget.
    Size-independent properties of the scoring layer's backward: (a) every row of softmax - onehot sums to zero, so the
    bias gradient sums to zero; (b) for sampled items the bias gradient and the kernel-gradient column equal a direct
    evaluation from the HIP path's own hidden rows and log-sum-exp (d-logits rounded to bf16 as the device rounds them)."""
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.engine import Engine
    from hiertcn_amd.params import init_params
    N, d, S, L, B = 300_000, 256, 20, 30, 56    # ~30,200 scored rows > the 28,608 rows of one 16 GiB chunk
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, max_activity_len=L,
                         max_session_num=S, batch_size=B)
    p = init_params(args, seed=2)
    rng = np.random.default_rng(11)
    p["hier/tcn/dense/bias"] = (0.5 * rng.standard_normal(N)).astype(np.float32)
    eng = Engine(args, hier=True, train=True)
    eng.set_weights(p)
    ids = np.minimum(1 + (rng.pareto(1.1, (B, S, L)) * 50).astype(np.int64), N - 1)
    ids[:, :, 20:] *= (rng.random((B, S, 1)) > 0.3)          # some sessions shorter
    x = np.zeros_like(ids); x[:, :, 1:] = ids[:, :, :-1]
    dev = lambda a, t=None: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0", dtype=t)  # noqa: E731
    xd, yd = dev(x, torch.int32), dev(ids, torch.int32)
    mask = dev((rng.random((B, S)) > 0.1).astype(np.float32))
    st0 = torch.zeros((B, eng.HT), dtype=torch.float32, device="cuda:0")
    eng.train_forward_backward(xd, yd, mask, st0)
    g = eng.export("grads")
    db = g["hier/tcn/dense/bias"].astype(np.float64)
    assert abs(db.sum()) <= 1e-4 * np.abs(db).sum()

    # the HIP path's own hidden rows / lse (evaluation entry points of the same handle)
    pooled = eng.session_pool(yd.view(B * S, L)).view(B, S, -1)
    sproj, _, _ = eng.gru_scan(pooled, mask, st0)
    hid = eng.tcn_forward(xd.view(B * S, L), sproj.view(B * S, -1))
    yv = yd.view(-1).long()
    sc = eng.score_fused(hid, yd.view(-1), want_target=True)
    valid = yv > 0
    lse = (sc["ce"] + sc["target"]).double()
    act = valid.view(B, -1).sum(1).double()
    users = (act > 0).sum().double()
    w = (1.0 / ((act + 1e-6) * users)).repeat_interleave(S * L) * valid
    top = torch.bincount(yv[valid]).argsort(descending=True)[:4].cpu().numpy()
    J = np.unique(np.concatenate([top, rng.integers(1, N, 12), [0]]))
    Wj = dev(p["hier/tcn/dense/kernel"][:, J]).bfloat16().double()
    bj = dev(p["hier/tcn/dense/bias"][J]).double()
    H = hid.double()
    z = H @ Wj + bj
    gj = w[:, None] * (torch.exp(z - lse[:, None]) - (yv[:, None] == dev(J)[None, :]).double())
    gj[~valid] = 0
    db_ref = gj.sum(0).cpu().numpy()
    assert np.abs(db[J] - db_ref).max() <= 2e-3 * np.abs(db_ref).max() + 1e-9
    dW_ref = (H.T @ gj.float().bfloat16().double()).cpu().numpy()          # [d, |J|]
    dW = g["hier/tcn/dense/kernel"][:, J].astype(np.float64)
    assert np.linalg.norm(dW - dW_ref) <= 5e-3 * np.linalg.norm(dW_ref), np.linalg.norm(dW - dW_ref) / np.linalg.norm(dW_ref)
    eng.close()


def test_width_padded_model_trains_d100():
    """BASELINE.json config 1's width (d = 100, zero-padded to 128 on upload): gradients vs autograd AT d = 100, the padded channels
    stay exactly zero through Adam steps, the loss falls, and the exported variables have the reference's shapes."""
    from hiertcn_amd.args import make_args
    from hiertcn_amd.engine import metrics_dict
    from hiertcn_amd.model import TcnModel
    from hiertcn_amd.params import init_params, pad_widths
    N = 700
    a = make_args([], model_type="hier", item_num=N, emb_dim=100, tcn_channel=[100, 100], kernel_size=3)
    a.model_type = "tcn"
    rng = np.random.default_rng(100)
    p = init_params(a, 2, "tcn")
    p = {k: (v + 0.05 * rng.standard_normal(v.shape).astype(np.float32)) if k.endswith("/bias") else v for k, v in p.items()}
    B, T = 6, 33
    lens = rng.integers(5, T + 1, B)
    y = (rng.integers(1, N, (B, T)) * (np.arange(T)[None, :] < lens[:, None])).astype(np.int32)
    x = np.zeros_like(y)
    x[:, 1:] = y[:, :-1]
    model = TcnModel(a, p, train=True)
    eng = model.engine
    assert eng.E == 128 and model._padded
    dev = lambda v: torch.as_tensor(np.ascontiguousarray(v)).cuda().contiguous()  # noqa: E731
    out = eng.train_forward_backward(dev(x).view(B, 1, T), dev(y).view(B, 1, T))
    g_pad = eng.export("grads")
    # autograd reference at the true width, on the values the device stores (bf16-rounded tables)
    pr = O.round_params_bf16(p, E=100)
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in pr.items()}
    ref_loss = torch_tcn_loss(a, P, x, y)
    ref_loss.backward()
    assert abs(metrics_dict(out["metrics8"])["loss"] - float(ref_loss.detach())) < 2e-3
    for k in sorted(P):
        r = P[k].grad.numpy()
        g = g_pad[k].astype(np.float64)
        cut = g[tuple(slice(0, n) for n in r.shape)]
        rest = g.copy()
        rest[tuple(slice(0, n) for n in r.shape)] = 0
        assert not rest.any(), k                                    # gradients of the padding: exactly zero
        rel = np.linalg.norm(cut - r) / max(np.linalg.norm(r), 1e-30)
        assert rel < 2e-2, (k, rel)
    losses = [model.train_step(x, y, 1e-2)["loss"] for _ in range(4)]
    assert losses[-1] < losses[0], losses
    w_pad = eng.export("weights")
    _, p0_pad = pad_widths(a, p, "tcn")
    for k, v in w_pad.items():                                       # padding still exactly zero after four Adam steps
        mask = np.ones(v.shape, bool)
        mask[tuple(slice(0, n) for n in p[k].shape)] = False
        assert not v[mask].any(), k
    got = model.params
    assert {k: v.shape for k, v in got.items()} == {k: v.shape for k, v in p.items()}
    assert np.abs(got["tcn/emb/kernel"] - p["tcn/emb/kernel"]).max() > 0          # it did train
    model.engine.close()


@pytest.mark.parametrize("d,overflow", [(128, False), (256, False), (256, True), (128, True)])
def test_fused_scoring_backward_matches_the_chunked_path(d, overflow):
    """csrc/score_bwd.hip (one pass over the logits for rank / ce / d hidden, a second for d Wo / d bo, nothing materialised, row count on
    the device) against the previous implementation (bf16 d-logits chunks + two 16-bit GEMMs per chunk: the ablation build's HIERTCN_TRAIN_LEGACY=1): loss, per-step
    outputs and every gradient.  overflow: scoring weights that put one item > 133 nats above every target, so the forward pass's
    running sum leaves fp32 range and both fix-up launches (rank / ce and d hidden) run."""
    from hiertcn_amd.engine import Engine, metrics_dict
    N, H, B, S, L = 1500, 32, 7, 3, 13
    args = small_args(N=N, d=d, H=H, S=S, L=L, B=B)
    p = make_params(args)
    rng = np.random.default_rng(d + overflow)
    if overflow:
        W = p["hier/tcn/dense/kernel"].copy()
        W[:, 1:N // 2] -= 6.0              # candidate targets far below zero (hidden rows are relu outputs: their channel sum is > 0)
        W[:, N - 7] += 6.0                 # one item far above
        p["hier/tcn/dense/kernel"] = W
        p["hier/tcn/temporal_conv_net/tblock_1/conv1/bias"] = p["hier/tcn/temporal_conv_net/tblock_1/conv1/bias"] + 0.5
    x, y, mask = make_batch(rng, B, S, L, N // 2 if overflow else N)
    st0 = (0.1 * rng.standard_normal((B, 2 * H))).astype(np.float32)
    dev = lambda v: torch.as_tensor(np.ascontiguousarray(v)).cuda().contiguous()  # noqa: E731
    res = {}
    for mode in ("fused", "legacy"):
        if mode == "legacy":
            os.environ["HIERTCN_TRAIN_LEGACY"] = "1"
        try:
            eng = Engine(args, hier=True, train=True, lib_variant="ablation" if mode == "legacy" else None)
            eng.set_weights(p)
            out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
            torch.cuda.synchronize()
            res[mode] = (metrics_dict(out["metrics8"]), eng.export("grads"))
            if mode == "fused" and overflow:   # the input does what it is meant to do: gaps beyond the fixed reference's range
                pooled = eng.session_pool(dev(y.reshape(B * S, L))).view(B, S, -1)
                sproj, _, _ = eng.gru_scan(pooled, dev(mask), dev(st0))
                hid = eng.tcn_forward(dev(x.reshape(B * S, L)), sproj.view(B * S, -1))
                z = eng.score_logits(hid).cpu().numpy()
                yy = y.reshape(-1)
                gap = (z.max(1) - z[np.arange(len(yy)), yy])[yy > 0]
                assert (gap * np.log2(np.e) > 200).any(), gap.max()
            eng.close()
        finally:
            os.environ.pop("HIERTCN_TRAIN_LEGACY", None)
    (mf, gf), (ml, gl) = res["fused"], res["legacy"]
    for k in ("loss", "recall10", "mrr", "rank_frac"):
        assert abs(mf[k] - ml[k]) < 1e-4 * max(1.0, abs(ml[k])), (k, mf[k], ml[k])
    for k in gl:
        a, b = gf[k].astype(np.float64), gl[k].astype(np.float64)
        rel = np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)
        assert np.isfinite(a).all() and rel < 2e-2, (k, rel)     # both round p to bf16 at different points: ~1e-2 apart at most


def test_scoring_backward_split_over_several_rounds_equals_the_unsplit_walk():
    """Past 128 padded row blocks the training forward of the scoring layer deals (row block, table part) over a grid of several rounds of the chip
    (score_bwd.hip: sg_dyn_grid, score_dyn_split's cost rule); HIERTCN_SGRAD_DYN_MAX_BLOCKS=127 keeps one workgroup per row block walking the whole
    table.  Same sums in another order: loss, carried state and every gradient of a 260-lane XING-sized step agree to 1e-5 (tests/tools/sgrad_split_probe.py,
    one process per setting: the variable is mapped onto hiertcn_tuning_t.sgrad_dyn_max_blocks when the Engine is created)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for blocks in (None, "127"):
        env = dict(os.environ)
        env.pop("HIERTCN_SGRAD_DYN_MAX_BLOCKS", None)
        if blocks:
            env["HIERTCN_SGRAD_DYN_MAX_BLOCKS"] = blocks
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "sgrad_split_probe.py")], env=env, capture_output=True, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("SGRAD_SPLIT_PROBE ")]
        assert r.returncode == 0 and line, r.stderr[-2000:]
        res.append(json.loads(line[0][len("SGRAD_SPLIT_PROBE "):]))
    a, b = res
    assert a["live_rows"] == b["live_rows"] and a["live_rows"] > 129 * 256          # more live row blocks than half the chip
    assert abs(a["loss"] - b["loss"]) < 1e-5 * abs(b["loss"])
    assert abs(a["state"] - b["state"]) <= 1e-6 * b["state"]
    for k in b["grads"]:
        ga, gb = np.asarray(a["grads"][k]), np.asarray(b["grads"][k])
        assert abs(ga[0] - gb[0]) <= 1e-4 * gb[0] + 1e-30, k                           # sum of squares
        scale = np.sqrt(gb[0] / max(1, len(gb) - 1)) if gb[0] > 0 else 1.0
        assert np.abs(ga[1:] - gb[1:]).max() <= 1e-4 * max(np.abs(gb[1:]).max(), scale) + 1e-12, k
