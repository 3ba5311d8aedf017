"""-m gpu: the fp32 TRAINING mode (Engine(dtype="fp32", train=True) = hiertcn_set_precision(FP32) + hiertcn_train_enable; csrc/train.hip): the step the
reference takes (model.py:135-141 on fp32 tensors, loss.py:20-21 on materialised logits) -- fp32 forward on the masters, logits materialised once,
every product of the backward an exact-fp32 GEMM.  Held here: loss, carried state and EVERY gradient against fp64 autograd on the UNROUNDED weights
(the 16-bit training path is checked against autograd on the bf16-rounded weights at 2e-2: tests/test_gpu_train.py); TF-form Adam on those
gradients; a short training run whose loss trajectory follows an fp64 torch run of the same steps; the scoring layer's row chunks; both low levels."""
import numpy as np
import pytest

from tests.gpu_common import make_batch, make_params, small_args

torch = pytest.importorskip("torch")
F = torch.nn.functional
pytestmark = pytest.mark.gpu

dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda().contiguous()  # noqa: E731


def torch_loss64(args, P, x, y, mask, state):
    """model_hier.py:39-93 + model.py:102-117 in fp64, nothing rounded (tests/test_gpu_train.py's torch_loss without its bf16 roundings)."""
    B, S, L = x.shape
    H, nl, N = args.hidden_dim, args.num_layer, args.item_num
    st = torch.as_tensor(state, dtype=torch.float64)
    E = P["hier/emb/kernel"]
    xt, yt = torch.as_tensor(x, dtype=torch.int64), torch.as_tensor(y, dtype=torch.int64)
    mk = torch.as_tensor(mask, dtype=torch.float64)
    logits = []
    for s in range(S):
        e = F.embedding(xt[:, s], E) * (xt[:, s] > 0)[..., None]
        a = (torch.cat([e, st[:, None, :].expand(-1, L, -1)], -1) @ P["hier/tcn/emb/kernel"]).transpose(1, 2)
        for i in range(len(args.tcn_channel)):
            pre = f"hier/tcn/temporal_conv_net/tblock_{i}"
            W = P[f"{pre}/conv1/kernel"].permute(2, 1, 0)
            d, K = 2 ** i, W.shape[-1]
            c = torch.relu(F.conv1d(F.pad(a, ((K - 1) * d, 0)), W, P[f"{pre}/conv1/bias"], dilation=d))
            res = a
            if f"{pre}/dense/kernel" in P:
                res = (a.transpose(1, 2) @ P[f"{pre}/dense/kernel"] + P[f"{pre}/dense/bias"]).transpose(1, 2)
            a = torch.relu(c + res)
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


def _trained_scale(p, args):
    """activations and logits of order 1-10 instead of glorot's 0.05: the regime in which 16-bit storage is visibly off"""
    q = {k: np.array(v, np.float32) for k, v in p.items()}
    for k in q:
        if k.endswith("/emb/kernel") and q[k].shape[0] == int(args.item_num):
            q[k] *= 3.0
        if "/conv1/kernel" in k:
            q[k] *= 1.6
        if k.endswith("tcn/dense/kernel"):
            q[k] *= 20.0
    return q


# (d, N, H, channels, K, (B, S, L)); the last: the reference's own batch shape (32 lanes x 10 slots x 20 steps) on a 3,000-item catalogue
SHAPES = [(64, 700, 32, [64, 64], 5, (7, 4, 9)), (128, 1500, 128, [128, 128], 5, (9, 3, 12)), (64, 900, 32, [64, 128, 128], 3, (6, 3, 20)),
          (256, 3000, 256, [256, 256], 5, (32, 10, 20))]


@pytest.fixture(scope="module", params=SHAPES, ids=["d64", "d128_h128", "d64_downsample_3blocks", "reference_batch_d256"])
def stepped(request):
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.engine import Engine, metrics_dict
    d, N, H, ch, K, (B, S, L) = request.param
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=H, num_layer=2, tcn_channel=ch, kernel_size=K, max_activity_len=L, max_session_num=S, batch_size=B)
    p = _trained_scale(make_params(args), args)
    eng = Engine(args, hier=True, train=True, dtype="fp32")
    eng.set_weights(p)
    rng = np.random.default_rng(d + N)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, H * 2))).astype(np.float32)
    out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    torch.cuda.synchronize()
    grads = eng.export("grads")
    loss = metrics_dict(out["metrics8"])["loss"]
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in p.items()}
    ref_loss, ref_state = torch_loss64(args, P, x, y, mask, st0)
    ref_loss.backward()
    ref = {k: v.grad.numpy() for k, v in P.items()}
    yield dict(args=args, p=p, eng=eng, grads=grads, ref=ref, loss=loss, ref_loss=float(ref_loss.detach()), state=out["state"].cpu().numpy(),
               ref_state=ref_state.detach().numpy(), batch=(x, y, mask, st0))
    eng.close()


def test_loss_and_state_are_fp32_grade(stepped):
    t = stepped
    assert abs(t["loss"] - t["ref_loss"]) < 2e-5 * max(1.0, abs(t["ref_loss"])), (t["loss"], t["ref_loss"])
    assert np.abs(t["state"] - t["ref_state"]).max() < 1e-5


def test_every_gradient_matches_fp64_autograd_on_the_unrounded_weights(stepped):
    t = stepped
    assert set(t["grads"]) == set(t["ref"])
    report = []
    for k in sorted(t["ref"]):
        g, r = t["grads"][k].astype(np.float64), t["ref"][k]
        assert g.shape == r.shape, k
        nr = np.linalg.norm(r)
        report.append((k, np.linalg.norm(g - r) / max(nr, 1e-30), nr))
    bad = [(k, rel) for k, rel, nr in report if nr > 1e-12 and rel > 1e-4]
    assert not bad, "\n".join(f"{k}: rel {rel:.3e}" for k, rel in bad) + "\nall:\n" + "\n".join(f"{k}: rel {rel:.3e} |ref| {nr:.3e}" for k, rel, nr in report)
    print("max relative gradient error %.2e (%s)" % max((rel, k) for k, rel, nr in report if nr > 1e-12))
    assert not t["grads"]["hier/emb/kernel"][0].any()      # the null id never receives a gradient (model.py:59)


def test_a_training_run_follows_the_fp64_trajectory():
    """Eight Adam steps (TF form, lr 1e-3) on one batch stream: the device's loss after every step against an fp64 torch run of the same steps that starts
    from the same fp32 weights -- the trajectory a TF run of the reference graph would take, to fp32 grade.  The 16-bit training path on the same stream
    is recorded beside it (it follows to ~1e-3)."""
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.engine import Engine, metrics_dict
    d, N, H, B, S, L = 64, 900, 32, 8, 3, 12
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=H, num_layer=2, tcn_channel=[d, d], kernel_size=5, max_activity_len=L, max_session_num=S, batch_size=B)
    p = _trained_scale(make_params(args), args)
    rng = np.random.default_rng(5)
    batches = [make_batch(rng, B, S, L, N) for _ in range(8)]
    st0 = np.zeros((B, 2 * H), np.float32)
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-8
    # fp64 reference: TF-form Adam (model.py:139: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); theta -= lr_t m / (sqrt(v) + eps))
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in p.items()}
    m1 = {k: torch.zeros_like(v) for k, v in P.items()}
    m2 = {k: torch.zeros_like(v) for k, v in P.items()}
    ref_losses, state = [], st0.astype(np.float64)
    for step, (x, y, mask) in enumerate(batches, 1):
        for v in P.values():
            v.grad = None
        loss, st = torch_loss64(args, P, x, y, mask, state)
        loss.backward()
        ref_losses.append(float(loss.detach()))
        state = st.detach().numpy()
        lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        with torch.no_grad():
            for k, v in P.items():
                g = v.grad if v.grad is not None else torch.zeros_like(v)
                m1[k] = b1 * m1[k] + (1 - b1) * g
                m2[k] = b2 * m2[k] + (1 - b2) * g * g
                v -= lr_t * m1[k] / (m2[k].sqrt() + eps)
    got = {}
    for mode in ("fp32", "bf16"):
        eng = Engine(args, hier=True, train=True, dtype=mode)
        eng.set_weights(p)
        st = dev(st0)
        losses = []
        for x, y, mask in batches:
            out = eng.train_forward_backward(dev(x), dev(y), dev(mask), st)
            eng.adam_step(lr)
            st = out["state"]
            losses.append(metrics_dict(out["metrics8"])["loss"])
        got[mode] = np.array(losses)
        eng.close()
    ref = np.array(ref_losses)
    err32, err16 = np.abs(got["fp32"] - ref).max(), np.abs(got["bf16"] - ref).max()
    print("loss trajectory (8 steps): fp64 reference %s; fp32 mode max |diff| %.2e; 16-bit path max |diff| %.2e" % (np.round(ref, 5), err32, err16))
    assert ref[-1] < ref[0]                     # it trains
    assert err32 < 5e-5 * max(1.0, ref.max()), (got["fp32"], ref)
    assert err32 < err16                          # ... and closer than the 16-bit path


def test_scoring_layer_in_row_chunks_equals_one_chunk():
    """The mode materialises the logits in ROW CHUNKS of at most hiertcn_tuning_t.train_fp32_chunk_mib MiB (default 8 GiB: the reference's own batch is one
    chunk); d Wo and d bo accumulate over the chunks.  A 324-row batch in 1 MiB chunks (273 rows + 51) against the same batch in one chunk: loss,
    state and every gradient agree to fp32 summation order."""
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.engine import Engine, metrics_dict
    d, N, H, B, S, L = 64, 900, 32, 9, 3, 12
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=H, num_layer=2, tcn_channel=[d, d], kernel_size=5, max_activity_len=L, max_session_num=S, batch_size=B)
    p = _trained_scale(make_params(args), args)
    rng = np.random.default_rng(9)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, 2 * H))).astype(np.float32)
    got = {}
    for name, tuning in (("one", None), ("chunks", {"train_fp32_chunk_mib": 1})):
        eng = Engine(args, hier=True, train=True, dtype="fp32", tuning=tuning)
        if tuning:
            assert eng.get_tuning()["train_fp32_chunk_mib"] == 1
        eng.set_weights(p)
        out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
        torch.cuda.synchronize()
        got[name] = (metrics_dict(out["metrics8"]), out["state"].cpu().numpy(), eng.export("grads"), out["rank"].cpu().numpy() if "rank" in out else None)
        eng.close()
    (m1, s1, g1, r1), (m2, s2, g2, r2) = got["one"], got["chunks"]
    for k in m1:
        assert abs(m1[k] - m2[k]) < 1e-6 * max(1.0, abs(m1[k])), (k, m1[k], m2[k])
    assert np.array_equal(s1, s2)
    if r1 is not None:
        assert np.array_equal(r1, r2)
    for k in g1:
        n = np.linalg.norm(g1[k])
        assert np.linalg.norm(g1[k] - g2[k]) <= 2e-6 * max(n, 1e-30), (k, np.linalg.norm(g1[k] - g2[k]) / max(n, 1e-30))


def test_fp32_training_mode_past_one_chunk_of_logits():
    """9,600 rows x 300,032 items x 4 B = 11.5 GB of logits = two 8 GiB row chunks (rounds 1-5a refused this batch): the step runs, and its loss and its
    scoring-layer bias gradient sit where the 16-bit training path puts them (that path is held to autograd in tests/test_gpu_train.py)."""
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.engine import Engine, metrics_dict
    big = hier_tcn_args(item_num=300_000, emb_dim=64, hidden_dim=32, num_layer=2, tcn_channel=[64, 64], kernel_size=3, max_activity_len=30, max_session_num=20,
                        batch_size=16)
    p = make_params(big)
    rng = np.random.default_rng(0)
    x, y, mask = make_batch(rng, 16, 20, 30, 300_000)
    res = {}
    for mode in ("fp32", "bf16"):
        eng = Engine(big, hier=True, train=True, dtype=mode)
        eng.set_weights(p)
        out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(np.zeros((16, 64), np.float32)))
        torch.cuda.synchronize()
        res[mode] = (metrics_dict(out["metrics8"])["loss"], eng.export("grads")["hier/tcn/dense/bias"])
        eng.close()
    (l32, g32), (l16, g16) = res["fp32"], res["bf16"]
    assert np.isfinite(l32) and abs(l32 - l16) < 2e-2 * max(1.0, abs(l16)), (l32, l16)
    assert np.linalg.norm(g32 - g16) < 3e-2 * np.linalg.norm(g32), np.linalg.norm(g32 - g16) / np.linalg.norm(g32)


@pytest.mark.parametrize("low", ["tcn", "gru"])
def test_run_hier_trains_in_the_fp32_mode(tmp_path, low):
    """`run_xing --train --storage_dtype fp32 [--model_low_type gru]` end to end on the synthetic XING CSV: three epochs of six batches through run_hier (pipelined train loop,
    TF-form Adam, checkpoint, evaluation between the epochs in the fp32 mode ON THE MASTERS the optimiser has just written): the loss falls, the
    checkpoint restores into a fresh fp32 evaluation model that evaluates identically."""
    import os
    from hiertcn_amd import checkpoint
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.data_loader import Dataloader_hier_model_xing
    from hiertcn_amd.data_prepare import load_xing
    from hiertcn_amd.model import HierTCNModel
    from hiertcn_amd.run_hier_xing import evaluate_hier, run_hier
    data = load_xing(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "interactions_synth.csv"))
    n_items = int(data[1][:, 1].max()) + 1
    H = 64 if low == "gru" else 32           # (HierGRU's hidden rows feed the scoring layer: one of its widths)
    args = hier_tcn_args(item_num=n_items, emb_dim=64, hidden_dim=H, num_layer=2, tcn_channel=[64, 64], kernel_size=5, batch_size=4, max_session_num=3,
                         max_activity_len=6, storage_dtype="fp32")
    args.model_low_type = low
    args.log_path, args.save_path = str(tmp_path) + "/log/", str(tmp_path) + "/model/"
    args.train, args.validate, args.test, args.save = True, False, True, True
    args.epoch_max, args.epoch_batches_train, args.save_epoch, args.test_epoch, args.learning_rate = 2, 6, 2, 2, 1e-2
    mk = lambda split: Dataloader_hier_model_xing(args, split, data=data)  # noqa: E731
    model = HierTCNModel(args)
    assert model.engine.exact and model.engine.train and model.engine.mode == "fp32" and model.engine.low_gru == (low == "gru")
    st0 = np.zeros((4, 2 * H))
    os.makedirs(args.log_path + "before/")
    before, *_ = evaluate_hier(args, mk("test"), st0, model, prefix="before")
    run_hier(args, mk("train"), mk("validate"), mk("test"), sess=model)
    txt = open(args.log_path + "test/results.txt").read()
    after_loss = float(txt.split("final\n")[1].split()[1].rstrip(","))
    assert after_loss < before["loss_epoch"] - 0.05, (before["loss_epoch"], after_loss)
    params = checkpoint.load(args.save_path + "2.ckpt")
    fresh = HierTCNModel(args, params, train=False)
    os.makedirs(args.log_path + "a/")
    os.makedirs(args.log_path + "b/")
    ra, *_ = evaluate_hier(args, mk("test"), st0, model, prefix="a")
    rb, *_ = evaluate_hier(args, mk("test"), st0, fresh, prefix="b")
    for k in ra:
        assert abs(ra[k] - rb[k]) < 1e-6 * max(1.0, abs(rb[k])), (k, ra[k], rb[k])
    model.engine.close()
    fresh.engine.close()


def test_single_level_model_trains_in_the_fp32_mode():
    """model_tcn (model.py:71-73) with unequal widths ([64, 64, 128, 128]: a Dense residual on block 2) and T = 70 -- BASELINE.json configs[0]'s model
    family -- in the fp32 training mode: loss and every gradient against fp64 autograd on the unrounded weights, then TcnModel.train_step lowers the loss."""
    from hiertcn_amd.args import make_args
    from hiertcn_amd.engine import Engine, metrics_dict
    from hiertcn_amd.model import TcnModel
    from hiertcn_amd.params import init_params
    N = 600
    a = make_args([], model_type="hier", item_num=N, emb_dim=64, tcn_channel=[64, 64, 128, 128], kernel_size=3, storage_dtype="fp32")
    a.model_type = "tcn"
    rng = np.random.default_rng(5)
    p = init_params(a, 2, "tcn")
    p = {k: ((v + 0.05 * rng.standard_normal(v.shape).astype(np.float32)) if k.endswith("/bias") else (v * (8.0 if k.endswith("tcn/dense/kernel") else 1.5)).astype(np.float32))
         for k, v in p.items()}
    B, T = 5, 70
    lens = rng.integers(5, T + 1, B)
    y = (rng.integers(1, N, (B, T)) * (np.arange(T)[None, :] < lens[:, None])).astype(np.int32)
    x = np.zeros_like(y)
    x[:, 1:] = y[:, :-1]
    eng = Engine(a, hier=False, train=True, dtype="fp32")
    eng.set_weights(p)
    out = eng.train_forward_backward(dev(x).view(B, 1, T), dev(y).view(B, 1, T))
    grads = eng.export("grads")
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in p.items()}
    xt, yt = torch.as_tensor(x, dtype=torch.int64), torch.as_tensor(y, dtype=torch.int64)
    h = (F.embedding(xt, P["tcn/emb/kernel"]) * (xt > 0)[..., None]).transpose(1, 2)
    for i in range(len(a.tcn_channel)):
        pre = f"tcn/temporal_conv_net/tblock_{i}"
        W = P[f"{pre}/conv1/kernel"].permute(2, 1, 0)
        dd, K = 2 ** i, W.shape[-1]
        c = torch.relu(F.conv1d(F.pad(h, ((K - 1) * dd, 0)), W, P[f"{pre}/conv1/bias"], dilation=dd))
        r = (h.transpose(1, 2) @ P[f"{pre}/dense/kernel"] + P[f"{pre}/dense/bias"]).transpose(1, 2) if f"{pre}/dense/kernel" in P else h
        h = torch.relu(c + r)
    z = h.transpose(1, 2) @ P["tcn/dense/kernel"] + P["tcn/dense/bias"]
    m = (yt > 0).double()
    ce = F.cross_entropy((z * m[..., None]).reshape(-1, N), yt.reshape(-1), reduction="none").reshape(B, -1) * m
    act = m.sum(1)
    ref_loss = (ce.sum(1) / (act + 1e-6)).sum() / (act > 0).double().sum()
    ref_loss.backward()
    assert abs(metrics_dict(out["metrics8"])["loss"] - float(ref_loss.detach())) < 2e-5 * max(1.0, float(ref_loss.detach()))
    assert set(grads) == set(P) and "tcn/temporal_conv_net/tblock_2/dense/kernel" in grads
    for k in sorted(P):
        g, r = grads[k].astype(np.float64), P[k].grad.numpy()
        rel = np.linalg.norm(g - r) / max(np.linalg.norm(r), 1e-30)
        assert rel < 1e-4, (k, rel)
    eng.close()
    model = TcnModel(a, p, train=True)
    assert model.engine.mode == "fp32" and model.engine.train
    losses = [model.train_step(x, y, 1e-2)["loss"] for _ in range(4)]
    assert losses[-1] < losses[0], losses
    model.engine.close()


# ---------------------------------------------------------------------------------------------------------------------
# HierGRU (model_low_type 'gru': the reference's DEFAULT low level, args.py:40 -> model_hier.py:67-68 -> model_gru.py:46-55) in the fp32 training mode
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("d,H,N,nl,shape", [(64, 64, 700, 2, (6, 3, 8)), (128, 64, 1500, 1, (6, 3, 8)), (128, 128, 900, 2, (9, 4, 7)),
                                            (256, 256, 600, 2, (5, 3, 6)), (128, 128, 3000, 2, (32, 10, 20))],
                         ids=["d64_nl2", "d128_h64_nl1", "d128_h128_nl2", "d256_h256_nl2", "reference_batch_d128"])
def test_hiergru_trains_in_the_fp32_mode(d, H, N, nl, shape):
    """The forward on the masters (fp32 'emb' projection, the exact-fp32 GRU cell chain with every record kept, materialised logits) and the
    backward (BPTT over the L steps as exact-fp32 GEMMs) against fp64 autograd on the UNROUNDED weights: loss, carried state, every gradient;
    then Adam steps lower the loss and the masters come back under the reference's variable names."""
    import copy
    from hiertcn_amd.engine import Engine, metrics_dict
    from tests.test_gpu_hiergru import torch_loss_gru
    args = copy.copy(small_args(N=N, d=d, H=H, nl=nl))
    args.model_low_type = "gru"
    p = {k: np.array(v, np.float32) for k, v in make_params(args).items()}
    p["hier/emb/kernel"] *= 3.0
    p["hier/gru/dense/kernel"] *= 12.0          # logits of a trained model's scale: the regime in which 16-bit storage is visibly off
    eng = Engine(args, hier=True, train=True, dtype="fp32")
    eng.set_weights(p)
    rng = np.random.default_rng(d + 11)
    B, S, L = shape
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, H * nl))).astype(np.float32)
    st0[:2] = 0.0
    out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    torch.cuda.synchronize()
    grads = eng.export("grads")
    loss = metrics_dict(out["metrics8"])["loss"]
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in p.items()}
    ref_loss, ref_state = torch_loss_gru(args, P, x, y, mask, st0, ste_bf16=lambda t: t)
    ref_loss.backward()
    assert abs(loss - float(ref_loss.detach())) < 2e-5 * max(1.0, abs(float(ref_loss.detach()))), (loss, float(ref_loss.detach()))
    assert np.abs(out["state"].cpu().numpy() - ref_state.detach().numpy()).max() < 1e-5
    assert set(grads) == set(P)
    report = []
    for k in sorted(P):
        g, r = grads[k].astype(np.float64), P[k].grad.numpy()
        nr = np.linalg.norm(r)
        report.append((k, np.linalg.norm(g - r) / max(nr, 1e-30), nr))
    bad = [(k, rel) for k, rel, nr in report if nr > 1e-12 and rel > 1e-4]
    assert not bad, "\n".join(f"{k}: rel {rel:.3e}" for k, rel in bad) + "\nall:\n" + "\n".join(f"{k}: rel {rel:.3e} |ref| {nr:.3e}" for k, rel, nr in report)
    print("HierGRU fp32 mode: max relative gradient error %.2e (%s)" % max((rel, k) for k, rel, nr in report if nr > 1e-12))
    assert not grads["hier/emb/kernel"][0].any()
    l0 = loss
    for _ in range(3):
        eng.adam_step(1e-3)          # (1e-2 overshoots at this logit scale)
        out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    assert metrics_dict(out["metrics8"])["loss"] < l0
    assert set(eng.export("weights")) == set(p)
    eng.close()

