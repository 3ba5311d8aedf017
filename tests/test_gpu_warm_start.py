"""-m gpu: warm-start runs (args.warm_start; model.py:40,102-103 `mask_y *= mask_warmstart`, fed from the events' timestamps at
run_hier_xing.py:125-129,286-289 / run_nn_xing.py:38-41,148-151) through hiertcn_set_score_ids: the loss, the ranks, the metrics and the
gradients see y with the masked steps set to 0, the session mean-pool keeps reading y."""
import copy
import os

import numpy as np
import pytest

from oracle import hiertcn_oracle as O
from tests.gpu_common import OracleSession, as_lists, make_batch, make_params, small_args

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda().contiguous()  # noqa: E731


def _case(low, d=64, H=64, N=900, B=7, S=3, L=9, seed=4):
    args = copy.copy(small_args(N=N, d=d, H=H, nl=2))
    args.model_low_type = low
    p = make_params(args)
    rng = np.random.default_rng(seed)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, 2 * H))).astype(np.float32)
    keep = rng.random(y.shape) < 0.55
    keep[0] = False                                   # a lane without any scored step: it is no "user" of this batch (model.py:112-113)
    ys = (y * keep).astype(np.int32)
    return args, p, x, y, ys, mask, st0


@pytest.mark.parametrize("low", ["tcn", "gru"])
@pytest.mark.parametrize("mode", ["bf16", "fp32", "split16"])
def test_evaluation_with_score_ids_matches_the_oracle(low, mode):
    from hiertcn_amd.engine import Engine, metrics_dict
    args, p, x, y, ys, mask, st0 = _case(low)
    B, S, L = x.shape
    N = args.item_num
    eng = Engine(args, hier=True, dtype=None if mode == "bf16" else mode)
    eng.set_weights(p)
    plain = eng.forward_eval(dev(x), dev(y), dev(mask), dev(st0), want_rows=True)
    st_plain, rank_plain = plain["state"].cpu().numpy(), plain["rank"].cpu().numpy().copy()
    out = eng.forward_eval(dev(x), dev(y), dev(mask), dev(st0), want_rows=True, y_score=dev(ys))
    m = metrics_dict(out["metrics8"])
    assert np.array_equal(out["state"].cpu().numpy(), st_plain)                       # the session mean-pool still read y
    rank = out["rank"].cpu().numpy()
    scored = ys.reshape(B, S * L) > 0
    assert not rank[~scored].any() and not out["ce"].cpu().numpy()[~scored].any()
    assert np.array_equal(rank[scored], rank_plain[scored])                           # a scored row's rank does not depend on which other rows are scored
    assert m["user_count"] == B - 1 and m["valid_steps"] == scored.sum()
    again = eng.forward_eval(dev(x), dev(y), dev(mask), dev(st0), want_rows=True)    # one-shot: the next call is an ordinary one
    assert np.array_equal(again["rank"].cpu().numpy(), rank_plain)
    xl, yl, ml = as_lists(x, y, mask)
    src = O.round_params_bf16(p, E=args.emb_dim) if mode == "bf16" else p
    pred, _ = O.model_hier(args, xl, yl, ml, st0, src, O.MIRROR if mode == "bf16" else O.EXACT64)
    ref = O.masked_loss_and_metrics(pred, ys.reshape(B, S * L).astype(np.int64), N)
    tol = 2e-3 if mode == "bf16" else 1e-5
    assert abs(m["loss"] - ref["loss"]) < tol * max(1.0, abs(ref["loss"])), (m["loss"], ref["loss"])
    for k in ("recall10", "mrr", "rank_frac"):
        assert abs(m[k] - ref[k]) < (2e-2 if mode == "bf16" else 1e-6), (k, m[k], ref[k])
    if mode != "bf16":
        assert (rank == ref["ranks"])[scored].mean() > 0.99
    eng.close()


def _torch_loss(args, P, x, y, ys, mask, state, low, ste):
    """model_hier.py:39-93 + model.py:102-117 on fp64 torch tensors with the warm-start mask: the pool reads y, the loss reads ys."""
    F = torch.nn.functional
    B, S, L = x.shape
    H, nl, N = args.hidden_dim, args.num_layer, args.item_num
    st = torch.as_tensor(state, dtype=torch.float64)
    E = P["hier/emb/kernel"]
    xt, yt = torch.as_tensor(x, dtype=torch.int64), torch.as_tensor(y, dtype=torch.int64)
    mk = torch.as_tensor(mask, dtype=torch.float64)

    def cell(scope, l, cur, h):
        pre = f"{scope}/multi_rnn_cell/cell_{l}/gru_cell"
        g = torch.sigmoid(torch.cat([cur, h], 1) @ P[f"{pre}/gates/kernel"] + P[f"{pre}/gates/bias"])
        r, uu = g.chunk(2, 1)
        c = torch.tanh(torch.cat([cur, r * h], 1) @ P[f"{pre}/candidate/kernel"] + P[f"{pre}/candidate/bias"])
        return uu * h + (1 - uu) * c

    logits = []
    for s in range(S):
        e = F.embedding(xt[:, s], E) * (xt[:, s] > 0)[..., None]
        cat = torch.cat([e, st[:, None, :].expand(-1, L, -1)], -1)
        if low == "gru":
            u = ste(cat @ P["hier/gru/emb/kernel"])
            length = (u.detach().abs().amax(2) > 0).sum(1)
            hs = torch.zeros((B, H * nl), dtype=torch.float64)
            outs = []
            for t in range(L):
                active = (t < length)[:, None]
                cur, new = u[:, t], []
                for l in range(nl):
                    cur = cell("hier/gru/rnn", l, cur, hs[:, l * H:(l + 1) * H])
                    new.append(cur)
                hs = torch.where(active, torch.cat(new, 1), hs)
                outs.append(torch.where(active, cur, torch.zeros_like(cur)))
            logits.append(ste(torch.stack(outs, 1)) @ P["hier/gru/dense/kernel"] + P["hier/gru/dense/bias"])
        else:
            a = ste(cat @ P["hier/tcn/emb/kernel"]).transpose(1, 2)
            for i in range(len(args.tcn_channel)):
                pre = f"hier/tcn/temporal_conv_net/tblock_{i}"
                W = P[f"{pre}/conv1/kernel"].permute(2, 1, 0)
                dd, K = 2 ** i, W.shape[-1]
                c = ste(torch.relu(F.conv1d(F.pad(a, ((K - 1) * dd, 0)), W, P[f"{pre}/conv1/bias"], dilation=dd)))
                a = ste(torch.relu(c + a))
            logits.append(a.transpose(1, 2) @ P["hier/tcn/dense/kernel"] + P["hier/tcn/dense/bias"])
        ey = F.embedding(yt[:, s], E) * (yt[:, s] > 0)[..., None]
        cur = ey.sum(1) / (yt[:, s] > 0).sum(1, keepdim=True) + P["hier/emb/bias"]
        new = []
        for l in range(nl):
            cur = cell("hier", l, cur, st[:, l * H:(l + 1) * H])
            new.append(cur)
        st = torch.cat(new, 1) * mk[:, s:s + 1]
    z = torch.cat(logits, 1)
    yy = torch.as_tensor(ys, dtype=torch.int64).reshape(B, S * L)          # <- the warm-start-masked targets
    m = (yy > 0).double()
    z = z * m[..., None]
    ce = F.cross_entropy(z.reshape(-1, N), yy.reshape(-1), reduction="none").reshape(B, -1) * m
    act = m.sum(1)
    users = (act > 0).double().sum()
    return ((ce.sum(1) / (act + 1e-6)).sum() / users), st


@pytest.mark.parametrize("low", ["tcn", "gru"])
def test_training_step_with_score_ids_matches_autograd_in_the_fp32_mode(low):
    """The fp32 training mode (nothing rounded): every gradient of the warm-start step against fp64 autograd -- the masked steps give no scoring
    gradient, but their targets still reach the item table through the session mean-pool's backward."""
    from hiertcn_amd.engine import Engine, metrics_dict
    args, p, x, y, ys, mask, st0 = _case(low, seed=6)
    eng = Engine(args, hier=True, train=True, dtype="fp32")
    eng.set_weights(p)
    out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0), y_score=dev(ys))
    torch.cuda.synchronize()
    grads = eng.export("grads")
    loss = metrics_dict(out["metrics8"])["loss"]
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in p.items()}
    ref, ref_state = _torch_loss(args, P, x, y, ys, mask, st0, low, lambda t: t)
    ref.backward()
    assert abs(loss - float(ref.detach())) < 2e-5 * max(1.0, abs(float(ref.detach())))
    assert np.abs(out["state"].cpu().numpy() - ref_state.detach().numpy()).max() < 1e-5
    bad = []
    for k in sorted(P):
        g, r = grads[k].astype(np.float64), P[k].grad.numpy()
        nr = np.linalg.norm(r)
        rel = np.linalg.norm(g - r) / max(nr, 1e-30)
        if nr > 1e-12 and rel > 1e-4:
            bad.append((k, rel))
    assert not bad, bad
    # ... and differs from the unmasked step's gradient
    eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    g2 = eng.export("grads")
    assert np.linalg.norm(g2["hier/emb/kernel"] - grads["hier/emb/kernel"]) > 1e-3 * np.linalg.norm(grads["hier/emb/kernel"])
    eng.close()


def test_training_step_with_score_ids_in_the_16_bit_mode_and_two_halves():
    """16-bit storage path (fused scoring backward): begin + finish with y_score == forward_backward with y_score, and the loss equals the
    evaluation loss of the same masked batch."""
    from hiertcn_amd.engine import Engine, metrics_dict
    args, p, x, y, ys, mask, st0 = _case("tcn", d=128, H=64, N=1500, seed=8)
    eng = Engine(args, hier=True, train=True)
    eng.set_weights(p)
    a = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0), y_score=dev(ys))
    ga = eng.export("grads")
    la = metrics_dict(a["metrics8"])["loss"]
    b = eng.train_step_begin(dev(x), dev(y), dev(mask), dev(st0), y_score=dev(ys))
    eng.train_step_finish(dev(x), dev(y), dev(mask))
    gb = eng.export("grads")
    assert abs(metrics_dict(b["metrics8"])["loss"] - la) <= 1e-6 * abs(la)        # (the scoring kernel's item walk starts where the XCD's walk stands: sums in another order)
    for k in ga:
        assert np.linalg.norm(ga[k] - gb[k]) <= 1e-5 * max(np.linalg.norm(ga[k]), 1e-30), k     # (fp32 atomics reorder sums)
    ev = eng.forward_eval(dev(x), dev(y), dev(mask), dev(st0), y_score=dev(ys))
    assert abs(metrics_dict(ev["metrics8"])["loss"] - la) < 2e-3 * max(1.0, abs(la))
    plain = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    assert metrics_dict(plain["metrics8"])["valid_steps"] > metrics_dict(a["metrics8"])["valid_steps"]
    eng.close()


def test_a_failed_step_never_leaves_score_ids_armed():
    """the one-shot ids are armed immediately before the C call, behind every check of the Python side, and any failure --
    a Python-side shape check, a rejected argument inside the library, a failed hiertcn_train_step_begin, a finish that raises -- leaves
    nothing behind: the next plain step is scored against its own targets."""
    import ctypes as C

    from hiertcn_amd.engine import Engine, HierTCNError, metrics_dict
    args, p, x, y, ys, mask, st0 = _case("tcn", d=64, H=64, N=900, seed=5)
    B, S, L = x.shape
    eng = Engine(args, hier=True, train=True)
    eng.set_weights(p)
    X, Y, YS, M, ST = dev(x), dev(y), dev(ys), dev(mask), dev(st0)
    plain = metrics_dict(eng.forward_eval(X, Y, M, ST)["metrics8"])
    masked = metrics_dict(eng.forward_eval(X, Y, M, ST, y_score=YS)["metrics8"])
    assert masked["valid_steps"] < plain["valid_steps"]
    full_steps = plain["valid_steps"]

    def next_plain_eval_is_plain(where):
        got = metrics_dict(eng.forward_eval(X, Y, M, ST)["metrics8"])
        assert got["valid_steps"] == full_steps and abs(got["loss"] - plain["loss"]) <= 1e-6 * abs(plain["loss"]), (where, got, plain)

    # (1) Python-side checks fail AFTER y_score itself was accepted: evaluation, training, the first half of a step
    bad_state = ST[:, :-1].contiguous()
    for name, call in (("forward_eval", lambda: eng.forward_eval(X, Y, M, bad_state, y_score=YS)),
                       ("train_forward_backward", lambda: eng.train_forward_backward(X, Y, M, bad_state, y_score=YS)),
                       ("train_step_begin", lambda: eng.train_step_begin(X, Y, M, bad_state, y_score=YS))):
        with pytest.raises(HierTCNError):
            call()
        next_plain_eval_is_plain(name + " / python check")
    # (2) the library rejects the call with the ids armed: consumed all the same (forward_eval, train_forward_backward), dropped (begin)
    lib, h = eng.lib, eng._h
    for name, call in (
            ("hiertcn_forward_eval", lambda: lib.hiertcn_forward_eval(h, None, None, None, None, B, S, L, None, 0, None, None, None, None)),
            ("hiertcn_train_forward_backward", lambda: lib.hiertcn_train_forward_backward(h, None, None, None, None, B, S, L, None, 0, None, None, None)),
            ("hiertcn_train_step_begin", lambda: lib.hiertcn_train_step_begin(h, None, None, None, None, B, S, L, None, 0, None, None, None, None))):
        eng.set_score_ids(YS)
        assert call() != 0
        next_plain_eval_is_plain(name + " / library check")
    # (3) a begin that succeeded keeps them for ITS finish; a finish that fails consumes them
    eng.train_step_begin(X, Y, M, ST, y_score=YS)
    assert lib.hiertcn_train_step_finish(h, None, None, None, B, S, L, None, 0, None) != 0
    next_plain_eval_is_plain("failed finish")
    # (4) and the ordinary pair still carries them from begin to finish
    a = eng.train_forward_backward(X, Y, M, ST, y_score=YS)
    ga = eng.export("grads")
    eng.train_step_begin(X, Y, M, ST, y_score=YS)
    eng.train_step_finish(X, Y, M)
    gb = eng.export("grads")
    for k in ga:
        assert np.linalg.norm(ga[k] - gb[k]) <= 1e-5 * max(np.linalg.norm(ga[k]), 1e-30), k
    assert metrics_dict(a["metrics8"])["valid_steps"] == masked["valid_steps"]
    next_plain_eval_is_plain("after an ordinary pair")
    eng.close()


def test_run_hier_and_run_nn_in_the_warm_start_protocol(tmp_path, monkeypatch):
    """`run_xing --warm_start` (run_xing.py:20-35: all three loaders on the 'train' users; training counts the events up to the warm-start time,
    evaluation the later ones): evaluate_hier on the device against the oracle session, then run_hier trains and the warm-start test loss falls;
    the single-level loop (run_nn) with a TcnModel likewise."""
    from hiertcn_amd import run_common
    from hiertcn_amd.args import hier_tcn_args, make_args
    from hiertcn_amd.data_loader import Dataloader_hier_model_xing, Dataloader_single_level_model_xing
    from hiertcn_amd.data_prepare import load_xing
    from hiertcn_amd.model import HierTCNModel, TcnModel
    from hiertcn_amd.run_hier_xing import evaluate_hier, run_hier
    from hiertcn_amd.run_nn_xing import evaluate_nn, run_nn
    data = load_xing(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "interactions_synth.csv"))
    monkeypatch.setattr(run_common, "WARM_START_TIME", float(np.quantile(data[1][:, 3], 0.3)))   # (every 8-lane batch then has scored users: tests/test_run_loop.py)
    N = int(data[1][:, 1].max()) + 1
    args = hier_tcn_args(item_num=N, emb_dim=64, hidden_dim=32, num_layer=2, tcn_channel=[64, 64], kernel_size=5, batch_size=8, max_session_num=3,
                         max_activity_len=6)
    args.warm_start = True
    args.log_path, args.save_path = str(tmp_path) + "/log/", str(tmp_path) + "/model/"
    for d in ("test", "oracle", "before"):
        os.makedirs(args.log_path + d + "/")
    p = make_params(args)
    model = HierTCNModel(args, p, train=True)
    mk = lambda: Dataloader_hier_model_xing(args, "train", data=data)  # noqa: E731
    st0 = np.zeros((8, 64))
    got, _, _, _, _ = evaluate_hier(args, mk(), st0, model, prefix="before")                       # args.warm_start is picked up
    want, _, _, _, _ = evaluate_hier(args, mk(), st0, OracleSession(args, O.round_params_bf16(p, E=64), O.MIRROR), prefix="oracle")
    assert np.isfinite(got["loss_epoch"]) and abs(got["loss_epoch"] - want["loss_epoch"]) < 2e-3
    assert abs(got["recall10_epoch_mean"] - want["recall10_epoch_mean"]) < 2e-2
    plain, _, _, _, _ = evaluate_hier(args, mk(), st0, model, prefix="before", warm_start=False)
    assert abs(plain["loss_epoch"] - got["loss_epoch"]) > 1e-4
    args.train, args.validate, args.test, args.save = True, False, True, False
    args.epoch_max, args.epoch_batches_train, args.test_epoch, args.learning_rate = 2, 10, 2, 1e-2
    run_hier(args, mk(), mk(), mk(), sess=model)
    after, _, _, _, _ = evaluate_hier(args, mk(), st0, model, prefix="test")
    import json
    tr = [json.loads(l)["value"] for l in open(args.log_path + "train/scalars.jsonl") if '"loss_epoch"' in l]
    assert len(tr) == 3 and tr[-1] < tr[0] - 0.02, tr                              # the loss over the events up to the warm-start time falls ...
    assert np.isfinite(after["loss_epoch"]) and after["loss_epoch"] != got["loss_epoch"]       # (the synthetic file holds no signal the later events could profit from)
    model.engine.close()
    # single-level loop
    a = make_args([], model_type="hier", item_num=N, emb_dim=64, tcn_channel=[64, 64], kernel_size=3)
    a.model_type, a.batch_size, a.warm_start = "tcn", 8, True
    a.log_path, a.save_path = str(tmp_path) + "/log_nn/", str(tmp_path) + "/model_nn/"
    os.makedirs(a.log_path + "test/")
    tm = TcnModel(a, train=True)
    mk1 = lambda: Dataloader_single_level_model_xing(a, "train", data=data)  # noqa: E731
    before, _, _, _ = evaluate_nn(a, mk1(), tm, prefix="test")
    plain1, _, _, _ = evaluate_nn(a, mk1(), tm, prefix="test", warm_start=False)
    assert np.isfinite(before["loss_epoch"]) and abs(before["loss_epoch"] - plain1["loss_epoch"]) > 1e-5
    a.train, a.validate, a.test, a.save = True, False, True, False
    a.epoch_max, a.epoch_batches_train, a.test_epoch, a.learning_rate = 2, 10, 2, 1e-2
    run_nn(a, mk1(), mk1(), mk1(), sess=tm)
    after1, _, _, _ = evaluate_nn(a, mk1(), tm, prefix="test")
    tr1 = [json.loads(l)["value"] for l in open(a.log_path + "train/scalars.jsonl") if '"loss_epoch"' in l]
    assert len(tr1) == 3 and tr1[-1] < tr1[0] - 0.02, tr1
    assert np.isfinite(after1["loss_epoch"]) and after1["loss_epoch"] != before["loss_epoch"]
    tm.engine.close()
