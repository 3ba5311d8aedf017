"""The single-level model_gru (`--model_type gru`: model.py:66-67 -> model_gru.py:16-63, the reference's GRU baseline on the run_nn loop):
oracle known answers and host-side width padding (CPU), then the HIP path against the oracle in every numerics mode, its training step
against autograd, and the GruModel / run_nn surface (-m gpu)."""
import os

import numpy as np
import pytest

from hiertcn_amd.args import make_args
from hiertcn_amd.params import init_params, pad_widths, param_shapes, unpad_widths
from oracle import hiertcn_oracle as O


def gru_args(N, d, H, nl=2, **kw):
    a = make_args([], model_type="hier", item_num=N, emb_dim=d, hidden_dim=H, num_layer=nl, **kw)   # 'hier' only to skip args_adjust's override
    a.model_type = "gru"
    return a


def noisy(params, seed=3):
    rng = np.random.default_rng(seed)
    return {k: (v + 0.05 * rng.standard_normal(v.shape).astype(np.float32)) if k.endswith("/bias") else v for k, v in params.items()}


def ids_batch(rng, B, T, N, full_rows=1):
    """the single-level loader's batch (data_loader.py:78-98): y = the user's events right-padded with 0, x = y shifted right by one (a null first step)."""
    lens = rng.integers(3, T + 1, B)
    lens[:full_rows] = T
    y = rng.integers(1, N, (B, T)) * (np.arange(T)[None, :] < lens[:, None])
    x = np.zeros_like(y)
    x[:, 1:] = y[:, :-1]
    return x.astype(np.int32), y.astype(np.int32)


def test_variables_and_args_adjust():
    a = gru_args(500, 128, 64)
    s = param_shapes(a, "gru")
    assert s["gru/emb/kernel"] == (500, 128) and s["gru/dense/kernel"] == (64, 500) and s["gru/dense/bias"] == (500,)
    assert s["gru/rnn/multi_rnn_cell/cell_0/gru_cell/gates/kernel"] == (128 + 64, 128)
    assert s["gru/rnn/multi_rnn_cell/cell_1/gru_cell/candidate/kernel"] == (64 + 64, 64)
    assert len(s) == 1 + 4 * 2 + 2
    p, po = init_params(a, 2, "gru"), O.init_params(a, 2, "gru")
    assert set(p) == set(po) == set(s) and all(p[k].shape == s[k] for k in s)
    assert np.all(p["gru/rnn/multi_rnn_cell/cell_0/gru_cell/gates/bias"] == 1.0)        # customed_gru_cell.py:312-314
    assert make_args([], model_type="gru").hidden_dim == 200                               # args.py:306-308


def test_sequence_length_is_a_count_used_as_a_prefix():
    """model_gru.py:51,109-118 on the loader's shifted x: the null first step is not counted, so every sequence is cut one step short -- the step
    that sees the user's last-but-one event outputs zeros and its logits are the dense bias; an all-null row outputs the bias everywhere."""
    a = gru_args(300, 64, 64)
    p = noisy(O.init_params(a, 2, "gru"))
    x = np.array([[0, 5, 7, 9, 0, 0], [0, 3, 8, 2, 6, 4], [0, 0, 0, 0, 0, 0]])
    z = O.model_gru(a, x, p)
    h = O.model_gru(a, x, p, return_hidden=True)
    assert h.shape == (3, 6, 64) and z.shape == (3, 6, 300)
    assert np.abs(h[0, :3]).max(-1).min() > 0 and not h[0, 3:].any()          # 3 non-null inputs -> steps 0..2 run, step 3 (id 9) does not
    assert np.abs(h[1, :5]).max(-1).min() > 0 and not h[1, 5:].any()
    assert not h[2].any()
    assert np.array_equal(z[0, 3], np.asarray(p["gru/dense/bias"], np.float64)) and np.array_equal(z[2, 0], z[0, 3])
    # step 0 runs on a zero input from the zero state: h' = (1 - u) tanh(bc) with u = sigmoid(bg_u)
    bg = np.asarray(p["gru/rnn/multi_rnn_cell/cell_0/gru_cell/gates/bias"], np.float64)
    bc = np.asarray(p["gru/rnn/multi_rnn_cell/cell_0/gru_cell/candidate/bias"], np.float64)
    h0 = (1 - O.sigmoid(bg[64:])) * np.tanh(bc)
    a1 = gru_args(300, 64, 64, nl=1)
    p1 = {k: v for k, v in p.items() if "cell_1" not in k}
    assert np.abs(O.model_gru(a1, x, p1, return_hidden=True)[0, 0] - h0).max() < 1e-12


def test_pad_gru_width_keeps_every_logit_and_round_trips():
    """hidden_dim 40 -> 64 and the reference's 200 -> 256 (args.py:306-308): a padded unit's state stays exactly zero, so every logit is unchanged."""
    for H0, H1 in ((40, 64), (200, 256)):
        a = gru_args(300, 64, H0)
        p = noisy(O.init_params(a, 2, "gru"))
        a2, p2 = pad_widths(a, p, "gru")
        assert a2.hidden_dim == H1 and {k: v.shape for k, v in p2.items()} == param_shapes(a2, "gru")
        x, _ = ids_batch(np.random.default_rng(0), 3, 11, 300)
        z0, z1 = O.model_gru(a, x, p), O.model_gru(a2, x, p2)
        assert np.abs(z0 - z1).max() < 1e-12
        h1 = O.model_gru(a2, x, p2, return_hidden=True)
        assert not h1[..., H0:].any()
        back = unpad_widths(a, p2, "gru")
        assert all(np.array_equal(back[k], p[k]) for k in p)
    with pytest.raises(ValueError):
        pad_widths(gru_args(50, 64, 600), O.init_params(gru_args(50, 64, 600), 2, "gru"), "gru")


# ---------------------------------------------------------------------------------------------------------
def _dev(v):
    import torch
    return torch.as_tensor(np.ascontiguousarray(v)).cuda().contiguous()


@pytest.mark.gpu
@pytest.mark.parametrize("d,H,N,nl,T", [(64, 64, 1000, 2, 9), (128, 128, 900, 1, 23), (128, 256, 777, 2, 70), (64, 64, 500, 2, 500)],
                         ids=["d64", "d128_nl1", "d128_h256_T70", "T500"])
def test_model_gru_hip_vs_oracle(d, H, N, nl, T):
    """16-bit storage path: hidden rows against the mirror oracle (zeros past the counted length exact), logits, loss, ranks; then the fp32 and
    split16 modes against the exact oracle on the unrounded weights."""
    from hiertcn_amd.engine import Engine, metrics_dict
    a = gru_args(N, d, H, nl)
    p = noisy(O.init_params(a, 3, "gru"))
    p["gru/dense/kernel"] = (p["gru/dense/kernel"] * 20).astype(np.float32)
    rng = np.random.default_rng(d + T)
    B = 7 if T < 100 else 3
    x, y = ids_batch(rng, B, T, N)
    eng = Engine(a, hier=False)
    assert eng.low_gru and not eng.hier and eng.kind == "gru" and eng.C == H
    eng.set_weights(p)
    pr = O.round_params_bf16(p, E=d)
    hid = eng.tcn_forward(_dev(x), None)
    got = hid.float().cpu().numpy().reshape(B, T, H)
    ref = O.model_gru(a, x, pr, "gru", O.MIRROR, return_hidden=True)
    assert np.array_equal(got == 0, ref == 0)
    assert not got[0, T - 1].any() and np.abs(got[0, T - 2]).max() > 0          # a full row: T - 1 non-null inputs -> its last step is cut
    diff = np.abs(got - ref)
    assert diff.max() <= 2.0 ** -7 and diff.mean() <= 2e-4
    out = eng.forward_eval(_dev(x).view(B, 1, T), _dev(y).view(B, 1, T), None, None, want_rows=True)
    m = metrics_dict(out["metrics8"])
    predm = O.model_gru(a, x, pr, "gru", O.MIRROR)
    logits = eng.score_logits(hid).cpu().numpy().reshape(B, T, N)
    valid = y > 0
    assert np.abs(logits - predm)[valid].max() < 2e-2 * max(1.0, np.abs(predm).max())
    refm = O.masked_loss_and_metrics(predm, y.astype(np.int64), N)
    assert abs(m["loss"] - refm["loss"]) < 5e-3 * max(1.0, abs(refm["loss"]))
    for k in ("recall10", "mrr", "rank_frac"):
        assert abs(m[k] - refm[k]) < 2e-2, (k, m[k], refm[k])
    eng.close()
    exact = O.model_gru(a, x, p, "gru", O.EXACT64)
    ref64 = O.masked_loss_and_metrics(exact, y.astype(np.int64), N)
    for mode in ("fp32", "split16"):
        ex = Engine(a, hier=False, dtype=mode)
        ex.set_weights(p)
        o = ex.forward_eval(_dev(x).view(B, 1, T), _dev(y).view(B, 1, T), None, None, want_rows=True)
        mm = metrics_dict(o["metrics8"])
        assert abs(mm["loss"] - ref64["loss"]) < 1e-4 * max(1.0, abs(ref64["loss"])), (mode, mm["loss"], ref64["loss"])
        assert (o["rank"].cpu().numpy().reshape(B, T) == ref64["ranks"])[valid].mean() > 0.99, mode
        if mode == "fp32":
            h32 = ex.tcn_forward(_dev(x), None)
            z32 = ex.score_logits(h32).cpu().numpy().reshape(B, T, N)
            assert np.abs(z32 - exact).max() < 2e-5 * max(1.0, np.abs(exact).max())
        ex.close()


def _torch_loss(a, P, x, y, ste):
    """model.py:56-62,66-67,102-117 with model_gru.py:46-57 on torch fp64 tensors; `ste`: the straight-through rounding of the rows the device stores in 16 bits."""
    import torch
    import torch.nn.functional as F
    B, T = x.shape
    H, nl, N = a.hidden_dim, a.num_layer, a.item_num
    xt, yt = torch.as_tensor(x, dtype=torch.int64), torch.as_tensor(y, dtype=torch.int64)
    u = ste(F.embedding(xt, P["gru/emb/kernel"]) * (xt > 0)[..., None])
    length = (xt > 0).sum(1)
    hs = torch.zeros((B, H * nl), dtype=torch.float64)
    outs = []
    for t in range(T):
        active = (t < length)[:, None]
        cur, new = u[:, t], []
        for l in range(nl):
            pre = f"gru/rnn/multi_rnn_cell/cell_{l}/gru_cell"
            h = hs[:, l * H:(l + 1) * H]
            g = torch.sigmoid(torch.cat([cur, h], 1) @ P[f"{pre}/gates/kernel"] + P[f"{pre}/gates/bias"])
            r, uu = g.chunk(2, 1)
            c = torch.tanh(torch.cat([cur, r * h], 1) @ P[f"{pre}/candidate/kernel"] + P[f"{pre}/candidate/bias"])
            cur = uu * h + (1 - uu) * c
            new.append(cur)
        hs = torch.where(active, torch.cat(new, 1), hs)
        outs.append(torch.where(active, cur, torch.zeros_like(cur)))
    z = ste(torch.stack(outs, 1)) @ P["gru/dense/kernel"] + P["gru/dense/bias"]
    m = (yt > 0).double()
    z = z * m[..., None]
    ce = F.cross_entropy(z.reshape(-1, N), yt.reshape(-1), reduction="none").reshape(B, T) * m
    act = m.sum(1)
    return (ce.sum(1) / (act + 1e-6)).sum() / (act > 0).double().sum()


@pytest.mark.gpu
@pytest.mark.parametrize("mode,d,H,N,nl,shape", [("bf16", 64, 64, 700, 2, (6, 12)), ("bf16", 128, 256, 600, 2, (5, 40)), ("bf16", 128, 128, 900, 1, (40, 9)),
                                                 ("fp32", 64, 64, 700, 2, (6, 12)), ("fp32", 128, 128, 900, 2, (9, 30))])
def test_model_gru_training_step_matches_autograd(mode, d, H, N, nl, shape):
    import torch
    from hiertcn_amd.engine import Engine, metrics_dict
    from tests.test_gpu_train import ste_bf16
    a = gru_args(N, d, H, nl)
    p = noisy(O.init_params(a, 2, "gru"))
    rng = np.random.default_rng(d + H)
    B, T = shape
    x, y = ids_batch(rng, B, T, N)
    eng = Engine(a, hier=False, train=True, dtype=None if mode == "bf16" else "fp32")
    eng.set_weights(p)
    out = eng.train_forward_backward(_dev(x).view(B, 1, T), _dev(y).view(B, 1, T))
    torch.cuda.synchronize()
    grads = eng.export("grads")
    loss = metrics_dict(out["metrics8"])["loss"]
    src = O.round_params_bf16(p, E=d) if mode == "bf16" else p
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in src.items()}
    ref = _torch_loss(a, P, x, y, ste_bf16 if mode == "bf16" else (lambda t: t))
    ref.backward()
    tol_l, tol_g = (2e-3, 2e-2) if mode == "bf16" else (2e-5, 1e-4)
    assert abs(loss - float(ref.detach())) < tol_l * max(1.0, abs(float(ref.detach()))), (loss, float(ref.detach()))
    assert set(grads) == set(P)
    bad, report = [], []
    for k in sorted(P):
        g, r = grads[k].astype(np.float64), P[k].grad.numpy()
        nr = np.linalg.norm(r)
        rel = np.linalg.norm(g - r) / max(nr, 1e-30)
        report.append(f"{k}: rel {rel:.3e} |ref| {nr:.3e}")
        if nr > 1e-9 and rel > tol_g:
            bad.append(k)
    assert not bad, "\n".join(report)
    assert not grads["gru/emb/kernel"][0].any()
    l0 = loss
    for _ in range(3):
        eng.adam_step(1e-2 if mode == "bf16" else 1e-3)
        out = eng.train_forward_backward(_dev(x).view(B, 1, T), _dev(y).view(B, 1, T))
    assert metrics_dict(out["metrics8"])["loss"] < l0
    assert set(eng.export("weights")) == set(p)
    eng.close()


@pytest.mark.gpu
def test_gru_model_and_run_nn_on_the_synthetic_xing_file(tmp_path):
    """`run_xing --model_type gru --train` end to end through run_nn (run_nn_xing.py:101-192) at a hidden width that is padded on upload (40 -> 64):
    evaluate_nn against the oracle loop, training lowers the test loss, the checkpoint holds the reference's variable names in the UNPADDED shapes
    and restores into a model that evaluates identically; the padded entries of the device masters are still exactly zero after training."""
    from hiertcn_amd import checkpoint
    from hiertcn_amd.data_loader import Dataloader_single_level_model_xing
    from hiertcn_amd.data_prepare import load_xing
    from hiertcn_amd.model import GruModel
    from hiertcn_amd.run_nn_xing import evaluate_nn, run_nn
    data = load_xing(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "interactions_synth.csv"))
    N = int(data[1][:, 1].max()) + 1
    a = gru_args(N, 64, 40)
    a.batch_size = 5
    a.log_path, a.save_path = str(tmp_path) + "/log/", str(tmp_path) + "/model/"
    os.makedirs(a.log_path + "test/")
    os.makedirs(a.log_path + "oracle/")
    p = noisy(init_params(a, 2, "gru"))
    model = GruModel(a, p, train=True)
    assert model.engine.H == 64 and model._padded
    a_pad, p_pad = pad_widths(a, p, "gru")
    pr = O.round_params_bf16(p_pad, E=64)

    class OracleNN:
        def eval_step(self, x, y):
            yi = np.asarray(y).astype(np.int64)
            m = O.masked_loss_and_metrics(O.model_gru(a_pad, np.asarray(x).astype(np.int64), pr, "gru", O.MIRROR), yi, N)
            return dict(loss=float(m["loss"]), ranks_float_mean_user=float(m["rank_frac"]), reciprocal_ranks_float_mean_user=float(m["mrr"]),
                        recall1_mean_user=float(m["recall1"]), recall5_mean_user=float(m["recall5"]), recall10_mean_user=float(m["recall10"]))

    mk = lambda split: Dataloader_single_level_model_xing(a, split, data=data)  # noqa: E731
    got, _, _, _ = evaluate_nn(a, mk("test"), model, prefix="test")
    want, _, _, _ = evaluate_nn(a, mk("test"), OracleNN(), prefix="oracle")
    assert abs(got["loss_epoch"] - want["loss_epoch"]) < 2e-3
    for k in ("recall10_epoch_mean", "reciprocal_ranks_float_epoch_mean", "ranks_float_epoch_mean"):
        assert abs(got[k] - want[k]) < 5e-3, (k, got[k], want[k])
    a.train, a.validate, a.test, a.save = True, False, True, True
    a.epoch_max, a.epoch_batches_train, a.save_epoch, a.test_epoch, a.learning_rate = 2, 8, 2, 2, 1e-2
    run_nn(a, mk("train"), mk("validate"), mk("test"), sess=model)
    after, _, _, _ = evaluate_nn(a, mk("test"), model, prefix="test")
    assert after["loss_epoch"] < got["loss_epoch"] - 0.05, (got["loss_epoch"], after["loss_epoch"])
    params = checkpoint.load(a.save_path + "2.ckpt")
    assert {k: v.shape for k, v in params.items()} == param_shapes(a, "gru")
    raw = model.engine.export("weights", root="gru")                      # the device masters, padded shapes
    back = pad_widths(a, params, "gru")[1]
    assert all(np.array_equal(raw[k], back[k]) for k in raw)               # ... whose padded entries are still exactly zero
    fresh = GruModel(a, params, train=False)
    again, _, _, _ = evaluate_nn(a, mk("test"), fresh, prefix="test")
    for k in after:
        assert abs(after[k] - again[k]) < 1e-6 * max(1.0, abs(again[k])), (k, after[k], again[k])
    model.engine.close()
    fresh.engine.close()


@pytest.mark.gpu
def test_model_gru_shim_signature():
    """hiertcn_amd.model_gru.model_gru(args, x, ...) -> pred with the reference's argument order (model_gru.py:16); pred[b, t] = logits."""
    from hiertcn_amd.model import reset_scopes
    from hiertcn_amd.model_gru import model_gru
    reset_scopes()
    a = gru_args(400, 64, 64)
    x, _ = ids_batch(np.random.default_rng(1), 4, 10, 400)
    pred = model_gru(a, x, None, None, None, False, "gru")
    assert pred.shape == (4, 10, 400)
    z = pred.materialize().cpu().numpy()
    assert z.shape == (4, 10, 400) and np.isfinite(z).all()
    assert np.array_equal(z[0, 9], z[0, 9 - 0]) and np.abs(z[0, 9] - z[0, 8]).max() > 0     # the full row's last step is cut (logits = the dense bias), the one before is not
    with pytest.raises(NotImplementedError):
        model_gru(a, x, return_state=True)
    reset_scopes()
