"""-m gpu: the mixture-output model (BASELINE.json configs[4]: model_tcn_gmm.py:12-44, loss.py:247-257) on the HIP path against
oracle/gmm_oracle.py.  Pinned: the trunk on one-hot input (row gathers), the head GEMM (every mean sampled through the
materialised mu_m, the fused |mu_m|^2 / mu_m[target] / prior logits), both likelihood losses and the masked lane mean.  The decode
itself is the build's own choice (parity unpinned: oracle/gmm_oracle.py)."""
import numpy as np
import pytest

from hiertcn_amd.args import make_args
from oracle import gmm_oracle as G
from oracle import hiertcn_oracle as O

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _args(N, M, channels, K, loss="gmm_prob"):
    return make_args([], model_type="tcn_gmm", item_num=N, output_dim=N, num_mixture=M, sigma=0.1, loss=loss, tcn_channel=list(channels),
                     kernel_size=K, adjust=False)


@pytest.mark.parametrize("table_pass", [False, True])
@pytest.mark.parametrize("N,M,channels,K,B,T", [(1500, 5, (64, 64), 5, 6, 23), (20_778, 5, (256, 256), 5, 4, 40), (3001, 3, (128, 256, 256), 3, 5, 17),
                                                (2500, 3, (64, 128), 3, 4, 19), (700, 2, (512,), 2, 3, 9)])
def test_tcn_gmm_forward_head_and_losses_vs_oracle(N, M, channels, K, B, T, table_pass, monkeypatch):
    """table_pass False: |mu_m|^2 through the Gram form (csrc/gmm_gram.hip: a C x C quadratic form of the hidden row; the default wherever
    3 C < D and C <= 256); True: HIERTCN_GMM_TABLE_PASS=1, the pass over the [D, C] mean tables.  Same assertions against the oracle."""
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    if table_pass:
        monkeypatch.setenv("HIERTCN_GMM_TABLE_PASS", "1")   # read at hiertcn_gmm_create
    else:
        monkeypatch.delenv("HIERTCN_GMM_TABLE_PASS", raising=False)
    a = _args(N, M, channels, K)
    p = G.init_params(a, seed=7)
    rng = np.random.default_rng(N + M)
    for k in p:   # non-zero biases so that a dropped or misplaced bias shows
        if k.endswith("bias") or k.endswith("biases"):
            p[k] = (0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    model = TcnGmmModel(a, p)
    ids = rng.integers(1, N, (B, T + 1))
    lens = rng.integers(2, T + 1, B)
    ids = ids * (np.arange(T + 1)[None, :] <= lens[:, None])           # right padding with the null id
    x, y = ids[:, :-1].astype(np.int32), ids[:, 1:].astype(np.int32)
    x[:, 0] = 0                                                          # a sequence starts from the null id (data_loader.py:246-261)
    dev = lambda v: torch.as_tensor(np.ascontiguousarray(v)).to("cuda:0")  # noqa: E731

    # trunk: bf16-mirror oracle (weights rounded as stored, activations rounded after every block)
    pr = {k: (O.round_bf16(v) if (k.endswith("kernel") or k.endswith("weights")) else v) for k, v in p.items()}
    pr["tcn_gmm/fully_connected/weights"][:, :M] = p["tcn_gmm/fully_connected/weights"][:, :M]   # the M prior columns stay fp32 on the device
    hid = model.forward(dev(x))
    h_ref = G.tcn_gmm_hidden(a, x, pr, "tcn_gmm", O.MIRROR)
    assert np.abs(hid.float().cpu().numpy().reshape(B, T, -1) - h_ref).max() < 2e-2 * max(1.0, np.abs(h_ref).max())   # one bf16 ulp of the largest activation
    # head on the DEVICE's hidden rows (so that the comparison below isolates the head): pred = h . W + b in fp64
    h_dev = hid.float().cpu().numpy().astype(np.float64).reshape(B, T, -1)
    pred = G.gmm_head(h_dev, pr, "tcn_gmm")                                                       # [B, T, M (1 + D)]
    log_pi, mu = G.decode(pred, M, N)
    loss, sq, tg, pl = model.loss_rows(hid, dev(y), "gmm_prob", parts=True)
    sq, tg, pl = sq.cpu().numpy().reshape(M, B, T), tg.cpu().numpy().reshape(M, B, T), pl.cpu().numpy().reshape(B, T, M)
    live = y > 0
    np.testing.assert_allclose(pl, pred[..., :M], atol=2e-5, rtol=1e-5)                           # prior logits
    for m in range(M):
        mu_dev = model.head_means(hid, m).cpu().numpy().reshape(B, T, N)
        assert np.abs(mu_dev - mu[:, :, m]).max() < 1e-4                                          # every output of the head GEMM
        np.testing.assert_allclose(sq[m], (mu[:, :, m] ** 2).sum(-1), rtol=2e-4)                  # |mu_m|^2
        t_ref = np.take_along_axis(mu[:, :, m], y[..., None].astype(np.int64), -1)[..., 0]
        assert np.abs(tg[m] - t_ref)[live].max() < 1e-4 and not tg[m][~live].any()                # mu_m[target]
        if table_pass or channels[-1] > 256:
            assert np.array_equal(tg[m][live], mu_dev[live, y[live]])                             # ... bitwise the streamed value
        else:   # Gram form: a gathered fp32 dot product instead of the MFMA sequence -- summation order only
            assert np.abs(tg[m][live] - mu_dev[live, y[live]]).max() < 2e-6 * max(1.0, np.abs(mu_dev).max())
    for mode in ("gmm_prob", "gmm_prob_max"):
        a.loss = mode
        got = model.loss_rows(hid, dev(y), mode).cpu().numpy().reshape(B, T)
        want = G.calc_loss_gmm_ids(pred, y, a)
        assert np.abs(got - want)[live].max() < 1e-4 * max(1.0, np.abs(want[live]).max()), (mode, np.abs(got - want)[live].max())
        assert not got[~live].any()
        assert abs(G.masked_mean_loss(got.astype(np.float64), y) - G.masked_mean_loss(want, y)) < 1e-4
    model.close()


def test_reference_surface_model_tcn_gmm_and_calc_loss_gmm():
    from hiertcn_amd import model_tcn_gmm as MG
    a = _args(900, 4, (64, 64), 3, loss="gmm_prob_max")
    MG.reset_scopes()
    rng = np.random.default_rng(1)
    x = rng.integers(0, 900, (3, 12))
    y = rng.integers(0, 900, (3, 12))
    pred = MG.model_tcn_gmm(a, x)
    assert pred.shape == (3, 12, 4 * 901)
    loss = MG.calc_loss_gmm(pred, y, None, None, None, None, None)
    model = MG._SCOPES["tcn_gmm"]
    h = pred.hidden.float().cpu().numpy().astype(np.float64).reshape(3, 12, -1)
    pr = {k: (O.round_bf16(v) if (k.endswith("kernel") or k.endswith("weights")) else v) for k, v in model.params.items()}
    pr["tcn_gmm/fully_connected/weights"][:, :4] = model.params["tcn_gmm/fully_connected/weights"][:, :4]
    want = G.calc_loss_gmm_ids(G.gmm_head(h, pr, "tcn_gmm"), y, a)
    assert np.abs(loss.cpu().numpy() - want)[y > 0].max() < 1e-4
    with pytest.raises(NotImplementedError):
        MG.calc_loss_gmm(pred, y, loss="gmm_prob_hinge")
    MG.reset_scopes()


def _gru_args(N, M, H, nl, loss="gmm_prob"):
    return make_args([], model_type="gru_gmm", item_num=N, output_dim=N, num_mixture=M, sigma=0.1, loss=loss, hidden_dim=H, num_layer=nl,
                     adjust=False)


@pytest.mark.parametrize("N,M,H,nl,B,T", [(900, 3, 64, 1, 5, 12), (2500, 5, 128, 2, 7, 21), (4001, 2, 256, 2, 3, 15), (600, 2, 512, 3, 4, 8)])
def test_gru_gmm_trunk_and_losses_vs_oracle(N, M, H, nl, B, T):
    """model_gru_gmm (model_gru_gmm.py:11-54): the GRU stack on one-hot rows (layer 0's input half as a row gather in the GEMM epilogue,
    dynamic_rnn's count-as-prefix sequence_length) against the oracle's LITERAL one-hot graph; then the shared head and both losses."""
    from hiertcn_amd.model_gru_gmm import GruGmmModel
    a = _gru_args(N, M, H, nl)
    p = G.init_params_gru(a, seed=11)
    rng = np.random.default_rng(N + H)
    for k in p:
        if k.endswith("candidate/bias") or k.endswith("biases"):
            p[k] = (0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    for k in p:   # the glorot limit of an [item_num + H, .] kernel is tiny: scale the GRU kernels up so that the gates actually move
        if k.endswith("/kernel"):
            p[k] = (p[k] * 6.0).astype(np.float32)
    model = GruGmmModel(a, p)
    ids = rng.integers(1, N, (B, T + 1))
    lens = rng.integers(2, T + 1, B)
    ids = ids * (np.arange(T + 1)[None, :] <= lens[:, None])
    x, y = ids[:, :-1].astype(np.int32), ids[:, 1:].astype(np.int32)
    x[:, 0] = 0                        # a null id INSIDE the counted prefix: the count-as-prefix quirk drops the last real step
    x[0, 3] = 0
    dev = lambda v: torch.as_tensor(np.ascontiguousarray(v)).to("cuda:0")  # noqa: E731
    hid = model.forward(dev(x))
    # the GRU runs in fp32 on unrounded weights; only its output rows are rounded to the storage type
    h_ref = G.gru_gmm_hidden(a, x, p, "rnn_gmm", O.MIRROR)
    got = hid.float().cpu().numpy().reshape(B, T, -1)
    assert np.abs(got - h_ref).max() < 1e-2 * max(1.0, np.abs(h_ref).max())      # one bf16 ulp of the largest output
    exact = G.gru_gmm_hidden(a, x, p, "rnn_gmm", O.EXACT64)
    assert np.abs(got - exact).max() < 8e-3 * max(1.0, np.abs(exact).max())
    # rows past a sequence's counted length are exactly zero (dynamic_rnn), the ones before are not
    length = (x > 0).sum(1)
    for b in range(B):
        assert not got[b, length[b]:].any() and got[b, :length[b]].any()
    # the shared head on the device's hidden rows
    pr = {k: (O.round_bf16(v) if k.endswith("weights") else v) for k, v in p.items()}
    pr["rnn_gmm/fully_connected/weights"][:, :M] = p["rnn_gmm/fully_connected/weights"][:, :M]
    pred = G.gmm_head(got.astype(np.float64), pr, "rnn_gmm")
    live = y > 0
    for mode in ("gmm_prob", "gmm_prob_max"):
        a.loss = mode
        dl = model.loss_rows(hid, dev(y), mode).cpu().numpy().reshape(B, T)
        want = G.calc_loss_gmm_ids(pred, y, a)
        assert np.abs(dl - want)[live].max() < 1e-4 * max(1.0, np.abs(want[live]).max()), mode
        assert not dl[~live].any()
    model.close()


def test_reference_surface_model_gru_gmm():
    from hiertcn_amd import model_gru_gmm as MG
    from hiertcn_amd.model_tcn_gmm import calc_loss_gmm
    a = _gru_args(700, 3, 64, 2)
    rng = np.random.default_rng(3)
    ids = rng.integers(1, 700, (4, 11)).astype(np.int32)
    pred = MG.model_gru_gmm(a, ids[:, :-1])
    assert pred.shape == (4, 10, 3 * (1 + 700))
    loss = calc_loss_gmm(pred, ids[:, 1:])
    assert loss.shape == (4, 10) and bool(torch.isfinite(loss).all())
    with pytest.raises(NotImplementedError):
        a.has_gap = True
        MG.model_gru_gmm(a, ids[:, :-1], x_gap=np.zeros((4, 10, 1)))
    MG.reset_scopes()


@pytest.mark.parametrize("sigma", [0.1, None], ids=["fixed_sigma", "learned_scale"])
def test_run_gmm_evaluation_loop_on_the_synthetic_xing_file(tmp_path, sigma):
    """run_gmm_xing.evaluate_gmm (the evaluate_nn-shaped loop behind `run_xing --model_type tcn_gmm`): per batch the masked lane mean
    (model.py:111-117) of the mixture loss through hiertcn_gmm_lane_loss, epoch value = mean over batches; against the oracle on the
    device's own hidden rows, batch for batch."""
    import os
    from hiertcn_amd.data_loader import Dataloader_single_level_model_xing
    from hiertcn_amd.data_prepare import load_xing
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    from hiertcn_amd.run_gmm_xing import evaluate_gmm
    data = load_xing(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "interactions_synth.csv"))
    N = int(data[1][:, 1].max()) + 1
    a = _args(N, 3, (64, 64), 3)
    a.sigma = sigma                       # None: `--sigma none`, M (1 + 2 D) head outputs
    a.batch_size = 5
    a.log_path = str(tmp_path) + "/log/"
    os.makedirs(a.log_path + "test/")
    p = G.init_params(a, seed=11)
    model = TcnGmmModel(a, p)
    pr = {k: (O.round_bf16(v) if (k.endswith("kernel") or k.endswith("weights")) else v) for k, v in p.items()}
    pr["tcn_gmm/fully_connected/weights"][:, :3] = p["tcn_gmm/fully_connected/weights"][:, :3]
    want = []
    loader = Dataloader_single_level_model_xing(a, "test", data=data)
    while True:
        x, y, _ = loader.get_batch()
        x, y = np.asarray(x).astype(np.int32), np.asarray(y).astype(np.int32)
        hid = model.forward(torch.as_tensor(np.ascontiguousarray(x)).to("cuda:0"))
        h_dev = hid.float().cpu().numpy().astype(np.float64).reshape(x.shape[0], x.shape[1], -1)
        pred = G.gmm_head(h_dev, pr, "tcn_gmm")
        want.append(G.masked_mean_loss(G.calc_loss_gmm_ids(pred, y, a), y))
        if loader.done:
            break
    got, _, _ = evaluate_gmm(a, Dataloader_single_level_model_xing(a, "test", data=data), model, prefix="test")
    assert len(want) >= 2
    assert abs(got["loss_epoch"] - float(np.mean(want))) < 1e-4 * max(1.0, abs(float(np.mean(want))))
    txt = open(a.log_path + "test/results.txt").read()
    assert txt.startswith("loss_epoch ") and "nan" in txt          # a density model has no rank metrics: written as nan
    model.close()


@pytest.mark.parametrize("C", [64, 128, 256])
def test_gram_form_equals_the_table_pass(C, monkeypatch):
    """The two ways to |mu_m|^2 and mu_m[target] on the same handle contents and the same hidden rows, with LARGE biases (the cross term
    2 h.W b and |b|^2 then dominate): |mu|^2 to 4e-6 relative (fp32 summation order: both are exact products of 16-bit operands), target
    mean to 2e-6, losses to 1e-5 -- at R = 700 rows (not a multiple of the 256-row workgroup) with null targets."""
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    N, M = 4000, 4
    a = _args(N, M, (C,), 2)
    p = G.init_params(a, seed=11)
    rng = np.random.default_rng(C)
    p["tcn_gmm/fully_connected/biases"] = (0.3 * rng.standard_normal(p["tcn_gmm/fully_connected/biases"].shape)).astype(np.float32)
    R = 700
    torch.manual_seed(C)   # (unseeded rows put the difference at 1.5e-6 .. 2.2e-6 from run to run: a flaky bound)
    hid = (torch.randn(R, C, device="cuda:0") * 0.7).relu().to(torch.bfloat16)
    y = torch.as_tensor(rng.integers(0, N, R).astype(np.int32)).to("cuda:0")
    y[::7] = 0
    out = {}
    for name, env in (("gram", None), ("table", "1")):
        if env:
            monkeypatch.setenv("HIERTCN_GMM_TABLE_PASS", env)
        else:
            monkeypatch.delenv("HIERTCN_GMM_TABLE_PASS", raising=False)
        model = TcnGmmModel(a, p)
        loss, sq, tg, pl = model.loss_rows(hid, y, "gmm_prob", parts=True)
        lmax = model.loss_rows(hid, y, "gmm_prob_max")
        out[name] = [v.cpu().numpy().astype(np.float64) for v in (loss, sq, tg, pl, lmax)]
        model.close()
    g, t = out["gram"], out["table"]
    assert np.abs(g[1] - t[1]).max() < 4e-6 * np.abs(t[1]).max(), np.abs(g[1] - t[1]).max() / np.abs(t[1]).max()   # (measured 1.5e-6 .. 2.2e-6 over random rows)
    assert np.abs(g[2] - t[2]).max() < 2e-6 * max(1.0, np.abs(t[2]).max())
    assert np.array_equal(g[3], t[3]) or np.abs(g[3] - t[3]).max() < 1e-6
    for i in (0, 4):
        assert np.abs(g[i] - t[i]).max() < 1e-5 * max(1.0, np.abs(t[i]).max()), (i, np.abs(g[i] - t[i]).max())
    assert not g[0][::7].any() and not g[2][:, ::7].any()


@pytest.mark.parametrize("N,M,channels,K,B,T", [(1500, 4, (64, 64), 3, 5, 21), (20_778, 5, (256, 256), 5, 3, 30), (3001, 3, (64, 128), 3, 4, 18)])
def test_learned_scale_head_vs_oracle(N, M, channels, K, B, T):
    """args.sigma None (model_tcn_gmm.py:39-40: n_out = M (1 + 2 D), sigma = exp of the third block -- the build's decode): the paired pass over
    the interleaved (means | log-scales) table (score.hip PMODE 22: Q_m = sum_d mu_d^2 e^{-2 s_d}), the linear sum_d s_d, the gathered mu[y] /
    s[y] and both losses against oracle/gmm_oracle.py on the device's hidden rows; every mean and every log-scale through head_means."""
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    a = _args(N, M, channels, K)
    a.sigma = None
    p = G.init_params(a, seed=5)
    rng = np.random.default_rng(N)
    hw = p["tcn_gmm/fully_connected/weights"]
    hw[:, M + M * N:] *= 6.0                                          # log-scales of order 1: sigma between ~0.3 and ~3
    p["tcn_gmm/fully_connected/biases"] = (0.05 * rng.standard_normal(p["tcn_gmm/fully_connected/biases"].shape)).astype(np.float32)
    model = TcnGmmModel(a, p)
    ids = rng.integers(1, N, (B, T + 1))
    lens = rng.integers(2, T + 1, B)
    ids = ids * (np.arange(T + 1)[None, :] <= lens[:, None])
    x, y = ids[:, :-1].astype(np.int32), ids[:, 1:].astype(np.int32)
    x[:, 0] = 0
    dev = lambda v: torch.as_tensor(np.ascontiguousarray(v)).to("cuda:0")  # noqa: E731
    pr = {k: (O.round_bf16(v) if (k.endswith("kernel") or k.endswith("weights")) else v) for k, v in p.items()}
    pr["tcn_gmm/fully_connected/weights"][:, :M] = p["tcn_gmm/fully_connected/weights"][:, :M]
    hid = model.forward(dev(x))
    h_dev = hid.float().cpu().numpy().astype(np.float64).reshape(B, T, -1)
    pred = G.gmm_head(h_dev, pr, "tcn_gmm")                           # [B, T, M (1 + 2 D)]
    assert pred.shape[-1] == M * (1 + 2 * N) and model_shape(model, B, T) == pred.shape
    _, mu = G.decode(pred, M, N)
    sc = G.decode_scale(pred, M, N)
    loss, q, tg, pl = model.loss_rows(hid, dev(y), "gmm_prob", parts=True)
    q, tg = q.cpu().numpy().reshape(M, B, T), tg.cpu().numpy().reshape(M, B, T)
    live = y > 0
    for m in range(M):
        assert np.abs(model.head_means(hid, m).cpu().numpy().reshape(B, T, N) - mu[:, :, m]).max() < 1e-4
        assert np.abs(model.head_means(hid, M + m).cpu().numpy().reshape(B, T, N) - sc[:, :, m]).max() < 1e-4
        np.testing.assert_allclose(q[m], (mu[:, :, m] ** 2 * np.exp(-2 * sc[:, :, m])).sum(-1), rtol=3e-4)      # Q_m (v_exp_f32 per output)
        t_ref = np.take_along_axis(mu[:, :, m], y[..., None].astype(np.int64), -1)[..., 0]
        assert np.abs(tg[m] - t_ref)[live].max() < 1e-4 and not tg[m][~live].any()
    for mode in ("gmm_prob", "gmm_prob_max"):
        a.loss = mode
        got = model.loss_rows(hid, dev(y), mode).cpu().numpy().reshape(B, T)
        want = G.calc_loss_gmm_ids(pred, y, a)
        assert np.abs(got - want)[live].max() < 2e-4 * max(1.0, np.abs(want[live]).max()), (mode, np.abs(got - want)[live].max())
        assert not got[~live].any()
    model.close()


def model_shape(model, B, T):
    from hiertcn_amd.model_tcn_gmm import GmmHead
    return GmmHead(model, None, B, T).shape


def test_learned_scale_is_refused_at_width_512():
    from hiertcn_amd._lib import HierTCNError
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    a = _args(700, 2, (512,), 2)
    a.sigma = None
    with pytest.raises(HierTCNError, match="learned scale"):
        TcnGmmModel(a, None)


@pytest.mark.parametrize("sigma", [0.1, None])
def test_model_hier_with_the_mixture_model_as_its_low_level(sigma):
    """model_hier(args with model_low_type='tcn_gmm') (model_hier.py:64-65): [emb(x), carried state] rows straight into model_tcn_gmm's
    TemporalConvNet (block 0: Cin = emb + hidden_dim * num_layer, Dense residual) + the mixture head, the across-session GRU as in HierTCN.
    Hidden rows, carried state and the likelihood loss against oracle/gmm_oracle.model_hier_tcn_gmm (weights and input rows rounded as
    stored), over ragged sessions with user boundaries, fixed and learned scale."""
    from hiertcn_amd import model_hier_gmm as HG
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.model_hier import model_hier
    from hiertcn_amd.model_tcn_gmm import calc_loss_gmm
    from tests.gpu_common import as_lists, make_batch
    N, E, H, nl, M = 900, 64, 32, 2, 3
    a = hier_tcn_args(item_num=N, output_dim=N, emb_dim=E, hidden_dim=H, num_layer=nl, tcn_channel=[64, 128], kernel_size=3, num_mixture=M,
                      model_low_type="tcn_gmm", loss="gmm_prob", sigma=0.1)
    a.sigma = sigma
    p = G.init_params_hier(a, seed=4)
    rng = np.random.default_rng(8)
    for k in p:
        if (k.endswith("/bias") and "gates" not in k) or k.endswith("biases"):
            p[k] = (p[k] + 0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    HG.reset_scopes()
    HG._SCOPES["hier"] = HG.HierGmmModel(a, p, "hier")
    B, S, L = 5, 4, 7
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, H * nl))).astype(np.float32)
    xl, yl, ml = as_lists(x, y, mask)
    pred, st = model_hier(a, xl, yl, ml, st0)
    assert pred.shape == (B, S * L, M * (1 + (1 if sigma else 2) * N))
    # oracle on the numbers the device holds: table / conv / head kernels rounded to bf16 (the M prior columns stay fp32), x_slice rows rounded
    pr = {k: (O.round_bf16(v) if (k.endswith("kernel") and "multi_rnn_cell" not in k) or k.endswith("weights") else v) for k, v in p.items()}
    pr["hier/tcn_gmm/fully_connected/weights"][:, :M] = p["hier/tcn_gmm/fully_connected/weights"][:, :M]
    h_ref, st_ref = G.model_hier_tcn_gmm(a, xl, yl, ml, st0, pr, O.MIRROR, return_hidden=True, round_input=O.round_bf16)
    h_dev = pred.hidden.float().cpu().numpy().reshape(B, S * L, -1)
    assert np.abs(h_dev - h_ref).max() < 2e-2 * max(1.0, np.abs(h_ref).max())          # one bf16 ulp of the largest activation
    np.testing.assert_allclose(st.cpu().numpy(), st_ref, atol=2e-5, rtol=0)              # the across-session scan is fp32
    yy = y.reshape(B, S * L)
    for mode in ("gmm_prob", "gmm_prob_max"):
        a.loss = mode
        got = calc_loss_gmm(pred, yy, loss=mode).cpu().numpy()
        want = G.calc_loss_gmm_ids(G.gmm_head(h_dev.astype(np.float64), pr, "hier/tcn_gmm"), yy, a)
        live = yy > 0
        assert np.abs(got - want)[live].max() < 2e-4 * max(1.0, np.abs(want[live]).max()), (mode, np.abs(got - want)[live].max())
        assert not got[~live].any()
    HG.reset_scopes()


@pytest.mark.parametrize("sigma,nl", [(0.1, 2), (None, 1)])
def test_model_hier_with_the_gru_mixture_model_as_its_low_level(sigma, nl):
    """model_hier(args with model_low_type='gru_gmm') (model_hier.py:68-69 -> model_gru_gmm.py:33-54): the [emb(x), carried state] rows into a
    MultiRNNCell of GRUCells under dynamic_rnn (zero state, sequence_length = the count of rows that are not all-zero, used as a prefix length)
    + the mixture head (hiertcn_gmm trunk 3).  Hidden rows -- exact zeros past the counted length included --, carried state and the likelihood
    losses against oracle/gmm_oracle.model_hier_gru_gmm on the numbers the device holds."""
    from hiertcn_amd import model_hier_gmm as HG
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.model_hier import model_hier
    from hiertcn_amd.model_tcn_gmm import calc_loss_gmm
    from tests.gpu_common import as_lists, make_batch
    N, E, H, M = 900, 64, 64, 3
    a = hier_tcn_args(item_num=N, output_dim=N, emb_dim=E, hidden_dim=H, num_layer=nl, tcn_channel=[64], kernel_size=3, num_mixture=M,
                      model_low_type="gru_gmm", loss="gmm_prob", sigma=0.1)
    a.sigma = sigma
    p = G.init_params_hier_gru(a, seed=4)
    rng = np.random.default_rng(8)
    for k in p:
        if (k.endswith("/bias") and "gates" not in k) or k.endswith("biases"):
            p[k] = (p[k] + 0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    HG.reset_scopes()
    HG._SCOPES["hier"] = HG.HierGmmModel(a, p, "hier")
    B, S, L = 5, 4, 7
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, H * nl))).astype(np.float32)
    st0[:2] = 0.0        # a user's first session: zero state, so the leading null-id row of x_slice is all-zero and is not counted
    xl, yl, ml = as_lists(x, y, mask)
    pred, st = model_hier(a, xl, yl, ml, st0)
    assert pred.shape == (B, S * L, M * (1 + (1 if sigma else 2) * N))
    # oracle on the numbers the device holds: the item table and the head kernel rounded to bf16 (the M prior columns and every GRU tensor stay
    # fp32: the cells run on exact-fp32 products), x_slice rows rounded to the storage type
    pr = {k: (O.round_bf16(v) if (k == "hier/emb/kernel" or k.endswith("weights")) else v) for k, v in p.items()}
    pr["hier/rnn_gmm/fully_connected/weights"][:, :M] = p["hier/rnn_gmm/fully_connected/weights"][:, :M]
    h_ref, st_ref = G.model_hier_gru_gmm(a, xl, yl, ml, st0, pr, O.MIRROR, return_hidden=True, round_input=O.round_bf16)
    h_dev = pred.hidden.float().cpu().numpy().reshape(B, S * L, -1)
    assert np.array_equal(h_dev == 0, h_ref == 0)                                        # [TF-sem] zeros past the counted length
    assert np.abs(h_dev - h_ref).max() <= 2.0 ** -7                                      # |h| <= 1: a storage-type ulp
    np.testing.assert_allclose(st.cpu().numpy(), st_ref, atol=2e-5, rtol=0)
    yy = y.reshape(B, S * L)
    for mode in ("gmm_prob", "gmm_prob_max"):
        a.loss = mode
        got = calc_loss_gmm(pred, yy, loss=mode).cpu().numpy()
        want = G.calc_loss_gmm_ids(G.gmm_head(h_dev.astype(np.float64), pr, "hier/rnn_gmm"), yy, a)
        live = yy > 0
        assert np.abs(got - want)[live].max() < 2e-4 * max(1.0, np.abs(want[live]).max()), (mode, np.abs(got - want)[live].max())
        assert not got[~live].any()
    HG.reset_scopes()


@pytest.mark.parametrize("seed", range(6))
def test_mixture_loss_fuzz_shapes(seed, monkeypatch):
    """Random head shapes (last width 64 / 128 / 256, D from just above 3 C to a few thousand and not a multiple of 32, 1..16 mixtures, row counts
    from 1 to ~900 that are no multiple of anything): the Gram form against the table pass on the same rows (fixed sigma), and the learned-scale
    pass against the density rebuilt on the host from the materialised means / log-scales (hiertcn_gmm_head_means) -- both losses, null targets."""
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    rng = np.random.default_rng(1000 + seed)
    C = int(rng.choice([64, 128, 256]))
    D = int(rng.integers(3 * C + 1, 3 * C + 2500))
    M = int(rng.integers(1, 17))
    R = int(rng.integers(1, 900))
    a = _args(D, M, (C,), 2)
    p = G.init_params(a, seed=seed)
    p["tcn_gmm/fully_connected/biases"] = (0.2 * rng.standard_normal(p["tcn_gmm/fully_connected/biases"].shape)).astype(np.float32)
    hid = (torch.as_tensor(rng.standard_normal((R, C)).astype(np.float32)).to("cuda:0") * 0.8).relu().to(torch.bfloat16)
    yn = rng.integers(0, D, R).astype(np.int32)
    yn[rng.random(R) < 0.2] = 0
    y = torch.as_tensor(yn).to("cuda:0")
    out = {}
    for name, env in (("gram", None), ("table", "1")):
        if env:
            monkeypatch.setenv("HIERTCN_GMM_TABLE_PASS", env)
        else:
            monkeypatch.delenv("HIERTCN_GMM_TABLE_PASS", raising=False)
        model = TcnGmmModel(a, p)
        out[name] = [model.loss_rows(hid, y, mode).cpu().numpy().astype(np.float64) for mode in ("gmm_prob", "gmm_prob_max")]
        model.close()
    monkeypatch.delenv("HIERTCN_GMM_TABLE_PASS", raising=False)
    for g_, t_ in zip(out["gram"], out["table"]):
        assert np.abs(g_ - t_).max() < 2e-5 * max(1.0, np.abs(t_).max()), (C, D, M, R, np.abs(g_ - t_).max())
        assert not g_[yn == 0].any()
    # learned scale on the same rows
    a.sigma = None
    p2 = G.init_params(a, seed=seed + 50)
    p2["tcn_gmm/fully_connected/weights"][:, M + M * D:] *= 5.0
    p2["tcn_gmm/fully_connected/biases"] = (0.1 * rng.standard_normal(p2["tcn_gmm/fully_connected/biases"].shape)).astype(np.float32)
    model = TcnGmmModel(a, p2)
    mu = np.stack([model.head_means(hid, m).cpu().numpy().astype(np.float64) for m in range(M)], 1)            # [R, M, D]
    sc = np.stack([model.head_means(hid, M + m).cpu().numpy().astype(np.float64) for m in range(M)], 1)
    _, _, _, pl = model.loss_rows(hid, y, "gmm_prob", parts=True)
    pred = np.concatenate([pl.cpu().numpy().astype(np.float64), mu.reshape(R, -1), sc.reshape(R, -1)], -1)[None]   # [1, R, M (1 + 2 D)]
    live = yn > 0
    for mode in ("gmm_prob", "gmm_prob_max"):
        a.loss = mode
        got = model.loss_rows(hid, y, mode).cpu().numpy()
        want = G.calc_loss_gmm_ids(pred, yn[None], a)[0]
        if live.any():
            assert np.abs(got - want)[live].max() < 3e-4 * max(1.0, np.abs(want[live]).max()), (C, D, M, R, mode)
        assert not got[~live].any()
    model.close()
