"""CPU tests pinning what can be pinned of the mixture-head oracle (oracle/gmm_oracle.py; the decode itself is the build's own
choice -- module docstring there): the one-hot trunk shortcut against the literal dense one-hot graph, the Gaussian log-density
against scipy, the id form of the loss against the one-hot form, and hand-derived known answers."""
import numpy as np
from scipy.stats import multivariate_normal

from hiertcn_amd.args import make_args
from oracle import gmm_oracle as G
from oracle import hiertcn_oracle as O


def _args(N=37, M=3, loss="gmm_prob", channels=(16, 16)):
    return make_args([], model_type="tcn_gmm", item_num=N, output_dim=N, num_mixture=M, sigma=0.1, loss=loss,
                     tcn_channel=list(channels), kernel_size=3, adjust=False)


def test_head_width_and_parameter_shapes():
    a = _args()
    p = G.init_params(a, seed=1)
    assert G.n_out(a) == 3 * (1 + 37)                                                   # model_tcn_gmm.py:42
    assert p["tcn_gmm/fully_connected/weights"].shape == (16, 3 * 38)
    assert p["tcn_gmm/temporal_conv_net/tblock_0/conv1/kernel"].shape == (3, 37, 16)     # Cin = item_num: no 'emb' (model_tcn_gmm.py:36)
    assert "tcn_gmm/temporal_conv_net/tblock_0/dense/kernel" in p                        # Cin != Cout (customized_tcn_cell.py:102-106)
    assert "tcn_gmm/temporal_conv_net/tblock_1/dense/kernel" not in p


def test_onehot_trunk_shortcut_equals_the_literal_one_hot_graph():
    a = _args()
    p = G.init_params(a, seed=2)
    for k in p:
        if k.endswith("bias") or k.endswith("biases"):
            p[k] = (0.1 * np.random.default_rng(3).standard_normal(p[k].shape)).astype(np.float32)
    rng = np.random.default_rng(4)
    ids = rng.integers(0, 37, (5, 9))
    ids[:, -2:] = 0                                                                      # right padding, null ids
    onehot = np.eye(37)[ids] * (ids > 0)[..., None]                                      # model.py:56-61
    lit = O.temporal_conv_net(onehot, p, "tcn_gmm", 2, O.EXACT64)
    np.testing.assert_allclose(G.tcn_gmm_hidden(a, ids, p), lit, atol=1e-12)
    pred = G.model_tcn_gmm(a, ids, p)
    np.testing.assert_allclose(pred, O.dense(lit, p["tcn_gmm/fully_connected/weights"], p["tcn_gmm/fully_connected/biases"]), atol=1e-12)


def test_mvn_log_prob_against_scipy():
    rng = np.random.default_rng(0)
    y, mu = rng.standard_normal(6), rng.standard_normal((2, 6))
    got = G.mvn_diag_log_prob(y, mu, 0.3)
    want = [multivariate_normal(mean=m, cov=0.09 * np.eye(6)).logpdf(y) for m in mu]
    np.testing.assert_allclose(got, want, rtol=1e-12)


def test_id_form_equals_one_hot_form_for_both_losses():
    rng = np.random.default_rng(5)
    for loss in ("gmm_prob", "gmm_prob_max"):
        a = _args(loss=loss)
        pred = 0.3 * rng.standard_normal((4, 7, G.n_out(a)))
        ids = rng.integers(0, 37, (4, 7))
        onehot = np.eye(37)[ids] * (ids > 0)[..., None]
        np.testing.assert_allclose(G.calc_loss_gmm_ids(pred, ids, a), G.calc_loss_gmm(pred, onehot, a), rtol=1e-10, atol=1e-12)


def test_known_answers():
    a = _args(N=10, M=2)
    D, s = 10, 0.1
    pred = np.zeros((1, 1, G.n_out(a)))                       # pi uniform, every mu = 0
    ids = np.array([[4]])
    # log_prob = log(2 * 1/2 * N(e_4; 0, s)) = -1/(2 s^2) - D log s - D/2 log 2pi ; loss = -log_prob / D
    want = -(-0.5 / s ** 2 - D * np.log(s) - 0.5 * D * np.log(2 * np.pi)) / D
    assert abs(G.calc_loss_gmm_ids(pred, ids, a)[0, 0] - want) < 1e-12
    a.loss = "gmm_prob_max"                                   # -log( 1/2 * exp(lp / D) ) = log 2 - lp / D
    assert abs(G.calc_loss_gmm_ids(pred, ids, a)[0, 0] - (np.log(2) + want)) < 1e-12
    # a mixture whose first mean IS the target one-hot: d2 = 0 for m = 0
    pred2 = pred.copy()
    pred2[0, 0, 2 + 4] = 1.0
    lp0 = -D * np.log(s) - 0.5 * D * np.log(2 * np.pi)
    lp1 = -0.5 / s ** 2 + lp0
    a.loss = "gmm_prob"
    want2 = -np.log(0.5 * np.exp(lp0) + 0.5 * np.exp(lp1)) / D
    assert abs(G.calc_loss_gmm_ids(pred2, ids, a)[0, 0] - want2) < 1e-10
    # masked mean (model.py:111-117)
    assert abs(G.masked_mean_loss(np.array([[2.0, 4.0], [9.0, 9.0]]), np.array([[3, 0], [0, 0]])) - 2.0 / (1 + 1e-6)) < 1e-9


def test_gru_gmm_oracle_shapes_and_dynamic_rnn_prefix():
    """model_gru_gmm.py:33-54 restated: parameter shapes, the count-as-prefix sequence_length, zero rows past it, a known answer with
    zero kernels (gate bias 1: h' = sigmoid(1) h + (1 - sigmoid(1)) tanh(0) = 0 from a zero state -> all-zero outputs, head = biases)."""
    a = make_args([], model_type="gru_gmm", item_num=23, output_dim=23, num_mixture=2, sigma=0.1, hidden_dim=8, num_layer=2, adjust=False)
    p = G.init_params_gru(a, seed=1)
    assert p["rnn_gmm/rnn/multi_rnn_cell/cell_0/gru_cell/gates/kernel"].shape == (23 + 8, 16)
    assert p["rnn_gmm/rnn/multi_rnn_cell/cell_1/gru_cell/candidate/kernel"].shape == (16, 8)
    assert p["rnn_gmm/fully_connected/weights"].shape == (8, 2 * 24) and (p["rnn_gmm/rnn/multi_rnn_cell/cell_0/gru_cell/gates/bias"] == 1).all()
    ids = np.array([[0, 5, 0, 7, 9, 0], [3, 4, 5, 6, 7, 8]])
    for k in p:
        if k.endswith("/kernel"):
            p[k] = p[k] * 8.0
    h = G.gru_gmm_hidden(a, ids, p)
    assert h.shape == (2, 6, 8)
    assert not h[0, 3:].any() and h[0, 1].any()          # three non-null ids -> steps 0..2 run, step 3 (a real id!) and later are zero
    assert h[1].any(-1).all()
    # step 0 of row 0 reads the null id from a zero state: x = 0, h = 0 -> c = tanh(bc) = 0, h' = 0
    assert not h[0, 0].any()
    z = {k: (np.zeros_like(v) if k.endswith("/kernel") else v) for k, v in p.items()}
    pred, hz = G.model_gru_gmm(a, ids, z, return_hidden=True)
    assert not hz.any() and np.allclose(pred, z["rnn_gmm/fully_connected/biases"])


def test_learned_scale_head_width_density_and_id_form():
    """args.sigma None (model_tcn_gmm.py:39-40): n_out = M (1 + 2 D); the per-dimension density against scipy; the id form the HIP
    epilogue computes (sum_d mu_d^2 e^{-2 s_d} + (1 - 2 mu[y]) e^{-2 s[y]}, sum_d s_d) against the one-hot form; a hand-derived value."""
    a = _args()
    a.sigma = None
    assert G.learned_sigma(a) and G.n_out(a) == 3 * (1 + 2 * 37)
    assert G.init_params(a, seed=1)["tcn_gmm/fully_connected/weights"].shape == (16, 3 * 75)
    rng = np.random.default_rng(0)
    y, mu, sc = rng.standard_normal(6), rng.standard_normal((2, 6)), 0.4 * rng.standard_normal((2, 6))
    got = G.mvn_diag_log_prob(y, mu, np.exp(sc))
    want = [multivariate_normal(mean=m, cov=np.diag(np.exp(2 * s))).logpdf(y) for m, s in zip(mu, sc)]
    np.testing.assert_allclose(got, want, rtol=1e-12)
    for loss in ("gmm_prob", "gmm_prob_max"):
        a.loss = loss
        pred = 0.3 * rng.standard_normal((4, 7, G.n_out(a)))
        ids = rng.integers(0, 37, (4, 7))
        onehot = np.eye(37)[ids] * (ids > 0)[..., None]
        np.testing.assert_allclose(G.calc_loss_gmm_ids(pred, ids, a), G.calc_loss_gmm(pred, onehot, a), rtol=1e-10, atol=1e-12)
    # every output zero: pi uniform, mu = 0, sigma = e^0 = 1 -> log_prob = -1/2 - D/2 log 2 pi
    b = _args(N=10, M=2)
    b.sigma = None
    assert abs(G.calc_loss_gmm_ids(np.zeros((1, 1, G.n_out(b))), np.array([[4]]), b)[0, 0] - (0.5 + 5 * np.log(2 * np.pi)) / 10) < 1e-12


def test_hier_with_mixture_low_level_shares_the_high_level_with_model_hier():
    """G.model_hier_tcn_gmm (model_hier.py:64-65): block 0 reads [emb, state] rows of width emb + H * nl with a Dense residual, the head has
    M (1 + D) outputs per step, and the carried state is exactly O.model_hier's on the same table / GRU variables (the high level never sees
    the low level's output, model_hier.py:83-93)."""
    from hiertcn_amd.args import hier_tcn_args
    a = hier_tcn_args(item_num=50, output_dim=50, emb_dim=16, hidden_dim=8, num_layer=2, tcn_channel=[16, 16], kernel_size=3, num_mixture=2,
                      model_low_type="tcn_gmm", loss="gmm_prob", sigma=0.1)
    p = G.init_params_hier(a, seed=3)
    assert p["hier/tcn_gmm/temporal_conv_net/tblock_0/conv1/kernel"].shape == (3, 16 + 16, 16)
    assert p["hier/tcn_gmm/temporal_conv_net/tblock_0/dense/kernel"].shape == (32, 16)
    rng = np.random.default_rng(0)
    x = [rng.integers(0, 50, (3, 5)) for _ in range(4)]
    y = [rng.integers(0, 50, (3, 5)) for _ in range(4)]
    for v in y:
        v[:, 0] = np.maximum(v[:, 0], 1)          # a session is never empty (data_loader.py:184-200)
    mask = [np.array([[1.0], [0.0], [1.0]]) for _ in range(4)]
    st0 = rng.standard_normal((3, 16))
    pred, st = G.model_hier_tcn_gmm(a, x, y, mask, st0, p)
    assert pred.shape == (3, 20, 2 * 51)
    b = hier_tcn_args(item_num=50, emb_dim=16, hidden_dim=8, num_layer=2, tcn_channel=[16, 16], kernel_size=3)
    q = O.init_params(b, seed=9, emb_dim=16)
    q.update({k: v for k, v in p.items() if k.startswith("hier/emb/") or k.startswith("hier/multi_rnn_cell/")})
    _, st_h = O.model_hier(b, x, y, mask, st0, q)
    np.testing.assert_allclose(st, st_h, atol=1e-12)
