"""CPU oracle for the mixture-output heads (BASELINE.json configs[4])  --  TEST INFRASTRUCTURE ONLY.

Restates, in NumPy fp64:
  * the trunk + head of model_tcn_gmm (model_tcn_gmm.py:27-44): NO 'emb' projection -- the TemporalConvNet runs on the model's
    input directly (on XING that input is one_hot(id) * sign(id), model.py:56-61, so block 0 has Cin = item_num and a Dense
    down-sample residual, customized_tcn_cell.py:102-106) -- then tf.contrib.layers.fully_connected(n_out, activation_fn=None)
    with n_out = M * (1 + D), D = args.output_dim, when args.sigma is set (model_tcn_gmm.py:39-44; model_gru_gmm.py:49-54 has
    the same head on a GRU trunk: gru_gmm_hidden / model_gru_gmm at the end of this file);
  * the two GMM likelihood losses of calc_loss_gmm that need no negative samples (loss.py:247-257):
        gmm_prob      loss = -log(exp(gmm.log_prob(y) / D))
        gmm_prob_max  loss = -log(max_m  pi_m * exp(mvn_m.log_prob(y) / D))

PARITY UNPINNED, twice over:
  (1) as for the whole model path, TensorFlow cannot run here and the reference ships no vectors;
  (2) the DECODE of the head's output into (pi, mu, sigma, gmm) exists nowhere in the reference: calc_loss_gmm has no caller,
      args.pred_mode is read only at loss.py:279, and model.py:74-77,108 would feed the raw head output into the cross entropy
      with a shape mismatch.  The decode used here is the build's own, documented choice:
          pi    = softmax(pred[..., :M])                                  (class prior of each Gaussian, loss.py:236)
          mu    = pred[..., M : M + M*D] reshaped [M, D], mixture-major   (loss.py:237)
          sigma = args.sigma (a scalar, the reference default 0.1, args.py:70-71), or with a learned scale (args.sigma None) sigma_d = exp(s_d), s = the
                  third block of M*D outputs, mixture-major (decode_scale below)
          gmm   = Mixture(Categorical(pi), [MultivariateNormalDiag(mu_m, sigma * 1)]) [TF-sem: tf.contrib.distributions]:
                  log_prob(y) = logsumexp_m( log pi_m + mvn_m.log_prob(y) ),
                  mvn.log_prob(y) = -1/2 sum_d ((y_d - mu_d) / sigma)^2 - D log(sigma) - D/2 log(2 pi).
  What CAN be pinned is pinned by tests: the head GEMM (shapes, bias, ordering), the formulas above against
  scipy.stats.multivariate_normal, and the one-hot shortcut  sum_d (y_d - mu_d)^2 = |mu|^2 - 2 mu[target] + 1.

Variable names [TF-sem]: <scope>/temporal_conv_net/tblock_{i}/{conv1,dense}/{kernel,bias},
<scope>/fully_connected/{weights [C_last, n_out], biases [n_out]} (tf.contrib.layers.fully_connected: xavier weights, zero
biases), scope = 'tcn_gmm' (model_tcn_gmm.py:12).
"""
from __future__ import annotations

import numpy as np

from . import hiertcn_oracle as O


def learned_sigma(args) -> bool:
    """model_tcn_gmm.py:39 / model_gru_gmm.py:50 `if not args.sigma`: None (or 0) -> the head also emits a scale per mean."""
    return not getattr(args, "sigma", 0.1)


def n_out(args) -> int:
    """model_tcn_gmm.py:39-42: M (1 + D + D) with a learned scale, M (1 + D) with args.sigma fixed."""
    return int(args.num_mixture) * (1 + (2 if learned_sigma(args) else 1) * int(args.output_dim))


def init_params(args, seed=2, scope="tcn_gmm", input_dim=None):
    """Shapes of model_tcn_gmm's variables; input_dim = width of the TCN's input (item_num for the single-level XING model)."""
    rng = np.random.default_rng(seed)
    K = int(args.kernel_size)
    cin = int(input_dim if input_dim is not None else args.item_num)
    p = {}
    for i, cout in enumerate(int(c) for c in args.tcn_channel):
        pre = f"{scope}/temporal_conv_net/tblock_{i}"
        p[f"{pre}/conv1/kernel"] = O._glorot(rng, (K, cin, cout), K * cin, K * cout)
        p[f"{pre}/conv1/bias"] = np.zeros(cout, np.float32)
        if cin != cout:
            p[f"{pre}/dense/kernel"] = O._glorot(rng, (cin, cout), cin, cout)
            p[f"{pre}/dense/bias"] = np.zeros(cout, np.float32)
        cin = cout
    no = n_out(args)
    p[f"{scope}/fully_connected/weights"] = O._glorot(rng, (cin, no), cin, no)
    p[f"{scope}/fully_connected/biases"] = np.zeros(no, np.float32)
    return p


def onehot_block0(ids, params, scope, num: O.Numerics):
    """TemporalBlock 0 of model_tcn_gmm on x = one_hot(id) * sign(id) (model.py:56-61), without building the one-hot: a one-hot
    row times a kernel is a row gather, so conv1 (customized_tcn_cell.py:46-49, dilation 1) is
    c[t] = b + sum_j W[j][id[t - (K-1-j)]] over the taps that fall inside the sequence and carry a non-null id, and the Dense
    residual (:102-106,123-124) is Wr[id[t]] + br (just br for the null id).  ids [B, T] -> [B, T, C0]."""
    ids = np.asarray(ids).astype(np.int64)
    pre = f"{scope}/temporal_conv_net/tblock_0"
    W = np.asarray(params[f"{pre}/conv1/kernel"], np.float64)      # [K, N, C0]
    b = np.asarray(params[f"{pre}/conv1/bias"], np.float64)
    Wr = np.asarray(params[f"{pre}/dense/kernel"], np.float64)     # [N, C0]
    br = np.asarray(params[f"{pre}/dense/bias"], np.float64)
    B, T = ids.shape
    K = W.shape[0]
    c = np.zeros((B, T, W.shape[2])) + b
    for j in range(K):
        s = K - 1 - j
        src = ids[:, : T - s] if s > 0 else ids
        c[:, s:] += W[j][src] * (src > 0)[..., None]
    r = Wr[ids] * (ids > 0)[..., None] + br
    return num.act(O.relu(O.relu(c) + r))


def tcn_gmm_hidden(args, ids, params, scope="tcn_gmm", num: O.Numerics = O.EXACT64):
    """model_tcn_gmm.py:27-36 on id input: block 0 by gathers, blocks 1.. as in customized_tcn_cell.py:147-161."""
    out = onehot_block0(ids, params, scope, num)
    for i in range(1, len(args.tcn_channel)):
        out = O.temporal_block(out, params, f"{scope}/temporal_conv_net/tblock_{i}", 2 ** i, num)
    return out


def gmm_head(hidden, params, scope="tcn_gmm"):
    """model_tcn_gmm.py:44 / model_gru_gmm.py:54: fully_connected(pred, n_out, activation_fn=None)."""
    return O.dense(hidden, params[f"{scope}/fully_connected/weights"], params[f"{scope}/fully_connected/biases"])


def model_tcn_gmm(args, ids, params, scope="tcn_gmm", num: O.Numerics = O.EXACT64, return_hidden=False):
    h = tcn_gmm_hidden(args, ids, params, scope, num)
    pred = gmm_head(h, params, scope)
    return (pred, h) if return_hidden else pred


def decode(pred, M: int, D: int):
    """The build's decode (module docstring): -> log_pi [..., M], mu [..., M, D]."""
    a = pred[..., :M]
    a = a - a.max(-1, keepdims=True)
    log_pi = a - np.log(np.exp(a).sum(-1, keepdims=True))
    mu = pred[..., M:M + M * D].reshape(pred.shape[:-1] + (M, D))
    return log_pi, mu


def decode_scale(pred, M: int, D: int):
    """Learned scale (args.sigma None): the THIRD block of the head output (docstring of model_tcn_gmm.py:16-18: prior, mean, variance parts),
    mixture-major like the means, as a LOG scale: sigma = exp(s) -> s [..., M, D].  The build's own choice like the rest of the decode
    (the reference never splits the head output): an unconstrained output needs some positive map, exp is the usual one."""
    return pred[..., M + M * D:M + 2 * M * D].reshape(pred.shape[:-1] + (M, D))


def mvn_diag_log_prob(y, mu, sigma):
    """tf.contrib.distributions.MultivariateNormalDiag(loc=mu, scale_diag=sigma).log_prob(y) [TF-sem]; y [..., D], mu [..., M, D];
    sigma: a float (args.sigma) or an array [..., M, D] (loss.py:250-251: scale_diag=sigma[:, :, i, :])."""
    D = y.shape[-1]
    if np.ndim(sigma) == 0:
        d2 = ((y[..., None, :] - mu) ** 2).sum(-1)
        return -0.5 * d2 / sigma ** 2 - D * np.log(sigma) - 0.5 * D * np.log(2 * np.pi)
    return -0.5 * (((y[..., None, :] - mu) / sigma) ** 2).sum(-1) - np.log(sigma).sum(-1) - 0.5 * D * np.log(2 * np.pi)


def calc_loss_gmm(pred, y, args):
    """loss.py:247-257 for args.loss in {'gmm_prob', 'gmm_prob_max'}.  pred [B, T, n_out] (raw head output), y [B, T, D]
    (one_hot(y_id) * sign(y_id) on XING) -> loss [B, T] (NOT yet masked: model.py:111 multiplies by mask_y afterwards)."""
    M, D = int(args.num_mixture), int(args.output_dim)
    pred = np.asarray(pred, np.float64)
    log_pi, mu = decode(pred, M, D)
    sigma = np.exp(decode_scale(pred, M, D)) if learned_sigma(args) else float(args.sigma)
    lp_m = mvn_diag_log_prob(np.asarray(y, np.float64), mu, sigma)            # [B, T, M]
    if args.loss == "gmm_prob":
        z = log_pi + lp_m
        mx = z.max(-1, keepdims=True)
        log_prob = (mx + np.log(np.exp(z - mx).sum(-1, keepdims=True)))[..., 0]
        return -np.log(np.exp(log_prob / D))
    if args.loss == "gmm_prob_max":
        return -np.log((np.exp(log_pi) * np.exp(lp_m / D)).max(-1))
    raise NotImplementedError(f"args.loss={args.loss!r}: the negative-sample GMM losses (loss.py:258-290) need impressions, which XING has none of")


def calc_loss_gmm_ids(pred, y_ids, args):
    """The same through the one-hot shortcut: sum_d (y_d - mu_d)^2 = |mu|^2 - 2 mu[target] + 1 for a target id > 0 and |mu|^2 for the
    null id (an all-zero y row).  This is the form the HIP epilogue computes; tests assert it equals calc_loss_gmm on real one-hots."""
    M, D = int(args.num_mixture), int(args.output_dim)
    pred = np.asarray(pred, np.float64)
    y_ids = np.asarray(y_ids).astype(np.int64)
    log_pi, mu = decode(pred, M, D)
    t = np.take_along_axis(mu, y_ids[..., None, None], -1)[..., 0]                        # mu_m[target]
    live = (y_ids > 0)[..., None]
    if learned_sigma(args):
        # sum_d ((y_d - mu_d) / sigma_d)^2 = sum_d mu_d^2 e^{-2 s_d} + (1 - 2 mu[target]) e^{-2 s[target]};  sum_d log sigma_d = sum_d s_d
        sc = decode_scale(pred, M, D)
        q = (mu ** 2 * np.exp(-2.0 * sc)).sum(-1)
        s_t = np.take_along_axis(sc, y_ids[..., None, None], -1)[..., 0]
        lp_m = -0.5 * (q + np.where(live, (1.0 - 2.0 * t) * np.exp(-2.0 * s_t), 0.0)) - sc.sum(-1) - 0.5 * D * np.log(2 * np.pi)
    else:
        sigma = float(args.sigma)
        s2 = (mu ** 2).sum(-1)                                                           # [B, T, M]
        d2 = s2 + np.where(live, 1.0 - 2.0 * t, 0.0)
        lp_m = -0.5 * d2 / sigma ** 2 - D * np.log(sigma) - 0.5 * D * np.log(2 * np.pi)
    if args.loss == "gmm_prob":
        z = log_pi + lp_m
        mx = z.max(-1, keepdims=True)
        return -((mx + np.log(np.exp(z - mx).sum(-1, keepdims=True)))[..., 0]) / D
    if args.loss == "gmm_prob_max":
        return -(log_pi + lp_m / D).max(-1)
    raise NotImplementedError(args.loss)


def init_params_hier(args, seed=2, scope="hier"):
    """Variables of model_hier with model_low_type 'tcn_gmm' (model_hier.py:28-93 + model_tcn_gmm.py:27-44 under scope 'hier')."""
    base = O.init_params(args, seed=seed, model="hier", emb_dim=int(getattr(args, "emb_dim", 128)))
    p = {k: v for k, v in base.items() if k.startswith(f"{scope}/emb/") or k.startswith(f"{scope}/multi_rnn_cell/")}
    cin = int(getattr(args, "emb_dim", 128)) + int(args.hidden_dim) * int(args.num_layer)
    p.update(init_params(args, seed=seed + 1, scope=f"{scope}/tcn_gmm", input_dim=cin))
    return p


def model_hier_tcn_gmm(args, x, y, mask, state, params, num: O.Numerics = O.EXACT64, return_hidden=False, round_input=None):
    """model_hier.py:21-94 with model_low_type 'tcn_gmm' (:64-65): per slot x_slice = [emb(x), state tiled] (:50,54-55) straight into
    model_tcn_gmm's TemporalConvNet (no 'emb' of its own) and its head; high level as in O.model_hier (:83-93).
    round_input: optional rounding of the x_slice rows (the device stores them in 16 bits).  -> (pred_all [B, sum L_i, n_out] or hidden, state)"""
    dt = num.dtype
    H, nl = int(args.hidden_dim), int(args.num_layer)
    state = np.asarray(state, dt)
    outs = []
    nb = len(args.tcn_channel)
    for i in range(len(x)):
        xi = np.asarray(x[i]).astype(np.int64)
        e = O.embed_ids(xi, params["hier/emb/kernel"], dt)
        feat = np.repeat(state[:, None, :], xi.shape[1], axis=1)
        xs = np.concatenate([e, feat], axis=-1)
        if round_input is not None:
            xs = round_input(xs)
        h = O.temporal_conv_net(xs, params, "hier/tcn_gmm", nb, num)
        outs.append(h if return_hidden else gmm_head(h, params, "hier/tcn_gmm"))
        y_slice = O.session_mean_pool(y[i], params, dt)
        _, state = O.multi_rnn_cell(y_slice, state, params, H, nl)
        state = state * np.asarray(mask[i], dt)
    return np.concatenate(outs, axis=1), state


def init_params_hier_gru(args, seed=2, scope="hier"):
    """Variables of model_hier with model_low_type 'gru_gmm' (model_hier.py:28-93 + model_gru_gmm.py:33-54 under scope 'hier/rnn_gmm')."""
    base = O.init_params(args, seed=seed, model="hier", emb_dim=int(getattr(args, "emb_dim", 128)))
    p = {k: v for k, v in base.items() if k.startswith(f"{scope}/emb/") or k.startswith(f"{scope}/multi_rnn_cell/")}
    cin = int(getattr(args, "emb_dim", 128)) + int(args.hidden_dim) * int(args.num_layer)
    rng = np.random.default_rng(seed + 1)
    H = int(args.hidden_dim)
    for l in range(int(args.num_layer)):
        pre = f"{scope}/rnn_gmm/rnn/multi_rnn_cell/cell_{l}/gru_cell"
        inp = cin if l == 0 else H
        p[f"{pre}/gates/kernel"] = O._glorot(rng, (inp + H, 2 * H), inp + H, 2 * H)
        p[f"{pre}/gates/bias"] = np.ones(2 * H, np.float32)
        p[f"{pre}/candidate/kernel"] = O._glorot(rng, (inp + H, H), inp + H, H)
        p[f"{pre}/candidate/bias"] = np.zeros(H, np.float32)
    no = n_out(args)
    p[f"{scope}/rnn_gmm/fully_connected/weights"] = O._glorot(rng, (H, no), H, no)
    p[f"{scope}/rnn_gmm/fully_connected/biases"] = np.zeros(no, np.float32)
    return p


def model_hier_gru_gmm(args, x, y, mask, state, params, num: O.Numerics = O.EXACT64, return_hidden=False, round_input=None):
    """model_hier.py:21-94 with model_low_type 'gru_gmm' (:68-69): per slot x_slice = [emb(x), state tiled] (:50,54-55) into model_gru_gmm's
    MultiRNNCell of GRUCells under dynamic_rnn from a zero state, sequence_length = the count of non-zero rows of x_slice used as a prefix length
    (model_gru_gmm.py:33-48,57-67; past it the state is carried and the output is zero [TF-sem]), then its head; high level as in O.model_hier.
    -> (pred_all [B, sum L_i, n_out] or hidden, state)"""
    dt = num.dtype
    H, nl = int(args.hidden_dim), int(args.num_layer)
    state = np.asarray(state, dt)
    outs = []
    for i in range(len(x)):
        xi = np.asarray(x[i]).astype(np.int64)
        e = O.embed_ids(xi, params["hier/emb/kernel"], dt)
        feat = np.repeat(state[:, None, :], xi.shape[1], axis=1)
        xs = np.concatenate([e, feat], axis=-1)
        if round_input is not None:
            xs = round_input(xs)
        B, T = xi.shape
        length = O.seq_length(xs)
        ls = np.zeros((B, H * nl), dt)
        h = np.zeros((B, T, H), dt)
        for t in range(T):
            active = (t < length)[:, None]
            cur, new_state = O.multi_rnn_cell(xs[:, t], ls, params, H, nl, scope="hier/rnn_gmm/rnn")
            ls = np.where(active, new_state, ls)
            h[:, t] = np.where(active, cur, 0.0)
        h = num.act(h)
        outs.append(h if return_hidden else gmm_head(h, params, "hier/rnn_gmm"))
        y_slice = O.session_mean_pool(y[i], params, dt)
        _, state = O.multi_rnn_cell(y_slice, state, params, H, nl)
        state = state * np.asarray(mask[i], dt)
    return np.concatenate(outs, axis=1), state


def masked_mean_loss(loss_bt, y_ids):
    """model.py:111-117 applied to a [B, T] loss."""
    mask = (np.asarray(y_ids) > 0).astype(np.float64)
    act = mask.sum(1)
    users = np.sign(act).sum()
    return float((((loss_bt * mask).sum(1)) / (act + 1e-6)).sum() / users)


# ---- model_gru_gmm (model_gru_gmm.py:11-54): the same head on a GRU trunk that reads the one-hot rows directly (no 'emb' projection) ----

def init_params_gru(args, seed=2, scope="rnn_gmm"):
    """Variables of model_gru_gmm [TF-sem names]: <scope>/rnn/multi_rnn_cell/cell_l/gru_cell/{gates,candidate}/{kernel,bias} with
    in_0 = item_num (the one-hot width) and in_l = H above; gate bias init 1.0 (customed_gru_cell.py:314); then the fully_connected head."""
    rng = np.random.default_rng(seed)
    N, H = int(args.item_num), int(args.hidden_dim)
    p = {}
    for l in range(int(args.num_layer)):
        pre = f"{scope}/rnn/multi_rnn_cell/cell_{l}/gru_cell"
        inp = N if l == 0 else H
        p[f"{pre}/gates/kernel"] = O._glorot(rng, (inp + H, 2 * H), inp + H, 2 * H)
        p[f"{pre}/gates/bias"] = np.ones(2 * H, np.float32)
        p[f"{pre}/candidate/kernel"] = O._glorot(rng, (inp + H, H), inp + H, H)
        p[f"{pre}/candidate/bias"] = np.zeros(H, np.float32)
    no = n_out(args)
    p[f"{scope}/fully_connected/weights"] = O._glorot(rng, (H, no), H, no)
    p[f"{scope}/fully_connected/biases"] = np.zeros(no, np.float32)
    return p


def gru_gmm_hidden(args, ids, params, scope="rnn_gmm", num: O.Numerics = O.EXACT64):
    """model_gru_gmm.py:33-48 on x = one_hot(id) * sign(id) (model.py:56-61), LITERALLY: the one-hot rows are built and multiplied.
    MultiRNNCell of GRUCells under dynamic_rnn from a zero state with sequence_length = length(x) (:57-67) -- the count of non-zero
    rows, used as a prefix length; past it the state is carried through and the output is zero [TF-sem].  ids [B, T] -> [B, T, H]."""
    ids = np.asarray(ids).astype(np.int64)
    N, H, nl = int(args.item_num), int(args.hidden_dim), int(args.num_layer)
    B, T = ids.shape
    x = np.zeros((B, T, N))
    bb, tt = np.nonzero(ids > 0)
    x[bb, tt, ids[bb, tt]] = 1.0
    length = O.seq_length(x)
    state = np.zeros((B, H * nl))
    outs = np.zeros((B, T, H))
    for t in range(T):
        active = (t < length)[:, None]
        cur, new_state = O.multi_rnn_cell(x[:, t], state, params, H, nl, scope=f"{scope}/rnn")
        state = np.where(active, new_state, state)
        outs[:, t] = np.where(active, cur, 0.0)
    return num.act(outs)


def model_gru_gmm(args, ids, params, scope="rnn_gmm", num: O.Numerics = O.EXACT64, return_hidden=False):
    h = gru_gmm_hidden(args, ids, params, scope, num)
    pred = gmm_head(h, params, scope)
    return (pred, h) if return_hidden else pred
