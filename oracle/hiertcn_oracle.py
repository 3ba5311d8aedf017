"""CPU oracle for the HierTCN forward + scoring path  --  TEST INFRASTRUCTURE ONLY.

This file is a NumPy restatement of the reference's arithmetic for the hot path named by
BASELINE.json:north_star.  It is the checker for the HIP path; it is never the thing that is
shipped or measured as the product.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import it.

PARITY UNPINNED (model numerics): the reference is TensorFlow-1.6 graph code; TensorFlow is
not installed in the build container (and cannot be: no network), the reference ships no tests
and no golden vectors, so no output of the reference's model code exists to pin this
restatement against.  What pins it instead:
  * every function follows the cited reference lines op-for-op (file:line under /root/reference);
  * hand-derivable known answers (tests/test_oracle_known_answers.py);
  * an independent PyTorch-CPU formulation of the same ops (tests/test_oracle_vs_torch.py);
  * the loader restatement IS pinned: fixtures captured by importing the reference's own
    data_loader.py / data_prepare.py in the build container (tests/golden/make_loader_fixtures.py).

TensorFlow semantics relied upon (marked [TF-sem]) are TF-1.6's documented behaviour of
tf.layers.dense / tf.nn.convolution / rnn.GRUCell / tf.nn.top_k / softmax_cross_entropy_with_logits.

Variable names follow the TF-1.x scoping the reference graph would produce:
  hier/emb/kernel [N,E]  hier/emb/bias [E]  hier/tcn/emb/kernel [E+H*nl, C0]
  hier/tcn/temporal_conv_net/tblock_{i}/conv1/{kernel [K,Cin,Cout], bias [Cout]}
  hier/tcn/temporal_conv_net/tblock_{i}/dense/{kernel [Cin,Cout], bias [Cout]}   (only if Cin != Cout)
  hier/tcn/dense/{kernel [C,N], bias [N]}
  hier/multi_rnn_cell/cell_{l}/gru_cell/{gates,candidate}/{kernel,bias}
Single-level model_tcn uses the same names under the 'tcn/' root (tcn/emb/kernel [N,E], ...).
"""
from __future__ import annotations

import numpy as np

# --------------------------------------------------------------------------------------
# numerics helpers
# --------------------------------------------------------------------------------------


def round_bf16(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (the storage rounding of the HIP path)."""
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    u = a.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    out = (r & 0xFFFFFFFF).astype(np.uint32).view(np.float32).reshape(a.shape)
    return out


def round_f16(x):
    """Round-to-nearest-even fp32 -> IEEE half -> fp32 (the storage rounding of the HIERTCN_F16 build)."""
    return np.asarray(x, dtype=np.float32).astype(np.float16).astype(np.float32)


class Numerics:
    """Where the oracle rounds.  mode='exact': nothing is rounded, everything in `dtype`
    (the literal restatement).  mode='bf16_mirror': activations are rounded to bf16 at exactly
    the points where the HIP path stores bf16 (DESIGN.md 'numerics contract'): after the input
    projection, after every TemporalBlock; weights are expected to be pre-rounded by the caller
    (round_params_bf16)."""

    def __init__(self, mode="exact", dtype=np.float64):
        assert mode in ("exact", "bf16_mirror", "f16_mirror")
        self.mode = mode
        self.dtype = dtype

    def act(self, x):
        if self.mode == "bf16_mirror":
            return round_bf16(x).astype(self.dtype)
        if self.mode == "f16_mirror":
            return round_f16(x).astype(self.dtype)
        return x


EXACT64 = Numerics("exact", np.float64)
EXACT32 = Numerics("exact", np.float32)
MIRROR = Numerics("bf16_mirror", np.float64)
MIRROR_F16 = Numerics("f16_mirror", np.float64)

def is_bf16_stored(name: str) -> bool:
    """Tensors the HIP path stores in bf16: the item table, the input projection (rows [0:E) of
    hier/tcn/emb/kernel; rows [E:) are consumed in fp32 by the GRU-side state projection, see
    round_params_bf16), conv kernels, down-sample kernels and the scoring kernel.  Biases and
    all GRU tensors stay fp32."""
    if name.endswith("/bias"):
        return False
    if "gru_cell" in name:
        return False
    return True


def round_params_bf16(params: dict, E: int | None = None, rnd=None) -> dict:
    """Return a copy of `params` whose bf16-stored tensors hold bf16-representable values (or, with
    rnd=round_f16, half-representable ones: the HIERTCN_F16 build).
    The state half of */tcn/emb/kernel (rows E:) stays fp32: the HIP path consumes it in the
    fp32 GRU-side kernel (hiertcn_gru_scan)."""
    round_bf16 = rnd or globals()["round_bf16"]  # noqa: F811
    out = {}
    for k, v in params.items():
        v = np.asarray(v, dtype=np.float32)
        if is_bf16_stored(k):
            if (k.endswith("tcn/emb/kernel") or k.endswith("gru/emb/kernel")) and k.startswith("hier/") and E is not None:
                w = v.copy()
                w[:E] = round_bf16(v[:E])
                out[k] = w
            else:
                out[k] = round_bf16(v)
        else:
            out[k] = v.copy()
    return out


# --------------------------------------------------------------------------------------
# parameter construction  ([TF-sem] glorot_uniform default of tf.layers; GRU gate bias 1.0
# customed_gru_cell.py:312-314; conv/dense bias zeros customized_convolution_layer.py:95,150,
# customized_dense_layer.py:101)
# --------------------------------------------------------------------------------------


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def init_params(args, seed=2, model="hier", emb_dim=None):
    """Random-init parameters with the reference's shapes.  `emb_dim` defaults to 128, the
    width the reference hard-codes (model_hier.py:50,85; model_tcn.py:35); the build
    parametrises it (BASELINE.json 'd')."""
    rng = np.random.default_rng(seed)
    N = int(args.item_num)
    E = int(emb_dim if emb_dim is not None else getattr(args, "emb_dim", 128))
    H, nl, K = int(args.hidden_dim), int(args.num_layer), int(args.kernel_size)
    ch = [int(c) for c in args.tcn_channel]
    p = {}
    if model == "hier" and getattr(args, "model_low_type", "tcn") == "gru":
        # HierGRU (model_hier.py:68 -> model_gru.py:16-63): 'emb' projection, MultiRNNCell of GRUCells under
        # dynamic_rnn's scope 'rnn' [TF-sem], scoring dense on the top layer's outputs
        p["hier/emb/kernel"] = _glorot(rng, (N, E), N, E)
        p["hier/emb/bias"] = np.zeros(E, np.float32)
        p["hier/gru/emb/kernel"] = _glorot(rng, (E + H * nl, E), E + H * nl, E)
        for scope, first_in in (("hier/gru/rnn", E), ("hier", E)):
            inp = first_in
            for l in range(nl):
                pre = f"{scope}/multi_rnn_cell/cell_{l}/gru_cell"
                p[f"{pre}/gates/kernel"] = _glorot(rng, (inp + H, 2 * H), inp + H, 2 * H)
                p[f"{pre}/gates/bias"] = np.ones(2 * H, np.float32)
                p[f"{pre}/candidate/kernel"] = _glorot(rng, (inp + H, H), inp + H, H)
                p[f"{pre}/candidate/bias"] = np.zeros(H, np.float32)
                inp = H
        p["hier/gru/dense/kernel"] = _glorot(rng, (H, N), H, N)
        p["hier/gru/dense/bias"] = np.zeros(N, np.float32)
        return p
    if model == "gru":
        # single-level model_gru (model.py:66-67; model_gru.py:16-63 under scope 'gru'): the 'emb' dense on one-hot rows is the item table
        p["gru/emb/kernel"] = _glorot(rng, (N, E), N, E)
        inp = E
        for l in range(nl):
            pre = f"gru/rnn/multi_rnn_cell/cell_{l}/gru_cell"
            p[f"{pre}/gates/kernel"] = _glorot(rng, (inp + H, 2 * H), inp + H, 2 * H)
            p[f"{pre}/gates/bias"] = np.ones(2 * H, np.float32)
            p[f"{pre}/candidate/kernel"] = _glorot(rng, (inp + H, H), inp + H, H)
            p[f"{pre}/candidate/bias"] = np.zeros(H, np.float32)
            inp = H
        p["gru/dense/kernel"] = _glorot(rng, (H, N), H, N)
        p["gru/dense/bias"] = np.zeros(N, np.float32)
        return p
    if model == "hier":
        root = "hier/tcn"
        p["hier/emb/kernel"] = _glorot(rng, (N, E), N, E)
        p["hier/emb/bias"] = np.zeros(E, np.float32)
        p["hier/tcn/emb/kernel"] = _glorot(rng, (E + H * nl, E), E + H * nl, E)
        cin = E
    else:
        root = "tcn"
        p["tcn/emb/kernel"] = _glorot(rng, (N, E), N, E)
        cin = E
    for i, cout in enumerate(ch):
        pre = f"{root}/temporal_conv_net/tblock_{i}"
        p[f"{pre}/conv1/kernel"] = _glorot(rng, (K, cin, cout), K * cin, K * cout)
        p[f"{pre}/conv1/bias"] = np.zeros(cout, np.float32)
        if cin != cout:  # customized_tcn_cell.py:102-106
            p[f"{pre}/dense/kernel"] = _glorot(rng, (cin, cout), cin, cout)
            p[f"{pre}/dense/bias"] = np.zeros(cout, np.float32)
        cin = cout
    p[f"{root}/dense/kernel"] = _glorot(rng, (cin, N), cin, N)
    p[f"{root}/dense/bias"] = np.zeros(N, np.float32)
    if model == "hier":
        inp = E
        for l in range(nl):
            pre = f"hier/multi_rnn_cell/cell_{l}/gru_cell"
            p[f"{pre}/gates/kernel"] = _glorot(rng, (inp + H, 2 * H), inp + H, 2 * H)
            p[f"{pre}/gates/bias"] = np.ones(2 * H, np.float32)  # customed_gru_cell.py:314
            p[f"{pre}/candidate/kernel"] = _glorot(rng, (inp + H, H), inp + H, H)
            p[f"{pre}/candidate/bias"] = np.zeros(H, np.float32)
            inp = H
    return p


def count_params(params):
    return int(sum(int(np.prod(v.shape)) for v in params.values()))


# --------------------------------------------------------------------------------------
# layer primitives
# --------------------------------------------------------------------------------------


def embed_ids(ids, E_table, dtype=np.float64):
    """model.py:59-61 + model_hier.py:50 (Dense.call customized_dense_layer.py:155-167):
    one_hot(id, N) * sign(id) followed by a bias-free dense == E[id] for id>0, exactly 0 for
    id==0 (row 0 of the table is never read)."""
    ids = np.asarray(ids).astype(np.int64)
    out = np.asarray(E_table, dtype=dtype)[ids]
    out = out * (ids > 0)[..., None]
    return out


def dense(x, W, b=None):
    """customized_dense_layer.py:155-172: tensordot over the last axis (+ bias_add)."""
    y = np.tensordot(x, np.asarray(W, dtype=x.dtype), axes=([x.ndim - 1], [0]))
    if b is not None:
        y = y + np.asarray(b, dtype=x.dtype)
    return y


def causal_conv1d(x, W, b, dilation):
    """customized_tcn_cell.py:46-49 (left-pad (K-1)*dil zeros) + _Conv.call
    customized_convolution_layer.py:171-198 (VALID cross-correlation, channels_last, bias_add).
    x [B,L,Cin], W [K,Cin,Cout] -> [B,L,Cout]:
    o[t] = b + sum_j x[t-(K-1-j)*dil] . W[j],  x[tau<0] = 0."""
    B, L, Cin = x.shape
    K = W.shape[0]
    pad = (K - 1) * dilation
    xp = np.concatenate([np.zeros((B, pad, Cin), x.dtype), x], axis=1)
    W = np.asarray(W, dtype=x.dtype)
    out = np.zeros((B, L, W.shape[2]), x.dtype)
    for j in range(K):
        out += np.tensordot(xp[:, j * dilation : j * dilation + L, :], W[j], axes=([2], [0]))
    return out + np.asarray(b, dtype=x.dtype)


def relu(x):
    return np.maximum(x, 0)


def temporal_block(x, params, prefix, dilation, num: Numerics):
    """customized_tcn_cell.py:109-127: relu(dropout(relu(conv1(x))) + res(x)); conv2 is
    constructed but never called (:120-122); dropout rate 0.0 (args.py:64) is the identity;
    res = identity or a biased Dense when Cin != Cout (:102-106,123-124)."""
    c = relu(causal_conv1d(x, params[f"{prefix}/conv1/kernel"], params[f"{prefix}/conv1/bias"], dilation))
    if f"{prefix}/dense/kernel" in params:
        r = dense(x, params[f"{prefix}/dense/kernel"], params[f"{prefix}/dense/bias"])
    else:
        r = x
    return num.act(relu(c + r))


def temporal_conv_net(x, params, root, n_blocks, num: Numerics):
    """customized_tcn_cell.py:147-161: block i has dilation 2**i (:150)."""
    out = x
    for i in range(n_blocks):
        out = temporal_block(out, params, f"{root}/temporal_conv_net/tblock_{i}", 2**i, num)
    return out


def tcn_hidden(args, x, params, root, num: Numerics):
    """model_tcn.py:35-36: bias-free 'emb' projection then the TemporalConvNet.  Returns the
    last hidden activations [B,L,C_last] (what the scoring GEMM consumes)."""
    u = num.act(dense(x, params[f"{root}/emb/kernel"]))
    return temporal_conv_net(u, params, root, len(args.tcn_channel), num)


def model_tcn(args, x, params, name="tcn", num: Numerics = EXACT64, return_hidden=False):
    """model_tcn.py:13-44.  `x` is either float [B,T,D] (the low level of model_hier) or integer
    ids [B,T] (single-level model: model.py:56,71-73 feeds one_hot(x_id)*sign(x_id), so the
    first dense is the embedding lookup with row 0 -> 0)."""
    x = np.asarray(x)
    if np.issubdtype(x.dtype, np.integer):
        u = num.act(embed_ids(x, params[f"{name}/emb/kernel"], num.dtype))
        h = temporal_conv_net(u, params, name, len(args.tcn_channel), num)
    else:
        h = tcn_hidden(args, x.astype(num.dtype), params, name, num)
    if return_hidden:
        return h
    return dense(h, params[f"{name}/dense/kernel"], params[f"{name}/dense/bias"])  # model_tcn.py:41


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def gru_cell(inp, h, Wg, bg, Wc, bc):
    """customed_gru_cell.py:309-337: value = sigmoid([inp,h].Wg + bg); r,u = split(value)
    (r FIRST, u second, :324); c = tanh([inp, r*h].Wc + bc); h' = u*h + (1-u)*c (:336)."""
    dt = inp.dtype
    H = h.shape[1]
    v = sigmoid(np.concatenate([inp, h], 1) @ np.asarray(Wg, dt) + np.asarray(bg, dt))
    r, u = v[:, :H], v[:, H:]
    c = np.tanh(np.concatenate([inp, r * h], 1) @ np.asarray(Wc, dt) + np.asarray(bc, dt))
    return u * h + (1 - u) * c


def multi_rnn_cell(inp, state, params, H, nl, scope="hier"):
    """customed_gru_cell.py:1050-1073 with state_is_tuple=False: layer l reads
    state[:, l*H:(l+1)*H]; the new state is the concat of the per-layer h'."""
    cur = inp
    new = []
    for l in range(nl):
        pre = f"{scope}/multi_rnn_cell/cell_{l}/gru_cell"
        h = state[:, l * H : (l + 1) * H]
        cur = gru_cell(cur, h, params[f"{pre}/gates/kernel"], params[f"{pre}/gates/bias"],
                       params[f"{pre}/candidate/kernel"], params[f"{pre}/candidate/bias"])
        new.append(cur)
    return cur, np.concatenate(new, 1)


def seq_length(x):
    """model_gru.py:109-118: the number of time steps whose feature vector is not all-zero -- a COUNT, used by
    dynamic_rnn as a prefix length (:51).  Inside model_hier it is taken on the projected [e, state] rows, so a padded
    step counts whenever the carried state is non-zero, and with a zero state (a user's first session) the leading
    null-id step does not: such a session of n events gets length n-1 and its last real step outputs zeros."""
    used = np.sign(np.abs(x).max(2))
    return used.sum(1).astype(np.int64)


def model_gru_low(args, xs, params, root="hier/gru", num: Numerics = EXACT64, return_hidden=False):
    """model_gru.py:16-63 as the low level of model_hier (model_hier.py:68): xs [B, L, E + H*nl] float.
    'emb' projection without bias (:46); MultiRNNCell of GRUCells run by dynamic_rnn from a zero state with
    sequence_length = length(projected x) (:47-53) -- [TF-sem] past a row's length the output is zero and the state is
    carried through unchanged; scoring dense on the outputs (:57)."""
    dt = num.dtype
    H, nl = int(args.hidden_dim), int(args.num_layer)
    xs = np.asarray(xs)
    if np.issubdtype(xs.dtype, np.integer):
        # single-level model_gru (model.py:56,66-67): x = one_hot(id) * sign(id), so the 'emb' dense (:46) is the embedding lookup with row 0 -> 0
        u = num.act(embed_ids(xs, params[f"{root}/emb/kernel"], dt))
    else:
        u = num.act(dense(xs.astype(dt), params[f"{root}/emb/kernel"]))
    length = seq_length(u)
    B, L, _ = u.shape
    state = np.zeros((B, H * nl), dt)
    outs = np.zeros((B, L, H), dt)
    for t in range(L):
        active = (t < length)[:, None]
        cur, new_state = multi_rnn_cell(u[:, t], state, params, H, nl, scope=f"{root}/rnn")
        state = np.where(active, new_state, state)
        outs[:, t] = np.where(active, cur, 0.0)
    outs = num.act(outs)
    if return_hidden:
        return outs
    return dense(outs, params[f"{root}/dense/kernel"], params[f"{root}/dense/bias"])


def model_gru(args, x_ids, params, name="gru", num: Numerics = EXACT64, return_hidden=False):
    """The single-level model_gru as model.py:66-67 builds it on XING (model_gru.py:16-63, scope 'gru'): integer ids [B, T] -> logits
    [B, T, N].  dynamic_rnn starts from the zero state with sequence_length = length(embedded x) (:47-53, :109-118): the COUNT of steps whose
    id is not null, used as a prefix length -- a sequence that opens with the loader's null step (data_loader.py: x is y shifted right) is cut
    one step short and its last real step outputs zeros (logits = the dense bias)."""
    return model_gru_low(args, np.asarray(x_ids).astype(np.int64), params, name, num, return_hidden)


def session_mean_pool(y_ids, params, dtype=np.float64):
    """model_hier.py:83-85: (sum_t one_hot(y)) / count, then dense(...,name='emb') WITH the
    default use_bias=True (asymmetric with :50) == mean of E[y] over valid steps + hier/emb/bias.
    The loader never emits an empty session (data_loader.py:184-200); an empty one would be
    0/0 = NaN in the reference and is reproduced as such."""
    y_ids = np.asarray(y_ids).astype(np.int64)
    e = embed_ids(y_ids, params["hier/emb/kernel"], dtype)
    cnt = (y_ids > 0).sum(1, keepdims=True).astype(dtype)
    with np.errstate(invalid="ignore", divide="ignore"):
        pooled = e.sum(1) / cnt
    return pooled + np.asarray(params["hier/emb/bias"], dtype)


def model_hier(args, x, y, mask, state, params, num: Numerics = EXACT64, return_hidden=False):
    """model_hier.py:21-94 with model_low_type 'tcn' (HierTCN) or 'gru' (HierGRU) and the default-off branches
    (gap, noise, impression, batchnorm) excluded.
    x, y: lists of S integer id arrays [B, L_i] (the one-hot of model.py:59-61 is undone: ids in,
    row-0-is-zero semantics kept); mask: list of [B,1]; state [B, H*nl].
    Returns (pred_all [B, sum L_i, N], state) -- or the hidden activations instead of pred_all
    when return_hidden (same concat order)."""
    dt = num.dtype
    H, nl = int(args.hidden_dim), int(args.num_layer)
    state = np.asarray(state, dt)
    preds = []
    for i in range(len(x)):
        xi = np.asarray(x[i]).astype(np.int64)
        e = embed_ids(xi, params["hier/emb/kernel"], dt)  # :50
        feat = np.repeat(state[:, None, :], xi.shape[1], axis=1)  # :54
        xs = np.concatenate([e, feat], axis=-1)  # :55
        if getattr(args, "model_low_type", "tcn") == "gru":  # :67-68 (the reference's default low level: HierGRU)
            preds.append(model_gru_low(args, xs, params, "hier/gru", num, return_hidden))
        else:
            h = tcn_hidden(args, xs, params, "hier/tcn", num)  # :63 -> model_tcn.py:35-36
            if return_hidden:
                preds.append(h)
            else:
                preds.append(dense(h, params["hier/tcn/dense/kernel"], params["hier/tcn/dense/bias"]))
        y_slice = session_mean_pool(y[i], params, dt)  # :83-85
        _, state = multi_rnn_cell(y_slice, state, params, H, nl)  # :91
        state = state * np.asarray(mask[i], dt)  # :93
    return np.concatenate(preds, axis=1), state  # :76-79,94


# --------------------------------------------------------------------------------------
# loss / metrics
# --------------------------------------------------------------------------------------


def log_sum_exp(z):
    m = z.max(-1, keepdims=True)
    return (m + np.log(np.exp(z - m).sum(-1, keepdims=True)))[..., 0]


def masked_loss_and_metrics(pred, y_id, item_num=None):
    """model.py:62,102-117,131-132 + loss.py:20-21 (cross_entropy) + calc_metric_fast
    loss.py:163-221.  pred [B,T,N] raw logits, y_id [B,T] int.
    Returns dict(loss, recall1, recall5, recall10, mrr, rank_frac, ranks_float, ranks, ce)."""
    pred = np.asarray(pred)
    dt = pred.dtype
    y_id = np.asarray(y_id).astype(np.int64)
    N = pred.shape[-1] if item_num is None else item_num
    mask_y = (y_id > 0).astype(dt)  # model.py:62
    pred = pred * mask_y[..., None]  # model.py:105 -> masked rows are all-zero logits
    # loss.py:21 [TF-sem]: labels = one_hot(y)*sign(y): one-hot for valid rows, all-zero for
    # masked rows -> CE = LSE - z[y] for valid rows, exactly 0 for masked rows.
    tgt = np.take_along_axis(pred, y_id[..., None], axis=-1)[..., 0] * mask_y  # sum(score*y) loss.py:179
    ce = (log_sum_exp(pred) - tgt) * mask_y
    loss_bt = ce * mask_y  # model.py:111
    activity_count = mask_y.sum(1)  # :112
    user_count = np.sign(activity_count).sum()  # :113
    activity_count = activity_count + dt.type(1e-6)  # :114
    loss = (loss_bt.sum(1) / activity_count).sum() / user_count  # :116-117
    ranks = (pred > tgt[..., None]).sum(-1).astype(dt)  # loss.py:179 strict >
    ranks_float = ranks / N  # :189-190
    rr = 1.0 / (1.0 + ranks)  # :191
    r1 = (ranks <= 0).astype(dt)  # :194-196
    r5 = (ranks <= 4).astype(dt)
    r10 = (ranks <= 9).astype(dt)
    ranks = ranks * mask_y  # :199-205
    ranks_float = ranks_float * mask_y
    rr, r1, r5, r10 = rr * mask_y, r1 * mask_y, r5 * mask_y, r10 * mask_y

    def user_mean(v):  # :208-219
        return (v.sum(1) / activity_count).sum() / user_count

    return dict(loss=loss, recall1=user_mean(r1), recall5=user_mean(r5), recall10=user_mean(r10),
                mrr=user_mean(rr), rank_frac=user_mean(ranks_float), ranks_float=ranks_float,
                ranks=ranks, ce=ce, target=tgt)


def top_k_indices(score, k):
    """[TF-sem] tf.nn.top_k as used at loss.py:120: values descending, ties -> lower index
    first.  A stable argsort of -score gives exactly that order."""
    order = np.argsort(-score, axis=-1, kind="stable")
    return order[..., :k]


def calc_metric_topk(pred, y_id):
    """calc_metric loss.py:108-160 (the slow variant): full top_k(k=N), rank = position of y_id
    in the index list (:124).  Returns (ranks_indices [B,T,N], ranks [B,T]) after the
    model.py:105 masking."""
    y_id = np.asarray(y_id).astype(np.int64)
    mask_y = (y_id > 0).astype(pred.dtype)
    pred = pred * mask_y[..., None]
    idx = top_k_indices(pred, pred.shape[-1])
    ranks = np.argmax(idx == y_id[..., None], axis=-1).astype(pred.dtype)
    return idx, ranks


# --------------------------------------------------------------------------------------
# the evaluate_hier aggregation (run_hier_xing.py:112-188): mean over batches of per-batch
# (mean over lanes of per-lane means)
# --------------------------------------------------------------------------------------


def evaluate_hier(args, batches, state, params, num: Numerics = EXACT64):
    """batches: iterable of (x_list, y_list, mask_list).  Carries `state` across batches
    (run_hier_xing.py:132,147).  Returns (epoch metrics dict, per-batch list, final state)."""
    acc = dict(loss=0.0, rank_frac=0.0, mrr=0.0, recall1=0.0, recall5=0.0, recall10=0.0)
    per_batch = []
    n = 0
    for x_list, y_list, mask_list in batches:
        pred, state = model_hier(args, x_list, y_list, mask_list, state, params, num)
        y_id = np.concatenate([np.asarray(v).astype(np.int64) for v in y_list], axis=1)  # :116
        m = masked_loss_and_metrics(pred, y_id, int(args.item_num))
        per_batch.append(m)
        for k in acc:
            acc[k] += float(m[k])  # :152-161
        n += 1
    return {k: v / max(n, 1) for k, v in acc.items()}, per_batch, state  # :181-188
