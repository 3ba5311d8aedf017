"""Parameter store of the hot path: TF-1.x variable names and shapes of the reference graph
(SURVEY.md 8(a)), and the initialisers the reference relies on.

  glorot-uniform for every kernel  ([TF-sem] default of tf.layers / add_variable(initializer=None),
                                    customized_dense_layer.py:133-135, customized_convolution_layer.py:137-141;
                                    for conv kernels fan = K*C)
  zeros for every bias             (customized_dense_layer.py:101, customized_convolution_layer.py:95,150)
  1.0 for the GRU gate bias        (customed_gru_cell.py:312-314), 0 for the candidate bias (:333)
"""
from __future__ import annotations

from typing import Dict

import numpy as np


def _glorot(rng, shape, fan_in, fan_out):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape).astype(np.float32)


def param_shapes(args, model: str = "hier") -> Dict[str, tuple]:
    N = int(args.item_num)
    E = int(getattr(args, "emb_dim", 128))
    H, nl, K = int(args.hidden_dim), int(args.num_layer), int(args.kernel_size)
    ch = [int(c) for c in args.tcn_channel]
    s: Dict[str, tuple] = {}
    if model == "hier" and getattr(args, "model_low_type", "tcn") == "gru":
        # HierGRU (the reference's default low level, args.py:40; model_hier.py:67-68 -> model_gru.py:16-63)
        s["hier/emb/kernel"] = (N, E)
        s["hier/emb/bias"] = (E,)
        s["hier/gru/emb/kernel"] = (E + H * nl, E)
        for scope in ("hier/gru/rnn", "hier"):
            inp = E
            for l in range(nl):
                pre = f"{scope}/multi_rnn_cell/cell_{l}/gru_cell"
                s[f"{pre}/gates/kernel"] = (inp + H, 2 * H)
                s[f"{pre}/gates/bias"] = (2 * H,)
                s[f"{pre}/candidate/kernel"] = (inp + H, H)
                s[f"{pre}/candidate/bias"] = (H,)
                inp = H
        s["hier/gru/dense/kernel"] = (H, N)
        s["hier/gru/dense/bias"] = (N,)
        return s
    if model == "gru":
        # single-level model_gru (model.py:66-67 -> model_gru.py:16-63 under scope 'gru'): 'emb' dense on the one-hot input = the item table,
        # the GRU stack under dynamic_rnn's scope 'rnn' [TF-sem], the scoring dense on the top layer's outputs
        s["gru/emb/kernel"] = (N, E)
        inp = E
        for l in range(nl):
            pre = f"gru/rnn/multi_rnn_cell/cell_{l}/gru_cell"
            s[f"{pre}/gates/kernel"] = (inp + H, 2 * H)
            s[f"{pre}/gates/bias"] = (2 * H,)
            s[f"{pre}/candidate/kernel"] = (inp + H, H)
            s[f"{pre}/candidate/bias"] = (H,)
            inp = H
        s["gru/dense/kernel"] = (H, N)
        s["gru/dense/bias"] = (N,)
        return s
    if model == "hier":
        root = "hier/tcn"
        s["hier/emb/kernel"] = (N, E)
        s["hier/emb/bias"] = (E,)  # created by the y-path dense only (model_hier.py:85 vs :50)
        s["hier/tcn/emb/kernel"] = (E + H * nl, E)
    else:
        root = "tcn"
        s["tcn/emb/kernel"] = (N, E)
    cin = E
    for i, cout in enumerate(ch):
        pre = f"{root}/temporal_conv_net/tblock_{i}"
        s[f"{pre}/conv1/kernel"] = (K, cin, cout)
        s[f"{pre}/conv1/bias"] = (cout,)
        if cin != cout:
            s[f"{pre}/dense/kernel"] = (cin, cout)
            s[f"{pre}/dense/bias"] = (cout,)
        cin = cout
    s[f"{root}/dense/kernel"] = (cin, N)
    s[f"{root}/dense/bias"] = (N,)
    if model == "hier":
        inp = E
        for l in range(nl):
            pre = f"hier/multi_rnn_cell/cell_{l}/gru_cell"
            s[f"{pre}/gates/kernel"] = (inp + H, 2 * H)
            s[f"{pre}/gates/bias"] = (2 * H,)
            s[f"{pre}/candidate/kernel"] = (inp + H, H)
            s[f"{pre}/candidate/bias"] = (H,)
            inp = H
    return s


def init_params(args, seed: int = 2, model: str = "hier") -> Dict[str, np.ndarray]:
    """numpy.random.default_rng(seed), tensors drawn in param_shapes() order."""
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(args, model).items():
        if name.endswith("/bias"):
            out[name] = (np.ones if name.endswith("gates/bias") else np.zeros)(shape, np.float32)
        elif len(shape) == 3:  # conv [K, Cin, Cout]
            out[name] = _glorot(rng, shape, shape[0] * shape[1], shape[0] * shape[2])
        else:
            out[name] = _glorot(rng, shape, shape[0], shape[1])
    return out


def count_params(params) -> int:
    return int(sum(int(np.prod(v.shape)) for v in params.values()))


def pad_widths(args, params: Dict[str, np.ndarray], model: str = "hier", multiple: int = 32):
    """Zero-pad the channel axes (emb_dim, tcn_channel) up to a multiple of one MFMA tile, e.g. the
    d=100 single-level configuration of BASELINE.json -> 128.  Padded channels are exactly zero
    everywhere (zero embedding columns, zero kernel rows/columns, zero biases; relu(0) = 0 and a zero
    row of the scoring kernel adds nothing), so every logit is unchanged.  Returns (args', params')
    with the same TF names.  hidden_dim (GRU) is not padded: its gate arithmetic is not zero-preserving
    (sigmoid(0) = 0.5), choose a multiple of 32."""
    import copy

    if model == "gru":
        return _pad_gru(args, params)

    def up(c):
        return -(-int(c) // multiple) * multiple

    a = copy.copy(args)
    E0, E1 = int(getattr(args, "emb_dim", 128)), up(getattr(args, "emb_dim", 128))
    ch0 = [int(c) for c in args.tcn_channel]
    ch1 = [up(c) for c in ch0]
    a.emb_dim, a.tcn_channel = E1, ch1
    want = param_shapes(a, model)
    have = param_shapes(args, model)
    if set(want) != set(have):   # padding must not change which blocks have a down-sample residual
        raise ValueError(f"padding {ch0} -> {ch1} with emb {E0} -> {E1} changes the variable set: {sorted(set(want) ^ set(have))}")
    out: Dict[str, np.ndarray] = {}
    for name, shape in want.items():
        src = np.asarray(params[name], dtype=np.float32)
        if name.endswith("tcn/emb/kernel") and model == "hier":   # [E + H*nl, E]: item rows then state rows
            dst = np.zeros(shape, np.float32)
            dst[:E0, :E0] = src[:E0]
            dst[E1:, :E0] = src[E0:]
        else:
            dst = np.zeros(shape, np.float32)
            dst[tuple(slice(0, n) for n in src.shape)] = src
        out[name] = dst
    return a, out


GRU_WIDTHS = (64, 128, 256, 512)   # what the scoring kernels, which read the stack's output rows, are built for


def _gru_index(shape0, in0, H0, in1, H1, kernel_cols):
    """index arrays that place a [in0 + H0, kernel_cols * H0] GRU kernel inside its [in1 + H1, kernel_cols * H1] padded form: input rows first,
    then the state rows; the columns in blocks of H (gates: r | u, customed_gru_cell.py:324)."""
    rows = np.concatenate([np.arange(in0), in1 + np.arange(H0)])
    cols = np.concatenate([b * H1 + np.arange(H0) for b in range(kernel_cols)])
    return rows, cols


def _pad_gru(args, params):
    """Single-level model_gru (model.py:66-67; args.py:306-308 sets hidden_dim = 200 for it): zero-pad the GRU width up to the next width the
    kernels are built for.  A padded unit stays exactly zero: its state starts at zero (dynamic_rnn's zero state, model_gru.py:47-53), its
    candidate is tanh(0) = 0, so h' = u * 0 + (1 - u) * 0 whatever its gates are; its rows of the next layer's kernels and of the scoring
    kernel are zero, so no logit changes and no gradient reaches a padded entry (zero gradients leave Adam's update at exactly zero)."""
    import copy
    a = copy.copy(args)
    H0, nl, E = int(args.hidden_dim), int(args.num_layer), int(getattr(args, "emb_dim", 128))
    H1 = next((w for w in GRU_WIDTHS if w >= H0), None)
    if H1 is None:
        raise ValueError(f"hidden_dim {H0}: the GRU stack is built for widths up to {GRU_WIDTHS[-1]}")
    a.hidden_dim = H1
    # the 'emb' width too, up to the stack's width (model_gru.py:46 hard-codes 128 against hidden_dim 200 -> 256): with E == H the session-tile /
    # layer-pipelined kernels of the HierGRU low level apply (lgru_tile.hip, lgru_wide.hip: one launch for all T steps instead of 2 T nl), and the
    # padded columns are exactly zero -- zero table columns against zero kernel rows: no logit changes, no gradient reaches them
    E1 = H1 if E < H1 else E
    a.emb_dim = E1
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(a, "gru").items():
        src = np.asarray(params[name], dtype=np.float32)
        dst = np.zeros(shape, np.float32)
        l = int(name.split("/cell_")[1].split("/")[0]) if "/cell_" in name else 0
        in0 = E if l == 0 else H0
        in1 = E1 if l == 0 else H1
        if name.endswith("gates/kernel") or name.endswith("candidate/kernel"):
            r, c = _gru_index(src.shape[0], in0, H0, in1, H1, 2 if "gates" in name else 1)
            dst[np.ix_(r, c)] = src
        elif name.endswith("gates/bias"):
            dst[np.concatenate([np.arange(H0), H1 + np.arange(H0)])] = src
        else:   # item table (columns E .. E1 zero), candidate bias, scoring kernel [H, N] / bias: leading corner
            dst[tuple(slice(0, n) for n in src.shape)] = src
        out[name] = dst
    return a, out


def _unpad_gru(args, padded):
    H0, E = int(args.hidden_dim), int(getattr(args, "emb_dim", 128))
    E1 = int(np.asarray(padded["gru/emb/kernel"]).shape[1])   # the padded 'emb' width (_pad_gru)
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(args, "gru").items():
        src = np.asarray(padded[name])
        H1 = src.shape[-1] // 2 if name.endswith("gates/kernel") or name.endswith("gates/bias") else (src.shape[-1] if "gru_cell" in name else src.shape[0])
        l = int(name.split("/cell_")[1].split("/")[0]) if "/cell_" in name else 0
        in0 = E if l == 0 else H0
        in1 = E1 if l == 0 else H1
        if name.endswith("gates/kernel") or name.endswith("candidate/kernel"):
            r, c = _gru_index(shape[0], in0, H0, in1, H1, 2 if "gates" in name else 1)
            out[name] = np.ascontiguousarray(src[np.ix_(r, c)])
        elif name.endswith("gates/bias"):
            out[name] = np.ascontiguousarray(src[np.concatenate([np.arange(H0), H1 + np.arange(H0)])])
        else:
            out[name] = np.ascontiguousarray(src[tuple(slice(0, n) for n in shape)])
    return out


def unpad_widths(args, padded: Dict[str, np.ndarray], model: str = "hier") -> Dict[str, np.ndarray]:
    """Inverse of pad_widths: cut the tensors of a width-padded model back to the shapes of `args` (what a checkpoint holds).  The cut
    entries are exactly zero for a model that was created by pad_widths and trained on the device: a zero-padded channel has zero
    activations, so every gradient into its weights and every gradient flowing back through it is exactly zero, and Adam on a
    zero gradient with zero moments is a zero update."""
    if model == "gru":
        return _unpad_gru(args, padded)
    E0 = int(getattr(args, "emb_dim", 128))
    out: Dict[str, np.ndarray] = {}
    for name, shape in param_shapes(args, model).items():
        src = np.asarray(padded[name])
        if name.endswith("tcn/emb/kernel") and model == "hier":
            E1 = src.shape[1]
            out[name] = np.concatenate([src[:E0, :E0], src[E1:, :E0]], 0)
        else:
            out[name] = src[tuple(slice(0, n) for n in shape)]
    return out
