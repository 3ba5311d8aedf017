"""model_hier with the mixture model as its LOW level (model_hier.py:64-65: model_low_type 'tcn_gmm'), backed by the HIP kernels.

The reference's slot loop (model_hier.py:39-93) with pred = model_tcn_gmm(args, x_slice, ...): x_slice = [dense 'emb' of the one-hot step,
the carried state tiled over the steps] (model_hier.py:50,54-55) goes STRAIGHT into the TemporalConvNet -- model_tcn_gmm has no 'emb'
projection of its own (model_tcn_gmm.py:27-36) -- so block 0 is an ordinary TemporalBlock with Cin = emb + hidden_dim * num_layer and a Dense
residual, followed by the mixture head (model_tcn_gmm.py:39-44).  The high level (session mean-pool + 'emb' bias, MultiRNNCell, user-boundary
mask: model_hier.py:83-93) is model_hier's own.

Two native handles: the main one (hiertcn_session_pool, hiertcn_gru_scan with state_before, hiertcn_embed_concat) for the item table and the
across-session GRU -- created with a minimal low level of its own that is never run -- and a hiertcn_gmm handle with trunk 2 (dense input rows)
for the TemporalConvNet and the head.  Variables (TF names under scope 'hier'): hier/emb/{kernel,bias}, hier/multi_rnn_cell/cell_l/gru_cell/...,
hier/tcn_gmm/temporal_conv_net/tblock_i/{conv1,dense}/{kernel,bias}, hier/tcn_gmm/fully_connected/{weights,biases}.
What the reference does with `pred` afterwards cannot be run (it feeds the [B, T, M (1 + D)] head output to the cross-entropy against
[B, T, item_num] targets, model.py:105-132): `pred` is a GmmHead, scored by calc_loss_gmm like model_tcn_gmm's.
model_low_type 'gru_gmm' (model_hier.py:68-69 -> model_gru_gmm.py:33-54) is the same with trunk 3: the MultiRNNCell of GRUCells under dynamic_rnn on
the x_slice rows (sequence_length = the count of rows that are not all-zero), variables hier/rnn_gmm/rnn/multi_rnn_cell/cell_l/gru_cell/... and
hier/rnn_gmm/fully_connected/{weights,biases} (model_gru_gmm's default scope name 'rnn_gmm' under 'hier').
Training (round 6): the WHOLE model -- item table, emb bias, across-session cells, the low level (TemporalConvNet or GRU stack), head -- trains as one
step on the mixture handle (csrc/gmm_train.hip, hiertcn_gmm_train_step_hier; HierGmmModel.enable_training / train_step_hier); evaluation afterwards
re-uploads the high level into the main handle (sync).
"""
from __future__ import annotations

import copy
import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import HierTCNError
from .model_tcn_gmm import GmmHead, TcnGmmModel, _ck

_SCOPES: Dict[str, "HierGmmModel"] = {}


def init_params(args, seed: int = 2, scope: str = "hier") -> Dict[str, np.ndarray]:
    """glorot-uniform kernels, zero biases, GRU gate bias 1 [TF-sem] in the shapes model_hier + model_tcn_gmm create."""
    from .params import init_params as hier_init
    rng = np.random.default_rng(seed)

    def glorot(shape, fi, fo):
        lim = np.sqrt(6.0 / (fi + fo))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    base = hier_init(_scan_args(args), seed=seed)
    p = {k: v for k, v in base.items() if k.startswith(f"{scope}/emb/") or k.startswith(f"{scope}/multi_rnn_cell/")}
    E, HT, K = int(args.emb_dim), int(args.hidden_dim) * int(args.num_layer), int(args.kernel_size)
    cin = E + HT
    n_out = int(args.num_mixture) * (1 + (1 if getattr(args, "sigma", 0.1) else 2) * int(args.output_dim))
    if getattr(args, "model_low_type", "tcn_gmm") == "gru_gmm":   # the GRU stack on the [e, state] rows (gate bias 1 [TF-sem]) + the head
        H = int(args.hidden_dim)
        for l in range(int(args.num_layer)):
            pre = f"{scope}/rnn_gmm/rnn/multi_rnn_cell/cell_{l}/gru_cell"
            inp = cin if l == 0 else H
            p[f"{pre}/gates/kernel"] = glorot((inp + H, 2 * H), inp + H, 2 * H)
            p[f"{pre}/gates/bias"] = np.ones(2 * H, np.float32)
            p[f"{pre}/candidate/kernel"] = glorot((inp + H, H), inp + H, H)
            p[f"{pre}/candidate/bias"] = np.zeros(H, np.float32)
        p[f"{scope}/rnn_gmm/fully_connected/weights"] = glorot((H, n_out), H, n_out)
        p[f"{scope}/rnn_gmm/fully_connected/biases"] = np.zeros(n_out, np.float32)
        return p
    for i, cout in enumerate(int(c) for c in args.tcn_channel):
        pre = f"{scope}/tcn_gmm/temporal_conv_net/tblock_{i}"
        p[f"{pre}/conv1/kernel"] = glorot((K, cin, cout), K * cin, K * cout)
        p[f"{pre}/conv1/bias"] = np.zeros(cout, np.float32)
        if cin != cout:
            p[f"{pre}/dense/kernel"] = glorot((cin, cout), cin, cout)
            p[f"{pre}/dense/bias"] = np.zeros(cout, np.float32)
        cin = cout
    p[f"{scope}/tcn_gmm/fully_connected/weights"] = glorot((cin, n_out), cin, n_out)
    p[f"{scope}/tcn_gmm/fully_connected/biases"] = np.zeros(n_out, np.float32)
    return p


def _scan_args(args):
    """args of the main handle: the same item table / GRU, with the smallest low level the library accepts (it is never run here)."""
    a = copy.copy(args)
    a.model_low_type, a.tcn_channel, a.kernel_size, a.loss = "tcn", [64], 1, "cross_entropy"   # (64: the narrowest last width the library builds)
    return a


class HierGmmModel(TcnGmmModel):
    """Main handle (table, session pool, across-session GRU) + gmm handle with trunk 2 (TemporalConvNet on [e, state] rows + mixture head)."""

    def __init__(self, args, params: Optional[dict] = None, name: str = "hier", seed: int = 2):
        from .engine import Engine, check_supported
        from .params import init_params as hier_init
        check_supported(_scan_args(args))
        if getattr(args, "loss", "gmm_prob") not in ("gmm_prob", "gmm_prob_max", "cross_entropy"):
            raise NotImplementedError(f"args.loss={args.loss!r}: the mixture low level evaluates 'gmm_prob' or 'gmm_prob_max' (loss.py:247-257)")
        if not torch.cuda.is_available():
            raise HierTCNError("no GPU visible: the compute path is HIP-only (no CPU fallback)")
        self.args, self.name = args, name
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.M, self.D, self.N = int(args.num_mixture), int(args.output_dim), int(args.item_num)
        self.low = getattr(args, "model_low_type", "tcn_gmm")
        if self.low not in ("tcn_gmm", "gru_gmm"):
            raise NotImplementedError(f"model_low_type={self.low!r}: the mixture low levels are 'tcn_gmm' and 'gru_gmm'")
        ch = [int(c) for c in args.tcn_channel]
        self.C = int(args.hidden_dim) if self.low == "gru_gmm" else ch[-1]
        self.E, self.HT = int(args.emb_dim), int(args.hidden_dim) * int(args.num_layer)
        self.params = params if params is not None else init_params(args, seed, name)
        # main handle: our table / emb bias / GRU cells; its own (unused) low level and scoring layer keep their initial values
        sa = _scan_args(args)
        ep = hier_init(sa, seed=seed)
        for k, v in self.params.items():
            if k.startswith(f"{name}/emb/") or k.startswith(f"{name}/multi_rnn_cell/"):
                ep[k] = v
        self.engine = Engine(sa, hier=True)
        self.engine.set_weights(ep)
        self._engine_params = ep     # (the main handle's own, never-run low level keeps these initial values when sync() re-uploads the high level)
        if self.low == "gru_gmm":
            cfg = _lib.GmmConfig(item_num=self.N, output_dim=self.D, num_mixture=self.M, kernel_size=1, n_blocks=0,   # (item_num: the high level's table, training only)
                                 sigma=float(args.sigma) if getattr(args, "sigma", 0.1) else 0.0, trunk=3, hidden_dim=int(args.hidden_dim),
                                 num_layer=int(args.num_layer), input_dim=self.E + self.HT)
        else:
            # (item_num / hidden_dim / num_layer: ignored by the evaluation of trunk 2, they describe the HIGH level to hiertcn_gmm_train_enable)
            cfg = _lib.GmmConfig(item_num=self.N, output_dim=self.D, num_mixture=self.M, kernel_size=int(args.kernel_size), n_blocks=len(ch),
                                 sigma=float(args.sigma) if getattr(args, "sigma", 0.1) else 0.0, trunk=2, hidden_dim=int(args.hidden_dim),
                                 num_layer=int(args.num_layer), input_dim=self.E + self.HT)
            for i, c in enumerate(ch):
                cfg.channels[i] = c
        cfg.table_pass = 1 if os.environ.get("HIERTCN_GMM_TABLE_PASS") else 0   # A/B runs, tests: never the Gram form (read here, once; the library reads no environment)
        self._h = C.c_void_p()
        _ck(self.lib.hiertcn_gmm_create(C.byref(self._h), C.byref(cfg)), self.lib)
        self._upload(self.params)
        self._ws = None

    def close(self):
        super().close()
        if getattr(self, "engine", None) is not None:
            self.engine.close()
            self.engine = None

    def _upload(self, params):
        keep = []

        def dev(k):
            t = torch.as_tensor(np.asarray(params[k], dtype=np.float32)).to(self.device).contiguous()
            keep.append(t)
            return t.data_ptr()

        root = f"{self.name}/rnn_gmm" if self.low == "gru_gmm" else f"{self.name}/tcn_gmm"
        w = _lib.GmmWeights()
        for l in range(int(self.args.num_layer) if self.low == "gru_gmm" else 0):
            pre = f"{root}/rnn/multi_rnn_cell/cell_{l}/gru_cell"
            w.gru_gate_kernel[l], w.gru_gate_bias[l] = dev(f"{pre}/gates/kernel"), dev(f"{pre}/gates/bias")
            w.gru_cand_kernel[l], w.gru_cand_bias[l] = dev(f"{pre}/candidate/kernel"), dev(f"{pre}/candidate/bias")
        for i in range(0 if self.low == "gru_gmm" else len(self.args.tcn_channel)):
            pre = f"{root}/temporal_conv_net/tblock_{i}"
            w.conv_kernel[i], w.conv_bias[i] = dev(f"{pre}/conv1/kernel"), dev(f"{pre}/conv1/bias")
            if f"{pre}/dense/kernel" in params:
                w.res_kernel[i], w.res_bias[i] = dev(f"{pre}/dense/kernel"), dev(f"{pre}/dense/bias")
        w.head_weights, w.head_biases = dev(f"{root}/fully_connected/weights"), dev(f"{root}/fully_connected/biases")
        _ck(self.lib.hiertcn_gmm_set_weights(self._h, C.byref(w), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        torch.cuda.current_stream().synchronize()

    # -- training (csrc/gmm_train.hip, hiertcn_gmm_train_step_hier): model_hier with the tcn_gmm low level as ONE step on the mixture handle ------------
    def _weights_struct(self, bufs):
        root = f"{self.name}/rnn_gmm" if self.low == "gru_gmm" else f"{self.name}/tcn_gmm"
        w = _lib.GmmWeights()
        p = lambda k: bufs[k].data_ptr()  # noqa: E731
        for l in range(int(self.args.num_layer) if self.low == "gru_gmm" else 0):
            pre = f"{root}/rnn/multi_rnn_cell/cell_{l}/gru_cell"
            w.gru_gate_kernel[l], w.gru_gate_bias[l] = p(f"{pre}/gates/kernel"), p(f"{pre}/gates/bias")
            w.gru_cand_kernel[l], w.gru_cand_bias[l] = p(f"{pre}/candidate/kernel"), p(f"{pre}/candidate/bias")
        for i in range(0 if self.low == "gru_gmm" else len(self.args.tcn_channel)):
            pre = f"{root}/temporal_conv_net/tblock_{i}"
            w.conv_kernel[i], w.conv_bias[i] = p(f"{pre}/conv1/kernel"), p(f"{pre}/conv1/bias")
            if f"{pre}/dense/kernel" in bufs:
                w.res_kernel[i], w.res_bias[i] = p(f"{pre}/dense/kernel"), p(f"{pre}/dense/bias")
        w.head_weights, w.head_biases = p(f"{root}/fully_connected/weights"), p(f"{root}/fully_connected/biases")
        w.hl_emb_kernel, w.hl_emb_bias = p(f"{self.name}/emb/kernel"), p(f"{self.name}/emb/bias")
        for l in range(int(self.args.num_layer)):
            pre = f"{self.name}/multi_rnn_cell/cell_{l}/gru_cell"
            w.hl_gate_kernel[l], w.hl_gate_bias[l] = p(f"{pre}/gates/kernel"), p(f"{pre}/gates/bias")
            w.hl_cand_kernel[l], w.hl_cand_bias[l] = p(f"{pre}/candidate/kernel"), p(f"{pre}/candidate/bias")
        return w

    def train_forward_backward(self, *a, **k):   # (TcnGmmModel's id-input step does not apply to this trunk)
        raise HierTCNError("HierGmmModel trains through train_forward_backward_hier / train_step_hier (ids [B, S, L], mask, state)")

    def train_forward_backward_hier(self, xd: torch.Tensor, yd: torch.Tensor, md: torch.Tensor, st: torch.Tensor, mode: Optional[str] = None,
                                    users_global: Optional[torch.Tensor] = None, want_rows: bool = False) -> dict:
        """packed ids xd, yd [B, S, L] int32, mask md [B, S] fp32, carried state st [B, HT] fp32 (all on the device) -> the gradients of EVERY variable of
        model_hier + model_tcn_gmm (item table, emb bias, across-session cells, TemporalConvNet, head) in the handle's gradient arena;
        returns dict(sums = [sum of lane means, users], state = the carried state behind the last slot[, loss_rows [B, S * L]])."""
        if not getattr(self, "_training", False):
            raise HierTCNError("enable_training() first")
        mode = mode or getattr(self.args, "loss", "gmm_prob")
        if mode not in self.MODES:
            raise NotImplementedError(f"args.loss={mode!r}: the mixture models train on 'gmm_prob' or 'gmm_prob_max' (loss.py:247-257)")
        B, S, L = xd.shape
        for t, name, dt, shape in ((xd, "x_ids", torch.int32, (B, S, L)), (yd, "y_ids", torch.int32, (B, S, L)), (md, "mask", torch.float32, (B, S)),
                                   (st, "state", torch.float32, (B, self.HT))):
            if t.dtype != dt or not t.is_cuda or not t.is_contiguous() or tuple(t.shape) != shape:
                raise HierTCNError(f"{name}: a contiguous {dt} device tensor {shape}")
        need = int(self.lib.hiertcn_gmm_train_hier_workspace_bytes(self._h, B, S, L))
        if self._train_ws is None or self._train_ws.numel() < need:
            self._train_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out2 = torch.empty(2, dtype=torch.float64, device=self.device)
        state_out = torch.empty((B, self.HT), dtype=torch.float32, device=self.device)
        rows = torch.empty((B, S * L), dtype=torch.float32, device=self.device) if want_rows else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _ck(self.lib.hiertcn_gmm_train_step_hier(self._h, p(xd), p(yd), p(md), p(st), B, S, L, self.MODES[mode], p(users_global), p(self._train_ws),
                                                 self._train_ws.numel(), p(state_out), p(out2), p(rows), C.c_void_p(torch.cuda.current_stream().cuda_stream)),
            self.lib)
        out = dict(sums=out2, state=state_out)
        if want_rows:
            out["loss_rows"] = rows
        return out

    def train_step_hier(self, xd, yd, md, st, lr: float, mode: Optional[str] = None) -> dict:
        """One optimizer step (run_hier_xing.py:275-301 with the mixture low level): forward + backward + [data parallel: global user count in the loss
        normaliser, one all-reduce of the flat gradient arena] + TF-form Adam."""
        from . import distributed as hd
        users = None
        if hd._is_dist():
            users = (yd.reshape(yd.shape[0], -1) > 0).any(dim=1).sum().to(torch.float32).reshape(1)
            hd._sum_async(users).wait()
        out = self.train_forward_backward_hier(xd, yd, md, st, mode, users_global=users)
        if hd._is_dist():
            hd._sum_async(self.arena(1)).wait()
            hd.all_reduce_metrics(out["sums"])
        self.adam_step(lr)
        return out

    def sync(self):
        """Evaluation reads the 16-bit tables of BOTH handles: the mixture handle's (hiertcn_gmm_train_sync) and the main handle's item table / emb bias /
        across-session cells, which the training step keeps as masters on the mixture handle."""
        if getattr(self, "_training", False) and getattr(self, "_stale", False):
            super().sync()
            w = self.export("weights")
            self.params = w
            ep = dict(self._engine_params)
            for k, v in w.items():
                if k.startswith(f"{self.name}/emb/") or k.startswith(f"{self.name}/multi_rnn_cell/"):
                    ep[k] = v
            self.engine.set_weights(ep)

    def set_parameters(self, params: dict):
        """Checkpoint restore (run_gmm_xing.run_hier_gmm): every variable into BOTH handles (and into the masters when training is enabled)."""
        missing = set(self.params) - set(params)
        if missing:
            raise HierTCNError(f"checkpoint lacks {sorted(missing)[:3]} ... ({len(missing)} variables)")
        self.params = {k: np.asarray(params[k], dtype=np.float32) for k in self.params}
        self._upload(self.params)
        ep = dict(self._engine_params)
        for k, v in self.params.items():
            if k.startswith(f"{self.name}/emb/") or k.startswith(f"{self.name}/multi_rnn_cell/"):
                ep[k] = v
        self.engine.set_weights(ep)
        self._engine_params = ep
        if getattr(self, "_training", False):
            self.restore("weights", self.params)
            self._stale = False     # (both handles' evaluation tables were just written from the same values)

    def eval_step_hier(self, xd: torch.Tensor, yd: torch.Tensor, md: torch.Tensor, st: torch.Tensor, mode: Optional[str] = None) -> dict:
        """One evaluation batch of the run loop (run_gmm_xing.evaluate_hier_gmm): packed ids [B, S, L], mask, carried state -> dict(sums = device
        doubles [sum of lane means, users] of the masked lane-mean mixture loss over the concatenated slots (model.py:111-117), state [B, HT])."""
        mode = mode or getattr(self.args, "loss", "gmm_prob")
        if mode not in self.MODES:
            raise NotImplementedError(f"args.loss={mode!r}: the mixture low level evaluates 'gmm_prob' or 'gmm_prob_max' (loss.py:247-257)")
        B, S, L = xd.shape
        hidden, state_out = self.forward_hier(xd, yd, md, st)
        yy = yd.view(B, S * L)
        rows = self.loss_rows(hidden, yy, mode)
        out2 = torch.empty(2, dtype=torch.float64, device=self.device)
        _ck(self.lib.hiertcn_gmm_lane_loss(self._h, C.c_void_p(rows.data_ptr()), C.c_void_p(yy.data_ptr()), B, S * L, C.c_void_p(out2.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        return dict(sums=out2, state=state_out)

    def forward_rows(self, rows: torch.Tensor, n_seq: int, L: int) -> torch.Tensor:
        """rows [n_seq*L, E + HT] (storage type) -> last TCN activations [n_seq*L, C]."""
        hidden = torch.empty((n_seq * L, self.C), dtype=torch.bfloat16, device=self.device)
        need = int(self.lib.hiertcn_gmm_forward_workspace_bytes(self._h, n_seq, L))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ws = self._ws
        _ck(self.lib.hiertcn_gmm_forward_rows(self._h, C.c_void_p(rows.data_ptr()), n_seq, L, C.c_void_p(hidden.data_ptr()), C.c_void_p(ws.data_ptr()),
                                              ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        return hidden

    def forward(self, ids):   # (TcnGmmModel's id-input forward does not apply to this trunk)
        raise HierTCNError("HierGmmModel reads [e, state] rows: use model_hier(args, x, y, mask, state) or forward_rows")

    def forward_hier(self, xd: torch.Tensor, yd: torch.Tensor, md: torch.Tensor, st: torch.Tensor):
        """packed ids xd, yd [B, S, L] int32, mask md [B, S] fp32, state st [B, HT] fp32 -> (hidden [B*S*L, C], state_out [B, HT])."""
        self.sync()
        eng = self.engine
        B, S, L = xd.shape
        pooled = eng.session_pool(yd.view(B * S, L)).view(B, S, -1)
        _, state_out, before = eng.gru_scan(pooled, md, st, want_state_before=True)
        rows = eng.embed_concat(xd.view(B * S, L), before.view(B * S, -1))
        return self.forward_rows(rows, B * S, L), state_out


def model_hier_tcn_gmm(args, x, y, mask, state, name="hier"):
    """-> (pred: GmmHead over the [B, S * Lmax] packed steps, state [B, HT] on the GPU); called by model_hier for model_low_type 'tcn_gmm' / 'gru_gmm'."""
    from .model import pack_slots
    if name not in _SCOPES:
        _SCOPES[name] = HierGmmModel(args, None, name)
    model = _SCOPES[name]
    xs, ys, ms, lens = pack_slots(x, y, mask)
    B, S, Lmax = xs.shape
    dev = model.device
    xd, yd, md = (torch.from_numpy(v).to(dev) for v in (xs, ys, ms))
    if isinstance(state, torch.Tensor):
        st = state.to(device=dev, dtype=torch.float32).contiguous()
    else:
        st = torch.as_tensor(np.asarray(state, dtype=np.float32)).to(dev).contiguous()
    hidden, state_out = model.forward_hier(xd, yd, md, st)
    head = GmmHead(model, hidden, B, S * Lmax)
    head.lens = lens
    return head, state_out


def reset_scopes():
    for m in _SCOPES.values():
        m.close()
    _SCOPES.clear()
