"""model_tcn_gmm with the reference's signature (model_tcn_gmm.py:12), backed by the HIP kernels (csrc/gmm.hip), plus
calc_loss_gmm for the two likelihood losses that need no negative samples (loss.py:247-257).

    model_tcn_gmm(args, x, x_gap=None, x_impression=None, name='tcn_gmm', reuse=True, training=False, mask=None) -> pred

x = ids [B, T] (the reference feeds one_hot(ids) * sign(ids), model.py:56-61; the TemporalConvNet reads it directly -- no 'emb'
projection in this model -- so block 0 is a row gather).  `pred` stands for the [B, T, M * (1 + D)] head output without
materialising it (5 x the item catalogue per step): GmmHead keeps the last TCN activations and scores them on demand.
The decode of `pred` into (pi, mu, sigma) is NOT in the reference (calc_loss_gmm has no caller); the one used here is documented
in oracle/gmm_oracle.py and include/hiertcn.h ("parity unpinned" for the decode).  model_gru_gmm (model_gru_gmm.py:11) shares the
head: hiertcn_amd/model_gru_gmm.py.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import HierTCNError

_SCOPES: Dict[str, "TcnGmmModel"] = {}


def _ck(rc, lib):
    if rc != 0:
        msg = lib.hiertcn_gmm_last_error()
        raise HierTCNError(f"hiertcn_gmm call failed ({rc}): {msg.decode() if msg else '?'}")


def init_params(args, seed: int = 2, scope: str = "tcn_gmm") -> Dict[str, np.ndarray]:
    """xavier/glorot-uniform kernels, zero biases [TF-sem] in the reference's shapes (model_tcn_gmm.py:28-44)."""
    rng = np.random.default_rng(seed)

    def glorot(shape, fi, fo):
        lim = np.sqrt(6.0 / (fi + fo))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    K, cin = int(args.kernel_size), int(args.item_num)
    p = {}
    for i, cout in enumerate(int(c) for c in args.tcn_channel):
        pre = f"{scope}/temporal_conv_net/tblock_{i}"
        p[f"{pre}/conv1/kernel"] = glorot((K, cin, cout), K * cin, K * cout)
        p[f"{pre}/conv1/bias"] = np.zeros(cout, np.float32)
        if cin != cout:
            p[f"{pre}/dense/kernel"] = glorot((cin, cout), cin, cout)
            p[f"{pre}/dense/bias"] = np.zeros(cout, np.float32)
        cin = cout
    n_out = int(args.num_mixture) * (1 + (1 if getattr(args, "sigma", 0.1) else 2) * int(args.output_dim))   # learned scale: M (1 + 2 D)
    p[f"{scope}/fully_connected/weights"] = glorot((cin, n_out), cin, n_out)
    p[f"{scope}/fully_connected/biases"] = np.zeros(n_out, np.float32)
    return p


class TcnGmmModel:
    """One hiertcn_gmm handle on the current device + its TF-named variables."""

    def __init__(self, args, params: Optional[dict] = None, name: str = "tcn_gmm", seed: int = 2):
        from .engine import check_supported
        check_supported(args, loss_ok=("gmm_prob", "gmm_prob_max", "cross_entropy"))
        if not torch.cuda.is_available():
            raise HierTCNError("no GPU visible: the compute path is HIP-only (no CPU fallback)")
        self.args, self.name = args, name
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.M, self.D, self.N = int(args.num_mixture), int(args.output_dim), int(args.item_num)
        ch = [int(c) for c in args.tcn_channel]
        self.C = ch[-1]
        cfg = _lib.GmmConfig(item_num=self.N, output_dim=self.D, num_mixture=self.M, kernel_size=int(args.kernel_size), n_blocks=len(ch),
                             sigma=float(args.sigma) if getattr(args, "sigma", 0.1) else 0.0)   # 0: learned scale (args.sigma None)
        for i, c in enumerate(ch):
            cfg.channels[i] = c
        cfg.table_pass = 1 if os.environ.get("HIERTCN_GMM_TABLE_PASS") else 0   # A/B runs, tests: never the Gram form (read here, once; the library reads no environment)
        self._h = C.c_void_p()
        _ck(self.lib.hiertcn_gmm_create(C.byref(self._h), C.byref(cfg)), self.lib)
        self.params = params if params is not None else init_params(args, seed, name)
        self._upload(self.params)
        self._ws = None

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.hiertcn_gmm_destroy(self._h)
            self._h = C.c_void_p()

    def _upload(self, params):
        keep = []

        def dev(k):
            t = torch.as_tensor(np.asarray(params[k], dtype=np.float32)).to(self.device).contiguous()
            keep.append(t)
            return t.data_ptr()

        w = _lib.GmmWeights()
        for i in range(len(self.args.tcn_channel)):
            pre = f"{self.name}/temporal_conv_net/tblock_{i}"
            w.conv_kernel[i], w.conv_bias[i] = dev(f"{pre}/conv1/kernel"), dev(f"{pre}/conv1/bias")
            if f"{pre}/dense/kernel" in params:
                w.res_kernel[i], w.res_bias[i] = dev(f"{pre}/dense/kernel"), dev(f"{pre}/dense/bias")
        w.head_weights, w.head_biases = dev(f"{self.name}/fully_connected/weights"), dev(f"{self.name}/fully_connected/biases")
        _ck(self.lib.hiertcn_gmm_set_weights(self._h, C.byref(w), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        torch.cuda.current_stream().synchronize()

    def _workspace(self, R):
        need = int(self.lib.hiertcn_gmm_workspace_bytes(self._h, R))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward(self, ids: torch.Tensor) -> torch.Tensor:
        """ids int32 [B, T] -> last TCN activations [B*T, C] (storage type)."""
        B, T = ids.shape
        self.sync()
        hidden = torch.empty((B * T, self.C), dtype=torch.bfloat16, device=self.device)
        need = int(self.lib.hiertcn_gmm_forward_workspace_bytes(self._h, B, T))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ws = self._ws
        _ck(self.lib.hiertcn_gmm_forward(self._h, C.c_void_p(ids.data_ptr()), B, T, C.c_void_p(hidden.data_ptr()), C.c_void_p(ws.data_ptr()),
                                         ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        return hidden

    def loss_rows(self, hidden: torch.Tensor, y_ids: torch.Tensor, mode: str = "gmm_prob", parts: bool = False):
        """-> loss [R] (0 where y == 0) [, |mu_m|^2 [M, R] (learned scale: sum_d mu_d^2 exp(-2 s_d)), mu_m[y] [M, R], prior logits [R, M]]."""
        R = hidden.shape[0]
        self.sync()
        y = y_ids.reshape(-1).contiguous()
        loss = torch.empty(R, dtype=torch.float32, device=self.device)
        sq = torch.empty((self.M, R), dtype=torch.float32, device=self.device) if parts else None
        tg = torch.empty((self.M, R), dtype=torch.float32, device=self.device) if parts else None
        pl = torch.empty((R, self.M), dtype=torch.float32, device=self.device) if parts else None
        ws = self._workspace(R)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _ck(self.lib.hiertcn_gmm_loss(self._h, p(hidden), p(y), R, {"gmm_prob": 0, "gmm_prob_max": 1}[mode], p(loss), p(sq), p(tg), p(pl),
                                      p(ws), ws.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        return (loss, sq, tg, pl) if parts else loss

    def eval_step(self, x, y) -> dict:
        """One evaluation step of the run loop (run_gmm_xing.evaluate_gmm): ids x, y [B, T] -> the batch's masked lane-mean mixture
        loss (model.py:111-117 on args.loss) as `loss_sum` / `user_count` (device doubles summed over lanes: ready for a cross-rank sum)."""
        ids = torch.as_tensor(np.ascontiguousarray(np.asarray(x).astype(np.int32))).to(self.device)
        yd = torch.as_tensor(np.ascontiguousarray(np.asarray(y).astype(np.int32))).to(self.device)
        B, T = ids.shape
        hidden = self.forward(ids)
        mode = getattr(self.args, "loss", "gmm_prob")
        if mode not in ("gmm_prob", "gmm_prob_max"):
            raise NotImplementedError(f"args.loss={mode!r}: the mixture models evaluate 'gmm_prob' or 'gmm_prob_max' (loss.py:247-257)")
        rows = self.loss_rows(hidden, yd, mode)
        out2 = torch.empty(2, dtype=torch.float64, device=self.device)
        _ck(self.lib.hiertcn_gmm_lane_loss(self._h, C.c_void_p(rows.data_ptr()), C.c_void_p(yd.data_ptr()), B, T, C.c_void_p(out2.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        return dict(sums=out2, loss_rows=rows.view(B, T))

    # -- training (include/hiertcn.h "training step of the mixture-output model", csrc/gmm_train.hip) -----------------------------------------
    MODES = {"gmm_prob": 0, "gmm_prob_max": 1}

    def _weights_struct(self, bufs: Dict[str, torch.Tensor]):
        w = _lib.GmmWeights()
        p = lambda k: bufs[k].data_ptr()  # noqa: E731
        for i in range(len(self.args.tcn_channel)):
            pre = f"{self.name}/temporal_conv_net/tblock_{i}"
            w.conv_kernel[i], w.conv_bias[i] = p(f"{pre}/conv1/kernel"), p(f"{pre}/conv1/bias")
            if f"{pre}/dense/kernel" in bufs:
                w.res_kernel[i], w.res_bias[i] = p(f"{pre}/dense/kernel"), p(f"{pre}/dense/bias")
        w.head_weights, w.head_biases = p(f"{self.name}/fully_connected/weights"), p(f"{self.name}/fully_connected/biases")
        return w

    def enable_training(self):
        """fp32 masters (filled from self.params), gradients and Adam moments on the device: hiertcn_gmm_train_enable.  The models on ids
        (model_tcn_gmm, model_gru_gmm) and model_hier with either as its low level (HierGmmModel), fixed or learned scale."""
        if getattr(self, "_training", False):
            return
        bufs = {k: torch.as_tensor(np.asarray(v, dtype=np.float32)).to(self.device).contiguous() for k, v in self.params.items()}
        w = self._weights_struct(bufs)
        _ck(self.lib.hiertcn_gmm_train_enable(self._h, C.byref(w), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        torch.cuda.current_stream().synchronize()
        self._training = True
        self._train_ws = None
        self._stale = False

    def train_forward_backward(self, x_ids: torch.Tensor, y_ids: torch.Tensor, mode: Optional[str] = None, users_global: Optional[torch.Tensor] = None,
                               want_rows: bool = False, sums_out: Optional[torch.Tensor] = None) -> dict:
        """ids [B, T] int32 on the device -> the gradients of every variable of the batch's lane-mean mixture loss (model.py:111-117 on args.loss)
        in the handle's gradient arena; returns dict(sums = device doubles [sum of lane means, users][, loss_rows [B, T]])."""
        if not getattr(self, "_training", False):
            raise HierTCNError("enable_training() first")
        mode = mode or getattr(self.args, "loss", "gmm_prob")
        if mode not in self.MODES:
            raise NotImplementedError(f"args.loss={mode!r}: the mixture models train on 'gmm_prob' or 'gmm_prob_max' (loss.py:247-257)")
        for t, name in ((x_ids, "x_ids"), (y_ids, "y_ids")):
            if t.dtype != torch.int32 or not t.is_cuda or not t.is_contiguous() or t.dim() != 2:
                raise HierTCNError(f"{name}: a contiguous int32 device tensor [B, T]")
        if x_ids.shape != y_ids.shape:
            raise HierTCNError("x_ids and y_ids differ in shape")
        B, T = x_ids.shape
        need = int(self.lib.hiertcn_gmm_train_workspace_bytes(self._h, B, T))
        if self._train_ws is None or self._train_ws.numel() < need:
            self._train_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        out2 = sums_out if sums_out is not None else torch.empty(2, dtype=torch.float64, device=self.device)
        rows = torch.empty((B, T), dtype=torch.float32, device=self.device) if want_rows else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _ck(self.lib.hiertcn_gmm_train_step(self._h, p(x_ids), p(y_ids), B, T, self.MODES[mode], p(users_global), p(self._train_ws),
                                            self._train_ws.numel(), p(out2), p(rows), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        out = dict(sums=out2)
        if want_rows:
            out["loss_rows"] = rows
        return out

    def adam_step(self, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, grad_scale: float = 1.0):
        """TF-form Adam (model.py:139) on the masters; the evaluation tables are re-derived lazily (sync) before the next evaluation."""
        _ck(self.lib.hiertcn_gmm_adam_step(self._h, float(lr), float(beta1), float(beta2), float(eps), float(grad_scale),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        self._stale = True

    def sync(self):
        """Evaluation tables <- masters (hiertcn_gmm_train_sync) where an optimizer step has run since the last one."""
        if getattr(self, "_training", False) and getattr(self, "_stale", False):
            _ck(self.lib.hiertcn_gmm_train_sync(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
            self._stale = False

    def train_step(self, x, y, lr: float, mode: Optional[str] = None) -> dict:
        """One optimizer step of the run loop (run_gmm_xing.run_gmm): ids x, y [B, T] (host or device) -> forward + backward + the data-parallel
        gradient sum (distributed.py: global user count in the loss normaliser, one all-reduce of the flat gradient arena) + Adam."""
        from . import distributed as hd
        ids = torch.as_tensor(np.ascontiguousarray(np.asarray(x).astype(np.int32))).to(self.device) if not isinstance(x, torch.Tensor) else x
        yd = torch.as_tensor(np.ascontiguousarray(np.asarray(y).astype(np.int32))).to(self.device) if not isinstance(y, torch.Tensor) else y
        users = None
        if hd._is_dist():
            users = (yd > 0).any(dim=1).sum().to(torch.float32).reshape(1)
            hd._sum_async(users).wait()
        out = self.train_forward_backward(ids, yd, mode, users_global=users)
        if hd._is_dist():
            hd._sum_async(self.arena(1)).wait()
            hd.all_reduce_metrics(out["sums"])
        self.adam_step(lr)
        return out

    def capture_train_graph(self, B: int, T: int, mode: Optional[str] = None) -> "GmmTrainGraph":
        """The forward + backward of one batch shape as a replayable hipGraph (torch.cuda.CUDAGraph on ROCm): model_gru_gmm's step is a chain of
        ~1,000 small launches at T = 200 (five per anti-diagonal of the (step, layer) wavefront), whose host cost the replay removes (measured: no gain -- the kernels' own ~10 us each bound the chain).  Adam stays outside (its bias-corrected rate is a
        host scalar that changes every step): GmmTrainGraph.step = copy ids -> replay -> adam_step."""
        return GmmTrainGraph(self, B, T, mode)

    def arena(self, which: int) -> torch.Tensor:
        """The flat fp32 arena `which` (0 masters, 1 gradients, 2 / 3 Adam moments) as a torch view (no copy)."""
        ptr, n = C.c_void_p(), C.c_size_t()
        _ck(self.lib.hiertcn_gmm_train_arena(self._h, int(which), C.byref(ptr), C.byref(n)), self.lib)

        class _Raw:
            __cuda_array_interface__ = {"shape": (int(n.value),), "typestr": "<f4", "data": (int(ptr.value), False), "version": 2}

        return torch.as_tensor(_Raw(), device=self.device)

    def export(self, which: str = "weights") -> Dict[str, np.ndarray]:
        """Masters ('weights'), gradients ('grads') or Adam moments ('m', 'v') as {TF variable name: fp32 ndarray in the reference layout}."""
        k = {"weights": 0, "grads": 1, "m": 2, "v": 3}[which]
        bufs = {name: torch.zeros(np.asarray(v).shape, dtype=torch.float32, device=self.device) for name, v in self.params.items()}
        w = self._weights_struct(bufs)
        _ck(self.lib.hiertcn_gmm_train_copy(self._h, k, 0, C.byref(w), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        torch.cuda.current_stream().synchronize()
        return {name: t.cpu().numpy() for name, t in bufs.items()}

    def restore(self, which: str, tensors: Dict[str, np.ndarray]):
        """The reverse of export (checkpoint restore: masters and Adam moments)."""
        k = {"weights": 0, "grads": 1, "m": 2, "v": 3}[which]
        bufs = {name: torch.as_tensor(np.asarray(tensors[name], dtype=np.float32)).to(self.device).contiguous() for name in self.params}
        w = self._weights_struct(bufs)
        _ck(self.lib.hiertcn_gmm_train_copy(self._h, k, 1, C.byref(w), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        torch.cuda.current_stream().synchronize()
        if k == 0:
            self._stale = True

    def adam_time(self, set_to: Optional[int] = None) -> int:
        get = C.c_int64(0)
        st = C.c_int64(int(set_to)) if set_to is not None else None
        _ck(self.lib.hiertcn_gmm_adam_time(self._h, C.byref(get), C.byref(st) if st is not None else None), self.lib)
        return int(get.value)

    def head_means(self, hidden: torch.Tensor, m: int) -> torch.Tensor:
        """parity/debug: mu_m [R, D] fp32 materialised (learned scale: m in [M, 2M) materialises the log-scales s_{m-M})."""
        R = hidden.shape[0]
        out = torch.empty((R, self.D), dtype=torch.float32, device=self.device)
        _ck(self.lib.hiertcn_gmm_head_means(self._h, C.c_void_p(hidden.data_ptr()), R, int(m), C.c_void_p(out.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        return out


class GmmTrainGraph:
    """Static-shape training step (forward + backward into the gradient arena) as a replayable graph.  Buffers: x, y int32 [B, T]; sums fp64 [2]."""

    def __init__(self, model: "TcnGmmModel", B: int, T: int, mode: Optional[str] = None):
        model.enable_training()
        self.model, self.mode = model, mode or getattr(model.args, "loss", "gmm_prob")
        dev = model.device
        self.x = torch.zeros((B, T), dtype=torch.int32, device=dev)
        self.y = torch.zeros((B, T), dtype=torch.int32, device=dev)
        self.sums = torch.zeros(2, dtype=torch.float64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):   # warm-up outside the capture (the workspace is allocated here)
            for _ in range(2):
                model.train_forward_backward(self.x, self.y, self.mode, sums_out=self.sums)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            model.train_forward_backward(self.x, self.y, self.mode, sums_out=self.sums)

    def step(self, x_ids: torch.Tensor, y_ids: torch.Tensor, lr: float) -> dict:
        self.x.copy_(x_ids, non_blocking=True)
        self.y.copy_(y_ids, non_blocking=True)
        self.graph.replay()
        self.model.adam_step(lr)
        return dict(sums=self.sums)


class GmmHead:
    """What `pred` is for a mixture model: the last TCN activations plus the handle that can score them."""

    def __init__(self, model: TcnGmmModel, hidden: torch.Tensor, B: int, T: int):
        self.model, self.hidden, self.B, self.T = model, hidden, B, T

    @property
    def shape(self):
        learned = not getattr(self.model.args, "sigma", 0.1)   # model_tcn_gmm.py:39-42
        return (self.B, self.T, self.model.M * (1 + (2 if learned else 1) * self.model.D))


def model_tcn_gmm(args, x, x_gap=None, x_impression=None, name="tcn_gmm", reuse=True, training=False, mask=None):
    ids = np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x).astype(np.int32)
    if ids.ndim != 2:
        raise NotImplementedError("model_tcn_gmm takes ids [B, T] (the one-hot of model.py:56 is undone)")
    if name not in _SCOPES:
        _SCOPES[name] = TcnGmmModel(args, None, name)
    model = _SCOPES[name]
    B, T = ids.shape
    hidden = model.forward(torch.from_numpy(np.ascontiguousarray(ids)).to(model.device))
    return GmmHead(model, hidden, B, T)


def calc_loss_gmm(pred: GmmHead, y, y_impression=None, pi=None, mu=None, sigma=None, gmm=None, loss: str = None):
    """loss.py:225-292 for args.loss in {'gmm_prob', 'gmm_prob_max'}: -> loss [B, T] (torch, on the GPU), already 0 on masked steps.
    y = target ids [B, T].  pi / mu / sigma / gmm are what the reference's (missing) decode would have passed; they are derived
    from `pred` inside the fused epilogue and must be None."""
    if any(v is not None for v in (pi, mu, sigma, gmm)):
        raise NotImplementedError("the decode lives in the fused epilogue (include/hiertcn.h, hiertcn_gmm_loss); pass None")
    mode = loss or getattr(pred.model.args, "loss", "gmm_prob")
    if mode not in ("gmm_prob", "gmm_prob_max"):
        raise NotImplementedError(f"args.loss={mode!r}: the negative-sample GMM losses (loss.py:258-290) need impressions (XING has none)")
    yd = torch.as_tensor(np.asarray(y.detach().cpu() if isinstance(y, torch.Tensor) else y).astype(np.int32)).to(pred.model.device)
    return pred.model.loss_rows(pred.hidden, yd, mode).view(pred.B, pred.T)


def reset_scopes():
    for m in _SCOPES.values():
        m.close()
    _SCOPES.clear()
