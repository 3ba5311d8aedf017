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

    def head_means(self, hidden: torch.Tensor, m: int) -> torch.Tensor:
        """parity/debug: mu_m [R, D] fp32 materialised (learned scale: m in [M, 2M) materialises the log-scales s_{m-M})."""
        R = hidden.shape[0]
        out = torch.empty((R, self.D), dtype=torch.float32, device=self.device)
        _ck(self.lib.hiertcn_gmm_head_means(self._h, C.c_void_p(hidden.data_ptr()), R, int(m), C.c_void_p(out.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        return out


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
