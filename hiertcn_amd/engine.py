"""Device-side engine: a thin, typed wrapper over the C-ABI (include/hiertcn.h).

PyTorch is used here for device memory, streams and (elsewhere) torch.distributed only; every
computation on the hot path is a hand-written HIP kernel reached through ctypes.  Tensors are
passed as raw device pointers; the engine checks dtype / device / contiguity / shape so that a
bad call fails in Python rather than as a GPU fault.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

import weakref

from . import _lib
from ._lib import HierTCNError, check

_LIVE_ENGINES = weakref.WeakSet()   # every open Engine: distributed.init_from_env retunes them when it finds ranks sharing a GPU

METRIC_KEYS = ("loss", "recall1", "recall5", "recall10", "mrr", "rank_frac", "user_count", "valid_steps")


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _req(t: torch.Tensor, dtype, name: str, shape=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if not t.is_cuda:
        raise HierTCNError(f"{name}: must live on the GPU (got {t.device})")
    if t.dtype != dtype:
        raise HierTCNError(f"{name}: dtype {t.dtype}, expected {dtype}")
    if not t.is_contiguous():
        raise HierTCNError(f"{name}: must be contiguous")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise HierTCNError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
    return t


UNSUPPORTED_FLAGS = ("has_gap", "has_impression", "has_batchnorm", "has_weightnorm", "has_layernorm", "l2_normalize")


def check_supported(args, train: bool = False, loss_ok=("cross_entropy",)):
    """The ONE place that refuses model-changing flags the hot path does not implement.  Every entry
    point builds an Engine, and Engine() builds its config here, so neither the model_hier / model_tcn shims nor the run loops
    nor the CLI can accept such a flag and silently run something else under the requested run name (args.py:232-240)."""
    for flag in UNSUPPORTED_FLAGS:
        if getattr(args, flag, False):
            raise NotImplementedError(f"args.{flag}=True is outside the hot path (reference default is False)")
    if getattr(args, "data_noise", None) is not None:
        raise NotImplementedError("args.data_noise is outside the hot path")
    if getattr(args, "loss", "cross_entropy") not in loss_ok:
        raise NotImplementedError(f"args.loss={args.loss!r}: this model is built for {loss_ok} (loss.py:20-21, 247-257)")
    if train and float(getattr(args, "dropout", 0.0) or 0.0) != 0.0:
        raise NotImplementedError("args.dropout != 0 while training: tf.layers.Dropout after conv1 (customized_tcn_cell.py:119) "
                                  "is not built; the reference default is 0.0")


def config_from_args(args, hier: bool = True, train: bool = False) -> _lib.Config:
    """args.py attribute names -> hiertcn_config_t."""
    check_supported(args, train)
    cfg = _lib.Config()
    cfg.item_num = int(args.item_num)
    cfg.emb_dim = int(getattr(args, "emb_dim", 128))
    cfg.hidden_dim = int(args.hidden_dim)
    cfg.num_layer = int(args.num_layer)
    cfg.kernel_size = int(args.kernel_size)
    ch = [int(c) for c in args.tcn_channel]
    if len(ch) > _lib.MAX_BLOCKS:
        raise HierTCNError("too many TCN blocks")
    cfg.n_blocks = len(ch)
    for i, c in enumerate(ch):
        cfg.channels[i] = c
    cfg.hier = 1 if hier else 0
    # hier: the low level of model_hier (model_hier.py:62-69); single level: model.py:66-73 dispatches on model_type ('gru': model_gru, else model_tcn)
    low = getattr(args, "model_low_type", "tcn") if hier else ("gru" if getattr(args, "model_type", "tcn") == "gru" else "tcn")
    if low not in ("tcn", "gru"):
        raise HierTCNError(f"model_low_type {low!r}: 'tcn' (HierTCN) and 'gru' (HierGRU) are built")
    cfg.low_type = 1 if low == "gru" else 0
    return cfg


class Engine:
    """One handle on the current CUDA(HIP) device."""

    def __init__(self, args, hier: bool = True, device: Optional[torch.device] = None, train: bool = False,
                 dtype: Optional[str] = None, tuning: Optional[dict] = None, lib_variant: Optional[str] = None):
        """dtype: storage type of the 16-bit tensors: 'bf16' (default; args.storage_dtype overrides) or 'fp16' (the
        libhiertcn_gfx950_f16.so build: BASELINE.json configs[3], inference only); 'fp32' = the exact numerics mode
        (hiertcn_set_precision: everything fp32 as the reference computes it, model_tcn.py:35-41; evaluation only, for
        parity runs on trained checkpoints -- the mode that meets "fp32 logits within 1e-3" at any logit scale);
        'split16' = the FAST parity-grade mode (HIERTCN_PRECISION_SPLIT16: fp32 low level, fused scoring on hi + lo IEEE-half
        operands, three MFMAs per product, csrc/score_x3.hip): the same tolerance at a third of the bf16 rate; evaluation only.
        tuning: hiertcn_tuning_t fields (include/hiertcn.h) to set on the handle, on top of what the environment asks for -- the library itself
        reads no environment variable: _lib.tuning_from_env() maps the HIERTCN_* switches of earlier rounds onto the fields ONCE, here.
        lib_variant: 'ablation' / 'testhooks' = another build of the bfloat16 library (_lib.LIB_VARIANTS; A/B runs and tests)."""
        dtype = dtype or getattr(args, "storage_dtype", "bf16")
        if dtype not in ("bf16", "fp16", "fp32", "split16"):
            raise HierTCNError(f"storage dtype {dtype!r}: 'bf16', 'fp16', 'fp32' (exact mode) or 'split16' (fast parity-grade mode)")
        if dtype == "split16" and train:
            raise HierTCNError("dtype='split16' is an EVALUATION mode; training runs on bf16 storage with fp32 masters, or in the fp32 mode (dtype='fp32')")
        if dtype == "fp32" and train and hier and getattr(args, "model_low_type", "tcn") not in ("tcn", "gru"):
            raise HierTCNError("dtype='fp32' with train=True (the fp32 training mode: fp32 forward on the masters, materialised logits) is built for "
                               "model_hier with model_low_type 'tcn' or 'gru' and the single-level model_tcn, at the reference's batch sizes")
        self.f16 = dtype == "fp16"
        self.mode = dtype if dtype in ("fp32", "split16") else "storage16"
        self.exact = self.mode != "storage16"       # fp32 hidden rows / fp32 low level (the FP32 and the SPLIT16 mode)
        self.act_dtype = torch.float32 if self.exact else (torch.float16 if self.f16 else torch.bfloat16)
        self.lib = _lib.load(f16=self.f16, variant=lib_variant)
        if not torch.cuda.is_available():
            raise HierTCNError("no GPU visible: the HierTCN compute path is HIP-only (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.hier = hier
        self.cfg = config_from_args(args, hier, train=bool(train))
        self._args_for_shapes = args
        self.N = int(self.cfg.item_num)
        self.E = int(self.cfg.emb_dim)
        self.low_gru = bool(self.cfg.low_type == 1)
        self.H = int(self.cfg.hidden_dim) if (hier or self.low_gru) else 0      # (single-level model_gru: the stack's width / depth)
        self.nl = int(self.cfg.num_layer) if (hier or self.low_gru) else 0
        self.HT = self.H * self.nl if hier else 0                                # the carried state of model_hier; single-level models carry none
        self.kind = "hier" if hier else ("gru" if self.low_gru else "tcn")      # scope of the TF variable names (model.py:66-88)
        self.C = self.H if self.low_gru else int(self.cfg.channels[self.cfg.n_blocks - 1])
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            self._ck(self.lib.hiertcn_create(C.byref(self._h), C.byref(self.cfg)))
            if self.exact:   # allocates the weight copies of the mode, which set_weights fills; must precede set_weights
                self._ck(self.lib.hiertcn_set_precision(self._h, _lib.PRECISION_SPLIT16 if self.mode == "split16" else _lib.PRECISION_FP32))
        self.train = bool(train)
        if self.train:   # fp32 masters / gradients / Adam moments; must precede set_weights
            with torch.cuda.device(self.device):
                self._ck(self.lib.hiertcn_train_enable(self._h))
        self._train_ws: Optional[torch.Tensor] = None
        self._ws: Optional[torch.Tensor] = None
        self._gru_ws: Optional[torch.Tensor] = None
        self._tcn_ws: Optional[torch.Tensor] = None
        self._weights_set = False
        # order of precedence, lowest first: a device shared between ranks (distributed.init_from_env's decision, whenever it is taken: live
        # engines are retuned there) < the HIERTCN_* environment switches < the `tuning` argument
        t = {}
        from . import distributed as _dist
        if _dist.shared_device():
            t.update(_dist.SHARED_DEVICE_TUNING)
        t.update(_lib.tuning_from_env())
        t.update(tuning or {})
        if t:
            self.set_tuning(**t)
        _LIVE_ENGINES.add(self)

    # ---------------------------------------------------------------------------------------
    def get_tuning(self) -> dict:
        """The handle's hiertcn_tuning_t as {field: value} (thresholds: the values in force)."""
        t = _lib.Tuning()
        self._ck(self.lib.hiertcn_get_tuning(self._h, C.byref(t)))
        return {f: int(getattr(t, f)) for f in _lib.TUNING_FIELDS}

    def set_tuning(self, **fields):
        """Change fields of the handle's hiertcn_tuning_t (launch selection: -1 automatic, 0 never, 1 wherever the shape allows; thresholds:
        0 = the default).  Cached workspaces are dropped: their sizes are functions of the tuning."""
        t = _lib.Tuning()
        self._ck(self.lib.hiertcn_get_tuning(self._h, C.byref(t)))
        for k, v in fields.items():
            if k not in _lib.TUNING_FIELDS:
                raise HierTCNError(f"hiertcn_tuning_t has no field {k!r}: {', '.join(_lib.TUNING_FIELDS)}")
            setattr(t, k, int(v))
        self._ck(self.lib.hiertcn_set_tuning(self._h, C.byref(t)))
        self._train_ws = self._ws = self._gru_ws = self._tcn_ws = None

    def _ck(self, rc: int):
        check(rc, self.lib)

    # ---------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self.lib.hiertcn_destroy(self._h)
            self._h = C.c_void_p()
            _LIVE_ENGINES.discard(self)

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream():
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def set_precision(self, dtype: str):
        """Switch between the numerics modes of the handle: 'fp32' (exact), 'split16' (fast parity-grade) and the 16-bit storage kernels
        ('bf16' / 'fp16', whichever library the handle was built from).  The FIRST switch to 'fp32' / 'split16' allocates that mode's
        weight copies and must precede set_weights; afterwards the modes can be switched between calls."""
        store = "fp16" if self.f16 else "bf16"
        if dtype not in ("fp32", "split16", store):
            raise HierTCNError(f"precision {dtype!r}: this handle offers 'fp32', 'split16' and {store!r}")
        code = {"fp32": _lib.PRECISION_FP32, "split16": _lib.PRECISION_SPLIT16}.get(dtype, _lib.PRECISION_STORAGE16)
        with torch.cuda.device(self.device):
            self._ck(self.lib.hiertcn_set_precision(self._h, code))
        self.mode = dtype if dtype in ("fp32", "split16") else "storage16"
        self.exact = self.mode != "storage16"
        self.act_dtype = torch.float32 if self.exact else (torch.float16 if self.f16 else torch.bfloat16)

    # ---------------------------------------------------------------------------------------
    def set_weights(self, params: Dict[str, "np.ndarray | torch.Tensor"], root: Optional[str] = None):
        """params: TF-named tensors in the reference layout; numpy or torch, any
        float dtype; uploaded as fp32 and converted on the device by hiertcn_set_weights."""
        root = root or self.kind
        tcn = (f"{root}/gru" if self.low_gru else f"{root}/tcn") if self.hier else root
        keep = []

        def dev(name, shape=None, optional=False):
            if name not in params:
                if optional:
                    return None
                raise HierTCNError(f"missing parameter {name}")
            v = params[name]
            t = torch.as_tensor(np.asarray(v, dtype=np.float32) if not isinstance(v, torch.Tensor) else v)
            t = t.to(device=self.device, dtype=torch.float32).contiguous()
            if shape is not None and tuple(t.shape) != tuple(shape):
                raise HierTCNError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")
            keep.append(t)
            return t.data_ptr()

        E, N, H, K = self.E, self.N, self.H, int(self.cfg.kernel_size)
        w = _lib.Weights()
        w.emb_kernel = dev(f"{root}/emb/kernel", (N, E))
        if self.hier:
            w.emb_bias = dev(f"{root}/emb/bias", (E,))
            w.proj_kernel = dev(f"{tcn}/emb/kernel", (E + self.HT, E))
        cin = H if self.low_gru else E
        for i in range(0 if self.low_gru else self.cfg.n_blocks):
            cout = int(self.cfg.channels[i])
            pre = f"{tcn}/temporal_conv_net/tblock_{i}"
            w.conv_kernel[i] = dev(f"{pre}/conv1/kernel", (K, cin, cout))
            w.conv_bias[i] = dev(f"{pre}/conv1/bias", (cout,))
            rk = dev(f"{pre}/dense/kernel", (cin, cout), optional=True)
            if rk is not None:
                w.res_kernel[i] = rk
                w.res_bias[i] = dev(f"{pre}/dense/bias", (cout,))
            cin = cout
        w.out_kernel = dev(f"{tcn}/dense/kernel", (cin, N))
        w.out_bias = dev(f"{tcn}/dense/bias", (N,))
        if self.hier:
            inp = E
            for l in range(self.nl):
                pre = f"{root}/multi_rnn_cell/cell_{l}/gru_cell"
                w.gru_gate_kernel[l] = dev(f"{pre}/gates/kernel", (inp + H, 2 * H))
                w.gru_gate_bias[l] = dev(f"{pre}/gates/bias", (2 * H,))
                w.gru_cand_kernel[l] = dev(f"{pre}/candidate/kernel", (inp + H, H))
                w.gru_cand_bias[l] = dev(f"{pre}/candidate/bias", (H,))
                inp = H
        if self.low_gru:   # dynamic_rnn's scope 'rnn' under model_gru's 'gru' (model_gru.py:47)
            inp = E
            for l in range(self.nl):
                lo = f"{tcn}/rnn/multi_rnn_cell/cell_{l}/gru_cell"
                w.low_gate_kernel[l] = dev(f"{lo}/gates/kernel", (inp + H, 2 * H))
                w.low_gate_bias[l] = dev(f"{lo}/gates/bias", (2 * H,))
                w.low_cand_kernel[l] = dev(f"{lo}/candidate/kernel", (inp + H, H))
                w.low_cand_bias[l] = dev(f"{lo}/candidate/bias", (H,))
                inp = H
        with torch.cuda.device(self.device):
            self._ck(self.lib.hiertcn_set_weights(self._h, C.byref(w), self._stream()))
            torch.cuda.current_stream().synchronize()  # the fp32 staging copies die with `keep`
        self._weights_set = True

    # ---------------------------------------------------------------------------------------
    def session_pool(self, y_ids: torch.Tensor) -> torch.Tensor:
        """y_ids [n_sessions, L] int32 -> [n_sessions, E] fp32   (model_hier.py:83-85)."""
        _req(y_ids, torch.int32, "y_ids")
        n, L = y_ids.shape
        out = torch.empty((n, self.E), dtype=torch.float32, device=self.device)
        self._ck(self.lib.hiertcn_session_pool(self._h, _ptr(y_ids), n, L, _ptr(out), self._stream()))
        return out

    def embed_concat(self, x_ids: torch.Tensor, state: torch.Tensor) -> torch.Tensor:
        """x_ids [n_sessions, L] int32, state [n_sessions, H*nl] fp32 -> rows [n_sessions*L, E + H*nl] (storage type): E[x_t] | state
        (model_hier.py:50,54-55 as rows: the input of a low level without an 'emb' projection of its own)."""
        _req(x_ids, torch.int32, "x_ids")
        n, L = x_ids.shape
        _req(state, torch.float32, "state", (n, self.HT))
        rows = torch.empty((n * L, self.E + self.HT), dtype=self.act_dtype, device=self.device)
        self._ck(self.lib.hiertcn_embed_concat(self._h, _ptr(x_ids), _ptr(state), n, L, _ptr(rows), self._stream()))
        return rows

    def gru_scan(self, pooled: torch.Tensor, mask: torch.Tensor, state_in: torch.Tensor, want_state_before=False):
        """pooled [B,S,E], mask [B,S], state_in [B,H*nl] -> (sproj [B,S,E], state_out, state_before|None)."""
        B, S, _ = pooled.shape
        _req(pooled, torch.float32, "pooled", (B, S, self.E))
        _req(mask, torch.float32, "mask", (B, S))
        _req(state_in, torch.float32, "state_in", (B, self.HT))
        sproj = torch.empty((B, S, self.E), dtype=torch.float32, device=self.device)
        state_out = torch.empty((B, self.HT), dtype=torch.float32, device=self.device)
        sb = torch.empty((B, S, self.HT), dtype=torch.float32, device=self.device) if want_state_before else None
        need = int(self.lib.hiertcn_gru_workspace_bytes(self._h, B, S))
        if self._gru_ws is None or self._gru_ws.numel() < need:
            self._gru_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._ck(self.lib.hiertcn_gru_scan(self._h, _ptr(pooled), _ptr(mask), _ptr(state_in), B, S, _ptr(sproj),
                                        _ptr(state_out), _ptr(sb), _ptr(self._gru_ws), self._gru_ws.numel(), self._stream()))
        return sproj, state_out, sb

    def tcn_forward(self, x_ids: torch.Tensor, sproj: Optional[torch.Tensor]) -> torch.Tensor:
        """x_ids [n_sessions, L] int32, sproj [n_sessions, E] fp32 -> hidden [n_sessions*L, C] bf16."""
        _req(x_ids, torch.int32, "x_ids")
        n, L = x_ids.shape
        if sproj is not None:
            _req(sproj, torch.float32, "sproj", (n, self.E))
        hidden = torch.empty((n * L, self.C), dtype=self.act_dtype, device=self.device)
        need = int(self.lib.hiertcn_tcn_workspace_bytes(self._h, n, L))  # 0 when the fused kernel applies
        if need and (self._tcn_ws is None or self._tcn_ws.numel() < need):
            self._tcn_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        ws = self._tcn_ws if need else None
        self._ck(self.lib.hiertcn_tcn_forward(self._h, _ptr(x_ids), _ptr(sproj), n, L, _ptr(hidden), _ptr(ws),
                                           ws.numel() if ws is not None else 0, self._stream()))
        return hidden

    def score_fused(self, hidden: torch.Tensor, y_ids: torch.Tensor, want_target=False, topk: int = 0):
        """hidden [R, C] bf16, y_ids [R] int32 -> dict(rank int32 [R], ce fp32 [R], target?,
        topk_idx int32 [R, topk], topk_val fp32 [R, topk] when topk > 0 (loss.py:120 order))."""
        R = hidden.shape[0]
        _req(hidden, self.act_dtype, "hidden", (R, self.C))
        y = y_ids.reshape(-1)
        _req(y, torch.int32, "y_ids", (R,))
        rank = torch.empty(R, dtype=torch.int32, device=self.device)
        ce = torch.empty(R, dtype=torch.float32, device=self.device)
        tgt = torch.empty(R, dtype=torch.float32, device=self.device) if want_target else None
        ti = torch.empty((R, topk), dtype=torch.int32, device=self.device) if topk > 0 else None
        tv = torch.empty((R, topk), dtype=torch.float32, device=self.device) if topk > 0 else None
        so = _lib.ScoreOut(rank=rank.data_ptr(), ce=ce.data_ptr(), target=tgt.data_ptr() if tgt is not None else None,
                           topk_idx=ti.data_ptr() if ti is not None else None,
                           topk_val=tv.data_ptr() if tv is not None else None, topk=int(topk))
        self._ck(self.lib.hiertcn_score_fused(self._h, _ptr(hidden), _ptr(y), R, C.byref(so), self._stream()))
        out = dict(rank=rank, ce=ce)
        if tgt is not None:
            out["target"] = tgt
        if topk > 0:
            out["topk_idx"], out["topk_val"] = ti, tv
        return out

    def score_logits(self, hidden: torch.Tensor) -> torch.Tensor:
        """parity/debug: materialised logits [R, N] fp32 (model_tcn.py:41, before the model.py:105 mask)."""
        R = hidden.shape[0]
        _req(hidden, self.act_dtype, "hidden", (R, self.C))
        logits = torch.empty((R, self.N), dtype=torch.float32, device=self.device)
        self._ck(self.lib.hiertcn_score_logits(self._h, _ptr(hidden), R, _ptr(logits), self._stream()))
        return logits

    def metrics(self, rank: torch.Tensor, ce: torch.Tensor, y_ids: torch.Tensor, B: int, T: int) -> torch.Tensor:
        """-> 8 doubles on the device: lane-mean SUMS + user_count + valid steps (hiertcn.h)."""
        _req(rank, torch.int32, "rank")
        _req(ce, torch.float32, "ce")
        y = y_ids.reshape(-1)
        _req(y, torch.int32, "y_ids", (B * T,))
        out = torch.empty(8, dtype=torch.float64, device=self.device)
        self._ck(self.lib.hiertcn_metrics(self._h, _ptr(rank), _ptr(ce), _ptr(y), B, T, _ptr(out), self._stream()))
        return out

    # ---------------------------------------------------------------------------------------
    def workspace(self, B: int, S: int, L: int) -> torch.Tensor:
        need = int(self.lib.hiertcn_workspace_bytes(self._h, B, S, L))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def set_score_ids(self, y_score: Optional[torch.Tensor]):
        """hiertcn_set_score_ids: the NEXT evaluation / training step of this handle scores against y_score [B, S, L] int32 (= y_ids with the steps
        that do not count set to 0: model.py:102-103, mask_y *= mask_warmstart) while the session mean-pool keeps reading that call's y_ids."""
        if y_score is not None:
            _req(y_score, torch.int32, "y_score")
        self._score_keep = y_score           # (alive until the step that consumes it has been enqueued and the next one replaces it)
        self._ck(self.lib.hiertcn_set_score_ids(self._h, _ptr(y_score)))

    def _armed(self, y_score, call):
        """Runs the C call of a step with the one-shot score ids armed IMMEDIATELY before it -- behind every argument check and workspace
        allocation of the Python side -- and disarmed again if anything raises, so that an exception never leaves ids behind for an unrelated
        later step (the library consumes them in the call itself, successful or not; a failed hiertcn_train_step_begin drops them too)."""
        if y_score is None:
            return call()
        self.set_score_ids(y_score)
        try:
            return call()
        except BaseException:
            try:
                self.set_score_ids(None)
            except Exception:  # noqa: BLE001  (the original error is the one to report)
                pass
            raise

    def forward_eval(self, x_ids, y_ids, mask, state_in, want_rows=False, state_out=None, metrics8=None, topk: int = 0, y_score=None):
        """One evaluation step on a packed batch: x_ids, y_ids [B,S,L] int32; mask [B,S] fp32;
        state_in [B,H*nl] fp32.  Returns dict(state, metrics8 (device, 8 doubles)[, rank, ce][, topk_idx, topk_val]).
        y_score (optional, [B,S,L] int32): warm-start evaluation -- see set_score_ids."""
        B, S, L = x_ids.shape
        if y_score is not None:
            _req(y_score, torch.int32, "y_score", (B, S, L))
        _req(x_ids, torch.int32, "x_ids", (B, S, L))
        _req(y_ids, torch.int32, "y_ids", (B, S, L))
        if self.hier:
            _req(mask, torch.float32, "mask", (B, S))
            _req(state_in, torch.float32, "state_in", (B, self.HT))
        else:  # single-level model_tcn: one sequence per lane, no carried state (run_nn_xing.py:36-41)
            if S != 1:
                raise HierTCNError("single-level model: x_ids is [B, 1, T]")
            mask = state_in = None
        ws = self.workspace(B, S, L)
        if state_out is None and self.hier:
            state_out = torch.empty((B, self.HT), dtype=torch.float32, device=self.device)
        if metrics8 is None:
            metrics8 = torch.empty(8, dtype=torch.float64, device=self.device)
        so = None
        out = dict(state=state_out, metrics8=metrics8)
        if want_rows or topk > 0:
            rank = torch.empty(B * S * L, dtype=torch.int32, device=self.device)
            ce = torch.empty(B * S * L, dtype=torch.float32, device=self.device)
            ti = torch.empty((B * S * L, topk), dtype=torch.int32, device=self.device) if topk > 0 else None
            tv = torch.empty((B * S * L, topk), dtype=torch.float32, device=self.device) if topk > 0 else None
            so = _lib.ScoreOut(rank=rank.data_ptr(), ce=ce.data_ptr(), target=None, topk_idx=ti.data_ptr() if topk > 0 else None,
                               topk_val=tv.data_ptr() if topk > 0 else None, topk=int(topk))
            out["rank"] = rank.view(B, S * L)
            out["ce"] = ce.view(B, S * L)
            if topk > 0:
                out["topk_idx"] = ti.view(B, S * L, topk)
                out["topk_val"] = tv.view(B, S * L, topk)
        self._armed(y_score, lambda: self._ck(self.lib.hiertcn_forward_eval(
            self._h, _ptr(x_ids), _ptr(y_ids), _ptr(mask), _ptr(state_in), B, S, L, _ptr(ws), ws.numel(), _ptr(state_out), _ptr(metrics8),
            C.byref(so) if so is not None else None, self._stream())))
        return out

    # -- hipGraph replay of the evaluation step (launch-bound at the reference's batch sizes) ------------------------
    def capture_eval_graph(self, B: int, S: int, L: int):
        """Captures hiertcn_forward_eval for one batch shape into a hipGraph (torch.cuda.CUDAGraph on ROCm) and returns
        an EvalGraph: copy ids / mask / state into its static buffers (or write them there directly), `replay()`,
        read `state_out` / `metrics8`.  A 32-lane XING batch is ~60 dependent launches; replay removes their host cost."""
        return EvalGraph(self, B, S, L)

    # -- training (hiertcn.h "training step") ------------------------------------------------------
    def train_forward_backward(self, x_ids, y_ids, mask=None, state_in=None, state_out=None, metrics8=None, y_score=None):
        """Forward + loss + gradients of every variable (into the handle's gradient arena).
        Returns dict(state, metrics8).  Single-level handles: x_ids / y_ids [B, 1, T], no mask / state.  y_score: see set_score_ids."""
        B, S, L = x_ids.shape
        if y_score is not None:
            _req(y_score, torch.int32, "y_score", (B, S, L))
        _req(x_ids, torch.int32, "x_ids", (B, S, L))
        _req(y_ids, torch.int32, "y_ids", (B, S, L))
        if self.hier:
            _req(mask, torch.float32, "mask", (B, S))
            _req(state_in, torch.float32, "state_in", (B, self.HT))
            if state_out is None:
                state_out = torch.empty((B, self.HT), dtype=torch.float32, device=self.device)
        else:
            mask = state_in = state_out = None
        need = int(self.lib.hiertcn_train_workspace_bytes(self._h, B, S, L))
        if self._train_ws is None or self._train_ws.numel() < need:
            self._train_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if metrics8 is None:
            metrics8 = torch.empty(8, dtype=torch.float64, device=self.device)
        self._armed(y_score, lambda: self._ck(self.lib.hiertcn_train_forward_backward(
            self._h, _ptr(x_ids), _ptr(y_ids), _ptr(mask), _ptr(state_in), B, S, L, _ptr(self._train_ws), self._train_ws.numel(), _ptr(state_out),
            _ptr(metrics8), self._stream())))
        return dict(state=state_out, metrics8=metrics8)

    # -- the step in two halves + helpers for data-parallel runs (hiertcn.h "data-parallel training") ---------------------
    def _train_ws_for(self, B, S, L) -> torch.Tensor:
        need = int(self.lib.hiertcn_train_workspace_bytes(self._h, B, S, L))
        if self._train_ws is None or self._train_ws.numel() < need:
            self._train_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._train_ws

    def train_step_begin(self, x_ids, y_ids, mask=None, state_in=None, state_out=None, metrics8=None, users_global=None, y_score=None):
        """Forward + loss + backward of the scoring layer; behind it the gradients of out_kernel / out_bias are final.
        users_global: device float32 [1], the lanes-with-a-target count of the GLOBAL batch (or None: this batch's own).
        y_score: see set_score_ids (it holds for the matching train_step_finish too)."""
        B, S, L = x_ids.shape
        if y_score is not None:
            _req(y_score, torch.int32, "y_score", (B, S, L))
        _req(x_ids, torch.int32, "x_ids", (B, S, L))
        _req(y_ids, torch.int32, "y_ids", (B, S, L))
        if self.hier:
            _req(mask, torch.float32, "mask", (B, S))
            _req(state_in, torch.float32, "state_in", (B, self.HT))
            if state_out is None:
                state_out = torch.empty((B, self.HT), dtype=torch.float32, device=self.device)
        else:
            mask = state_in = state_out = None
        if users_global is not None:
            _req(users_global, torch.float32, "users_global", (1,))
        ws = self._train_ws_for(B, S, L)
        if metrics8 is None:
            metrics8 = torch.empty(8, dtype=torch.float64, device=self.device)
        self._armed(y_score, lambda: self._ck(self.lib.hiertcn_train_step_begin(
            self._h, _ptr(x_ids), _ptr(y_ids), _ptr(mask), _ptr(state_in), B, S, L, _ptr(ws), ws.numel(), _ptr(state_out), _ptr(metrics8),
            _ptr(users_global), self._stream())))   # (on success the ids stay armed for the matching train_step_finish)
        return dict(state=state_out, metrics8=metrics8)

    def train_step_finish(self, x_ids, y_ids, mask=None):
        B, S, L = x_ids.shape
        try:
            ws = self._train_ws_for(B, S, L)
            self._ck(self.lib.hiertcn_train_step_finish(self._h, _ptr(x_ids), _ptr(y_ids), _ptr(mask if self.hier else None), B, S, L,
                                                     _ptr(ws), ws.numel(), self._stream()))
        except BaseException:
            self.lib.hiertcn_set_score_ids(self._h, None)   # whatever the matching begin armed must not reach an unrelated later step
            raise

    def count_users(self, y_ids: torch.Tensor) -> torch.Tensor:
        """-> device float32 [1]: lanes of y [B, ...] with at least one target (model.py:113)."""
        B = y_ids.shape[0]
        y = y_ids.reshape(B, -1)
        _req(y, torch.int32, "y_ids")
        out = torch.empty(1, dtype=torch.float32, device=self.device)
        self._ck(self.lib.hiertcn_count_users(self._h, _ptr(y), B, y.shape[1], _ptr(out), self._stream()))
        return out

    def grad_segment(self, which: int):
        """(offset, count) in floats of part `which` (0 item table, 1 scoring kernel + bias, 2 the rest) of the flat arenas."""
        off, n = C.c_size_t(), C.c_size_t()
        self._ck(self.lib.hiertcn_train_segment(self._h, int(which), C.byref(off), C.byref(n)))
        return int(off.value), int(n.value)

    def grad_rows_list(self, x_ids, y_ids, cap=None):
        """Ascending distinct ids of the batch -> (ids int32 [cap], count int32 [1]) on the device; cap >= the most rows the batch
        can touch (known on the host: no read-back)."""
        n_ids = int(x_ids.numel())
        cap = max(int(cap or 0), min(self.N - 1, n_ids * (2 if y_ids is not None else 1)))
        ids = torch.zeros(max(cap, self.N), dtype=torch.int32, device=self.device)   # (the compaction writes up to N - 1 entries)
        cnt = torch.zeros(1, dtype=torch.int32, device=self.device)
        need = int(self.lib.hiertcn_grad_rows_workspace_bytes(self._h))
        if getattr(self, "_rows_ws", None) is None or self._rows_ws.numel() < need:
            self._rows_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._ck(self.lib.hiertcn_grad_rows_list(self._h, _ptr(x_ids), _ptr(y_ids), n_ids, _ptr(ids), ids.numel(), _ptr(cnt),
                                              _ptr(self._rows_ws), self._rows_ws.numel(), self._stream()))
        return ids[:cap], cnt, cap

    def grad_rows_op(self, op: int, ids, count, rows=None):
        cap = int(ids.numel())
        self._ck(self.lib.hiertcn_grad_rows_op(self._h, int(op), _ptr(ids), _ptr(count), cap, _ptr(rows), self._stream()))

    def grad_arena(self) -> torch.Tensor:
        """The flat fp32 gradient buffer as a torch view (no copy): the all-reduce target of data-parallel training."""
        ptr = C.c_void_p()
        n = C.c_size_t()
        self._ck(self.lib.hiertcn_grad_arena(self._h, C.byref(ptr), C.byref(n)))

        class _Raw:
            __cuda_array_interface__ = {"shape": (int(n.value),), "typestr": "<f4", "data": (int(ptr.value), False), "version": 2}

        return torch.as_tensor(_Raw(), device=self.device)

    def _arena(self, which: int) -> torch.Tensor:
        ptr, n = C.c_void_p(), C.c_size_t()
        self._ck(self.lib.hiertcn_train_arena(self._h, which, C.byref(ptr), C.byref(n)))

        class _Raw:
            __cuda_array_interface__ = {"shape": (int(n.value),), "typestr": "<f4", "data": (int(ptr.value), False), "version": 2}

        return torch.as_tensor(_Raw(), device=self.device)

    def optimizer_state(self) -> Dict[str, np.ndarray]:
        """Adam slots (flat, internal layout: valid for this model config only) + time step, for checkpoints."""
        t = C.c_int64(0)
        self._ck(self.lib.hiertcn_adam_time(self._h, C.byref(t), None))
        torch.cuda.current_stream().synchronize()
        return {"adam_m": self._arena(2).cpu().numpy(), "adam_v": self._arena(3).cpu().numpy(), "adam_t": np.array([t.value], np.int64)}

    def load_optimizer_state(self, state: Dict[str, np.ndarray]):
        m, v = self._arena(2), self._arena(3)
        if state["adam_m"].size != m.numel():
            raise HierTCNError("optimizer state belongs to a different model configuration")
        m.copy_(torch.from_numpy(np.ascontiguousarray(state["adam_m"], dtype=np.float32)).to(self.device))
        v.copy_(torch.from_numpy(np.ascontiguousarray(state["adam_v"], dtype=np.float32)).to(self.device))
        t = C.c_int64(int(np.asarray(state["adam_t"]).reshape(-1)[0]))
        self._ck(self.lib.hiertcn_adam_time(self._h, None, C.byref(t)))

    def adam_step(self, lr: float, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8, grad_scale: float = 1.0):
        """tf.train.AdamOptimizer(lr) defaults (model.py:139)."""
        self._ck(self.lib.hiertcn_adam_step(self._h, float(lr), float(beta1), float(beta2), float(eps), float(grad_scale), self._stream()))

    def export(self, which: str = "weights", root: Optional[str] = None) -> Dict[str, np.ndarray]:
        """Masters ('weights') or gradients ('grads') as {TF variable name: fp32 ndarray in the reference layout}."""
        from .params import param_shapes
        kind = self.kind
        root = root or kind
        shapes = param_shapes(self._args_for_shapes, kind)
        bufs = {k: torch.zeros(v, dtype=torch.float32, device=self.device) for k, v in shapes.items()}
        w = _lib.Weights()
        tcn = ("hier/gru" if self.low_gru else "hier/tcn") if self.hier else kind
        g = lambda k: bufs[k].data_ptr()  # noqa: E731
        w.emb_kernel = g(f"{kind}/emb/kernel")
        if self.hier:
            w.emb_bias, w.proj_kernel = g("hier/emb/bias"), g(f"{tcn}/emb/kernel")
        for i in range(0 if self.low_gru else self.cfg.n_blocks):
            pre = f"{tcn}/temporal_conv_net/tblock_{i}"
            w.conv_kernel[i], w.conv_bias[i] = g(f"{pre}/conv1/kernel"), g(f"{pre}/conv1/bias")
            if f"{pre}/dense/kernel" in bufs:
                w.res_kernel[i], w.res_bias[i] = g(f"{pre}/dense/kernel"), g(f"{pre}/dense/bias")
        w.out_kernel, w.out_bias = g(f"{tcn}/dense/kernel"), g(f"{tcn}/dense/bias")
        for l in range(self.nl):
            if self.hier:
                pre = f"hier/multi_rnn_cell/cell_{l}/gru_cell"
                w.gru_gate_kernel[l], w.gru_gate_bias[l] = g(f"{pre}/gates/kernel"), g(f"{pre}/gates/bias")
                w.gru_cand_kernel[l], w.gru_cand_bias[l] = g(f"{pre}/candidate/kernel"), g(f"{pre}/candidate/bias")
            if self.low_gru:
                lo = f"{tcn}/rnn/multi_rnn_cell/cell_{l}/gru_cell"
                w.low_gate_kernel[l], w.low_gate_bias[l] = g(f"{lo}/gates/kernel"), g(f"{lo}/gates/bias")
                w.low_cand_kernel[l], w.low_cand_bias[l] = g(f"{lo}/candidate/kernel"), g(f"{lo}/candidate/bias")
        self._ck(self.lib.hiertcn_export_weights(self._h, C.byref(w), 0 if which == "weights" else 1, self._stream()))
        torch.cuda.current_stream().synchronize()
        return {(root + k[len(kind):] if k.startswith(kind) else k): v.cpu().numpy() for k, v in bufs.items()}

    # ---------------------------------------------------------------------------------------
    def timing_enable(self, on: bool = True):
        self._ck(self.lib.hiertcn_timing_enable(self._h, 1 if on else 0))

    def health_probe(self):
        """Asynchronous read of the persistent launches' health word (hiertcn_persistent_health_async): enqueues an 8-byte device-to-host copy
        into pinned memory on the current stream and returns a probe; probe.done() says whether it has landed (no wait), probe.timed_out()
        waits for it and says whether a bounded wait ran out in a launch enqueued before the probe.  The run loops take one where the
        reference writes results.txt, every 20 batches (run_hier_xing.py)."""
        return HealthProbe(self)

    def persistent_timed_out(self) -> bool:
        """True if a bounded wait inside a persistent launch of this handle ever ran out (include/hiertcn.h); synchronises."""
        f = C.c_int32(0)
        self._ck(self.lib.hiertcn_persistent_health(self._h, C.byref(f)))
        return bool(f.value)

    def timing_read(self, reset: bool = True):
        ms = C.c_double(0.0)
        n = C.c_int64(0)
        self._ck(self.lib.hiertcn_timing_read(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value


class HealthProbe:
    def __init__(self, eng: "Engine"):
        self.word = torch.zeros(1, dtype=torch.int64).pin_memory()
        self.event = torch.cuda.Event()
        with torch.cuda.device(eng.device):
            eng._ck(eng.lib.hiertcn_persistent_health_async(eng._h, self.word.data_ptr(), eng._stream()))
            self.event.record()

    def done(self) -> bool:
        return self.event.query()

    def timed_out(self) -> bool:
        self.event.synchronize()
        return bool(int(self.word[0]) != 0)


class EvalGraph:
    """Static-shape evaluation step as a replayable graph.  Buffers: x, y int32 [B,S,L]; mask fp32 [B,S]; state_in /
    state_out fp32 [B, H*nl]; metrics8 fp64 [8].  `chain_state()` copies state_out into state_in on the device (the carried
    state of run_hier_xing.py:132,147 never visits the host)."""

    def __init__(self, eng: "Engine", B: int, S: int, L: int):
        self.eng, self.shape = eng, (B, S, L)
        dev = eng.device
        self.x = torch.zeros((B, S, L), dtype=torch.int32, device=dev)
        self.y = torch.zeros((B, S, L), dtype=torch.int32, device=dev)
        self.mask = torch.zeros((B, S), dtype=torch.float32, device=dev)
        self.state_in = torch.zeros((B, eng.HT), dtype=torch.float32, device=dev)
        self.state_out = torch.zeros((B, eng.HT), dtype=torch.float32, device=dev)
        self.metrics8 = torch.zeros(8, dtype=torch.float64, device=dev)
        eng.workspace(B, S, L)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):   # warm-up outside the capture: lazy module / attribute initialisation happens here
            for _ in range(2):
                eng.forward_eval(self.x, self.y, self.mask, self.state_in, state_out=self.state_out, metrics8=self.metrics8)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            eng.forward_eval(self.x, self.y, self.mask, self.state_in, state_out=self.state_out, metrics8=self.metrics8)

    def replay(self):
        self.graph.replay()

    def chain_state(self):
        self.state_in.copy_(self.state_out, non_blocking=True)


def metrics_dict(m8) -> Dict[str, float]:
    """8 doubles (host or device) -> the reference's batch metrics (model.py:117, loss.py:215-219)."""
    v = m8.detach().cpu().numpy() if isinstance(m8, torch.Tensor) else np.asarray(m8)
    users = v[6]
    d = {k: float(v[i] / users) if users > 0 else float("nan") for i, k in enumerate(METRIC_KEYS[:6])}
    d["user_count"] = float(users)
    d["valid_steps"] = float(v[7])
    return d
