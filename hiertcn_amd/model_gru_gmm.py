"""model_gru_gmm with the reference's signature (model_gru_gmm.py:11), backed by the HIP kernels (csrc/gmm.hip trunk 1, csrc/pool_gru.hip
launch_onehot_gru) and the mixture head shared with model_tcn_gmm.

    model_gru_gmm(args, x, x_gap=None, x_impression=None, name='rnn_gmm', reuse=True) -> pred

x = ids [B, T] (the reference feeds one_hot(ids) * sign(ids), model.py:56-61).  This model has NO 'emb' projection (model_gru_gmm.py:33-48,
unlike model_gru.py:46): the MultiRNNCell of GRUCells reads the one-hot rows directly, so layer 0's kernels are [item_num + H, .] and the
input half of [x, h] . W is the row W[id] -- a gather in the GEMM epilogue.  dynamic_rnn runs from a zero state with
sequence_length = length(x) = the COUNT of non-zero input rows (= of ids > 0; model_gru_gmm.py:57-67), used as a prefix length: past it
the state is carried through and the output row is zero [TF-sem].  `pred` stands for the [B, T, M * (1 + D)] head output
(tf.contrib.layers.fully_connected, :54) without materialising it: model_tcn_gmm.GmmHead.  has_gap / has_impression / has_batchnorm
inputs are refused (off by default and never fed on XING).  Decode of the head: see model_tcn_gmm ("parity unpinned").
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from ._lib import HierTCNError
from .model_tcn_gmm import GmmHead, TcnGmmModel, _ck

_SCOPES: Dict[str, "GruGmmModel"] = {}


def init_params(args, seed: int = 2, scope: str = "rnn_gmm") -> Dict[str, np.ndarray]:
    """glorot-uniform kernels [TF-sem]; gate bias 1.0, candidate bias 0 (customed_gru_cell.py:314,330); xavier head, zero head biases."""
    rng = np.random.default_rng(seed)

    def glorot(shape, fi, fo):
        lim = np.sqrt(6.0 / (fi + fo))
        return rng.uniform(-lim, lim, size=shape).astype(np.float32)

    N, H = int(args.item_num), int(args.hidden_dim)
    p = {}
    for l in range(int(args.num_layer)):
        pre = f"{scope}/rnn/multi_rnn_cell/cell_{l}/gru_cell"
        inp = N if l == 0 else H
        p[f"{pre}/gates/kernel"] = glorot((inp + H, 2 * H), inp + H, 2 * H)
        p[f"{pre}/gates/bias"] = np.ones(2 * H, np.float32)
        p[f"{pre}/candidate/kernel"] = glorot((inp + H, H), inp + H, H)
        p[f"{pre}/candidate/bias"] = np.zeros(H, np.float32)
    n_out = int(args.num_mixture) * (1 + (1 if getattr(args, "sigma", 0.1) else 2) * int(args.output_dim))   # learned scale: M (1 + 2 D)
    p[f"{scope}/fully_connected/weights"] = glorot((H, n_out), H, n_out)
    p[f"{scope}/fully_connected/biases"] = np.zeros(n_out, np.float32)
    return p


class GruGmmModel(TcnGmmModel):
    """One hiertcn_gmm handle with the GRU trunk + its TF-named variables; the head methods are TcnGmmModel's."""

    def __init__(self, args, params: Optional[dict] = None, name: str = "rnn_gmm", seed: int = 2):
        from .engine import check_supported
        check_supported(args, loss_ok=("gmm_prob", "gmm_prob_max", "cross_entropy"))
        if not torch.cuda.is_available():
            raise HierTCNError("no GPU visible: the compute path is HIP-only (no CPU fallback)")
        self.args, self.name = args, name
        self.lib = _lib.load()
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.M, self.D, self.N = int(args.num_mixture), int(args.output_dim), int(args.item_num)
        self.C = int(args.hidden_dim)
        self.nl = int(args.num_layer)
        cfg = _lib.GmmConfig(item_num=self.N, output_dim=self.D, num_mixture=self.M, kernel_size=0, n_blocks=0,
                             sigma=float(args.sigma) if getattr(args, "sigma", 0.1) else 0.0,   # 0: learned scale (args.sigma None)
                             trunk=1, hidden_dim=self.C, num_layer=self.nl)
        cfg.table_pass = 1 if os.environ.get("HIERTCN_GMM_TABLE_PASS") else 0   # A/B runs, tests: never the Gram form (read here, once; the library reads no environment)
        self._h = C.c_void_p()
        _ck(self.lib.hiertcn_gmm_create(C.byref(self._h), C.byref(cfg)), self.lib)
        self.params = params if params is not None else init_params(args, seed, name)
        self._upload(self.params)
        self._ws = None

    def _weights_struct(self, bufs):
        """hiertcn_gmm_weights_t over {TF name: device tensor} (training: masters in / gradients, moments out -- TcnGmmModel.export / restore)."""
        w = _lib.GmmWeights()
        p = lambda k: bufs[k].data_ptr()  # noqa: E731
        for l in range(self.nl):
            pre = f"{self.name}/rnn/multi_rnn_cell/cell_{l}/gru_cell"
            w.gru_gate_kernel[l], w.gru_gate_bias[l] = p(f"{pre}/gates/kernel"), p(f"{pre}/gates/bias")
            w.gru_cand_kernel[l], w.gru_cand_bias[l] = p(f"{pre}/candidate/kernel"), p(f"{pre}/candidate/bias")
        w.head_weights, w.head_biases = p(f"{self.name}/fully_connected/weights"), p(f"{self.name}/fully_connected/biases")
        return w

    def _upload(self, params):
        keep = []

        def dev(k):
            t = torch.as_tensor(np.asarray(params[k], dtype=np.float32)).to(self.device).contiguous()
            keep.append(t)
            return t.data_ptr()

        w = _lib.GmmWeights()
        for l in range(self.nl):
            pre = f"{self.name}/rnn/multi_rnn_cell/cell_{l}/gru_cell"
            w.gru_gate_kernel[l], w.gru_gate_bias[l] = dev(f"{pre}/gates/kernel"), dev(f"{pre}/gates/bias")
            w.gru_cand_kernel[l], w.gru_cand_bias[l] = dev(f"{pre}/candidate/kernel"), dev(f"{pre}/candidate/bias")
        w.head_weights, w.head_biases = dev(f"{self.name}/fully_connected/weights"), dev(f"{self.name}/fully_connected/biases")
        _ck(self.lib.hiertcn_gmm_set_weights(self._h, C.byref(w), C.c_void_p(torch.cuda.current_stream().cuda_stream)), self.lib)
        torch.cuda.current_stream().synchronize()


def model_gru_gmm(args, x, x_gap=None, x_impression=None, name="rnn_gmm", reuse=True):
    if (getattr(args, "has_gap", False) and x_gap is not None) or (getattr(args, "has_impression", False) and x_impression is not None):
        raise NotImplementedError("gap / impression inputs (model_gru_gmm.py:27-31) are off by default and never fed on XING: not built")
    ids = np.asarray(x.detach().cpu() if isinstance(x, torch.Tensor) else x).astype(np.int32)
    if ids.ndim != 2:
        raise NotImplementedError("model_gru_gmm takes ids [B, T] (the one-hot of model.py:56 is undone)")
    if name not in _SCOPES:
        _SCOPES[name] = GruGmmModel(args, None, name)
    model = _SCOPES[name]
    B, T = ids.shape
    hidden = model.forward(torch.from_numpy(np.ascontiguousarray(ids)).to(model.device))
    return GmmHead(model, hidden, B, T)


def reset_scopes():
    for m in _SCOPES.values():
        m.close()
    _SCOPES.clear()
