"""The contract between the run loop and the model (what /root/reference/model.py is to
run_hier_xing.py): parameter scopes, the lazy `pred` handle, batch packing and one evaluation step.

The reference builds one TF graph at import time (model.py:19-182) and the run loops fetch module
globals by name.  Here a `HierTCNModel` owns the same things explicitly: the variables of a scope
(TF names), an Engine (HIP kernels behind the C-ABI) and `eval_step`, which returns
the fetch list of run_hier_xing.py:145-149.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from .engine import Engine, metrics_dict
from .params import init_params

_SCOPES: Dict[str, object] = {}


def _to_ids(a) -> np.ndarray:
    """loader arrays are float64 holding integer ids (data_loader.py:246); model.py:50-51 feeds
    them to int32 placeholders."""
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.asarray(a).astype(np.int32)


def pack_slots(x_list: Sequence, y_list: Sequence, mask_list: Sequence):
    """S arrays [B, L_i] (ragged over slots) -> ids [B, S, Lmax] int32 zero padded, mask [B, S].
    Zero padding is exactly the loader's own right-padding (data_loader.py:246-252): id 0 embeds to
    zero and is masked out of loss/metrics (model.py:59-62)."""
    S = len(x_list)
    xs = [_to_ids(v) for v in x_list]
    ys = [_to_ids(v) for v in y_list]
    B = xs[0].shape[0]
    lens = [v.shape[1] for v in xs]
    Lmax = max(lens)
    x = np.zeros((B, S, Lmax), np.int32)
    y = np.zeros((B, S, Lmax), np.int32)
    for i in range(S):
        x[:, i, :lens[i]] = xs[i]
        y[:, i, :lens[i]] = ys[i]
    mask = np.concatenate([np.asarray(m.detach().cpu() if isinstance(m, torch.Tensor) else m, dtype=np.float32).reshape(B, 1)
                           for m in mask_list], axis=1)
    return x, y, mask, lens


def warm_score_ids(y: np.ndarray, lens, mask_warmstart):
    """y [B, S, Lmax] (pack_slots) and the warm-start feed [B, sum L_i] in the concat order of y_id (run_hier_xing.py:116,125-129) -> the ids the
    loss and the metrics see: y where the mask is set, 0 elsewhere (model.py:102-103).  None in, None out."""
    if mask_warmstart is None:
        return None
    mw = np.asarray(mask_warmstart) > 0
    B, S, Lmax = y.shape
    if mw.shape != (B, int(sum(lens))):
        raise ValueError(f"mask_warmstart {mw.shape}: expected {(B, int(sum(lens)))} (one entry per step of y_id)")
    out = np.zeros_like(y)
    off = 0
    for i, l in enumerate(lens):
        out[:, i, :l] = y[:, i, :l] * mw[:, off:off + l]
        off += l
    return out


class PendingEval:
    """One evaluation batch in flight (HierTCNModel.eval_step_packed_async): result() waits for ITS read-backs only and builds the dict
    eval_step_packed returns (without "state": the carried state stays on the device until eval_end)."""

    def __init__(self, rank_h, m8_h, event, y, lens, n_items):
        self.rank_h, self.m8_h, self.event, self.y, self.lens, self.N = rank_h, m8_h, event, y, lens, n_items

    def result(self):
        self.event.synchronize()
        y = self.y
        B, S, Lmax = y.shape
        m = metrics_dict(self.m8_h)
        cols = np.concatenate([i * Lmax + np.arange(l) for i, l in enumerate(self.lens)])
        rank = self.rank_h.numpy()[:, cols].astype(np.float32)
        mask_y = (y.reshape(B, S * Lmax)[:, cols] > 0).astype(np.float32)
        return dict(loss=m["loss"], ranks_float_mean_user=m["rank_frac"], ranks_float=rank / float(self.N), mask_y=mask_y,
                    reciprocal_ranks_float_mean_user=m["mrr"], recall1_mean_user=m["recall1"], recall5_mean_user=m["recall5"],
                    recall10_mean_user=m["recall10"], ranks=rank, user_count=m["user_count"])


class PendingEvalNN:
    """The single-level model's evaluation batch in flight (TcnModel.eval_step_async)."""

    def __init__(self, rank_h, m8_h, event, yi, n_items):
        self.rank_h, self.m8_h, self.event, self.yi, self.N = rank_h, m8_h, event, yi, n_items

    def result(self):
        self.event.synchronize()
        m = metrics_dict(self.m8_h)
        rank = self.rank_h.numpy().astype(np.float32)
        return dict(loss=m["loss"], ranks_float_mean_user=m["rank_frac"], ranks_float=rank / float(self.N), mask_y=(self.yi > 0).astype(np.float32),
                    reciprocal_ranks_float_mean_user=m["mrr"], recall1_mean_user=m["recall1"], recall5_mean_user=m["recall5"],
                    recall10_mean_user=m["recall10"], ranks=rank, user_count=m["user_count"])


class PendingTrain:
    def __init__(self, m8_h, event):
        self.m8_h, self.event = m8_h, event

    def result(self):
        self.event.synchronize()
        m = metrics_dict(self.m8_h)
        return dict(loss=m["loss"], user_count=m["user_count"])


class LazyLogits:
    """What `pred` is in the reference (model_hier.py:79: [B, sum L_i, N] logits) without the
    logits: the last TCN activations plus the engine that can score them.  calc_loss /
    calc_metric_fast consume it through the fused kernel; .materialize() writes real logits
    (parity / debug only: 4*N bytes per step)."""

    def __init__(self, engine: Engine, hidden: torch.Tensor, B: int, S: int, Lmax: int, lens: List[int]):
        self.engine, self.hidden = engine, hidden
        self.B, self.S, self.Lmax, self.lens = B, S, Lmax, list(lens)
        self._cols = torch.as_tensor(np.concatenate([i * Lmax + np.arange(l) for i, l in enumerate(lens)]), device=hidden.device)
        self._scored = None

    @property
    def shape(self):
        return (self.B, int(sum(self.lens)), self.engine.N)

    def unpad(self, rows: torch.Tensor) -> torch.Tensor:
        """[B*S*Lmax, ...] row-major tensor -> [B, sum L_i, ...] in the reference's concat order."""
        v = rows.view(self.B, self.S * self.Lmax, *rows.shape[1:])
        return v.index_select(1, self._cols)

    def pad_ids(self, y) -> torch.Tensor:
        """[B, sum L_i] ids (the reference's y_id = concat(y_list), run_hier_xing.py:116) -> [B*S*Lmax] int32."""
        y = torch.as_tensor(_to_ids(y), device=self.hidden.device)
        full = torch.zeros((self.B, self.S * self.Lmax), dtype=torch.int32, device=self.hidden.device)
        full.index_copy_(1, self._cols, y)
        return full.reshape(-1).contiguous()

    def scored(self, y, topk: int = 0) -> dict:
        if self._scored is None or (topk > 0 and self._scored.get("topk", 0) < topk):
            yp = self.pad_ids(y)
            out = self.engine.score_fused(self.hidden, yp, want_target=True, topk=topk)
            out["y_padded"] = yp
            out["topk"] = topk
            self._scored = out
        return self._scored

    def materialize(self) -> torch.Tensor:
        return self.unpad(self.engine.score_logits(self.hidden))


class HierTCNModel:
    def __init__(self, args, params: Optional[dict] = None, name: str = "hier", seed: int = 2, train: Optional[bool] = None):
        self.args, self.name = args, name
        self.hier = name.startswith("hier")
        self.trainable = bool(getattr(args, "train", False)) if train is None else bool(train)
        self.engine = Engine(args, hier=self.hier, train=self.trainable)
        self._params = params if params is not None else init_params(args, seed, "hier" if self.hier else "tcn")
        self.engine.set_weights(self._params, root=name)

    @property
    def params(self) -> dict:
        """TF-named fp32 variables.  A trainable model reads them back from the device masters."""
        if self.trainable:
            self._params = self.engine.export("weights", root=self.name)
        return self._params

    @params.setter
    def params(self, value: dict):
        self._params = value

    def set_parameters(self, params: dict):
        self._params = params
        self.engine.set_weights(params, root=self.name)

    # one sess.run of the evaluation fetch list (run_hier_xing.py:145-149) ------------------------
    def eval_step(self, x_list, y_list, mask_list, state_np, want_pred: bool = False, mask_warmstart=None):
        """Slot lists as the loader emits them (S arrays [B, L_i]) -> the fetch list.  mask_warmstart (optional, [B, sum L_i] in the concat order
        of y_id): the feed of model.py:40 -- `mask_y *= mask_warmstart` (:102-103): the steps it zeroes count neither in the loss nor in the metrics."""
        x, y, mask, lens = pack_slots(x_list, y_list, mask_list)
        return self.eval_step_packed(x, y, mask, lens, state_np, want_pred, y_score=warm_score_ids(y, lens, mask_warmstart))

    def eval_step_packed(self, x, y, mask, lens, state_np, want_pred: bool = False, y_score=None):
        """x, y int32 [B, S, Lmax]; mask float32 [B, S]; lens = the slot widths L_i.  Per-step
        outputs come back in the reference's concat order [B, sum L_i] (model_hier.py:79).
        y_score (optional): y with the steps a warm-start mask drops set to 0 (warm_score_ids)."""
        eng = self.engine
        B, S, Lmax = x.shape
        dev = eng.device
        xd, yd, md = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (x, y, mask))
        st = torch.as_tensor(np.asarray(state_np, dtype=np.float32)).to(dev).contiguous()
        ysd = None if y_score is None else torch.from_numpy(np.ascontiguousarray(y_score, dtype=np.int32)).to(dev)
        out = eng.forward_eval(xd, yd, md, st, want_rows=True, y_score=ysd)
        m = metrics_dict(out["metrics8"])
        cols = np.concatenate([i * Lmax + np.arange(l) for i, l in enumerate(lens)])
        rank = out["rank"].cpu().numpy()[:, cols].astype(np.float32)
        mask_y = ((y if y_score is None else y_score).reshape(B, S * Lmax)[:, cols] > 0).astype(np.float32)   # (the fetch follows model.py:103)
        res = dict(loss=m["loss"], state=out["state"].cpu().numpy(), ranks_float_mean_user=m["rank_frac"],
                   ranks_float=rank / float(eng.N), mask_y=mask_y, reciprocal_ranks_float_mean_user=m["mrr"],
                   recall1_mean_user=m["recall1"], recall5_mean_user=m["recall5"], recall10_mean_user=m["recall10"],
                   ranks=rank, user_count=m["user_count"])
        if want_pred:
            pooled = eng.session_pool(yd.view(B * S, Lmax)).view(B, S, -1)
            sproj, _, _ = eng.gru_scan(pooled, md, st)
            hid = eng.tcn_forward(xd.view(B * S, Lmax), sproj.view(B * S, -1))
            res["pred"] = LazyLogits(eng, hid, B, S, Lmax, lens).materialize().cpu().numpy()
        return res

    # ---- the same fetches with the device left running (round 4) -----------------------------------------------------------------
    # At the reference's batch (32 lanes) a device step is 0.23 ms and the host side of one sess.run-shaped call -- three uploads, the
    # carried state down and up again, three read-backs that each wait for the device, the numpy post-processing of the ranks -- is
    # another 0.3 ms, strictly in series.  The pipelined form keeps the carried state ON THE DEVICE between batches (two buffers,
    # ping-pong), reads ranks / metrics back through pinned buffers with an event, and hands out a PendingEval whose result() the
    # loop asks for one batch later: the host works on batch i - 1 while the device runs batch i.  Same numbers, same dict.
    def eval_begin(self, state_np):
        eng = self.engine
        st = torch.as_tensor(np.asarray(state_np, dtype=np.float32)).to(eng.device).contiguous()
        self._pipe_state = [st, torch.empty_like(st)]
        self._pipe_cur = 0
        self._pipe_pool = {}

    def eval_end(self):
        """-> the carried state after the last batch (numpy), as eval_step_packed's step["state"] would have been."""
        torch.cuda.current_stream().synchronize()
        st = self._pipe_state[self._pipe_cur].cpu().numpy()
        self._pipe_state = None
        return st

    def _pipe_buffers(self, B, W):
        """Pinned read-back buffers (ranks [B, W], 8 metric sums) + an event, one of three rotating sets per shape."""
        key = (B, W)
        pool = self._pipe_pool.setdefault(key, dict(i=0, bufs=[]))
        if len(pool["bufs"]) < 3:   # (at most two batches are in flight: three sets so that one is always free)
            pool["bufs"].append((torch.empty((B, W), dtype=torch.int32).pin_memory(), torch.empty(8, dtype=torch.float64).pin_memory(), torch.cuda.Event()))
            pool["i"] = len(pool["bufs"]) - 1   # (so that the rotation continues behind the set just handed out: 0, 1, 2, 0, 1, 2, ...)
            return pool["bufs"][-1]
        pool["i"] = (pool["i"] + 1) % 3
        return pool["bufs"][pool["i"]]

    def _pipe_upload(self, x, y, mask):
        """ids and mask to the device through PINNED staging and asynchronous copies.  A pageable `.to(device)` is a blocking copy queued
        behind everything the stream already holds -- three of them per batch made the host wait for the previous batch's kernels
        (42 us each, the largest host cost of the loop).  Three rotating sets: a set is reused three batches later, after the result() of
        the batch that used it has waited for an event recorded behind its copies."""
        dev = self.engine.device
        n, nm = x.size, mask.size
        key = ("up", n, nm)
        pool = self._pipe_pool.setdefault(key, dict(i=0, bufs=[]))
        if len(pool["bufs"]) < 3:
            pool["bufs"].append((torch.empty(2 * n, dtype=torch.int32).pin_memory(), torch.empty(nm, dtype=torch.float32).pin_memory(),
                                 torch.empty(2 * n, dtype=torch.int32, device=dev), torch.empty(nm, dtype=torch.float32, device=dev)))
            pool["i"] = len(pool["bufs"]) - 1
            ih, mh, idv, mdv = pool["bufs"][-1]
        else:
            pool["i"] = (pool["i"] + 1) % 3
            ih, mh, idv, mdv = pool["bufs"][pool["i"]]
        ihn = ih.numpy()
        ihn[:n] = x.reshape(-1)
        ihn[n:] = y.reshape(-1)
        mh.numpy()[:] = mask.reshape(-1)
        idv.copy_(ih, non_blocking=True)
        mdv.copy_(mh, non_blocking=True)
        return idv[:n].view(x.shape), idv[n:].view(y.shape), mdv.view(mask.shape)

    def eval_step_packed_async(self, x, y, mask, lens):
        """Enqueues one evaluation batch (state from / to the device buffers of eval_begin) and returns a PendingEval."""
        eng = self.engine
        B, S, Lmax = x.shape
        xd, yd, md = self._pipe_upload(x, y, mask)
        st_in, st_out = self._pipe_state[self._pipe_cur], self._pipe_state[1 - self._pipe_cur]
        out = eng.forward_eval(xd, yd, md, st_in, want_rows=True, state_out=st_out)
        self._pipe_cur ^= 1
        rank_h, m8_h, ev = self._pipe_buffers(B, S * Lmax)
        rank_h.copy_(out["rank"], non_blocking=True)
        m8_h.copy_(out["metrics8"], non_blocking=True)
        ev.record()
        return PendingEval(rank_h, m8_h, ev, y, lens, eng.N)

    def train_begin(self, state_np):
        self.eval_begin(state_np)

    def train_end(self):
        return self.eval_end()

    def train_step_packed_async(self, x, y, mask, lr):
        """One optimizer step with the carried state kept on the device; -> PendingTrain (result(): dict(loss, user_count))."""
        xd, yd, md = self._pipe_upload(x, y, mask)
        st_in, st_out = self._pipe_state[self._pipe_cur], self._pipe_state[1 - self._pipe_cur]
        out = self.train_step_device(xd, yd, md, st_in, lr, state_out=st_out)
        self._pipe_cur ^= 1
        _rank_h, m8_h, ev = self._pipe_buffers(1, 1)
        m8_h.copy_(out["metrics8"], non_blocking=True)
        ev.record()
        return PendingTrain(m8_h, ev)

    def train_step(self, x_list, y_list, mask_list, state_np, lr, mask_warmstart=None):
        """One sess.run([train_op, loss, state]) of run_hier_xing.py:301: forward, masked lane-mean cross
        entropy (model.py:102-117), gradients of every variable, [RCCL all-reduce of the gradient arena
        when a process group is up], TF-form Adam (model.py:139-141).  Returns dict(loss, state).
        mask_warmstart: as in eval_step (run_hier_xing.py:286-289 feeds the events at or before the warm-start time)."""
        if not self.trainable:
            raise RuntimeError("this model was created without training state (args.train = False)")
        x, y, mask, lens = pack_slots(x_list, y_list, mask_list)
        return self.train_step_packed(x, y, mask, state_np, lr, y_score=warm_score_ids(y, lens, mask_warmstart))

    def train_step_packed(self, x, y, mask, state_np, lr, y_score=None):
        eng = self.engine
        dev = eng.device
        xd, yd, md = (torch.from_numpy(np.ascontiguousarray(v)).to(dev) for v in (x, y, mask))
        st = torch.as_tensor(np.asarray(state_np, dtype=np.float32)).to(dev).contiguous()
        ysd = None if y_score is None else torch.from_numpy(np.ascontiguousarray(y_score, dtype=np.int32)).to(dev)
        out = self.train_step_device(xd, yd, md, st, lr, y_score=ysd)
        m = metrics_dict(out["metrics8"])
        return dict(loss=m["loss"], state=out["state"].cpu().numpy(), user_count=m["user_count"])

    def train_step_device(self, xd, yd, md, st, lr, state_out=None, metrics8=None, y_score=None):
        """The step on device tensors, nothing read back: [global user count ->] forward + loss + scoring-layer backward ->
        [collective of the scoring layer's gradient, overlapped with] the rest of the backward -> [the other collectives] ->
        TF-form Adam.  `metrics8` comes back summed over ranks (its loss is the global batch's, model.py:116-117)."""
        from .distributed import GradientExchange, all_reduce_metrics
        eng = self.engine
        if getattr(self, "_exchange", None) is None:
            # ids per rank and step, known on the host and the same on every rank (ranks pad to their own longest session, so the local
            # tensor's size is neither): lanes x slots x longest kept session (hier), lanes x longest kept sequence (single-level)
            a = self.args
            bound = (int(a.batch_size) * int(a.max_session_num) * int(a.max_activity_len) if self.hier
                     else int(a.batch_size) * int(getattr(a, "max_seq_len", 500)))
            self._exchange = GradientExchange(eng, ids_bound=max(bound, 1))
        ex = self._exchange
        users = ex.global_users(yd if y_score is None else y_score)      # lanes with a SCORED target (model.py:112-113 on the warm-start-masked mask_y)
        out = eng.train_step_begin(xd, yd, md, st, state_out=state_out, metrics8=metrics8, users_global=users, y_score=y_score)
        ex.after_scoring()
        eng.train_step_finish(xd, yd, md)
        ex.after_backward(xd, yd if self.hier else None)
        eng.adam_step(float(lr), grad_scale=1.0)
        all_reduce_metrics(out["metrics8"])
        return out


class TcnModel:
    """The single-level models (model.py:56-57,66-73): ids [B, T] -> embedding (<scope>/emb/kernel) -> low level -> scoring, no carried state.
    KIND 'tcn' (run with --model_type tcn): TemporalConvNet; widths that are not a multiple of 32 (BASELINE.json config 1: d = 100) are zero-padded
    on upload (params.pad_widths).  KIND 'gru' (GruModel below, --model_type gru): the GRU stack of model_gru.py:33-55."""
    KIND = "tcn"

    def _to_kind(self, params: dict) -> dict:
        K, n = self.KIND, self.name
        return {(K + k[len(n):] if k.startswith(n + "/") else k): v for k, v in params.items()}

    def __init__(self, args, params: Optional[dict] = None, name: Optional[str] = None, seed: int = 2, train: Optional[bool] = None):
        from .params import pad_widths
        K = self.KIND
        self.args, self.name, self.hier = args, (name or K), False
        self.trainable = bool(getattr(args, "train", False)) if train is None else bool(train)
        self._params = params if params is not None else init_params(args, seed, K)
        self._dev_args, dev_params = pad_widths(args, self._to_kind(self._params), K)
        if K == "gru":   # (Engine picks the single-level low level from args.model_type, as model.py:66-73 does)
            self._dev_args.model_type = "gru"
        # width-padded models (d = 100 -> 128; model_gru's hidden_dim 200 -> 256) train as they are: the padded channels stay exactly zero (params.unpad_widths)
        self._padded = (self._dev_args.emb_dim != int(getattr(args, "emb_dim", 128)) or
                        list(self._dev_args.tcn_channel) != [int(c) for c in args.tcn_channel] or
                        int(self._dev_args.hidden_dim) != int(args.hidden_dim))
        self.engine = Engine(self._dev_args, hier=False, train=self.trainable)
        self.engine.set_weights(dev_params, root=K)

    @property
    def params(self) -> dict:
        if self.trainable:
            K = self.KIND
            dev = self.engine.export("weights", root=self.name)
            if self._padded:
                from .params import unpad_widths
                cut = unpad_widths(self.args, self._to_kind(dev), K)
                dev = {(self.name + k[len(K):] if k.startswith(K + "/") else k): v for k, v in cut.items()}
            self._params = dev
        return self._params

    @params.setter
    def params(self, value: dict):
        self._params = value

    def set_parameters(self, params: dict):
        from .params import pad_widths
        self._params = params
        self.engine.set_weights(pad_widths(self.args, self._to_kind(params), self.KIND)[1], root=self.KIND)

    def eval_step(self, x, y, want_pred: bool = False, mask_warmstart=None):
        """One sess.run of evaluate_nn's fetch list (run_nn_xing.py:43-47): x, y [B, T] ids.  mask_warmstart [B, T] (model.py:40,102-103): the
        targets it zeroes are not scored -- a single-level model reads y nowhere else, so the mask is applied to the ids."""
        eng = self.engine
        xi, yi = _to_ids(x), _to_ids(y)
        if mask_warmstart is not None:
            yi = (yi * (np.asarray(mask_warmstart) > 0)).astype(yi.dtype)
        B, T = xi.shape
        xd = torch.from_numpy(np.ascontiguousarray(xi)).to(eng.device).view(B, 1, T)
        yd = torch.from_numpy(np.ascontiguousarray(yi)).to(eng.device).view(B, 1, T)
        out = eng.forward_eval(xd, yd, None, None, want_rows=True)
        m = metrics_dict(out["metrics8"])
        rank = out["rank"].cpu().numpy().astype(np.float32)
        res = dict(loss=m["loss"], ranks_float_mean_user=m["rank_frac"], ranks_float=rank / float(eng.N),
                   mask_y=(yi > 0).astype(np.float32), reciprocal_ranks_float_mean_user=m["mrr"],
                   recall1_mean_user=m["recall1"], recall5_mean_user=m["recall5"], recall10_mean_user=m["recall10"],
                   ranks=rank, user_count=m["user_count"])
        if want_pred:
            hid = eng.tcn_forward(xd.view(B, T), None)
            res["pred"] = eng.score_logits(hid).view(B, T, -1).cpu().numpy()
        return res

    def eval_step_async(self, x, y):
        """evaluate_nn's fetches with the device left running (as HierTCNModel.eval_step_packed_async: pinned staging, asynchronous copies,
        read-backs behind an event); -> a pending object whose result() is eval_step's dict."""
        eng = self.engine
        xi, yi = _to_ids(x), _to_ids(y)
        B, T = xi.shape
        if getattr(self, "_pipe_pool", None) is None:
            self._pipe_pool = {}
        dummy = np.zeros(1, np.float32)
        xd, yd, _md = HierTCNModel._pipe_upload(self, np.ascontiguousarray(xi, np.int32), np.ascontiguousarray(yi, np.int32), dummy)
        out = eng.forward_eval(xd.view(B, 1, T), yd.view(B, 1, T), None, None, want_rows=True)
        rank_h, m8_h, ev = HierTCNModel._pipe_buffers(self, B, T)
        rank_h.copy_(out["rank"].view(B, T), non_blocking=True)
        m8_h.copy_(out["metrics8"], non_blocking=True)
        ev.record()
        return PendingEvalNN(rank_h, m8_h, ev, yi, eng.N)

    def train_step(self, x, y, lr, mask_warmstart=None):
        """One sess.run([train_op, loss]) of run_nn_xing.py: forward, loss, gradients, [collectives], TF-form Adam."""
        if not self.trainable:
            raise RuntimeError("this model was created without training state (args.train = False)")
        eng = self.engine
        xi, yi = _to_ids(x), _to_ids(y)
        if mask_warmstart is not None:
            yi = (yi * (np.asarray(mask_warmstart) > 0)).astype(yi.dtype)
        B, T = xi.shape
        xd = torch.from_numpy(np.ascontiguousarray(xi)).to(eng.device).view(B, 1, T)
        yd = torch.from_numpy(np.ascontiguousarray(yi)).to(eng.device).view(B, 1, T)
        out = HierTCNModel.train_step_device(self, xd, yd, None, None, lr)
        m = metrics_dict(out["metrics8"])
        return dict(loss=m["loss"], user_count=m["user_count"])


class GruModel(TcnModel):
    """The single-level model_gru (model.py:66-67 -> model_gru.py:16-63; run with --model_type gru): ids [B, T] -> embedding (gru/emb/kernel; the
    'emb' dense of model_gru.py:46 on one-hot rows) -> MultiRNNCell of GRUCells under dynamic_rnn from the zero state, sequence_length = the count of
    non-null input steps (:51, :109-118) -> scoring dense (:57).  hidden_dim is zero-padded to the next kernel width (args.py:306-308 sets 200)."""
    KIND = "gru"


def get_scope(args, name: str):
    """tf.variable_scope(name, reuse=tf.AUTO_REUSE) (model_hier.py:28, model_tcn.py:26): the first
    use of a scope creates its variables (glorot / zeros / gate-bias 1.0), later uses share them."""
    if name not in _SCOPES:
        _SCOPES[name] = HierTCNModel(args, None, name) if name.startswith("hier") else (GruModel if name.startswith("gru") else TcnModel)(args, None, name)
    return _SCOPES[name]


def set_scope(name: str, model: HierTCNModel):
    _SCOPES[name] = model


def reset_scopes():
    for m in _SCOPES.values():
        m.engine.close()
    _SCOPES.clear()
