"""User-batch data parallelism: one process per GPU, a full weight replica per
GPU, users sharded across ranks.  Lanes never interact in the forward (model_hier.py:39-93 is
per-row of B), so the evaluation path has exactly one collective: a sum all-reduce of the 8 metric
accumulators (lane-mean SUMS + user count + valid steps, include/hiertcn.h `hiertcn_metrics`).
Backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests.
"""
from __future__ import annotations

import os
from typing import Tuple

import torch
import torch.distributed as dist


_FORCE = False   # init_from_env(force_group=True): run every collective even at world size 1 (a one-rank RCCL communicator)
_SHARED_DEVICE = False   # init_from_env found two ranks on one GPU: the persistent (co-resident) launches are switched off on every handle

# what a shared device changes in hiertcn_tuning_t (include/hiertcn.h): no launch that waits on other workgroups -- the across-session scan (its
# training form and its backward follow gru_scan_persistent) and the HierGRU low level run as launch chains
SHARED_DEVICE_TUNING = {"gru_scan_persistent": 0, "lgru_persistent": 0}


def shared_device() -> bool:
    """True once init_from_env has seen two ranks of the job on one GPU.  Engine.__init__ consults it (an Engine created AFTER the decision
    starts with SHARED_DEVICE_TUNING), and init_from_env applies the same fields to every Engine already alive, so the order of
    `Engine(...)` and `init_from_env(...)` does not matter."""
    return _SHARED_DEVICE


def _mark_shared_device():
    global _SHARED_DEVICE
    _SHARED_DEVICE = True
    try:
        from . import engine as _engine
    except Exception:  # noqa: BLE001  (no library / no GPU: there is no live handle to retune)
        return
    from ._lib import tuning_from_env
    fields = {k: v for k, v in SHARED_DEVICE_TUNING.items() if k not in tuning_from_env()}   # (an explicit HIERTCN_* switch outranks the guard, as in Engine.__init__)
    for eng in list(getattr(_engine, "_LIVE_ENGINES", ())):
        try:
            if fields:
                eng.set_tuning(**fields)
        except Exception:  # noqa: BLE001  (a closed handle)
            pass


def _active() -> bool:
    """A process group whose collectives this module should enter: more than one rank, or a group forced at world size 1."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCE)


def device_ident(device=None) -> str:
    """hostname | (uuid or pci address) of a rank's GPU: equal on two ranks exactly when they share a device.  Never raises: a runtime that
    reports neither gives a sentinel that is unique per process (the count-based answer then decides)."""
    import socket
    try:
        prop = torch.cuda.get_device_properties(device if device is not None else torch.cuda.current_device())
        dev = str(getattr(prop, "uuid", "")) or "%04x:%02x:%02x" % (getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", 0),
                                                                    getattr(prop, "pci_device_id", 0))
    except Exception:  # noqa: BLE001
        dev = "unknown-%d" % os.getpid()
    return socket.gethostname() + "|" + dev


def init_from_env(backend: str | None = None, device: torch.device | None = None, force_group: bool = False) -> Tuple[int, int, int]:
    """Reads RANK / WORLD_SIZE / LOCAL_RANK (torch.distributed.run) -> (rank, world, local_rank).
    world == 1 needs no process group; force_group=True creates one all the same (backend "nccl": a one-rank RCCL communicator) and makes
    every collective of this module run through it -- the whole multi-GPU path short of a second GPU (tests/test_gpu_rccl_one_rank.py,
    `bench.py --rccl-one-rank`)."""
    global _FORCE
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if device is not None and torch.cuda.is_available() and torch.device(device).type == "cuda":
        torch.cuda.set_device(device)   # object collectives of the nccl backend stage through the CURRENT device
    if (world > 1 or force_group) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl" and device is not None:
            kw["device_id"] = device
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    if force_group:
        _FORCE = True
    if world > 1 and torch.cuda.is_available():
        # Do ranks share a GPU (a gloo smoke run, HIP_VISIBLE_DEVICES mapping several ranks to one device, ...)?  The persistent launches (the
        # across-session scan and the HierGRU low level of small batches, csrc/gru_persist.hip) assume their workgroups are co-resident on a
        # device of their own: decided from the devices actually in use -- every rank's hostname | (uuid or pci address) -- not from the rank
        # count, and the launch chains run instead where a device is shared.  The identity cannot fail (device_ident); the collective is
        # entered by EVERY rank, outside any try: a rank that skipped it would leave the others waiting, and the answer -- hence the kernel
        # choice -- is the same on all ranks.
        idents = [None] * world
        dist.all_gather_object(idents, device_ident(device))
        shared = world > torch.cuda.device_count() * max(1, len({i.split("|")[0] for i in idents})) or len(set(idents)) < world
        if shared:
            _mark_shared_device()
    return rank, world, local_rank


def rank_world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def is_main() -> bool:
    return rank_world()[0] == 0


def barrier():
    if _active():
        dist.barrier()


def user_shard(n_users: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced shard of the user axis: (first_user, n_users_of_rank).  The first
    n_users % world ranks hold one user more."""
    base, extra = divmod(int(n_users), int(world))
    count = base + (1 if rank < extra else 0)
    first = rank * base + min(rank, extra)
    return first, count


def all_reduce_metrics(m8: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks of the 8 accumulators; afterwards out[k] / out[6] is the metric
    over the GLOBAL batch (a mean over all lanes of all ranks, model.py:116-117 / loss.py:215-219)."""
    if _active():
        if m8.is_cuda and dist.get_backend() == "gloo":   # (tests / smoke runs: gloo reduces host tensors)
            host = m8.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            m8.copy_(host)
        else:
            dist.all_reduce(m8, op=dist.ReduceOp.SUM)
    return m8


def max_over_ranks(seconds: float, device=None) -> float:
    if not _active():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device or ("cuda" if dist.get_backend() == "nccl" else "cpu"))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _is_dist() -> bool:
    import os
    if os.environ.get("HIERTCN_DP_OFF"):   # bench.py: time the step without its collectives (replicas drift: timing only)
        return False
    return _active()


class _Done:
    def wait(self):
        return None


def _sum_async(t: torch.Tensor):
    """Sum `t` over ranks in place; returns a handle with .wait().  RCCL ("nccl"): asynchronous on the backend's own stream (the
    collective starts when the work queued on the current stream so far is done; wait() makes the current stream wait for it,
    the host never blocks).  gloo (CPU tests / a smoke run with ranks sharing a GPU): staged through the host, synchronous."""
    if t.is_cuda and dist.get_backend() == "gloo":
        host = t.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM)
        t.copy_(host)
        return _Done()
    return dist.all_reduce(t, op=dist.ReduceOp.SUM, async_op=True)


def _gather_cat(t: torch.Tensor) -> torch.Tensor:
    """[n, ...] per rank -> [world, n, ...] (same shape on every rank)."""
    world = dist.get_world_size()
    if t.is_cuda and dist.get_backend() == "gloo":
        parts = [torch.empty_like(t, device="cpu") for _ in range(world)]
        dist.all_gather(parts, t.cpu())
        return torch.stack(parts).to(t.device)
    out = torch.empty((world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)   # concatenation along dim 0
    dist.all_gather_into_tensor(out, t.contiguous())
    return out.view((world,) + tuple(t.shape))


class GradientExchange:
    """The gradient collectives of one data-parallel training step (model.py:135-141 on the global batch), shaped
    for what the gradient arena holds rather than one post-hoc dense all-reduce:

      1 scoring kernel + bias  [NP, C] + [NP] fp32 (308 MB at cfg-3), DENSE: summed right behind the scoring layer's backward,
        while the TemporalBlocks / projection / GRU backward still computes (`after_scoring`).  On RCCL as reduce-scatter +
        all-gather (HIERTCN_DP_COLLECTIVE=allreduce: one all-reduce), which over the fully connected xGMI node moves
        2 (W-1)/W of the buffer per GPU over all seven links;
      0 item table [N, E] fp32 (307 MB), ROW-SPARSE: only ids of the batch have a row.  When the batch can touch fewer rows
        than a dense ring all-reduce would move (host-known bound: 2 ids per step, no read-back), the ranks all-gather
        (row id, row) pairs and every rank adds them up in rank order (bit-identical replicas); otherwise dense like 1;
      2 every other variable (a few MB): one all-reduce.

    The loss normaliser is the GLOBAL user count (all-reduced before the step: `global_users`), so the summed gradients need no
    1/world factor and ragged shards (ranks with different numbers of live lanes) are weighted exactly as model.py:113-117 would
    weight them in one big batch."""

    def __init__(self, engine, sparse_rows_max_fraction: float = None, ids_bound: int = None):
        """ids_bound: a HOST-KNOWN, RANK-INDEPENDENT upper bound on the number of ids one rank feeds per step (x_ids.numel(): e.g.
        batch_size * max_session_num * max_activity_len from args).  It sizes the one all-gather capacity every rank must agree on and
        drives the sparse-versus-dense choice.  Ranks pad to their OWN longest session (pack_slots), so a bound taken from the local
        batch would differ between ranks and steps; without ids_bound the ranks agree on the step's maximum with one small MAX
        all-reduce EVERY step -- unconditionally: a collective only some ranks enter would pair mismatched collectives or hang."""
        import os
        self.eng = engine
        self.ids_bound = None if ids_bound is None else int(ids_bound)
        self.mode = os.environ.get("HIERTCN_DP_COLLECTIVE", "rs_ag")
        self.sparse_frac = float(os.environ.get("HIERTCN_DP_SPARSE_FRACTION", "-1")) if sparse_rows_max_fraction is None else sparse_rows_max_fraction
        self._pending = []
        self._shard = None
        self.last = {}
        self.dense_taken = None      # what _dense actually ran last: 'rs_ag', 'allreduce_bucketed' or 'allreduce' (rs_ag needs numel % world == 0)
        self._over_host = None       # pinned host copy of the step's summed "some rank exceeds ids_bound" flag + the event behind its copy
        self._over_event = None
        self._over_fresh = False     # set by global_users, cleared by _ids_per_rank: a step that skipped global_users checks its own count

    @property
    def world(self) -> int:
        return dist.get_world_size() if _is_dist() else 1

    def global_users(self, y_ids: torch.Tensor):
        """Device float [1]: lanes with a target over all ranks (None at world 1: the kernel counts its own batch).  The same all-reduce
        carries one more float: 1 from every rank that feeds more ids this step than `ids_bound` allows.  Its sum reaches the host through
        a pinned buffer (no stream synchronisation) and is read in `_ids_per_rank`, where ALL ranks then raise together -- a rank that
        raised alone would leave the others waiting in the next collective."""
        if not _is_dist():
            return None
        u = self.eng.count_users(y_ids)
        over = 1.0 if (self.ids_bound is not None and int(y_ids.numel()) > self.ids_bound) else 0.0
        t = torch.cat([u, torch.full((1,), over, dtype=torch.float32, device=u.device)])
        _sum_async(t).wait()
        if self.ids_bound is not None:
            if t.is_cuda:
                if self._over_host is None:
                    self._over_host = torch.zeros(1, dtype=torch.float32).pin_memory()
                    self._over_event = torch.cuda.Event()
                self._over_host.copy_(t[1:2], non_blocking=True)
                self._over_event.record()
            else:
                self._over_host = t[1:2].clone()
            self._over_fresh = True
        return t[0:1]

    def _ids_per_rank(self, n_local: int) -> int:
        """The number every rank uses for the all-gather capacity and the sparse-versus-dense decision of this step: the same on all
        ranks by construction (the host-known bound), or agreed on by a collective that every rank enters every step."""
        if self.ids_bound is not None:
            if self._over_fresh:                     # the flag summed over ranks by THIS step's global_users: the same value everywhere
                self._over_fresh = False
                if self._over_event is not None:
                    self._over_event.synchronize()
                n_over = float(self._over_host[0])
                self._over_host[0] = 0.0
                if n_over > 0:
                    raise ValueError(f"{int(n_over)} rank(s) feed more ids in one step than ids_bound = {self.ids_bound} (this rank: {n_local}): the bound "
                                     "must hold on every rank and step (batch_size * max_session_num * max_activity_len); raised on every rank")
            elif n_local > self.ids_bound:           # (after_backward called without global_users: nothing was agreed on)
                raise ValueError(f"this rank feeds {n_local} ids in one step, more than ids_bound = {self.ids_bound}: the bound must hold on every "
                                 "rank and step (batch_size * max_session_num * max_activity_len)")
            return self.ids_bound
        t = torch.tensor([n_local], dtype=torch.int64)
        if dist.get_backend() == "nccl":
            t = t.to(self.eng.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return int(t.item())

    def _dense(self, seg: torch.Tensor):
        world = self.world
        if (self.mode == "rs_ag" and dist.get_backend() == "nccl" and seg.numel() % world == 0):
            n = seg.numel() // world
            if self._shard is None or self._shard.numel() != n:
                self._shard = torch.empty(n, dtype=seg.dtype, device=seg.device)
            # both run on the backend's stream, in issue order; neither makes the compute stream wait here
            w1 = dist.reduce_scatter_tensor(self._shard, seg, op=dist.ReduceOp.SUM, async_op=True)
            w2 = dist.all_gather_into_tensor(seg, self._shard, async_op=True)
            self.dense_taken = "rs_ag"

            class _Both:
                def wait(self_inner):
                    w1.wait()
                    w2.wait()
            return _Both()
        if self.mode == "allreduce" and dist.get_backend() == "nccl":
            self.dense_taken = "allreduce_bucketed"
            return _sum_async_bucketed(seg)
        self.dense_taken = "allreduce"   # (gloo, or rs_ag asked for a segment that does not divide by the world size)
        return _sum_async(seg)

    def after_scoring(self):
        """Call between train_step_begin and train_step_finish."""
        self._pending = []
        if not _is_dist():
            return
        off, n = self.eng.grad_segment(1)
        self._pending.append(self._dense(self.eng.grad_arena()[off:off + n]))

    def after_backward(self, x_ids: torch.Tensor, y_ids):
        """Call behind train_step_finish; returns when every gradient of the arena is the global sum (stream-ordered)."""
        if not _is_dist():
            return
        eng, world = self.eng, self.world
        g = eng.grad_arena()
        off2, n2 = eng.grad_segment(2)
        self._pending.append(_sum_async(g[off2:off2 + n2]))
        off0, n0 = eng.grad_segment(0)
        n_ids = self._ids_per_rank(int(x_ids.numel())) * (2 if y_ids is not None else 1)   # the same on every rank
        bound = min(eng.N - 1, n_ids)
        # all-gather of pairs receives (W-1) * bound rows; a ring all-reduce moves 2 (W-1)/W * N rows
        frac = self.sparse_frac if self.sparse_frac >= 0 else 2.0 / world
        sparse = bound < frac * eng.N
        self.last = dict(item_table="row-sparse" if sparse else "dense", rows_bound=bound)
        if not sparse:
            self._pending.append(_sum_async(g[off0:off0 + n0]))
        else:
            ids, cnt, cap = eng.grad_rows_list(x_ids, y_ids, cap=bound)
            rows = torch.empty((cap, eng.E), dtype=torch.float32, device=eng.device)
            eng.grad_rows_op(0, ids, cnt, rows)                 # pack own rows
            eng.grad_rows_op(1, ids, cnt)                       # and clear them: every rank rebuilds the sum in rank order
            all_ids, all_cnt, all_rows = _gather_cat(ids), _gather_cat(cnt), _gather_cat(rows)
            for r in range(world):
                eng.grad_rows_op(2, all_ids[r], all_cnt[r], all_rows[r])
        for wk in self._pending:
            wk.wait()
        self._pending = []


def _sum_async_bucketed(flat: torch.Tensor, bucket_bytes: int = 256 << 20):
    """HIERTCN_DP_COLLECTIVE=allreduce on RCCL: the dense segment as a few large all-reduces in flight at once (a bucket's ring can start
    while the next one is queued) instead of reduce-scatter + all-gather.  Returns a handle with .wait()."""
    n = max(1, bucket_bytes // flat.element_size())
    works = [dist.all_reduce(flat[i:i + n], op=dist.ReduceOp.SUM, async_op=True) for i in range(0, flat.numel(), n)]

    class _All:
        def wait(self):
            for wk in works:
                wk.wait()
    return _All()
