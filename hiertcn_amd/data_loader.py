"""Batch packers for the XING path: the loader protocol the run loops consume
(`get_batch()`, `.done`, `.viewer_table`; run_hier_xing.py:114-115,177) plus a packed form that
feeds the HIP path directly.

Stands where /root/reference/data_loader.py stands:
  Dataloader_single_level_model_xing  :31-106   pad each user's whole history to the batch max
  Dataloader_hier_model_xing          :115-272  B lane queues of sessions, S slots per batch
and reproduces their batches bit for bit (tests/test_data_loader.py against fixtures captured
from the reference loader).  The organisation is different: every user is cut into sessions ONCE,
up front, with array passes (session boundaries, truncation to max_activity_len, last-session
mask); a lane queue then holds session *numbers*, not Python lists of events, and a batch is
assembled by fancy indexing.  `get_packed_batch()` emits ids [B][S][Lmax] int32 + mask [B][S]
(the layout of include/hiertcn.h) without building the float64 per-slot arrays at all.
"""
from __future__ import annotations

import ctypes as C
import heapq
import os
from typing import List, Optional, Tuple

import numpy as np

from .data_prepare import load_xing

DATA_PATH = "dataset/interactions.csv"  # data_loader.py:54,140 (relative to the working directory)

_PACKER = None
PK_ABI_VERSION = 2      # csrc/packer.c's PK_ABI_VERSION


def active_packer() -> str:
    """'c' or 'python': which batch packer get_packed_batch runs on (reported by bench.py's run_loop leg: the C one is ~10 x faster at 32 lanes)."""
    return "c" if _packer_lib() is not None else "python"


def _packer_lib():
    """libhiertcn_packer.so (csrc/packer.c, built by hiertcn_amd.build): the lane queues and the batch gather in C.  None when the library
    has not been built or HIERTCN_PY_PACKER is set: the NumPy / heapq form below does the same, ~10 x slower at 32 lanes."""
    global _PACKER
    if os.environ.get("HIERTCN_PY_PACKER"):
        return None
    if _PACKER is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path, src = os.path.join(here, "libhiertcn_packer.so"), os.path.join(here, "csrc", "packer.c")
        if os.path.exists(src) and (not os.path.exists(path) or os.path.getmtime(src) > os.path.getmtime(path)):
            try:   # absent, or older than its source: (re)build it (one gcc call); a box without gcc keeps the NumPy packer
                from .build import build_packer
                build_packer()
            except Exception:  # noqa: BLE001
                pass
        if not os.path.exists(path):
            _PACKER = False
        else:
            lib = C.CDLL(path)
            try:
                lib.pk_abi_version.restype = C.c_int
                ver = lib.pk_abi_version()
            except AttributeError:
                ver = 1
            if ver != PK_ABI_VERSION:
                raise RuntimeError(f"{path}: packer interface {ver}, this binding expects {PK_ABI_VERSION}: rebuild it (python -m hiertcn_amd.build)")
            p64, pd, pi32, pf = C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_float)
            lib.pk_create.restype = C.c_void_p
            lib.pk_create.argtypes = [C.c_int, C.c_int, p64, p64, pd, p64, p64, p64]
            lib.pk_destroy.argtypes = [C.c_void_p]
            lib.pk_fill.argtypes = [C.c_void_p, p64, C.c_int64, p64, C.c_int]
            lib.pk_peek_lmax.restype = C.c_int64
            lib.pk_peek_lmax.argtypes = [C.c_void_p]
            lib.pk_pop.argtypes = [C.c_void_p, p64, pi32, pi32, pf, pi32, C.c_int64]
            lib.pk_queue_len.restype = C.c_int64
            lib.pk_queue_len.argtypes = [C.c_void_p, C.c_int]
            _PACKER = lib
    return _PACKER or None


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct))


def _split(viewer_table: np.ndarray, split: str, rank: int = 0, world: int = 1) -> np.ndarray:
    """data_loader.py:55-60,141-146: first 80 % of the users train, next 10 % validate, last 10 % test.  Data-parallel runs
    (world > 1) then take this rank's contiguous share of the split's users (the path shards by user)."""
    n = viewer_table.shape[0]
    if split == "train":
        vt = viewer_table[0:int(n * 0.8), :]
    elif split == "validate":
        vt = viewer_table[int(n * 0.8):int(n * 0.9), :]
    elif split == "test":
        vt = viewer_table[int(n * 0.9):, :]
    else:
        vt = viewer_table
    if world > 1:
        from .distributed import user_shard
        first, cnt = user_shard(vt.shape[0], rank, world)
        vt = vt[first:first + cnt]
    return vt


def _check_item_ids(viewer_data: np.ndarray, args):
    """load_xing re-indexes the items to 1..n_items, and n_items depends on the filters; args.item_num is a fixed default
    (args.py:47).  An id >= item_num has no table row: the device treats it as the null id, which is not what was asked for."""
    if viewer_data.shape[0] and int(viewer_data[:, 1].max()) >= int(args.item_num):
        raise ValueError("the data holds item id {} but args.item_num is {} (ids must be < item_num; row 0 is the null id): "
                         "pass --item_num {}".format(int(viewer_data[:, 1].max()), int(args.item_num), int(viewer_data[:, 1].max()) + 1))


class _UserWalk:
    """The user order of one pass over a split (data_loader.py:70-75,159-170): arange, or a fresh
    numpy global-RNG permutation per pass when args.shuffle."""

    def __init__(self, n_users: int, shuffle: bool):
        self.n, self.shuffle = n_users, shuffle
        self.refresh()

    def refresh(self):
        self.index = np.random.permutation(self.n) if self.shuffle else np.arange(self.n)
        self.index_pointer = 0


class Dataloader_single_level_model_xing:
    """get_batch() -> (x, y, info): x, y float64 [B, T], info [B, T, 5]; T = longest history in the
    batch; rows past the last user of a short final batch stay zero (data_loader.py:78-106)."""

    def __init__(self, args, type: str = "train", data: Optional[Tuple[np.ndarray, np.ndarray]] = None, rank: int = 0, world: int = 1):
        self.args = args
        self.done = False
        vt, self.viewer_data = data if data is not None else load_xing(DATA_PATH, max_activity=int(args.max_seq_len))
        _check_item_ids(self.viewer_data, args)
        self.viewer_table = _split(vt, type, rank, world)
        self._ids = self.viewer_data[:, 1].astype(np.int64)
        self._walk = _UserWalk(self.viewer_table.shape[0], bool(args.shuffle))

    def refresh(self):
        self._walk.refresh()

    def _rows(self):
        B = int(self.args.batch_size)
        w = self._walk
        users = w.index[w.index_pointer:w.index_pointer + B]
        start = self.viewer_table[users, 0].astype(np.int64)
        count = self.viewer_table[users, 1].astype(np.int64)
        T = int(count.max())
        t = np.arange(T)[None, :]
        valid = t < count[:, None]
        src = np.where(valid, start[:, None] + t, 0)
        self.done = False
        w.index_pointer += B
        if w.index_pointer >= w.index.shape[0]:
            w.refresh()
            self.done = True
        return B, T, valid, src

    def get_batch(self):
        B, T, valid, src = self._rows()
        n = valid.shape[0]
        seq = np.zeros((B, T + 1))
        seq[:n, 1:] = np.where(valid, self._ids[src], 0)
        info = np.zeros((B, T, 5))
        info[:n] = np.where(valid[..., None], self.viewer_data[src], 0.0)
        return seq[:, :-1], seq[:, 1:], info

    def get_packed_batch(self):
        """-> x, y int32 [B, T] (the single-level model's placeholders, model.py:42-43)."""
        B, T, valid, src = self._rows()
        n = valid.shape[0]
        y = np.zeros((B, T), np.int32)
        y[:n] = np.where(valid, self._ids[src], 0)
        x = np.zeros_like(y)
        x[:, 1:] = y[:, :-1]
        return x, y


class Dataloader_hier_model_xing:
    """B lane queues of sessions (data_loader.py:115-272).

    A user's sessions go to the currently shortest lane (first minimum, :203); users keep being
    enqueued -- at least one per batch -- until every lane holds MORE than S sessions (:229); a
    batch pops one session per lane for each of the S slots (:233-265).  `done` reports that the
    user index wrapped while filling (:214-216, :227), not that the queues drained."""

    def __init__(self, args, type: str = "train", data: Optional[Tuple[np.ndarray, np.ndarray]] = None, rank: int = 0, world: int = 1):
        self.args = args
        self.done = False
        vt, self.viewer_data = data if data is not None else load_xing(DATA_PATH, max_activity=int(args.max_seq_len))
        _check_item_ids(self.viewer_data, args)
        self.viewer_table = _split(vt, type, rank, world)
        self._ids = self.viewer_data[:, 1].astype(np.int64)
        self._cut_sessions()
        self._walk = _UserWalk(self.viewer_table.shape[0], bool(args.shuffle))
        B = int(args.batch_size)
        self._lanes: List[list] = [[] for _ in range(B)]   # session numbers queued per lane; _heads = first unread entry
        self._heads = [0] * B
        self._queue_len = np.zeros(B)
        # (sessions queued, lane) ordered like np.argmin(queue_len) picks: shortest lane, lowest index on ties.  Lengths in
        # the heap are relative to _popped (every batch takes S from every lane, which keeps the order).
        self._heap = [(0, b) for b in range(B)]
        self._popped = 0
        # the same queues in C where the packer library is built (csrc/packer.c); the arrays it reads stay referenced here
        self._pk, self._pk_lib = None, _packer_lib()
        if self._pk_lib is not None:
            self._pk_arrays = [np.ascontiguousarray(a, dtype=t) for a, t in (
                (self._s_begin, np.int64), (self._s_len, np.int64), (self._s_mask, np.float64), (self._ids, np.int64), (self._u_first, np.int64),
                (self._u_nsess, np.int64))]
            a = self._pk_arrays
            self._pk = C.c_void_p(self._pk_lib.pk_create(B, int(args.max_session_num), _p(a[0], C.c_int64), _p(a[1], C.c_int64), _p(a[2], C.c_double),
                                                         _p(a[3], C.c_int64), _p(a[4], C.c_int64), _p(a[5], C.c_int64)))
            if not self._pk.value:
                raise MemoryError("pk_create")
            self._walk_c = None

    def __del__(self):  # pragma: no cover
        for pre in list(getattr(self, "_prefetchers", ())):   # no background thread inside pk_pop when the queues are freed
            pre.close()
        if getattr(self, "_pk", None) is not None and self._pk.value:
            self._pk_lib.pk_destroy(self._pk)
            self._pk = None

    @property
    def queue_len(self) -> np.ndarray:
        """Sessions queued per lane (data_loader.py:150,203)."""
        if self._pk is not None:
            return np.array([self._pk_lib.pk_queue_len(self._pk, b) for b in range(len(self._lanes))], dtype=np.float64)
        return self._queue_len

    def _fill_native(self) -> bool:
        """enqueue_loop in C: at least one user, then until every lane holds more than S sessions; the walk (and its refresh at the end of a
        pass, which is what `done` reports) stays here."""
        w = self._walk
        stop, force = False, 1
        while True:
            if self._walk_c is None or self._walk_c[0] is not w.index:
                self._walk_c = (w.index, np.ascontiguousarray(w.index, dtype=np.int64))
            ptr = C.c_int64(int(w.index_pointer))
            rc = self._pk_lib.pk_fill(self._pk, _p(self._walk_c[1], C.c_int64), int(self._walk_c[1].shape[0]), C.byref(ptr), force)
            w.index_pointer = int(ptr.value)
            if rc < 0:
                raise MemoryError("pk_fill")
            if rc == 0:
                return stop
            w.refresh()
            stop = True
            if rc == 2:
                return stop
            force = 0

    # -- one array pass over the event table ------------------------------------------------------
    def _cut_sessions(self):
        """Session table: first event row, kept length (truncated to max_activity_len, :198) and
        mask (1 except a user's last session, :186-187) of every session; a user's sessions are the
        consecutive entries whose first row lies inside the user's [start, start + count)."""
        vd = self.viewer_data
        n = vd.shape[0]
        boundary = np.ones(n, bool)
        boundary[1:] = (vd[1:, 4] != vd[:-1, 4]) | (vd[1:, 0] != vd[:-1, 0])
        begins = np.flatnonzero(boundary)
        ends = np.append(begins[1:], n)
        last_of_user = np.ones(begins.shape[0], bool)
        last_of_user[:-1] = vd[begins[1:], 0] != vd[begins[:-1], 0]
        self._s_begin = begins
        self._s_len = np.minimum(ends - begins, int(self.args.max_activity_len))
        self._s_mask = np.where(last_of_user, 0.0, 1.0)
        start = self.viewer_table[:, 0].astype(np.int64)
        count = self.viewer_table[:, 1].astype(np.int64)
        self._u_first = np.searchsorted(begins, start, side="left")
        self._u_nsess = np.searchsorted(begins, start + count, side="left") - self._u_first

    def refresh(self, i=None):
        self._walk.refresh()

    # -- queue mechanics ---------------------------------------------------------------------------
    def enqueue(self) -> bool:
        w = self._walk
        u = int(w.index[w.index_pointer])
        first, n = int(self._u_first[u]), int(self._u_nsess[u])
        qlen, lane = heapq.heappop(self._heap)      # == np.argmin(self.queue_len), data_loader.py:203
        self._lanes[lane].extend(range(first, first + n))
        heapq.heappush(self._heap, (qlen + n, lane))
        self._queue_len[lane] += n
        w.index_pointer += 1
        if w.index_pointer >= w.index.shape[0]:
            w.refresh()
            return True
        return False

    def enqueue_loop(self) -> bool:
        if self._pk is not None:
            return self._fill_native()
        stop = False
        while True:
            stop = self.enqueue() or stop
            if self._heap[0][0] - self._popped > int(self.args.max_session_num):   # == np.amin(self.queue_len) > S (:229)
                return stop

    def _pop_slots(self) -> np.ndarray:
        """-> session numbers [B, S] of the next batch."""
        S = int(self.args.max_session_num)
        if self._pk is not None:
            out = np.empty((len(self._lanes), S), np.int64)
            self._pk_lib.pk_pop(self._pk, _p(out, C.c_int64), None, None, None, None, 0)
            return out
        out = np.array([lane[h:h + S] for lane, h in zip(self._lanes, self._heads)], dtype=np.int64)
        for b, lane in enumerate(self._lanes):
            h = self._heads[b] + S
            if h > 4096:          # drop the consumed prefix now and then
                del lane[:h]
                h = 0
            self._heads[b] = h
        self._queue_len -= S
        self._popped += S
        return out

    def dequeue(self):
        """The reference's per-slot float64 arrays (data_loader.py:233-265)."""
        sess = self._pop_slots()
        B, S = sess.shape
        x_out, y_out, mask_out, info_out = [], [], [], []
        for i in range(S):
            sid = sess[:, i]
            L = int(self._s_len[sid].max())
            t = np.arange(L)[None, :]
            valid = t < self._s_len[sid][:, None]
            src = np.where(valid, self._s_begin[sid][:, None] + t, 0)
            seq = np.zeros((B, L + 1))
            seq[:, 1:] = np.where(valid, self._ids[src], 0)
            x_out.append(seq[:, :-1])
            y_out.append(seq[:, 1:])
            info_out.append(np.where(valid[..., None], self.viewer_data[src], 0.0))
            mask_out.append(self._s_mask[sid][:, None].copy())
        return x_out, y_out, mask_out, info_out

    def get_batch(self):
        self.done = self.enqueue_loop()
        return self.dequeue()

    def get_packed_batch(self):
        """-> x, y int32 [B, S, Lmax], mask float32 [B, S], lens (per-slot max length, i.e. the L_i
        of the reference's ragged slot arrays; needed only to un-pad `pred`-shaped outputs)."""
        self.done = self.enqueue_loop()
        if self._pk is not None:
            B, S = len(self._lanes), int(self.args.max_session_num)
            Lmax = int(self._pk_lib.pk_peek_lmax(self._pk))
            x, y = np.empty((B, S, Lmax), np.int32), np.empty((B, S, Lmax), np.int32)
            mask, widths, sess = np.empty((B, S), np.float32), np.empty(S, np.int32), np.empty((B, S), np.int64)
            self._pk_lib.pk_pop(self._pk, _p(sess, C.c_int64), _p(x, C.c_int32), _p(y, C.c_int32), _p(mask, C.c_float), _p(widths, C.c_int32), Lmax)
            return x, y, mask, [int(v) for v in widths]
        sess = self._pop_slots()
        slen = self._s_len[sess]                                   # [B, S]
        Lmax = int(slen.max())
        t = np.arange(Lmax)[None, None, :]
        valid = t < slen[..., None]
        src = np.where(valid, self._s_begin[sess][..., None] + t, 0)
        y = np.where(valid, self._ids[src], 0).astype(np.int32)
        x = np.zeros_like(y)
        x[:, :, 1:] = y[:, :, :-1]
        slot_len = slen.max(axis=0)                                # the reference's per-slot widths L_i
        x *= t < slot_len[None, :, None]                           # nothing exists past a slot's own width
        mask = self._s_mask[sess].astype(np.float32)
        return x, y, mask, [int(v) for v in slot_len]


import atexit as _atexit
import weakref as _weakref

_LIVE_PREFETCHERS = _weakref.WeakSet()   # every running Prefetcher: one exit hook stops their threads before the interpreter tears down


def _close_live_prefetchers():
    for pre in list(_LIVE_PREFETCHERS):
        try:
            pre.close()
        except Exception:  # noqa: BLE001  (interpreter exit)
            pass


_atexit.register(_close_live_prefetchers)


class Prefetcher:
    """Runs a batch source (`loader.get_packed_batch`, `loader.get_batch`) on a background thread, `depth` batches ahead, so that
    packing batch i+1 overlaps the GPU work on batch i (the packers are NumPy array passes that release the GIL for most of their
    time).  Order and contents are exactly the source's; `done` travels with each batch when the source has the attribute.

        pre = Prefetcher(loader.get_packed_batch, depth=4, owner=loader)
        x, y, mask, widths = pre.get();  pre.done  # == loader.done as it was right after that batch
    """

    def __init__(self, source, depth: int = 2, owner=None):
        import queue
        import threading
        weakref = _weakref
        self._q = queue.Queue(maxsize=max(1, int(depth)))
        self._stop = threading.Event()
        self._source = source
        # the owner (a loader whose C packer the thread calls into) is held WEAKLY: a prefetcher must not keep its loader -- and the loader's
        # queues -- alive, and the loader's list of prefetchers holds this object weakly in turn (no cycle, nothing pinned for the process's life)
        self._owner = weakref.ref(owner) if owner is not None else None
        if owner is not None and getattr(source, "__self__", None) is owner:
            self._source = weakref.WeakMethod(source)          # (a bound method of the owner would hold it strongly)
        self.done = False
        self._t = threading.Thread(target=self._run, daemon=True)
        # the thread is stopped and joined before the owner's packer is destroyed (owner.__del__ -> close) and before the interpreter tears down
        # (ONE module-level atexit hook over the live prefetchers: a daemon thread inside pk_pop while pk_destroy frees the queues would read
        # freed memory)
        if owner is not None:
            try:
                owner.__dict__.setdefault("_prefetchers", weakref.WeakSet()).add(self)
            except AttributeError:
                pass
        _LIVE_PREFETCHERS.add(self)
        self._t.start()

    def _call_source(self):
        src = self._source
        if isinstance(src, _weakref.WeakMethod):
            src = src()
            if src is None:
                raise RuntimeError("Prefetcher: the loader it reads from has been collected")
        return src()

    def _run(self):
        while not self._stop.is_set():
            try:
                batch = self._call_source()
                owner = self._owner() if self._owner is not None else None
                item = (batch, bool(getattr(owner, "done", False)), None)
                del owner
            except Exception as e:  # noqa: BLE001  (re-raised in the consumer)
                item = (None, False, e)
            while not self._stop.is_set():
                try:
                    self._q.put(item, timeout=0.1)
                    break
                except Exception:  # queue.Full
                    continue
            if item[2] is not None:
                return

    def get(self):
        batch, done, err = self._q.get()
        if err is not None:
            raise err
        self.done = done
        return batch

    def close(self):
        """Stops and joins the thread, drops the batches still queued and leaves every registry (idempotent)."""
        import threading
        self._stop.set()
        if self._t.is_alive() and threading.current_thread() is not self._t:   # (the loader's last reference may die on the thread itself)
            self._t.join(timeout=5.0)
        try:
            while True:
                self._q.get_nowait()
        except Exception:  # queue.Empty
            pass
        _LIVE_PREFETCHERS.discard(self)
        owner = self._owner() if self._owner is not None else None
        if owner is not None:
            try:
                owner.__dict__.get("_prefetchers", set()).discard(self)
                if owner.__dict__.get("_prefetcher") is self:
                    owner.__dict__.pop("_prefetcher")
            except AttributeError:
                pass

    def __del__(self):  # pragma: no cover
        try:
            self._stop.set()
        except Exception:  # noqa: BLE001
            pass
