"""Pieces shared by the hier and single-level run loops: the epoch accumulator behind
`results.txt`, the scalar log (TensorBoard event files + `scalars.jsonl`) in the place of tf.summary.FileWriter, and the
learning-rate schedule.

Reference: run_hier_xing.py:96-106,152-201 / run_nn_xing.py:20-27,50-95 (the six running sums, the
every-20-batches and final blocks of results.txt), run_hier_xing.py:233-235,311-314 (three
FileWriters receiving hand-built `loss_epoch`-style scalars), :265-269 (lr / 5 at epochs 60, 120).
"""
from __future__ import annotations

import json
import os
import time
from typing import Dict, Optional

import numpy as np

# (key in a step's result dict, tag written to results.txt / the event log)
SCALARS = (("loss", "loss_epoch"), ("ranks_float_mean_user", "ranks_float_epoch_mean"),
           ("reciprocal_ranks_float_mean_user", "reciprocal_ranks_float_epoch_mean"),
           ("recall1_mean_user", "recall1_epoch_mean"), ("recall5_mean_user", "recall5_epoch_mean"),
           ("recall10_mean_user", "recall10_epoch_mean"))


class EpochMeter:
    """Mean over batches of the per-batch (lane-mean) scalars -- the reference's epoch metric is a
    mean of means and therefore depends on the batching."""

    def __init__(self, results_path: Optional[str], final_header: bool):
        self.sums = {k: 0.0 for k, _ in SCALARS}
        self.batch_count = 0
        self.results_path = results_path
        self.final_header = final_header  # evaluate_hier writes a 'final' line (:198), evaluate_nn does not (:91)

    def add(self, step: Dict[str, float]):
        for k, _ in SCALARS:
            self.sums[k] += float(step[k])
        self.batch_count += 1
        if self.batch_count % 20 == 0:  # :170
            self._write(self.means(), head="batch count {}:".format(self.batch_count))

    def means(self) -> Dict[str, float]:
        n = max(self.batch_count, 1)
        return {tag: self.sums[k] / n for k, tag in SCALARS}

    def finish(self) -> Dict[str, float]:
        self._reduce_over_ranks()
        m = self.means()
        self._write(m, head="final\n" if self.final_header else "")
        return m

    def _reduce_over_ranks(self):
        """Data-parallel evaluation: every rank walked its own shard of the users; the epoch metric is the mean over ALL batches
        of all ranks (the reference's mean of per-batch means, run_hier_xing.py:181-188, with the shards' batches pooled)."""
        import torch
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
            return
        keys = [k for k, _ in SCALARS]
        t = torch.tensor([self.sums[k] for k in keys] + [float(self.batch_count)], dtype=torch.float64)
        if dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        t = t.cpu()
        for i, k in enumerate(keys):
            self.sums[k] = float(t[i])
        self.batch_count = int(round(float(t[-1])))

    def _write(self, m: Dict[str, float], head: str):
        if not self.results_path:
            return
        os.makedirs(os.path.dirname(self.results_path), exist_ok=True)
        t = [tag for _, tag in SCALARS]
        with open(self.results_path, "a") as f:
            f.write(head)
            f.write("{} {}, {} {}\n".format(t[0], m[t[0]], t[1], m[t[1]]))
            f.write("{} {}, {} {}\n".format(t[2], m[t[2]], t[3], m[t[3]]))
            f.write("{} {}, {} {}\n".format(t[4], m[t[4]], t[5], m[t[5]]))


# ---- TensorBoard event files (what tf.summary.FileWriter writes, run_hier_xing.py:233-235,311-345), without TensorFlow ------------------------------
# A file is a sequence of TFRecords -- uint64 length, masked CRC-32C of those 8 bytes, the payload, masked CRC-32C of the payload -- whose payloads
# are serialised `Event` protos: {1: wall_time double, 2: step int64, 3: file_version string | 5: Summary {1: repeated Value {1: tag, 2: simple_value float}}}.
_CRC32C_TABLE = []


def crc32c(data: bytes) -> int:
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78); crc32c(b"123456789") == 0xE3069283."""
    if not _CRC32C_TABLE:
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            _CRC32C_TABLE.append(c)
    c = 0xFFFFFFFF
    for byte in data:
        c = _CRC32C_TABLE[(c ^ byte) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _masked_crc(data: bytes) -> int:
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def _varint(n: int) -> bytes:
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b7 = n & 0x7F
        n >>= 7
        out.append(b7 | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _ld(field: int, payload: bytes) -> bytes:   # length-delimited field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def tf_event(wall_time: float, step: int = 0, scalars: Dict[str, float] = None, file_version: str = None) -> bytes:
    import struct
    ev = b"\x09" + struct.pack("<d", float(wall_time)) + b"\x10" + _varint(int(step))
    if file_version is not None:
        ev += _ld(3, file_version.encode())
    if scalars is not None:
        summary = b"".join(_ld(1, _ld(1, str(tag).encode()) + b"\x15" + struct.pack("<f", float(v))) for tag, v in scalars.items())
        ev += _ld(5, summary)
    return ev


def tf_record(payload: bytes) -> bytes:
    import struct
    head = struct.pack("<Q", len(payload))
    return head + struct.pack("<I", _masked_crc(head)) + payload + struct.pack("<I", _masked_crc(payload))


class ScalarLog:
    """One directory per split, like tf.summary.FileWriter(args.log_path + 'train/') (run_hier_xing.py:233-235): `add_summary(summary, step)` appends
    the scalars to a TensorBoard event file `events.out.tfevents.<time>.<host>` (the format above: TensorBoard reads the directory as it reads the
    reference's) and to `scalars.jsonl` as {"step", "tag", "value", "wall"} (greppable).  Creating it creates the directory, which is what makes
    `<log_path>/<prefix>/results.txt` writable in the reference."""

    def __init__(self, logdir: str):
        import socket
        self.logdir = logdir
        os.makedirs(logdir, exist_ok=True)
        self._path = os.path.join(logdir, "scalars.jsonl")
        now = time.time()
        self._events = os.path.join(logdir, "events.out.tfevents.%010d.%s" % (int(now), socket.gethostname()))
        with open(self._events, "ab") as f:
            f.write(tf_record(tf_event(now, 0, file_version="brain.Event:2")))

    def add_summary(self, summary: Dict[str, float], step: int):
        now = time.time()
        with open(self._path, "a") as f:
            for tag, value in summary.items():
                f.write(json.dumps({"step": int(step), "tag": tag, "value": float(value), "wall": now}) + "\n")
        with open(self._events, "ab") as f:
            f.write(tf_record(tf_event(now, int(step), scalars=summary)))


WARM_START_TIME = 1445617130   # run_hier_xing.py:128,288 / run_nn_xing.py:40,150: the XING timestamp that splits a user's events into "before" and "after"


def warm_start_mask(info, train: bool) -> np.ndarray:
    """The `mask_warmstart` feed of a warm-start run (args.warm_start; model.py:40,102-103).  info: the loader's per-event rows aligned with y --
    a list of S arrays [B, L_i, 5] (hier loader; concatenated along the steps as run_hier_xing.py:117 does) or one array [B, T, 5]; column 3 is
    the event's timestamp (data_prepare.load_xing).  Evaluation counts the events AFTER the time (run_hier_xing.py:128, run_nn_xing.py:40),
    training the real events at or before it (:288 / :150: `<= T` and `> 0`, padding has timestamp 0).  -> float32 [B, sum L_i]."""
    ts = (np.concatenate([np.asarray(i) for i in info], axis=1) if isinstance(info, (list, tuple)) else np.asarray(info))[:, :, 3]
    m = (ts <= WARM_START_TIME) & (ts > 0) if train else ts > WARM_START_TIME
    return m.astype(np.float32)


def scheduled_lr(args, epoch: int):
    """run_hier_xing.py:265-269 / run_nn_xing.py: divide by 5 when epoch 60 and 120 START
    (mutates args.learning_rate, as the reference does)."""
    if args.lr_schedule and epoch in (60, 120):
        args.learning_rate /= 5
    return args.learning_rate
