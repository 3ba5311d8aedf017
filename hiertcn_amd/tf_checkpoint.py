"""Reader for a TF-1.x `tf.train.Saver` checkpoint (the V2 "tensor bundle": `<prefix>.index` + `<prefix>.data-NNNNN-of-MMMMM`), without
TensorFlow -- so that a HierTCN trained with the reference (`saver.save(sess, args.save_path + str(epoch) + '.ckpt')`,
run_hier_xing.py:317-321; restored at :220-226 / model.py:146-147) can be evaluated on the MI355X path on a machine that has no TensorFlow.

    params = read_tf_checkpoint("run/model/<name>/20.ckpt")        # {variable name: fp32 ndarray}, Adam slot variables skipped
    HierTCNModel(args).set_parameters(params)                       # names are the ones SURVEY.md 8(a) lists

Format, as documented in tensorflow/core/util/tensor_bundle/ and tensorflow/core/lib/io/format.cc (a LevelDB-style table):
  index file = data blocks | metaindex block | index block | 48-byte footer (two block handles, padding, magic 0xdb4775248b80fb57);
  a block = prefix-compressed entries (varint shared, varint non_shared, varint value_len, key suffix, value) + restart array + count,
  followed in the file by a 5-byte trailer (compression type, masked crc32c); the index block maps separator keys to data-block handles;
  key "" holds the BundleHeaderProto (num_shards, endianness, version), every other key a BundleEntryProto (dtype, shape, shard_id, offset,
  size, crc32c) that says where the tensor's bytes sit in which data shard.
NOT VERIFIED AGAINST A FILE WRITTEN BY TENSORFLOW: there is none in the reference tree and TensorFlow cannot be installed here; the reader
is pinned by a writer of the same format in tests/ (round trip, hand-built known-answer bytes, crc checks) -- a real checkpoint may
still surprise it (snappy-compressed index blocks and sliced / partitioned variables are refused with a clear message).
"""
from __future__ import annotations

import os
import struct
from typing import Dict, Tuple

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
_DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 9: np.int64, 19: np.float16}   # tensorflow/core/framework/types.proto


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7


_CRC_TABLE = None


def crc32c(data: bytes) -> int:
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tab.append(c)
        _CRC_TABLE = np.array(tab, dtype=np.uint32)
    c = 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        c = int(tab[(c ^ b) & 0xFF]) ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def _gf2_times(mat, vec: int) -> int:
    out, i = 0, 0
    while vec:
        if vec & 1:
            out ^= mat[i]
        vec >>= 1
        i += 1
    return out


def _gf2_compose(a, b):
    """the operator 'apply b, then a' (both 32 x 32 over GF(2), stored as 32 column words)"""
    return [_gf2_times(a, b[n]) for n in range(32)]


def _zeros_operator(n_bytes: int):
    """The 32 x 32 GF(2) operator that advances a (finalised) CRC-32C over n_bytes zero bytes -- zlib's crc32_combine, kept as a matrix so
    that one operator serves every equally long chunk."""
    odd = [0x82F63B78] + [1 << n for n in range(31)]      # one zero bit
    even = _gf2_compose(odd, odd)                          # two
    odd = _gf2_compose(even, even)                         # four
    op = [1 << n for n in range(32)]                       # identity
    cur = odd
    while n_bytes:
        cur = _gf2_compose(cur, cur)                       # 8, 16, 32 ... zero bits: 1, 2, 4 ... bytes
        if n_bytes & 1:
            op = _gf2_compose(cur, op)
        n_bytes >>= 1
    return op


def crc32c_fast(data) -> int:
    """CRC-32C of a large buffer: the buffer is cut into up to 65,536 equally long chunks whose CRCs advance TOGETHER, one table lookup per
    byte position over the whole array of chunk states (NumPy), and the chunk CRCs are then combined with the zero-advance operator.
    Bit-identical to crc32c() (tests/test_tf_checkpoint.py); ~1 s per 300 MB instead of minutes, so the item tables are verified too."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.reshape(-1).view(np.uint8)
    n = buf.size
    if n < (1 << 16):
        return crc32c(buf.tobytes())
    crc32c(b"")                                            # builds the table
    m = min(1 << 16, n // 1024)
    lc = n // m
    cols = np.ascontiguousarray(buf[:m * lc].reshape(m, lc).T)      # [lc][m]: byte j of every chunk contiguous
    c = np.full(m, 0xFFFFFFFF, dtype=np.uint32)
    tab = _CRC_TABLE
    for j in range(lc):
        c = tab[(c ^ cols[j]) & 0xFF] ^ (c >> 8)
    c ^= np.uint32(0xFFFFFFFF)
    op = _zeros_operator(lc)
    total = int(c[0])
    for v in c[1:].tolist():
        total = _gf2_times(op, total) ^ v
    rest = buf[m * lc:]
    if rest.size:
        total = _gf2_times(_zeros_operator(int(rest.size)), total) ^ crc32c(rest.tobytes())
    return total


def masked_crc32c(data: bytes) -> int:
    c = crc32c_fast(data) if len(data) >= (1 << 16) else crc32c(data)
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def _block(buf: bytes, offset: int, size: int, verify: bool) -> bytes:
    raw = buf[offset:offset + size]
    ctype = buf[offset + size]
    if verify:
        want = struct.unpack("<I", buf[offset + size + 1:offset + size + 5])[0]
        if masked_crc32c(buf[offset:offset + size + 1]) != want:
            raise ValueError("checkpoint index: block checksum mismatch (corrupt file?)")
    if ctype == 1:
        raise NotImplementedError("checkpoint index: snappy-compressed block (not supported without the snappy library)")
    if ctype != 0:
        raise ValueError(f"checkpoint index: unknown block compression type {ctype}")
    return raw


def _entries(block: bytes):
    n_restarts = struct.unpack("<I", block[-4:])[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + block[pos:pos + non_shared]
        pos += non_shared
        yield key, block[pos:pos + vlen]
        pos += vlen


def _proto_fields(buf: bytes):
    """(field number, wire type, value) of a serialized protobuf message; nested messages come back as bytes."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = buf[pos:pos + n]
            pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"protobuf wire type {wt} not expected in a bundle entry")
        yield field, wt, v


def _entry(buf: bytes) -> dict:
    e = dict(dtype=0, shape=[], shard_id=0, offset=0, size=0, crc=None, sliced=False)
    for f, _wt, v in _proto_fields(buf):
        if f == 1:
            e["dtype"] = v
        elif f == 2:   # TensorShapeProto: repeated Dim dim = 2 { int64 size = 1 }
            for f2, _w2, v2 in _proto_fields(v):
                if f2 == 2:
                    size = 0
                    for f3, _w3, v3 in _proto_fields(v2):
                        if f3 == 1:
                            size = v3
                    e["shape"].append(int(size))
        elif f == 3:
            e["shard_id"] = v
        elif f == 4:
            e["offset"] = v
        elif f == 5:
            e["size"] = v
        elif f == 6:
            e["crc"] = struct.unpack("<I", v)[0]
        elif f == 7:
            e["sliced"] = True
    return e


def read_index(prefix: str, verify: bool = True) -> Tuple[dict, Dict[str, dict]]:
    """-> (header {num_shards, endianness, version}, {tensor name: entry})."""
    with open(prefix + ".index", "rb") as f:
        buf = f.read()
    if len(buf) < 48 or struct.unpack("<Q", buf[-8:])[0] != TABLE_MAGIC:
        raise ValueError(f"{prefix}.index is not a TensorFlow checkpoint index (bad magic)")
    footer = buf[-48:]
    _mo, p = _varint(footer, 0)
    _ms, p = _varint(footer, p)
    io, p = _varint(footer, p)
    isz, p = _varint(footer, p)
    header, entries = dict(num_shards=1, endianness=0, version=None), {}
    for _sep, handle in _entries(_block(buf, io, isz, verify)):
        off, q = _varint(handle, 0)
        size, q = _varint(handle, q)
        for key, val in _entries(_block(buf, off, size, verify)):
            if key == b"":
                for f, _wt, v in _proto_fields(val):
                    if f == 1:
                        header["num_shards"] = v
                    elif f == 2:
                        header["endianness"] = v
            else:
                entries[key.decode()] = _entry(val)
    if header["endianness"] != 0:
        raise NotImplementedError("big-endian checkpoint")
    return header, entries


def is_optimizer_slot(name: str) -> bool:
    """tf.train.AdamOptimizer's slot / accumulator variables (model.py:139): '<var>/Adam', '<var>/Adam_1', 'beta1_power', 'beta2_power'."""
    last = name.rsplit("/", 1)[-1]
    return last in ("Adam", "Adam_1", "beta1_power", "beta2_power") or name in ("beta1_power", "beta2_power")


def read_tf_checkpoint(prefix: str, with_slots: bool = False, verify: bool = True) -> Dict[str, np.ndarray]:
    """{variable name: ndarray (float tensors as float32)} of a V2 checkpoint written by tf.train.Saver.  `prefix` is what was passed to
    saver.save / saver.restore (e.g. '.../20.ckpt')."""
    header, entries = read_index(prefix, verify)
    shards: Dict[int, np.memmap] = {}
    out: Dict[str, np.ndarray] = {}
    for name, e in entries.items():
        if not with_slots and is_optimizer_slot(name):
            continue
        if e["sliced"]:
            raise NotImplementedError(f"{name}: partitioned / sliced variable (the reference has none)")
        if e["dtype"] not in _DTYPES:
            raise NotImplementedError(f"{name}: dtype enum {e['dtype']} not supported")
        sid = e["shard_id"]
        if sid not in shards:
            path = "%s.data-%05d-of-%05d" % (prefix, sid, header["num_shards"])
            if not os.path.exists(path):
                raise FileNotFoundError(path)
            shards[sid] = np.memmap(path, dtype=np.uint8, mode="r")
        if e["offset"] + e["size"] > shards[sid].size:
            raise ValueError(f"{name}: bytes [{e['offset']}, {e['offset'] + e['size']}) lie outside its data shard ({shards[sid].size} bytes): truncated checkpoint")
        raw = bytes(shards[sid][e["offset"]:e["offset"] + e["size"]])
        dt_size = np.dtype(_DTYPES[e["dtype"]]).itemsize
        if int(np.prod(e["shape"], dtype=np.int64)) * dt_size != len(raw):
            raise ValueError(f"{name}: {len(raw)} bytes do not hold a tensor of shape {e['shape']}")
        # every tensor is verified, the item tables included (crc32c_fast: chunk CRCs advanced together, then combined)
        if verify and e["crc"] is not None and masked_crc32c(raw) != e["crc"]:
            raise ValueError(f"{name}: tensor checksum mismatch")
        dt = np.dtype(_DTYPES[e["dtype"]])
        arr = np.frombuffer(raw, dtype=dt).reshape(e["shape"])
        out[name] = arr.astype(np.float32) if dt.kind == "f" else arr.copy()
    return out
