"""CPU: hiertcn_amd.tf_checkpoint (reader of TF-1.x V2 checkpoints without TensorFlow) against a WRITER of the same documented format
kept here in the tests -- round trip through the files, multi-block indexes with prefix-compressed keys, shards, checksums, the Adam slot
filter -- and against hand-assembled bytes.  No file written by TensorFlow itself is available (module docstring of the reader)."""
import struct

import numpy as np
import pytest

from hiertcn_amd import tf_checkpoint as T


def _vi(n):
    out = b""
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


def _field(num, wt, payload):
    return _vi((num << 3) | wt) + payload


def _entry_proto(dtype, shape, shard, offset, size, crc):
    dims = b"".join(_field(2, 2, _vi(len(d)) + d) for d in (_field(1, 0, _vi(s)) for s in shape))
    return (_field(1, 0, _vi(dtype)) + _field(2, 2, _vi(len(dims)) + dims) + (_field(3, 0, _vi(shard)) if shard else b"")
            + _field(4, 0, _vi(offset)) + _field(5, 0, _vi(size)) + _field(6, 5, struct.pack("<I", crc)))


def _block(items, restart_interval=4):
    """LevelDB-style block with prefix compression and restart points."""
    out, restarts, last = b"", [], b""
    for i, (k, v) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(k), len(last)) and k[shared] == last[shared]:
                shared += 1
        out += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        last = k
    out += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    return out


def write_bundle(prefix, tensors, num_shards=1, block_entries=5):
    """tensors: {name: ndarray}; entries sorted by key as a table requires; round-robin over shards."""
    names = sorted(tensors)
    data = [b"" for _ in range(num_shards)]
    items = [(b"", _field(1, 0, _vi(num_shards)) + _field(2, 0, _vi(0)))]        # BundleHeaderProto under the empty key
    dt = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9, np.dtype(np.float64): 2}
    for i, n in enumerate(names):
        a = np.ascontiguousarray(tensors[n])
        raw = a.tobytes()
        sid = i % num_shards
        items.append((n.encode(), _entry_proto(dt[a.dtype], a.shape, sid, len(data[sid]), len(raw), T.masked_crc32c(raw))))
        data[sid] += raw
    for sid in range(num_shards):
        with open("%s.data-%05d-of-%05d" % (prefix, sid, num_shards), "wb") as f:
            f.write(data[sid])
    body, index_items = b"", []
    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        blk = _block(chunk)
        index_items.append((chunk[-1][0], _vi(len(body)) + _vi(len(blk))))
        body += blk + b"\x00" + struct.pack("<I", T.masked_crc32c(blk + b"\x00"))
    meta = _block([])
    meta_h = _vi(len(body)) + _vi(len(meta))
    body += meta + b"\x00" + struct.pack("<I", T.masked_crc32c(meta + b"\x00"))
    idx = _block(index_items, restart_interval=1)
    idx_h = _vi(len(body)) + _vi(len(idx))
    body += idx + b"\x00" + struct.pack("<I", T.masked_crc32c(idx + b"\x00"))
    footer = meta_h + idx_h
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", T.TABLE_MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(body + footer)


def test_crc32c_known_answers():
    assert T.crc32c(b"123456789") == 0xE3069283                       # the CRC-32C check value (RFC 3720 appendix B.4)
    assert T.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert T.masked_crc32c(b"") == 0xA282EAD8


def test_round_trip_hier_model_names_shards_and_slot_filter(tmp_path):
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.params import init_params
    args = hier_tcn_args(item_num=53, emb_dim=32, hidden_dim=32, tcn_channel=[32, 32])
    p = init_params(args, 3)
    extra = {k + "/Adam": np.zeros_like(v) for k, v in p.items()}
    extra.update({k + "/Adam_1": np.ones_like(v) for k, v in p.items()})
    extra["beta1_power"] = np.array(0.9, np.float32)
    extra["beta2_power"] = np.array(0.999, np.float32)
    extra["global_step"] = np.array(7, np.int64)
    prefix = str(tmp_path / "20.ckpt")
    write_bundle(prefix, {**p, **extra}, num_shards=2, block_entries=4)
    header, entries = T.read_index(prefix)
    assert header["num_shards"] == 2 and len(entries) == len(p) + len(extra)
    got = T.read_tf_checkpoint(prefix)
    assert set(got) == set(p) | {"global_step"}                        # optimizer slots skipped
    for k in p:
        assert got[k].dtype == np.float32 and np.array_equal(got[k], p[k]), k
    assert int(got["global_step"].reshape(-1)[0]) == 7
    full = T.read_tf_checkpoint(prefix, with_slots=True)
    assert np.array_equal(full["hier/emb/kernel/Adam_1"], np.ones_like(p["hier/emb/kernel"]))
    assert abs(float(full["beta2_power"].reshape(-1)[0]) - 0.999) < 1e-7


def test_corruption_and_foreign_files_are_refused(tmp_path):
    prefix = str(tmp_path / "1.ckpt")
    write_bundle(prefix, {"a/kernel": np.arange(12, dtype=np.float32).reshape(3, 4)})
    raw = bytearray(open(prefix + ".index", "rb").read())
    raw[5] ^= 0xFF
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(ValueError):
        T.read_tf_checkpoint(prefix)
    open(prefix + ".index", "wb").write(b"not a table" * 10)
    with pytest.raises(ValueError):
        T.read_tf_checkpoint(prefix)
    prefix2 = str(tmp_path / "2.ckpt")
    write_bundle(prefix2, {"a/kernel": np.arange(12, dtype=np.float32).reshape(3, 4)})
    d = bytearray(open(prefix2 + ".data-00000-of-00001", "rb").read())
    d[0] ^= 1
    open(prefix2 + ".data-00000-of-00001", "wb").write(bytes(d))
    with pytest.raises(ValueError):
        T.read_tf_checkpoint(prefix2)                                   # tensor checksum


def test_checkpoint_load_accepts_a_tf_prefix(tmp_path):
    from hiertcn_amd import checkpoint
    prefix = str(tmp_path / "20.ckpt")
    w = {"hier/emb/bias": np.linspace(0, 1, 8, dtype=np.float32)}
    write_bundle(prefix, w)
    params = checkpoint.load(prefix)
    assert np.array_equal(params["hier/emb/bias"], w["hier/emb/bias"])


def test_crc32c_fast_equals_the_scalar_crc_and_large_tensors_are_verified(tmp_path):
    """crc32c_fast (chunk CRCs advanced together + zero-advance combine) is bit-identical to the byte-serial CRC at sizes around its chunking
    boundaries, and the reader now verifies tensors of every size: a flipped byte inside a multi-megabyte table is caught, and a data shard
    cut short is refused instead of being read past its end."""
    from hiertcn_amd import tf_checkpoint as T
    rng = np.random.default_rng(0)
    assert T.crc32c(b"123456789") == 0xE3069283                      # the CRC-32C check value
    for n in (1 << 16, (1 << 16) + 1, 70_001, 1_048_577, 3_000_017):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert T.crc32c_fast(data) == T.crc32c(data), n
    big = rng.standard_normal((700, 600)).astype(np.float32)          # 1.68 MB: above the old 1 MB verification limit
    prefix = str(tmp_path / "big.ckpt")
    write_bundle(prefix, {"hier/emb/kernel": big})
    got = T.read_tf_checkpoint(prefix)
    assert np.array_equal(got["hier/emb/kernel"], big)
    shard = prefix + ".data-00000-of-00001"
    raw = bytearray(open(shard, "rb").read())
    raw[len(raw) // 2] ^= 0x40
    open(shard, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="checksum"):
        T.read_tf_checkpoint(prefix)
    open(shard, "wb").write(bytes(raw[: len(raw) // 2]))
    with pytest.raises(ValueError, match="truncated"):
        T.read_tf_checkpoint(prefix)
