"""The native ingestion + packers against vectors captured from the REFERENCE's own
data_prepare.py / data_loader.py (tests/golden/make_loader_fixtures.py): bit-for-bit."""
import os

import numpy as np
import pytest

from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.data_loader import Dataloader_hier_model_xing, Dataloader_single_level_model_xing
from hiertcn_amd.data_prepare import load_xing

HERE = os.path.dirname(os.path.abspath(__file__))
CSV = os.path.join(HERE, "golden", "interactions_synth.csv")


@pytest.fixture(scope="module")
def fx():
    return np.load(os.path.join(HERE, "golden", "loader_fixtures.npz"))


@pytest.fixture(scope="module")
def data():
    return load_xing(CSV)


def test_load_xing_matches_reference(fx, data):
    vt, vd = data
    assert vd.dtype == np.float64 and vd.shape[1] == 5
    np.testing.assert_array_equal(vt.astype(np.float64), fx["viewer_table"])
    np.testing.assert_array_equal(vd, fx["viewer_data"])
    # the filters of data_prepare.py:21,32,47 did something on this input
    assert (vd[:, 2] != 4).all() and vt[:, 1].min() >= 10 and vt[:, 1].max() <= 500
    assert vd[:, 1].min() == 1 and len(np.unique(vd[:, 1])) == int(vd[:, 1].max())


@pytest.mark.parametrize("split", ["train", "test"])
def test_hier_loader_matches_reference(fx, data, split):
    B, S, max_act, _, n_train, n_test, _ = [int(v) for v in fx["meta"]]
    args = hier_tcn_args(batch_size=B, max_session_num=S, max_activity_len=max_act)
    ld = Dataloader_hier_model_xing(args, split, data=data)
    np.testing.assert_array_equal(ld.viewer_table.astype(np.float64), fx[f"hier_{split}_table"])
    dones = []
    for b in range(n_train if split == "train" else n_test):
        x, y, m, info = ld.get_batch()
        dones.append(ld.done)
        assert len(x) == len(y) == len(m) == len(info) == S
        for s in range(S):
            for got, key in ((x[s], "x"), (y[s], "y"), (m[s], "m"), (info[s], "i")):
                want = fx[f"hier_{split}_b{b}_{key}{s}"]
                assert got.dtype == want.dtype == np.float64 and got.shape == want.shape, (b, s, key)
                np.testing.assert_array_equal(got, want, err_msg=f"batch {b} slot {s} {key}")
    np.testing.assert_array_equal(np.array(dones), fx[f"hier_{split}_done"])
    if split == "test":
        assert any(dones), "the fixture must cover the wrap-around / done path"


def test_packed_batch_equals_slot_lists(fx, data):
    """get_packed_batch() is the same batch as get_batch(), already in the C-ABI layout."""
    B, S, max_act = [int(v) for v in fx["meta"][:3]]
    args = hier_tcn_args(batch_size=B, max_session_num=S, max_activity_len=max_act)
    a = Dataloader_hier_model_xing(args, "train", data=data)
    b = Dataloader_hier_model_xing(args, "train", data=data)
    for _ in range(10):
        xl, yl, ml, _ = a.get_batch()
        x, y, mask, lens = b.get_packed_batch()
        assert a.done == b.done and x.dtype == np.int32 and mask.dtype == np.float32
        assert lens == [v.shape[1] for v in xl] and x.shape == (B, S, max(lens))
        for s in range(S):
            L = lens[s]
            np.testing.assert_array_equal(x[:, s, :L], xl[s])
            np.testing.assert_array_equal(y[:, s, :L], yl[s])
            assert not x[:, s, L:].any() and not y[:, s, L:].any()
            np.testing.assert_array_equal(mask[:, s:s + 1], ml[s])


def test_single_level_loader_matches_reference(fx, data):
    Bs, n = int(fx["meta"][3]), int(fx["meta"][6])
    args = hier_tcn_args(batch_size=Bs)
    ld = Dataloader_single_level_model_xing(args, "train", data=data)
    ld2 = Dataloader_single_level_model_xing(args, "train", data=data)
    dones = []
    for b in range(n):
        x, y, info = ld.get_batch()
        dones.append(ld.done)
        np.testing.assert_array_equal(x, fx[f"single_b{b}_x"])
        np.testing.assert_array_equal(y, fx[f"single_b{b}_y"])
        np.testing.assert_array_equal(info, fx[f"single_b{b}_i"])
        xp, yp = ld2.get_packed_batch()
        np.testing.assert_array_equal(xp, x.astype(np.int32))
        np.testing.assert_array_equal(yp, y.astype(np.int32))
    np.testing.assert_array_equal(np.array(dones), fx["single_done"])


@pytest.mark.parametrize("B,S,L,shuffle", [(4, 3, 6, False), (32, 10, 20, False), (7, 2, 5, True), (64, 4, 9, True)])
def test_native_packer_equals_the_numpy_packer(B, S, L, shuffle):
    """csrc/packer.c (lane queues + batch gather in C, libhiertcn_packer.so) against the NumPy / heapq form of the same loader: identical
    packed batches, slot-list batches, `done` flags and queue lengths over several passes of the split (the walk wraps -- with a fresh
    permutation when shuffling -- inside a fill; both loaders draw the permutations from the same seeded global RNG state)."""
    import os
    from hiertcn_amd import data_loader as DL
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.data_prepare import load_xing
    if DL._packer_lib() is None:
        pytest.skip("libhiertcn_packer.so not built (python -m hiertcn_amd.build)")
    here = os.path.dirname(os.path.abspath(__file__))
    data = load_xing(os.path.join(here, "golden", "interactions_synth.csv"))
    n_items = int(data[1][:, 1].max()) + 1
    args = hier_tcn_args(item_num=n_items, batch_size=B, max_session_num=S, max_activity_len=L, shuffle=shuffle)
    runs = []
    for native in (True, False):
        if not native:
            os.environ["HIERTCN_PY_PACKER"] = "1"
        try:
            np.random.seed(11)
            ld = DL.Dataloader_hier_model_xing(args, "all", data=data)
            assert (ld._pk is not None) == native
            out, dones = [], 0
            for i in range(60):
                if i % 3 == 2:
                    xs, ys, ms, _info = ld.get_batch()
                    out.append(("slots", [np.asarray(v) for v in xs], [np.asarray(v) for v in ys], [np.asarray(v) for v in ms]))
                else:
                    x, y, m, w = ld.get_packed_batch()
                    out.append(("packed", x, y, m, w))
                out.append(("state", bool(ld.done), np.asarray(ld.queue_len, np.float64).copy()))
                dones += int(ld.done)
            runs.append((out, dones))
        finally:
            os.environ.pop("HIERTCN_PY_PACKER", None)
    (a, da), (b, db) = runs
    assert da == db and da >= 1          # the split was walked more than once
    for ea, eb in zip(a, b):
        assert ea[0] == eb[0]
        for va, vb in zip(ea[1:], eb[1:]):
            if isinstance(va, list) and va and isinstance(va[0], np.ndarray):
                assert len(va) == len(vb) and all(np.array_equal(p, q) and p.dtype == q.dtype for p, q in zip(va, vb))
            elif isinstance(va, np.ndarray):
                assert np.array_equal(va, vb) and va.dtype == vb.dtype
            else:
                assert va == vb


def test_prefetcher_order_and_lifecycle(fx, data):
    """Prefetcher: the source's batches in order; nothing pins a prefetcher, its queue or its loader for the life of the process:
    close() leaves every registry, and a loader dropped WITHOUT close() takes its prefetcher's thread with it."""
    import gc
    import time
    import weakref

    from hiertcn_amd.data_loader import _LIVE_PREFETCHERS, Prefetcher

    B, S, max_act = [int(v) for v in fx["meta"][:3]]
    args = hier_tcn_args(batch_size=B, max_session_num=S, max_activity_len=max_act)
    want_ld = Dataloader_hier_model_xing(args, "train", data=data)
    want = [want_ld.get_packed_batch() for _ in range(4)]
    live0 = len(_LIVE_PREFETCHERS)
    ld = Dataloader_hier_model_xing(args, "train", data=data)
    pre = ld._prefetcher = Prefetcher(ld.get_packed_batch, depth=2, owner=ld)
    for w in want:
        got = pre.get()
        for a, b in zip(got[:3], w[:3]):
            np.testing.assert_array_equal(a, b)
    assert len(_LIVE_PREFETCHERS) == live0 + 1 and pre in ld._prefetchers
    thread = pre._t
    pre.close()
    pre.close()                                              # idempotent
    assert not thread.is_alive() and pre._q.empty()
    assert len(_LIVE_PREFETCHERS) == live0 and pre not in ld._prefetchers and "_prefetcher" not in ld.__dict__
    # a second prefetcher on the same loader, then the loader is dropped without close()
    pre2 = ld._prefetcher = Prefetcher(ld.get_packed_batch, depth=2, owner=ld)
    pre2.get()
    t2, wl, wp = pre2._t, weakref.ref(ld), weakref.ref(pre2)
    del ld, pre2, pre
    for _ in range(50):
        gc.collect()
        if not t2.is_alive() and wl() is None and wp() is None:
            break
        time.sleep(0.1)
    assert wl() is None, "the prefetcher (or its thread) kept the loader alive"
    assert not t2.is_alive() and wp() is None
    assert len(_LIVE_PREFETCHERS) == live0
