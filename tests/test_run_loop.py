"""run_hier / evaluate_hier control flow (run_hier_xing.py:83-207,210-350) on the synthetic XING
CSV, with the CPU oracle as the model handle: batching, state carry, mean-of-means aggregation,
results.txt text, eval-once-when-not-training."""
import json
import os
import re

import numpy as np
import pytest

from hiertcn_amd import checkpoint
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.data_loader import Dataloader_hier_model_xing
from hiertcn_amd.data_prepare import load_xing
from hiertcn_amd.run_hier_xing import evaluate_hier, ranks_analysis, run_hier
from oracle import hiertcn_oracle as O
from tests.gpu_common import OracleSession, make_params

HERE = os.path.dirname(os.path.abspath(__file__))
CSV = os.path.join(HERE, "golden", "interactions_synth.csv")


@pytest.fixture(scope="module")
def data():
    return load_xing(CSV)


def _args(tmp_path, data, **kw):
    n_items = int(data[1][:, 1].max()) + 1
    a = hier_tcn_args(item_num=n_items, emb_dim=16, hidden_dim=8, num_layer=2, tcn_channel=[16, 16], kernel_size=3,
                      batch_size=4, max_session_num=3, max_activity_len=6, **kw)
    a.log_path = str(tmp_path) + "/log/"
    a.save_path = str(tmp_path) + "/model/"
    return a


def test_evaluate_hier_aggregation_and_results_file(tmp_path, data):
    args = _args(tmp_path, data)
    params = make_params(args)
    sess = OracleSession(args, params)
    os.makedirs(args.log_path + "test/")
    ld = Dataloader_hier_model_xing(args, "test", data=data)
    st0 = np.zeros((4, 16))
    summary, summary_val, t_load, t_run, st = evaluate_hier(args, ld, st0, sess, prefix="test")
    # independent replay: same batches through oracle.evaluate_hier
    ld2 = Dataloader_hier_model_xing(args, "test", data=data)
    batches = []
    while True:
        x, y, m, _ = ld2.get_batch()
        batches.append((x, y, m))
        if ld2.done:
            break
    ref, per_batch, st_ref = O.evaluate_hier(args, batches, np.zeros((4, 16)), params)
    assert len(batches) == 5  # the fixture's `done` pattern: wrap in the 5th batch
    assert summary["loss_epoch"] == pytest.approx(ref["loss"], rel=1e-12)
    assert summary["recall10_epoch_mean"] == pytest.approx(ref["recall10"], rel=1e-12)
    assert summary["reciprocal_ranks_float_epoch_mean"] == pytest.approx(ref["mrr"], rel=1e-12)
    assert summary["ranks_float_epoch_mean"] == pytest.approx(ref["rank_frac"], rel=1e-12)
    np.testing.assert_allclose(st, st_ref, rtol=0, atol=0)
    txt = open(args.log_path + "test/results.txt").read().splitlines()
    assert txt[0] == "final"
    assert re.fullmatch(r"loss_epoch \S+, ranks_float_epoch_mean \S+", txt[1])
    assert re.fullmatch(r"reciprocal_ranks_float_epoch_mean \S+, recall1_epoch_mean \S+", txt[2])
    assert re.fullmatch(r"recall5_epoch_mean \S+, recall10_epoch_mean \S+", txt[3])
    assert float(txt[1].split()[1].rstrip(",")) == pytest.approx(ref["loss"])


def test_results_file_every_20_batches(tmp_path, data):
    args = _args(tmp_path, data)

    class Loader:  # never-ending stub with the loader protocol; done on the 41st batch
        def __init__(self):
            self.inner = Dataloader_hier_model_xing(args, "train", data=data)
            self.n, self.done, self.viewer_table = 0, False, self.inner.viewer_table

        def get_batch(self):
            self.n += 1
            out = self.inner.get_batch()
            self.done = self.n >= 41
            return out

    os.makedirs(args.log_path + "validate/")
    evaluate_hier(args, Loader(), np.zeros((4, 16)), OracleSession(args, make_params(args)), prefix="validate")
    txt = open(args.log_path + "validate/results.txt").read()
    assert txt.count("batch count 20:loss_epoch") == 1 and txt.count("batch count 40:loss_epoch") == 1
    assert txt.count("final\n") == 1


def test_ranks_analysis_matches_loop_definition():
    rng = np.random.default_rng(0)
    widths = [3, 5, 2]
    xs = [np.zeros((4, w)) for w in widths]
    rf = rng.random((4, sum(widths)))
    mk = (rng.random((4, sum(widths))) > 0.3).astype(float)
    got, got_n = ranks_analysis(rf, xs, mk, np.zeros(7), np.zeros(7))
    want, want_n = np.zeros(7), np.zeros(7)
    first = 0
    for w in widths:   # run_hier_xing.py:21-30, slot by slot
        for i in range(w):
            want[i] += rf[:, first + i].sum()
            want_n[i] += mk[:, first + i].sum()
        first += w
    np.testing.assert_allclose(got, want)
    np.testing.assert_allclose(got_n, want_n)


def test_ranks_analysis_matches_the_references_own_function_on_captured_vectors():
    """tests/golden/run_loop_fixtures.npz: inputs and outputs of the reference's ranks_analysis (run_hier_xing.py:11-31) executed in the build
    container (make_run_loop_fixtures.py) -- ragged slot widths, width-1 slots, accumulators that are non-zero at entry."""
    import os
    fx = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "run_loop_fixtures.npz"))
    assert int(fx["n_cases"]) >= 6
    for c in range(int(fx["n_cases"])):
        widths = [int(w) for w in fx[f"c{c}_widths"]]
        B = fx[f"c{c}_ranks_float"].shape[0]
        acc, cnt = ranks_analysis(fx[f"c{c}_ranks_float"], [np.zeros((B, w)) for w in widths], fx[f"c{c}_mask_y"], fx[f"c{c}_acc_in"].copy(),
                                  fx[f"c{c}_cnt_in"].copy())
        np.testing.assert_allclose(acc, fx[f"c{c}_acc_out"], rtol=1e-13, atol=0)      # (a different summation order: column sums first)
        np.testing.assert_array_equal(cnt, fx[f"c{c}_cnt_out"])


def test_run_hier_eval_only_runs_once_and_keeps_per_loader_state(tmp_path, data):
    args = _args(tmp_path, data)
    args.train, args.validate, args.test = False, True, True
    sess = OracleSession(args, make_params(args))
    lv = Dataloader_hier_model_xing(args, "validate", data=data)
    lt = Dataloader_hier_model_xing(args, "test", data=data)
    run_hier(args, None, lv, lt, sess=sess)
    for split in ("validate", "test"):
        assert open(args.log_path + split + "/results.txt").read().count("final") == 1
        assert os.path.exists(args.log_path + split + "/scalars.jsonl")
    assert os.path.isdir(args.log_path + "train/")


def test_checkpoint_roundtrip_with_carried_state(tmp_path, data):
    args = _args(tmp_path, data)
    p = make_params(args)
    f = checkpoint.save(args.save_path + "20.ckpt", p, extra={"state_np": np.ones((4, 16))})
    assert f.endswith("20.ckpt.npz") and os.path.exists(f)
    q, extra = checkpoint.load(args.save_path + "20.ckpt", with_extra=True)
    assert set(q) == set(p) and all(np.array_equal(q[k], p[k]) for k in p)
    assert "hier/multi_rnn_cell/cell_1/gru_cell/gates/kernel" in q and extra["state_np"].shape == (4, 16)


def test_run_xing_dispatches_the_mixture_models_and_refuses_training(tmp_path):
    """model.py:74-77: --model_type tcn_gmm / gru_gmm reach run_gmm_xing.run_gmm: the evaluation flow over a model handle with eval_step ->
    'sums', and (round 6) run_nn's training cadence over enable_training / train_step / export; a model whose training step is not built
    raises from enable_training and the loop lets that through."""
    import numpy as np
    import pytest
    import torch
    from hiertcn_amd.args import make_args
    from hiertcn_amd.run_gmm_xing import evaluate_gmm, run_gmm

    class FakeGmm:
        def __init__(self):
            self.calls = 0

        def eval_step(self, x, y):
            self.calls += 1
            mask = (np.asarray(y) > 0)
            lanes = mask.any(1).sum()
            return dict(sums=torch.tensor([2.5 * lanes * self.calls, float(lanes)], dtype=torch.float64))

    class Loader:
        def __init__(self, n):
            self.n, self.i, self.done = n, 0, False

        def get_batch(self):
            self.i += 1
            self.done = self.i >= self.n
            y = np.ones((4, 6), np.int32)
            return y, y, None

    a = make_args([], model_type="tcn_gmm", item_num=50, num_mixture=2, sigma=0.1, loss="gmm_prob", adjust=False)
    a.log_path = str(tmp_path) + "/log/"
    a.name = "t"
    fake = FakeGmm()
    m, _, _ = evaluate_gmm(a, Loader(3), fake, prefix="test")
    assert fake.calls == 3 and abs(m["loss_epoch"] - 2.5 * (1 + 2 + 3) / 3) < 1e-12      # mean over batches of the per-batch lane means
    assert np.isnan(m["recall10_epoch_mean"])
    a.train, a.validate, a.test, a.load, a.save = True, False, True, False, False
    a.epoch_max, a.epoch_batches_train, a.test_epoch, a.lr_schedule, a.learning_rate = 2, 3, 2, False, 0.5

    class Refuses(FakeGmm):
        def enable_training(self):
            raise NotImplementedError("model_gru_gmm trains nowhere in this build")

    with pytest.raises(NotImplementedError, match="trains nowhere"):
        run_gmm(a, Loader(10 ** 9), None, Loader(2), sess=Refuses())

    class Trains(FakeGmm):
        enabled, steps, lrs = False, 0, []

        def enable_training(self):
            self.enabled = True

        def train_step(self, x, y, lr):
            assert self.enabled
            self.steps += 1
            self.lrs.append(lr)
            return dict(sums=torch.tensor([4.0 / self.steps, 2.0], dtype=torch.float64))

        def export(self, which):
            return {"w": np.full(1, self.steps, np.float32)}

    tr = Trains()
    assert run_gmm(a, Loader(10 ** 9), None, Loader(2), sess=tr) is tr
    assert tr.steps == 3 * 3 and set(tr.lrs) == {0.5}                    # epochs 0, 1, 2 x epoch_batches_train steps (run_nn_xing.py:147-153)
    assert tr.calls == 2                                                    # one test pass, at epoch 2 (test_epoch 2; epoch 0 never evaluates)
    assert tr.params["w"][0] == 9                                           # the weights the loop hands back are the trained ones
    lines = open(a.log_path + "train/scalars.jsonl").read().splitlines()
    assert len(lines) == 3 and json.loads(lines[0])["tag"] == "loss_epoch"
    assert abs(json.loads(lines[0])["value"] - np.mean([2.0 / k for k in (1, 2, 3)])) < 1e-12
    a.train, a.validate, a.test, a.load = False, False, True, False
    assert run_gmm(a, None, None, Loader(2), sess=fake) is fake


def test_scalar_log_writes_a_tensorboard_event_file(tmp_path):
    """run_hier_xing.py:233-235,311-345 write tf.Summary scalars through tf.summary.FileWriter.  ScalarLog writes the same file format without
    TensorFlow: TFRecord framing (length, masked CRC-32C of the length, payload, masked CRC-32C of the payload) around serialised Event protos.  Checked
    here with an independent reader: CRC-32C against its published check value, every record's two checksums, and the payloads parsed by
    google.protobuf against message types declared with event.proto / summary.proto's field numbers."""
    import struct
    from google.protobuf import descriptor_pb2, descriptor_pool, message_factory
    from hiertcn_amd.run_common import ScalarLog, crc32c
    assert crc32c(b"123456789") == 0xE3069283 and crc32c(b"") == 0
    log = ScalarLog(str(tmp_path) + "/train/")
    log.add_summary({"loss_epoch": 6.25, "recall10_epoch_mean": 0.125}, 3)
    log.add_summary({"loss_epoch": 5.5}, 4)
    files = [f for f in os.listdir(log.logdir) if f.startswith("events.out.tfevents.")]
    assert len(files) == 1 and os.path.exists(log.logdir + "/scalars.jsonl")
    raw = open(os.path.join(log.logdir, files[0]), "rb").read()
    fd = descriptor_pb2.FileDescriptorProto(name="hiertcn_test_event.proto", package="hiertcn_test", syntax="proto3")
    val = fd.message_type.add(name="Value")
    val.field.add(name="tag", number=1, type=9, label=1)
    val.field.add(name="simple_value", number=2, type=2, label=1)
    summ = fd.message_type.add(name="Summary")
    summ.field.add(name="value", number=1, type=11, label=3, type_name=".hiertcn_test.Value")
    evd = fd.message_type.add(name="Event")
    evd.field.add(name="wall_time", number=1, type=1, label=1)
    evd.field.add(name="step", number=2, type=3, label=1)
    evd.field.add(name="file_version", number=3, type=9, label=1)
    evd.field.add(name="summary", number=5, type=11, label=1, type_name=".hiertcn_test.Summary")
    pool = descriptor_pool.DescriptorPool()
    pool.Add(fd)
    Event = message_factory.GetMessageClass(pool.FindMessageTypeByName("hiertcn_test.Event"))

    def unmask(m):
        r = (m - 0xA282EAD8) & 0xFFFFFFFF
        return ((r >> 17) | (r << 15)) & 0xFFFFFFFF

    events, off = [], 0
    while off < len(raw):
        (n,) = struct.unpack_from("<Q", raw, off)
        assert unmask(struct.unpack_from("<I", raw, off + 8)[0]) == crc32c(raw[off:off + 8])
        payload = raw[off + 12:off + 12 + n]
        assert unmask(struct.unpack_from("<I", raw, off + 12 + n)[0]) == crc32c(payload)
        e = Event()
        e.ParseFromString(payload)
        events.append(e)
        off += 16 + n
    assert off == len(raw) and len(events) == 3
    assert events[0].file_version == "brain.Event:2" and events[0].wall_time > 1e9
    assert events[1].step == 3 and [(v.tag, v.simple_value) for v in events[1].summary.value] == [("loss_epoch", 6.25), ("recall10_epoch_mean", 0.125)]
    assert events[2].step == 4 and [(v.tag, v.simple_value) for v in events[2].summary.value] == [("loss_epoch", 5.5)]


def test_warm_start_masks_and_evaluate_hier(tmp_path, data, monkeypatch):
    """args.warm_start (args.py:10-11): model.py:102-103 multiplies mask_y by the fed `mask_warmstart`, which run_hier_xing.py builds from the events'
    timestamps -- evaluation: after 1445617130 (:128), training: real events at or before it (:288).  The masked steps drop out of the loss and of every
    metric; the model itself (the session mean-pool over ALL targets, the carried state) is untouched."""
    from hiertcn_amd import run_common
    from hiertcn_amd.model import pack_slots, warm_score_ids
    from hiertcn_amd.run_common import warm_start_mask
    assert run_common.WARM_START_TIME == 1445617130                  # run_hier_xing.py:128,288
    # the reference's time leaves 4.9 % of the synthetic file's events "after": with 4 lanes most batches would have no scored user at all, and such
    # a batch is 0 / 0 = NaN in the reference's graph (model.py:113,117) and here alike.  8 lanes and the 30 % quantile keep every batch populated.
    WARM_START_TIME = float(np.quantile(data[1][:, 3], 0.3))
    monkeypatch.setattr(run_common, "WARM_START_TIME", WARM_START_TIME)
    args = _args(tmp_path, data)
    args.batch_size = 8
    params = make_params(args)
    sess = OracleSession(args, params)
    os.makedirs(args.log_path + "test/")
    os.makedirs(args.log_path + "plain/")
    st0 = np.zeros((8, 16))
    warm, _, _, _, st_w = evaluate_hier(args, Dataloader_hier_model_xing(args, "train", data=data), st0, sess, prefix="test", warm_start=True)
    plain, _, _, _, st_p = evaluate_hier(args, Dataloader_hier_model_xing(args, "train", data=data), st0, sess, prefix="plain")
    np.testing.assert_array_equal(st_w, st_p)                      # the mask never reaches the model
    assert warm["loss_epoch"] != plain["loss_epoch"]
    # independent replay
    ld = Dataloader_hier_model_xing(args, "train", data=data)
    state, sums, n, n_eval, n_train, n_real = np.zeros((8, 16)), np.zeros(2), 0, 0, 0, 0
    while True:
        xl, yl, ml, il = ld.get_batch()
        ts = np.concatenate(il, axis=1)[:, :, 3]
        y_id = np.concatenate(yl, axis=1).astype(np.int64)
        me, mt = warm_start_mask(il, train=False), warm_start_mask(il, train=True)
        assert np.array_equal(me > 0, ts > WARM_START_TIME) and np.array_equal(mt > 0, (ts <= WARM_START_TIME) & (ts > 0))
        assert not (me * mt).any() and np.array_equal((me + mt) > 0, y_id > 0)       # the two masks split the real events
        n_eval, n_train, n_real = n_eval + int(me.sum()), n_train + int(mt.sum()), n_real + int((y_id > 0).sum())
        pred, state = O.model_hier(args, xl, yl, ml, state, params)
        m = O.masked_loss_and_metrics(pred, y_id * (me > 0), int(args.item_num))
        assert np.isfinite(m["loss"])
        sums += [m["loss"], m["recall10"]]
        n += 1
        x, y, _, lens = pack_slots(xl, yl, ml)
        ys = warm_score_ids(y, lens, me)
        assert ys.shape == y.shape and np.array_equal(ys > 0, (y > 0) & (ys == y)) and int((ys > 0).sum()) == int(me.sum())
        if ld.done:
            break
    assert n_eval > 0 and n_train > 0 and n_eval + n_train == n_real
    assert warm["loss_epoch"] == pytest.approx(sums[0] / n, rel=1e-12) and warm["recall10_epoch_mean"] == pytest.approx(sums[1] / n, rel=1e-12)
    with pytest.raises(ValueError):
        warm_score_ids(y, lens, me[:, :-1])
    # a batch in which no lane has a scored step: user_count = 0 -> 0 / 0, as in the reference's graph
    none = O.masked_loss_and_metrics(pred, y_id * 0, int(args.item_num))
    assert np.isnan(none["loss"])



def test_run_hier_gmm_keeps_one_carried_state_per_loader_and_run_hiers_cadence(tmp_path):
    """run_gmm_xing.run_hier_gmm / evaluate_hier_gmm over a fake model handle (no GPU): run_hier's cadence (run_hier_xing.py:261-349) -- per epoch
    epoch_batches_train steps at the scheduled rate, each fed the state the PREVIOUS step returned (:301), one state per loader surviving across
    evaluations and epochs (:241-243,332,343), evaluation every test_epoch, a checkpoint every save_epoch."""
    import torch
    from hiertcn_amd import checkpoint
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.run_gmm_xing import run_hier_gmm

    class Fake:
        device = torch.device("cpu")
        params = {"hier/emb/kernel": np.zeros((3, 2), np.float32)}

        def __init__(self):
            self.train_states, self.eval_states, self.lrs, self.enabled = [], [], [], False

        def enable_training(self):
            self.enabled = True

        def train_step_hier(self, xd, yd, md, st, lr):
            assert self.enabled and xd.dtype == torch.int32 and md.dtype == torch.float32
            self.train_states.append(float(st[0, 0]))
            self.lrs.append(lr)
            return dict(state=st + 1.0, sums=torch.tensor([3.0, 2.0], dtype=torch.float64))

        def eval_step_hier(self, xd, yd, md, st):
            self.eval_states.append(float(st[0, 0]))
            return dict(state=st + 100.0, sums=torch.tensor([5.0, 2.0], dtype=torch.float64))

        def export(self, which):
            return {"hier/emb/kernel": np.full((3, 2), float(len(self.train_states)), np.float32)}

    class Loader:
        def __init__(self, n):
            self.n, self.i, self.done = n, 0, False

        def get_packed_batch(self):
            self.i += 1
            self.done = self.i % self.n == 0
            z = np.ones((4, 2, 3), np.int32)
            return z, z, np.ones((4, 2), np.float32), [3, 3]

    a = hier_tcn_args(item_num=50, emb_dim=64, hidden_dim=64, num_layer=2, batch_size=4, model_low_type="tcn_gmm", loss="gmm_prob")
    a.log_path, a.save_path, a.load_path, a.name = str(tmp_path) + "/log/", str(tmp_path) + "/model/", str(tmp_path) + "/model/", "t"
    os.makedirs(a.save_path)
    a.train, a.validate, a.test, a.load, a.save = True, False, True, False, True
    a.epoch_max, a.epoch_batches_train, a.test_epoch, a.save_epoch, a.lr_schedule, a.learning_rate = 2, 3, 1, 2, False, 0.5
    fake = Fake()
    run_hier_gmm(a, Loader(10 ** 9), None, Loader(2), sess=fake)
    assert fake.train_states == [float(i) for i in range(9)]            # 3 epochs x 3 steps, the state handed on from step to step and epoch to epoch
    assert fake.lrs == [0.5] * 9
    assert fake.eval_states == [0.0, 100.0, 200.0, 300.0]               # epochs 1 and 2 (not 0), two batches each: the test loader's own state, carried across evaluations
    saved = checkpoint.load(a.save_path + "2.ckpt")
    assert float(saved["hier/emb/kernel"][0, 0]) == 9.0                 # written after epoch 2's steps
    assert not os.path.exists(a.save_path + "1.ckpt")
    lines = open(a.log_path + "test/scalars.jsonl").read().splitlines()
    assert len(lines) == 2 and all(abs(__import__("json").loads(l)["value"] - 2.5) < 1e-12 for l in lines)
