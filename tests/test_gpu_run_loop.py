"""-m gpu: the HIP path driven through the reference-shaped run loop (evaluate_hier over the
lane-queue loader on the synthetic XING CSV) against the CPU oracle driven through the same loop."""
import os

import numpy as np
import pytest

from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.data_loader import Dataloader_hier_model_xing
from hiertcn_amd.data_prepare import load_xing
from hiertcn_amd.run_hier_xing import evaluate_hier
from oracle import hiertcn_oracle as O
from tests.gpu_common import OracleSession, make_params

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CSV = os.path.join(HERE, "golden", "interactions_synth.csv")


class _SlotListsOnly:
    """Hides get_packed_batch so that evaluate_hier takes the reference's slot-list protocol."""

    def __init__(self, inner):
        self._inner = inner
        self.viewer_table = inner.viewer_table

    def get_batch(self):
        return self._inner.get_batch()

    @property
    def done(self):
        return self._inner.done


@pytest.mark.parametrize("packed", [True, False])
def test_evaluate_hier_hip_vs_oracle(tmp_path, packed):
    from hiertcn_amd.model import HierTCNModel

    data = load_xing(CSV)
    n_items = int(data[1][:, 1].max()) + 1
    args = hier_tcn_args(item_num=n_items, emb_dim=64, hidden_dim=32, num_layer=2, tcn_channel=[64, 64], kernel_size=5,
                         batch_size=4, max_session_num=3, max_activity_len=6)
    args.log_path = str(tmp_path) + "/log/"
    os.makedirs(args.log_path + "test/")
    params = make_params(args)
    model = HierTCNModel(args, params)
    oracle = OracleSession(args, O.round_params_bf16(params, E=64), O.MIRROR)
    st0 = np.zeros((4, 64))
    ld = Dataloader_hier_model_xing(args, "test", data=data)
    got, got_val, _, _, st = evaluate_hier(args, ld if packed else _SlotListsOnly(ld), st0, model, prefix="test")
    want, _, _, _, st_ref = evaluate_hier(args, _SlotListsOnly(Dataloader_hier_model_xing(args, "test", data=data)), st0, oracle,
                                          prefix="oracle")
    # tolerances: loss 1e-3 absolute (fp32 accumulation order of the fused LSE vs the fp64 mirror); carried GRU state
    # 1e-4 after 5 batches x 3 slots; rank-derived means may differ only through near-ties of bf16-mirrored logits
    assert abs(got["loss_epoch"] - want["loss_epoch"]) < 1e-3
    np.testing.assert_allclose(st, st_ref, atol=1e-4, rtol=0)
    for k in ("recall1_epoch_mean", "recall5_epoch_mean", "recall10_epoch_mean", "reciprocal_ranks_float_epoch_mean",
              "ranks_float_epoch_mean"):
        assert abs(got[k] - want[k]) < 5e-3, (k, got[k], want[k])
    assert os.path.exists(args.log_path + "test/results.txt")
    model.engine.close()


@pytest.mark.parametrize("dtype", ["fp32", "split16"])
def test_evaluate_hier_in_the_exact_mode_equals_the_exact_oracle(tmp_path, dtype):
    """`--storage_dtype fp32` (exact.hip) and `--storage_dtype split16` (the fast parity-grade mode, score_x3.hip) through the run loop: the epoch metrics of evaluate_hier on the fp32 mode equal those of the exact oracle
    (fp64, unrounded weights) driven through the same loop -- loss to 1e-5, every rank-derived mean exactly or within one near-tie --
    with a scoring layer scaled to a trained model's logit range, where the 16-bit path would be visibly off."""
    from hiertcn_amd.model import HierTCNModel

    data = load_xing(CSV)
    n_items = int(data[1][:, 1].max()) + 1
    args = hier_tcn_args(item_num=n_items, emb_dim=64, hidden_dim=32, num_layer=2, tcn_channel=[64, 64], kernel_size=5,
                         batch_size=4, max_session_num=3, max_activity_len=6, storage_dtype=dtype)
    args.log_path = str(tmp_path) + "/log/"
    os.makedirs(args.log_path + "test/")
    params = make_params(args)
    params["hier/tcn/dense/kernel"] = (params["hier/tcn/dense/kernel"] * 60).astype(np.float32)     # logits of order 10
    model = HierTCNModel(args, params)
    assert model.engine.exact and model.engine.act_dtype.is_floating_point and model.engine.act_dtype.itemsize == 4
    oracle = OracleSession(args, params, O.EXACT64)
    st0 = np.zeros((4, 64))
    got, _, _, _, st = evaluate_hier(args, Dataloader_hier_model_xing(args, "test", data=data), st0, model, prefix="test")
    want, _, _, _, st_ref = evaluate_hier(args, _SlotListsOnly(Dataloader_hier_model_xing(args, "test", data=data)), st0, oracle,
                                          prefix="oracle")
    assert abs(got["loss_epoch"] - want["loss_epoch"]) < 1e-5 * max(1.0, abs(want["loss_epoch"]))
    np.testing.assert_allclose(st, st_ref, atol=1e-5, rtol=0)
    for k in ("recall1_epoch_mean", "recall5_epoch_mean", "recall10_epoch_mean", "reciprocal_ranks_float_epoch_mean",
              "ranks_float_epoch_mean"):
        assert abs(got[k] - want[k]) < 1e-4, (k, got[k], want[k])
    model.engine.close()


def test_pipelined_loops_equal_the_blocking_ones(tmp_path):
    """Round 4: evaluate_hier / run_hier leave the device running -- carried state device to device, a batch's ranks and metrics read one batch
    later through pinned buffers (model.py: PendingEval / PendingTrain).  HIERTCN_EVAL_SERIAL=1 / the *_packed calls are the blocking
    sess.run-shaped form: same epoch summary, same per-position rank table, same carried state (evaluation: bitwise); training: the same
    losses and carried state to 1e-5 (atomics reorder the gradient sums)."""
    from hiertcn_amd.model import HierTCNModel

    data = load_xing(CSV)
    n_items = int(data[1][:, 1].max()) + 1
    args = hier_tcn_args(item_num=n_items, emb_dim=64, hidden_dim=64, num_layer=2, tcn_channel=[64, 64], kernel_size=5,
                         batch_size=4, max_session_num=3, max_activity_len=6)
    args.log_path = None
    params = make_params(args)
    model = HierTCNModel(args, params)
    st0 = (0.1 * np.random.default_rng(0).standard_normal((4, 128))).astype(np.float32)
    outs = []
    for serial in (False, True):
        if serial:
            os.environ["HIERTCN_EVAL_SERIAL"] = "1"
        try:
            outs.append(evaluate_hier(args, Dataloader_hier_model_xing(args, "test", data=data), st0, model, prefix="test"))
        finally:
            os.environ.pop("HIERTCN_EVAL_SERIAL", None)
    (s_a, v_a, _, _, st_a), (s_b, v_b, _, _, st_b) = outs
    assert s_a == s_b
    assert np.array_equal(v_a["ranks_by_position"], v_b["ranks_by_position"])
    assert {k: v for k, v in v_a.items() if k != "ranks_by_position"} == {k: v for k, v in v_b.items() if k not in ("ranks_by_position", "state")}
    assert np.array_equal(np.asarray(st_a, np.float32), np.asarray(st_b, np.float32))
    # training: n steps blocking vs pipelined on two models with the same weights
    losses, states = [], []
    for pipelined in (False, True):
        m = HierTCNModel(args, params, train=True)
        ld = Dataloader_hier_model_xing(args, "train", data=data)
        st, ls = st0.copy(), []
        if pipelined:
            m.train_begin(st)
            pend = None
            for _ in range(6):
                x, y, mask, _w = ld.get_packed_batch()
                nxt = m.train_step_packed_async(x, y, mask, 1e-3)
                if pend is not None:
                    ls.append(pend.result()["loss"])
                pend = nxt
            ls.append(pend.result()["loss"])
            st = m.train_end()
        else:
            for _ in range(6):
                x, y, mask, _w = ld.get_packed_batch()
                out = m.train_step_packed(x, y, mask, st, 1e-3)
                st = out["state"]
                ls.append(out["loss"])
        losses.append(ls)
        states.append(np.asarray(st, np.float32))
        m.engine.close()
    np.testing.assert_allclose(losses[0], losses[1], rtol=1e-5)
    np.testing.assert_allclose(states[0], states[1], atol=1e-5)
    assert losses[0][-1] < losses[0][0]


@pytest.mark.parametrize("loop", ["hier", "gmm"])
def test_a_timed_out_persistent_launch_stops_the_epoch_within_twenty_batches(tmp_path, loop):
    """run_hier_xing.py's health watch: evaluate_hier (and run_gmm_xing.evaluate_hier_gmm, which shares it) over 60 batches against the test-hooks library,
    the scan of batch 3 timing out -- the loop raises at the first check behind it (batches 20-22), not at the end of the epoch, and the handle works again
    afterwards (tests/tools/health_watch_probe.py)."""
    import subprocess
    import sys
    from hiertcn_amd.build import LIB_TESTHOOKS
    root = os.path.dirname(HERE)
    env = dict(os.environ, HIERTCN_LIB=LIB_TESTHOOKS)
    env.pop("HIERTCN_EVAL_SERIAL", None)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "health_watch_probe.py"), str(tmp_path)] + (["gmm"] if loop == "gmm" else []), env=env,
                       capture_output=True, text=True,
                       timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("HEALTH_WATCH_PROBE")]
    assert r.returncode == 0 and len(lines) == 2, (r.stdout[-1500:], r.stderr[-1500:])
    batch = int(lines[0].split()[1])
    assert 20 <= batch <= 22, lines
    assert lines[1].endswith("recovered")
