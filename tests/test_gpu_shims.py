"""-m gpu: the reference-signature surface (model.py:85,105-132 as a caller would write it): model_hier / model_tcn
builders, calc_loss, calc_metric_fast, calc_metric (slow, full sort), calc_metric_topk -- against the oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import hiertcn_oracle as O  # noqa: E402
from tests.gpu_common import as_lists, make_batch, make_params, small_args  # noqa: E402

pytestmark = pytest.mark.gpu


def test_model_hier_and_loss_functions_like_model_py():
    from hiertcn_amd import model as M
    from hiertcn_amd.loss import activity_and_user_count, calc_loss, calc_metric, calc_metric_fast, calc_metric_topk
    from hiertcn_amd.model_hier import model_hier
    M.reset_scopes()
    args = small_args(N=600, d=64, H=32)
    p = make_params(args)
    M.set_scope("hier", M.HierTCNModel(args, p, "hier"))
    rng = np.random.default_rng(2)
    B, S, L = 5, 3, 8
    x, y, mask = make_batch(rng, B, S, L, 600)
    # ragged slot widths, as the loader emits them (data_loader.py:246-261)
    widths = [int(max(1, (y[:, s] > 0).sum(1).max())) for s in range(S)]
    xl = [x[:, s, :w] for s, w in enumerate(widths)]
    yl = [y[:, s, :w] for s, w in enumerate(widths)]
    ml = [mask[:, s:s + 1] for s in range(S)]
    st0 = (0.1 * rng.standard_normal((B, 64))).astype(np.float32)
    # ---- model.py:85
    pred, state = model_hier(args, xl, yl, ml, st0)
    y_id = np.concatenate(yl, 1)                                   # run_hier_xing.py:116
    assert pred.shape == (B, sum(widths), 600)
    # ---- model.py:105-117
    mask_y, activity_count, user_count = activity_and_user_count(y_id)
    loss_bt = calc_loss(pred, y_id).cpu()
    loss = float(((loss_bt * mask_y).sum(1) / activity_count).sum() / user_count)
    # ---- model.py:131-132
    r1, r5, r10, mrr, rf, ranks_float, ranks = calc_metric_fast(pred, mask_y, activity_count, user_count, y_id)
    pr = O.round_params_bf16(p, E=64)
    ref_pred, ref_state = O.model_hier(args, xl, yl, ml, st0, pr, O.MIRROR)
    ref = O.masked_loss_and_metrics(ref_pred, y_id, 600)
    assert np.abs(state.cpu().numpy() - ref_state).max() < 1e-5
    assert abs(loss - ref["loss"]) < 1e-3
    logits = pred.materialize().cpu().numpy()                      # parity/debug: real [B, sum L_i, N] logits
    assert np.abs(logits - ref_pred).max() < 1e-3
    own = logits * (y_id[..., None] > 0)
    cnt = (own > np.take_along_axis(own, y_id[..., None].astype(np.int64), -1)).sum(-1) * (y_id > 0)
    assert np.array_equal(ranks.cpu().numpy().astype(np.int64), cnt)
    assert abs(float(r10) - ref["recall10"]) < 2e-2 and abs(float(mrr) - ref["mrr"]) < 2e-2
    # ---- loss.py:120 ranked candidates: fused (k <= 16) and the slow full sort agree with a stable sort of the logits
    ti, tv = calc_metric_topk(pred, y_id, 10)
    want = O.top_k_indices(own, 10)
    v = y_id > 0
    assert np.array_equal(ti.cpu().numpy()[v], want[v])
    slow = calc_metric(pred, mask_y, activity_count, user_count, y_id)
    assert np.array_equal(slow[6].cpu().numpy()[..., :10][v], want[v])          # ranks_indices
    assert np.array_equal(slow[8].cpu().numpy()[v], ranks.cpu().numpy()[v])     # no ties above the target in random data
    M.reset_scopes()


def test_model_tcn_single_level_like_model_py():
    from hiertcn_amd import model as M
    from hiertcn_amd.args import make_args
    from hiertcn_amd.loss import activity_and_user_count, calc_loss, calc_metric_fast
    from hiertcn_amd.model_tcn import model_tcn
    from hiertcn_amd.params import init_params
    M.reset_scopes()
    a = make_args([], model_type="hier", item_num=500, emb_dim=64, tcn_channel=[64, 64, 128], kernel_size=3)
    a.model_type = "tcn"
    p = init_params(a, 2, "tcn")
    M.set_scope("tcn", M.TcnModel(a, p, "tcn"))
    rng = np.random.default_rng(4)
    B, T = 4, 45
    y = (rng.integers(1, 500, (B, T)) * (np.arange(T)[None, :] < rng.integers(5, T + 1, B)[:, None])).astype(np.int64)
    x = np.zeros_like(y)
    x[:, 1:] = y[:, :-1]
    pred = model_tcn(a, x)                                         # model.py:73
    mask_y, activity_count, user_count = activity_and_user_count(y)
    loss_bt = calc_loss(pred, y).cpu()
    loss = float(((loss_bt * mask_y).sum(1) / activity_count).sum() / user_count)
    r1, r5, r10, mrr, rf, ranks_float, ranks = calc_metric_fast(pred, mask_y, activity_count, user_count, y)
    z = O.model_tcn(a, x, O.round_params_bf16(p, E=64), "tcn", O.MIRROR)
    ref = O.masked_loss_and_metrics(z, y, 500)
    assert abs(loss - ref["loss"]) < 1e-3 and abs(float(rf) - ref["rank_frac"]) < 5e-3
    assert np.abs(pred.materialize().cpu().numpy() - z).max() < 1e-3
    M.reset_scopes()


def test_the_tuning_of_a_call_in_progress_never_outlives_the_call():
    """csrc/common.h TuneScope.  tests/tools/tune_scope_probe.py runs against the test-hooks build (the only one that exports the
    probe symbol) in a process of its own; the shipped libraries do not export it."""
    import os
    import subprocess
    import sys

    from hiertcn_amd.build import LIB, LIB_F16, LIB_TESTHOOKS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.path.exists(LIB_TESTHOOKS), "python -m hiertcn_amd.build (or __graft_entry__.build()) builds the test-hooks library"
    env = dict(os.environ, HIERTCN_LIB=LIB_TESTHOOKS)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "tune_scope_probe.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "tune_scope_probe ok" in r.stdout
    for path in (LIB, LIB_F16):
        assert b"hiertcn_test_tune_in_scope" not in open(path, "rb").read(), path
