"""Pins oracle/cpu_baseline.py (torch-CPU, the timed 'reference CPU path') and
hiertcn_amd/params.py (product-side initialiser) to the NumPy oracle."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from hiertcn_amd import params as P  # noqa: E402
from hiertcn_amd.args import hier_tcn_args, make_args  # noqa: E402
from oracle import hiertcn_oracle as O  # noqa: E402
from oracle.cpu_baseline import CpuHierTCN, time_cpu_baseline  # noqa: E402
from tests.gpu_common import as_lists, make_batch, make_params  # noqa: E402


def test_product_init_matches_oracle_init():
    args = hier_tcn_args(item_num=301, emb_dim=32, hidden_dim=16, tcn_channel=[32, 32])
    a, b = P.init_params(args, seed=2), O.init_params(args, seed=2)
    assert list(a) == list(b)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    assert P.count_params(P.init_params(hier_tcn_args(), seed=0)) == 5_750_698
    t = make_args([], model_type="tcn")
    assert list(P.init_params(t, 0, "tcn")) == list(O.init_params(t, 0, "tcn"))
    # HierGRU, the reference's default low level (args.py:40): same variable names / shapes / values on both sides, and the
    # trainable-parameter count of its default graph: E 2,659,584 + b_emb 128 + gru/emb 49,152 + low GRU 197,376 + high GRU
    # 197,376 + scoring dense 2,680,362
    g = make_args([], model_type="hier")
    assert g.model_low_type == "gru"
    a, b = P.init_params(g, seed=2), O.init_params(g, seed=2)
    assert sorted(a) == sorted(b) and all(np.array_equal(a[k], b[k]) for k in a)
    assert P.count_params(a) == 5_783_978
    assert a["hier/gru/rnn/multi_rnn_cell/cell_1/gru_cell/candidate/kernel"].shape == (256, 128)


def test_cpu_baseline_matches_oracle():
    args = hier_tcn_args(item_num=211, emb_dim=16, hidden_dim=8, tcn_channel=[16, 16])
    p = make_params(args)
    rng = np.random.default_rng(3)
    B, S, L = 6, 3, 7
    x, y, mask = make_batch(rng, B, S, L, 211)
    st0 = rng.standard_normal((B, 16)).astype(np.float32)
    m, st = CpuHierTCN(args, p).eval_batch(torch.from_numpy(x.astype(np.int64)), torch.from_numpy(y.astype(np.int64)),
                                           torch.from_numpy(mask), torch.from_numpy(st0))
    xl, yl, ml = as_lists(x, y, mask)
    pred, st_ref = O.model_hier(args, xl, yl, ml, st0, p, O.EXACT64)
    ref = O.masked_loss_and_metrics(pred, y.reshape(B, S * L), 211)
    assert np.abs(st.numpy() - st_ref).max() < 1e-5
    for k in ("loss", "recall1", "recall5", "recall10", "mrr", "rank_frac"):
        assert abs(m[k] - float(ref[k])) < 1e-4, k


def test_time_cpu_baseline_runs_bounded():
    """The budget shrinks the SAMPLE (slots, then lanes in steps of 32), never the repeats: always 3 timed runs after a warm-up."""
    args = hier_tcn_args(item_num=500, emb_dim=16, hidden_dim=8, tcn_channel=[16, 16])
    p = make_params(args)
    x, y, mask = make_batch(np.random.default_rng(0), 64, 2, 5, 500, ragged=False)
    r = time_cpu_baseline(args, p, x, y, mask, variants=((32, 2), (64, 1)), repeats=3, budget_s=1e-6)
    assert r["value"] > 0 and r["cores"] >= 1
    assert [(v["lanes"], v["slots"], v["timed_runs"]) for v in r["variants"]] == [(32, 1, 3), (32, 1, 3)]   # both shrunk to the floor
    r = time_cpu_baseline(args, p, x, y, mask, variants=((32, 2), (64, 1)), repeats=1, budget_s=60.0)
    assert [(v["lanes"], v["slots"], v["timed_runs"]) for v in r["variants"]] == [(32, 2, 3), (64, 1, 3)]   # fits: full sample, >= 3 repeats
    assert r["sessions"] == 64
