"""-m gpu: BASELINE.json configs[3] under pytest -- model_hier d = 512, 1,000,000-item catalogue, IEEE-half storage
(libhiertcn_gfx950_f16.so), the scoring-GEMM stress shape.

* the fused scoring kernels at the full catalogue width (a few thousand rows): rank and ranked top-10 bit-exact against a
  count / stable sort over the device's own materialised logits (loss.py:179, loss.py:120), cross entropy against an fp64
  log-sum-exp of them (loss.py:21), logits of a row sample against the oracle rounding to half at the same points
  (model_tcn.py:41);
* one whole `forward_eval` at d = 512 in half precision against the oracle at a small catalogue (every kernel of the step
  at this width: pool, scan, layered low level, scoring, lane metrics).
"""
import numpy as np
import pytest

from oracle import hiertcn_oracle as O
from tests.gpu_common import as_lists, make_batch, make_params, small_args

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")


def test_cfg4_scoring_fp16_d512_one_million_items():
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.engine import Engine
    N, d, R, K, SAMPLE = 1_000_000, 512, 3072, 10, 48
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5)
    rng = np.random.default_rng(4)
    # only the scoring layer matters here: glorot Wo / small non-zero bias; the other tensors keep the engine's contract
    p = O.init_params(args, seed=2)
    p["hier/tcn/dense/bias"] = (0.05 * rng.standard_normal(N)).astype(np.float32)
    # exact ties, forced through duplicated ITEMS: item b is a copy of item a (same Wo column, same bias); rows 1..16 have a or b as
    # their target -- the rank must not count the twin (strict '>', loss.py:179), a ranked list holds the lower index first
    # (loss.py:120 [TF-sem])
    twins = [(int(a), int(b)) for a, b in zip(rng.integers(1, N // 2, 8), rng.integers(N // 2, N, 8))]
    for a, b in twins:
        p["hier/tcn/dense/kernel"][:, b] = p["hier/tcn/dense/kernel"][:, a]
        p["hier/tcn/dense/bias"][b] = p["hier/tcn/dense/bias"][a]
    eng = Engine(args, hier=True, dtype="fp16")
    assert eng.lib.hiertcn_storage_dtype() == 1
    eng.set_weights(p)
    hid_f = (0.5 * rng.standard_normal((R, d))).astype(np.float32)
    hid = dev(hid_f).to(torch.float16).contiguous()
    y = rng.integers(1, N, R).astype(np.int32)
    y[::7] = 0                                       # masked rows inside the launch
    for i, (a, b) in enumerate(twins):
        y[1 + 2 * i], y[2 + 2 * i] = a, b
    out = eng.score_fused(hid, dev(y), want_target=True, topk=K)
    rank, ce, tgt = (out[k].cpu().numpy() for k in ("rank", "ce", "target"))
    ti, tv = out["topk_idx"].cpu().numpy(), out["topk_val"].cpu().numpy()
    plain = eng.score_fused(hid, dev(y), want_target=True)       # the production loss + rank flavour at this width (k_score_pipe<512,...>)
    assert np.array_equal(plain["rank"].cpu().numpy(), rank)
    assert np.array_equal(plain["target"].cpu().numpy(), tgt)
    # two kernels, two summation orders over 10^6 fp32 terms (log-sum-exp ~ 14): sqrt(n) * eps * lse ~ 1e-4
    assert np.abs(plain["ce"].cpu().numpy() - ce).max() < 5e-4, np.abs(plain["ce"].cpu().numpy() - ce).max()

    rows = np.unique(np.concatenate([np.arange(17), rng.integers(0, R, SAMPLE)]))
    logits = eng.score_logits(hid[torch.as_tensor(rows, device="cuda")].contiguous()).cpu().numpy()    # [rows, N] fp32
    yy = y[rows]
    v = yy > 0
    t_own = logits[np.arange(len(rows)), yy]
    assert np.array_equal(tgt[rows][v], t_own[v])                                  # fused target == streamed logit, bitwise
    assert np.array_equal(rank[rows][v], (logits > t_own[:, None]).sum(1)[v])      # loss.py:179 over own logits
    assert np.array_equal(ti[rows][v], O.top_k_indices(logits, K)[v])              # loss.py:120 over own logits
    assert np.array_equal(tv[rows][v], np.take_along_axis(logits, O.top_k_indices(logits, K), 1)[v])
    l64 = logits.astype(np.float64)
    mx = l64.max(1, keepdims=True)
    lse = (mx + np.log(np.exp(l64 - mx).sum(1, keepdims=True)))[:, 0]
    assert np.abs(ce[rows][v] - (lse - t_own)[v]).max() < 5e-4, np.abs(ce[rows][v] - (lse - t_own)[v]).max()   # loss.py:21
    assert not rank[~(y > 0)].any() and not ce[~(y > 0)].any()                     # model.py:105: masked rows
    assert np.array_equal(ti[~(y > 0)], np.broadcast_to(np.arange(K), ti[~(y > 0)].shape))

    # logits of the sample against the oracle: half-rounded Wo and hidden rows, fp64 accumulate (model_tcn.py:41)
    Wo = O.round_f16(p["hier/tcn/dense/kernel"]).astype(np.float64)                # [d, N]
    h16 = hid[torch.as_tensor(rows, device="cuda")].float().cpu().numpy().astype(np.float64)
    ref = h16 @ Wo + p["hier/tcn/dense/bias"].astype(np.float64)
    assert np.abs(logits - ref).max() < 1e-3, np.abs(logits - ref).max()           # tolerance: north star, fp32 logits
    eng.close()


def test_cfg4_width_forward_eval_fp16_vs_oracle():
    """Every kernel of one evaluation step at d = 512 in the IEEE-half build against the f16-mirror oracle (small catalogue)."""
    from hiertcn_amd.engine import Engine, metrics_dict
    N, d, B, S, L = 3000, 512, 5, 3, 12
    args = small_args(N=N, d=d, H=d, S=S, L=L, B=B)
    p = make_params(args)
    eng = Engine(args, hier=True, dtype="fp16")
    eng.set_weights(p)
    pr = O.round_params_bf16(p, E=d, rnd=O.round_f16)
    rng = np.random.default_rng(512)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.2 * rng.standard_normal((B, 2 * d))).astype(np.float32)
    out = eng.forward_eval(dev(x), dev(y), dev(mask), dev(st0), want_rows=True, topk=10)
    m = metrics_dict(out["metrics8"])
    xl, yl, ml = as_lists(x, y, mask)
    pred, st = O.model_hier(args, xl, yl, ml, st0, pr, O.MIRROR_F16)
    ref = O.masked_loss_and_metrics(pred, y.reshape(B, S * L), N)
    assert np.abs(out["state"].cpu().numpy() - st).max() < 1e-5
    assert abs(m["loss"] - ref["loss"]) < 1e-3
    for k in ("recall1", "recall5", "recall10", "mrr"):
        assert abs(m[k] - ref[k]) < 2e-2, (k, m[k], ref[k])
    pooled = eng.session_pool(dev(y.reshape(B * S, L))).view(B, S, -1)
    sproj, _, _ = eng.gru_scan(pooled, dev(mask), dev(st0))
    hid = eng.tcn_forward(dev(x.reshape(B * S, L)), sproj.view(B * S, -1))
    logits = eng.score_logits(hid).cpu().numpy().reshape(B, S * L, N)
    assert np.abs(logits - pred).max() < 1e-3
    yy = y.reshape(-1)
    v = yy > 0
    flat = logits.reshape(-1, N)
    assert np.array_equal(out["rank"].cpu().numpy().reshape(-1)[v], (flat > flat[np.arange(len(yy)), yy][:, None]).sum(1)[v])
    assert np.array_equal(out["topk_idx"].cpu().numpy().reshape(-1, 10)[v], O.top_k_indices(flat, 10)[v])
    eng.close()
