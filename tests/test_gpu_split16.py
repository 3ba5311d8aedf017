"""-m gpu: the split-operand scoring kernels (HIERTCN_PRECISION_SPLIT16, Engine(dtype='split16'); csrc/score_x3.hip at a last width of 64 / 128 /
256, csrc/score_x3w.hip at 512 = BASELINE.json configs[3]) -- the fused
model_tcn.py:41 + model.py:105 + loss.py:21,120,179 on hi + lo IEEE-half parts, three MFMAs per product.

What is held here: (1) the fused outputs are EXACTLY the rank / ranked top-k of the kernel's own materialised logits (same MFMA sequence),
ties and masked rows included; (2) those logits are fp32-grade against an fp64 product, subnormal parts included; (3) the overflow fix-up,
the ragged-batch compaction, the small-batch catalogue split and the per-XCD item walk change no rank.  The comparison with the exact oracle
at the logit scale of a trained model (logits < 1e-3, ranks, top-10) is tests/test_gpu_exact_mode.py, parametrised over this mode.
"""
import numpy as np
import pytest

from oracle import hiertcn_oracle as O
from tests.gpu_common import make_batch, make_params, small_args

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")  # noqa: E731


def _engine(N, d, Wo, bo, seed=2):
    from hiertcn_amd.engine import Engine
    args = small_args(N=N, d=d, H=32, nl=1, blocks=1, K=3)
    p = make_params(args, seed=seed)
    p["hier/tcn/dense/kernel"], p["hier/tcn/dense/bias"] = Wo.astype(np.float32), bo.astype(np.float32)
    eng = Engine(args, hier=True, dtype="split16")
    eng.set_weights(p)
    return eng


@pytest.mark.parametrize("d,N,R", [(64, 1000, 700), (128, 2053, 1300), (256, 777, 515), (256, 5000, 2049), (512, 1500, 515), (512, 5000, 1300)])
@pytest.mark.parametrize("k", [0, 5, 10, 16])
def test_fused_outputs_are_exactly_those_of_the_kernels_own_logits(d, N, R, k):
    """rank = #{z_n > z_y} (loss.py:179), target = z_y, ranked top-k (loss.py:120: ties -> lower index) of the fused launch against NumPy on the
    logits flavour of the same kernel; ce against an fp64 log-sum-exp of those logits; rows without a target (0, out of range) -> 0 / 0 / 0..k-1.
    Exact ties are forced by duplicated item columns, R is no multiple of the 256-row tile (128 rows at d = 512)."""
    rng = np.random.default_rng(d + N + k)
    Wo = (rng.standard_normal((d, N)) * (4.0 / np.sqrt(d))).astype(np.float32)
    bo = (0.3 * rng.standard_normal(N)).astype(np.float32)
    for _ in range(40):
        i, j = rng.integers(0, N, 2)
        Wo[:, j], bo[j] = Wo[:, i], bo[i]
    for boost, (i, j) in zip((9.0, 8.5), rng.integers(1, N, (2, 2))):   # two duplicate pairs that lead every list
        bo[i] = boost
        Wo[:, j], bo[j] = Wo[:, i], bo[i]
    eng = _engine(N, d, Wo, bo)
    hid = np.maximum(rng.standard_normal((R, d)), 0).astype(np.float32)     # post-ReLU rows
    y = rng.integers(1, N, R).astype(np.int32)
    y[::9] = 0
    y[5] = N + 3
    hid_t = dev(hid)
    logits = eng.score_logits(hid_t).cpu().numpy()
    out = eng.score_fused(hid_t, dev(y), want_target=True, topk=k)
    valid = (y > 0) & (y < N)
    yy = np.where(valid, y, 0)
    tgt = logits[np.arange(R), yy]
    assert np.array_equal(out["rank"].cpu().numpy()[valid], (logits > tgt[:, None]).sum(1)[valid])
    assert np.array_equal(out["target"].cpu().numpy()[valid], tgt[valid])
    ce = out["ce"].cpu().numpy()
    ref = O.log_sum_exp(logits.astype(np.float64)) - tgt
    assert np.abs(ce - ref)[valid].max() < 2e-5
    assert not out["rank"].cpu().numpy()[~valid].any() and not ce[~valid].any()
    if k:
        want = O.top_k_indices(logits, k)
        ti, tv = out["topk_idx"].cpu().numpy(), out["topk_val"].cpu().numpy()
        assert np.array_equal(ti[valid], want[valid])
        assert np.array_equal(tv[valid], np.take_along_axis(logits, want, 1)[valid])
        vals = np.take_along_axis(logits, want, 1)[valid]
        assert (vals[:, :-1] == vals[:, 1:]).any(), "no tie in any list"
        assert np.array_equal(ti[~valid], np.broadcast_to(np.arange(k), ((~valid).sum(), k))) and not tv[~valid].any()
        plain = eng.score_fused(hid_t, dev(y))       # the list flavour's rank / ce agree with the plain flavour's (another item walk: ce to rounding)
        assert np.array_equal(plain["rank"].cpu().numpy(), out["rank"].cpu().numpy())
        assert np.abs(plain["ce"].cpu().numpy() - ce).max() < 2e-5
    eng.close()


@pytest.mark.parametrize("d", [64, 128, 256, 512])
@pytest.mark.parametrize("wmag,hmag", [(1.0, 1.0), (1e-6, 1e-5), (3e5, 1e6)], ids=["unit", "tiny", "huge"])
def test_split_logits_are_fp32_grade(d, wmag, hmag):
    """The three-product logits against the fp64 product of the SAME fp32 inputs: error of fp32 order relative to sum |h||w|, whatever the
    magnitude of the operands -- a half has 5 exponent bits, the kernel scales the table and every row by powers of two before the split
    ('tiny': every factor far below a half's normal range; 'huge': far above its maximum of 65504)."""
    N, R = 3000, 300
    rng = np.random.default_rng(d)
    Wo = (wmag * rng.standard_normal((d, N)) / np.sqrt(d)).astype(np.float32)
    bo = (wmag * hmag * rng.standard_normal(N)).astype(np.float32)
    hid = (hmag * rng.standard_normal((R, d))).astype(np.float32)
    hid[:50] *= 100.0                   # rows of different magnitude: the row scale is per row
    hid[50:100] *= 0.01
    eng = _engine(N, d, Wo, bo)
    logits = eng.score_logits(dev(hid)).cpu().numpy().astype(np.float64)
    exact = hid.astype(np.float64) @ Wo.astype(np.float64) + bo.astype(np.float64)
    scale = np.abs(hid).astype(np.float64) @ np.abs(Wo).astype(np.float64) + np.abs(bo)
    rel = np.abs(logits - exact) / scale
    print("d=%d |w|~%g |h|~%g: max error / sum|h||w| = %.3e (fp32 epsilon 6.0e-8)" % (d, wmag, hmag, rel.max()))
    assert np.isfinite(logits).all()
    assert rel.max() < 2e-6                    # 2^-23 per rounded part, 2^-22 per dropped product + the MFMA's fp32 accumulation (a plain fp32 dot product: <= d 2^-24)
    out = eng.score_fused(dev(hid), dev(rng.integers(1, N, R).astype(np.int32)), want_target=True, topk=5)   # and the fused flavour scales alike
    assert np.array_equal(out["topk_idx"].cpu().numpy(), O.top_k_indices(logits.astype(np.float32), 5))
    assert np.isfinite(out["ce"].cpu().numpy()).all()
    eng.close()


def test_factors_far_below_the_tensor_maximum_sit_on_kept_subnormals():
    """A factor below 2^-28 of the table's maximum lands on a half SUBNORMAL after the scaling: carried to 2^-39 of that maximum absolute.  The
    f16 MFMA must keep subnormal inputs for that (flushing them would zero these products altogether): columns whose weights are 1e-9 of the
    others', zero bias, against fp64 -- a few per cent error if kept, 100 % if flushed."""
    N, R, d = 2000, 200, 128
    rng = np.random.default_rng(5)
    Wo = (rng.standard_normal((d, N)) / np.sqrt(d)).astype(np.float32)
    Wo[:, 1000:1100] *= 1e-9
    bo = rng.standard_normal(N).astype(np.float32)
    bo[1000:1100] = 0
    hid = rng.standard_normal((R, d)).astype(np.float32)
    eng = _engine(N, d, Wo, bo)
    logits = eng.score_logits(dev(hid)).cpu().numpy().astype(np.float64)[:, 1000:1100]
    exact = (hid.astype(np.float64) @ Wo.astype(np.float64))[:, 1000:1100]
    floor = np.abs(Wo).max() * 2.0 ** -39 * np.abs(hid).sum(1)[:, None]       # sum_k |h_k| * (half a subnormal quantum / table scale)
    err = np.abs(logits - exact)
    print("subnormal columns: |z| ~ %.2e, max error %.2e, bound %.2e" % (np.abs(exact).mean(), err.max(), floor.max()))
    assert np.all(err <= floor + 1e-7 * np.abs(exact))
    assert np.median(err / np.abs(exact)) < 0.2       # kept (flushed: every one of them 1.0)
    eng.close()


@pytest.mark.parametrize("d", [64, 256, 512])
@pytest.mark.parametrize("case", ["single_term", "sum_of_terms", "late_spike"])
def test_softmax_sum_past_fp32_is_redone_by_the_fixup_launch(case, d):
    """The production flavour keeps the target logit + 64 as its softmax reference and leaves ce = +inf where the running sum leaves fp32
    (a logit > 133 nats above the target); the fix-up launch behind it redoes those row tiles with a running maximum (tests/test_gpu_parity.py
    has the same three shapes for the 16-bit kernels)."""
    N, R = 1500, 600
    rng = np.random.default_rng(11)
    u = np.zeros(d, np.float32)
    u[:8] = 1.0
    W = (0.01 * rng.standard_normal((d, N))).astype(np.float32)
    y = rng.integers(1, N // 2, R).astype(np.int32)
    W[:, 1:N // 2] += -8.0 * u[:, None] / 8.0
    if case == "single_term":
        W[:, N - 5] += 8.0 * u / 8.0
    elif case == "sum_of_terms":
        W[:, N // 2:N // 2 + 600] += 5.2 * u[:, None] / 8.0
    else:
        W[:, N // 2:N // 2 + 300] += 4.0 * u[:, None] / 8.0
        W[:, N - 3] += 9.0 * u / 8.0
    eng = _engine(N, d, W, np.zeros(N, np.float32))
    hid = np.zeros((R, d), np.float32)
    hid[:, :8] = 10.5 * (1.0 + 0.1 * rng.random((R, 8)))
    hid[300:] *= 0.05                                   # rows 300.. stay far inside fp32: their tiles (256-row tiles 1 and 2) mix both kinds
    hid_t = dev(hid)
    logits = eng.score_logits(hid_t).cpu().numpy().astype(np.float64)
    out = eng.score_fused(hid_t, dev(y), want_target=True, topk=5)
    ce, rank = out["ce"].cpu().numpy(), out["rank"].cpu().numpy()
    rows = np.arange(R)
    gap_log2 = (logits.max(1) - logits[rows, y]) * np.log2(np.e)
    assert gap_log2[:300].min() > 200 and gap_log2[300:].max() < 100
    ref = O.log_sum_exp(logits) - logits[rows, y]
    assert np.isfinite(ce).all()
    assert np.abs(ce - ref).max() < 1e-4 * max(1.0, np.abs(ref).max() / 100), np.abs(ce - ref).max()
    assert np.array_equal(rank, (logits > logits[rows, y][:, None]).sum(1))
    assert np.array_equal(out["topk_idx"].cpu().numpy(), O.top_k_indices(logits.astype(np.float32), 5))
    eng.close()


@pytest.mark.parametrize("d", [256, 512])
def test_forward_eval_compaction_and_catalogue_split_change_no_rank(d):
    """hiertcn_forward_eval in this mode: the fp32 low level, the device-side list of rows that carry a target, and -- at this batch size --
    (row tile x catalogue part) workgroups merged by k_score_merge, against the step-by-step calls on dense rows (one workgroup per row tile):
    identical ranks, ce to rounding, identical metrics; padded steps 0."""
    from hiertcn_amd.engine import Engine, metrics_dict
    N, B, S, L = 20_778, 32, 10, 20
    args = small_args(N=N, d=d, H=d, nl=2, blocks=2, K=5, L=L, S=S, B=B)
    p = make_params(args)
    p["hier/tcn/dense/kernel"] = (p["hier/tcn/dense/kernel"] * 40).astype(np.float32)
    rng = np.random.default_rng(7)
    x, y, mask = make_batch(rng, B, S, L, N, ragged=True, zipf=True)
    st0 = (0.2 * rng.standard_normal((B, 2 * d))).astype(np.float32)
    eng = Engine(args, hier=True, dtype="split16")
    eng.set_weights(p)
    out = eng.forward_eval(dev(x), dev(y), dev(mask), dev(st0), want_rows=True)
    pooled = eng.session_pool(dev(y.reshape(B * S, L))).view(B, S, -1)
    sproj, state, _ = eng.gru_scan(pooled, dev(mask), dev(st0))
    hid = eng.tcn_forward(dev(x.reshape(B * S, L)), sproj.view(B * S, -1))
    assert hid.dtype == torch.float32
    step = eng.score_fused(hid, dev(y.reshape(-1)))
    valid = y.reshape(-1) > 0
    assert np.array_equal(out["rank"].cpu().numpy().reshape(-1), step["rank"].cpu().numpy())
    assert np.abs(out["ce"].cpu().numpy().reshape(-1) - step["ce"].cpu().numpy()).max() < 2e-5
    assert not out["rank"].cpu().numpy().reshape(-1)[~valid].any() and not out["ce"].cpu().numpy().reshape(-1)[~valid].any()
    assert np.array_equal(out["state"].cpu().numpy(), state.cpu().numpy())
    m8 = eng.metrics(step["rank"], step["ce"], dev(y), B, S * L)
    a, b = metrics_dict(out["metrics8"]), metrics_dict(m8)
    for key in a:
        assert abs(a[key] - b[key]) < 1e-6 * max(1.0, abs(b[key])), (key, a[key], b[key])
    # and with lists: the list flavour scores the compacted rows without the catalogue split
    outk = eng.forward_eval(dev(x), dev(y), dev(mask), dev(st0), want_rows=True, topk=10)
    assert np.array_equal(outk["rank"].cpu().numpy(), out["rank"].cpu().numpy())
    logits = eng.score_logits(hid).cpu().numpy()
    want = O.top_k_indices(logits, 10)
    ti = outk["topk_idx"].cpu().numpy().reshape(-1, 10)
    assert np.array_equal(ti[valid], want[valid])
    assert np.array_equal(ti[~valid], np.broadcast_to(np.arange(10), ti[~valid].shape))
    eng.close()


@pytest.mark.parametrize("d,R", [(256, 300_000), (512, 100_000)])
def test_ranks_do_not_depend_on_the_item_walk_or_the_launch(d, R):
    """Large launch (several rounds of workgroups: later ones start their item walk where their XCD's neighbours are): ranks and targets
    identical across repeated launches and equal to the fixed-walk list flavour's; ce equal to rounding."""
    N = 60_000
    rng = np.random.default_rng(3)
    Wo = (rng.standard_normal((d, N)) * (3.0 / np.sqrt(d))).astype(np.float32)
    bo = (0.2 * rng.standard_normal(N)).astype(np.float32)
    eng = _engine(N, d, Wo, bo)
    hid = torch.relu(torch.randn((R, d), device="cuda:0", generator=torch.Generator("cuda:0").manual_seed(1)))
    y = torch.randint(1, N, (R,), device="cuda:0", dtype=torch.int32, generator=torch.Generator("cuda:0").manual_seed(2))
    a = eng.score_fused(hid, y, want_target=True)
    b = eng.score_fused(hid, y, want_target=True)
    c = eng.score_fused(hid, y, want_target=True, topk=8)      # lists: the walk starts at item 0 in every workgroup
    for o in (b, c):
        assert torch.equal(a["rank"], o["rank"]) and torch.equal(a["target"], o["target"])
        assert (a["ce"] - o["ce"]).abs().max().item() < 2e-5
    # a sample of rows against the kernel's own logits
    rows = torch.arange(0, R, 997, device="cuda:0")
    logits = eng.score_logits(hid[rows].contiguous())
    tgt = logits.gather(1, y[rows].long()[:, None])
    assert torch.equal((logits > tgt).sum(1).int(), a["rank"][rows])
    assert torch.equal(torch.topk(logits, 8, dim=1).values, c["topk_val"][rows])
    eng.close()


def test_eval_graph_replay_in_this_mode_matches_the_direct_call():
    """hiertcn_forward_eval in the split16 mode captured into a hipGraph (Engine.capture_eval_graph): nothing in it synchronises or allocates,
    and a replay gives the direct call's state and metrics bit for bit."""
    from hiertcn_amd.engine import Engine
    N, d, B, S, L = 3000, 128, 8, 4, 12
    args = small_args(N=N, d=d, H=d, nl=2, blocks=2, K=5, L=L, S=S, B=B)
    eng = Engine(args, hier=True, dtype="split16")
    eng.set_weights(make_params(args))
    rng = np.random.default_rng(1)
    x, y, mask = make_batch(rng, B, S, L, N, ragged=True)
    st0 = (0.2 * rng.standard_normal((B, 2 * d))).astype(np.float32)
    want = eng.forward_eval(dev(x), dev(y), dev(mask), dev(st0))
    g = eng.capture_eval_graph(B, S, L)
    g.x.copy_(dev(x)); g.y.copy_(dev(y)); g.mask.copy_(dev(mask)); g.state_in.copy_(dev(st0))
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(g.state_out, want["state"])
    assert torch.equal(g.metrics8[:6], want["metrics8"][:6]) or (g.metrics8 - want["metrics8"]).abs().max().item() < 1e-6   # (ce: the item walk's entry point)
    eng.close()


def test_split16_contract_errors():
    from hiertcn_amd.engine import Engine
    from hiertcn_amd._lib import HierTCNError
    args = small_args(N=300, d=64, H=64, nl=1, blocks=1, K=3, L=8, S=2, B=4)
    with pytest.raises(HierTCNError, match="EVALUATION"):
        Engine(args, hier=True, dtype="split16", train=True)
    wide = small_args(N=300, d=512, H=64, nl=1, blocks=1, K=3, L=8, S=2, B=4)   # every last width hiertcn_create takes has a split kernel (512: score_x3w.hip)
    Engine(wide, hier=True, dtype="split16").close()
    eng = Engine(args, hier=True)                      # a 16-bit handle whose weights are set has no split table
    eng.set_weights(make_params(args))
    with pytest.raises(HierTCNError, match="precede hiertcn_set_weights"):
        eng.set_precision("split16")
    eng.close()
    both = Engine(args, hier=True, dtype="split16")    # both parity modes on one handle: allocate before the weights, then switch freely
    both.set_precision("fp32")
    both.set_weights(make_params(args))
    hid = dev(np.random.default_rng(0).standard_normal((40, 64)).astype(np.float32))
    z32 = both.score_logits(hid)
    both.set_precision("split16")
    zs = both.score_logits(hid)
    assert (z32 - zs).abs().max().item() < 1e-5
    both.set_precision("bf16")
    with pytest.raises(HierTCNError):
        both.score_logits(hid)                         # the 16-bit kernels take 16-bit rows
    both.close()
