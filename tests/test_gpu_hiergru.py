"""HierGRU (model_low_type='gru', the reference's default low level: args.py:40, model_hier.py:67-68 -> model_gru.py:16-63)
on the HIP path against the CPU oracle, through the C-ABI."""
import copy

import numpy as np
import pytest

from oracle import hiertcn_oracle as O
from tests.gpu_common import as_lists, make_batch, make_params, small_args

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    return t.to("cuda:0", dtype=dtype) if dtype else t.to("cuda:0")


def gru_args(**kw):
    a = copy.copy(small_args(**kw))
    a.model_low_type = "gru"
    return a


@pytest.fixture(scope="module", params=[(64, 64, 1000, 2), (128, 64, 2053, 1), (256, 128, 777, 2)])
def setup(request):
    from hiertcn_amd.engine import Engine
    d, H, N, nl = request.param
    args = gru_args(N=N, d=d, H=H, nl=nl)
    p = make_params(args)
    assert "hier/gru/emb/kernel" in p and "hier/gru/rnn/multi_rnn_cell/cell_0/gru_cell/gates/kernel" in p
    eng = Engine(args, hier=True)
    eng.set_weights(p)
    pr = O.round_params_bf16(p, E=d)
    rng = np.random.default_rng(d + 1)
    B, S, L = 7, 3, 9
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, H * nl))).astype(np.float32)
    st0[:3] = 0.0   # a user's first session: zero state -> the leading null step does not count (model_gru.py:109-118)
    yield dict(args=args, p=p, pr=pr, eng=eng, x=x, y=y, mask=mask, st0=st0, B=B, S=S, L=L, N=N, d=d, H=H)
    eng.close()


def gpu_hidden(s):
    B, S, L = s["B"], s["S"], s["L"]
    eng = s["eng"]
    pooled = eng.session_pool(dev(s["y"].reshape(B * S, L))).view(B, S, -1)
    sproj, st_out, _ = eng.gru_scan(pooled, dev(s["mask"]), dev(s["st0"]))
    hid = eng.tcn_forward(dev(s["x"].reshape(B * S, L)), sproj.view(B * S, -1))
    return hid, st_out


def test_low_gru_hidden_matches_mirror_oracle(setup):
    s = setup
    hid, st = gpu_hidden(s)
    assert hid.shape == (s["B"] * s["S"] * s["L"], s["H"])
    got = hid.float().cpu().numpy().reshape(s["B"], s["S"], s["L"], -1)
    xl, yl, ml = as_lists(s["x"], s["y"], s["mask"])
    ref, st_ref = O.model_hier(s["args"], xl, yl, ml, s["st0"], s["pr"], O.MIRROR, return_hidden=True)
    ref = ref.reshape(got.shape)
    # [TF-sem] rows past the counted length are EXACT zeros on both sides
    assert np.array_equal(got == 0, ref == 0)
    # and the zero-state quirk is present in this fixture: a first-slot session whose last real step is zeroed
    first = np.abs(ref[:3, 0]).max(-1) > 0
    assert np.array_equal(first.sum(1), (s["x"][:3, 0] > 0).sum(1))   # length = COUNT of non-zero projected rows
    assert first[:, 0].all()                                            # ... applied as a prefix: step 0 (null id) is active
    diff = np.abs(got - ref)
    assert diff.max() <= 2.0 ** -7 and diff.mean() <= 1e-4   # |h| <= 1 (GRU output): a rare 1-ulp flip of the storage type
    assert np.abs(st.cpu().numpy() - st_ref).max() < 1e-5


def test_hiergru_forward_eval_matches_oracle(setup):
    from hiertcn_amd.engine import metrics_dict
    s = setup
    eng = s["eng"]
    B, S, L, N = s["B"], s["S"], s["L"], s["N"]
    out = eng.forward_eval(dev(s["x"]), dev(s["y"]), dev(s["mask"]), dev(s["st0"]), want_rows=True)
    m = metrics_dict(out["metrics8"])
    xl, yl, ml = as_lists(s["x"], s["y"], s["mask"])
    predm, st = O.model_hier(s["args"], xl, yl, ml, s["st0"], s["pr"], O.MIRROR)
    hid, _ = gpu_hidden(s)
    logits = eng.score_logits(hid).cpu().numpy().reshape(B, S * L, N)
    ymask = s["y"].reshape(B, S * L) > 0
    assert np.abs(logits - predm)[ymask].max() < 1e-3
    pred, _ = O.model_hier(s["args"], xl, yl, ml, s["st0"], s["pr"], O.EXACT64)
    assert np.abs(logits - pred)[ymask].max() < 4e-3 * max(1.0, np.abs(pred).max())
    assert np.abs(out["state"].cpu().numpy() - st).max() < 1e-5
    ref = O.masked_loss_and_metrics(predm, s["y"].reshape(B, S * L), N)
    assert abs(m["loss"] - ref["loss"]) < 1e-3
    tol = 2e-3
    tgt = np.take_along_axis(predm, s["y"].reshape(B, S * L, 1).astype(np.int64), -1)
    lo, hi = (predm > tgt + tol).sum(-1), (predm > tgt - tol).sum(-1)
    rk = out["rank"].cpu().numpy()
    assert np.all((rk >= lo)[ymask]) and np.all((rk <= hi)[ymask])
    for k in ("recall1", "recall5", "recall10", "mrr", "rank_frac"):
        assert abs(m[k] - ref[k]) < 2e-2, (k, m[k], ref[k])


def test_hiergru_model_hier_surface(setup):
    """model_hier(args, ...) with the reference's default model_low_type goes through the same surface."""
    from hiertcn_amd.model import HierTCNModel
    s = setup
    model = HierTCNModel(s["args"], params=s["p"])
    xl, yl, ml = as_lists(s["x"], s["y"], s["mask"])
    r = model.eval_step(xl, yl, ml, s["st0"])
    predm, st = O.model_hier(s["args"], xl, yl, ml, s["st0"], s["pr"], O.MIRROR)
    ref = O.masked_loss_and_metrics(predm, s["y"].reshape(s["B"], -1), s["N"])
    assert abs(r["loss"] - ref["loss"]) < 1e-3
    assert np.abs(np.asarray(r["state"]) - st).max() < 1e-5


# ---------------------------------------------------------------------------------------------------------------------
# training step of HierGRU against autograd (same contract as tests/test_gpu_train.py)
# ---------------------------------------------------------------------------------------------------------------------
def torch_loss_gru(args, P, x, y, mask, state, ste_bf16=None):
    """model_hier.py:39-93 with the model_gru.py:16-63 low level + model.py:102-117, on torch fp64 tensors; activations are
    rounded (straight-through) where the device rounds them: the projected rows and the GRU outputs fed to scoring
    (ste_bf16: that rounding; the fp32 training mode's test passes the identity -- nothing is rounded there)."""
    import torch.nn.functional as F
    if ste_bf16 is None:
        from tests.test_gpu_train import ste_bf16
    B, S, L = x.shape
    H, nl, N = args.hidden_dim, args.num_layer, args.item_num
    st = torch.as_tensor(state, dtype=torch.float64)
    E = P["hier/emb/kernel"]
    xt, yt = torch.as_tensor(x, dtype=torch.int64), torch.as_tensor(y, dtype=torch.int64)
    mk = torch.as_tensor(mask, dtype=torch.float64)

    def cell(scope, l, cur, h):
        pre = f"{scope}/multi_rnn_cell/cell_{l}/gru_cell"
        g = torch.sigmoid(torch.cat([cur, h], 1) @ P[f"{pre}/gates/kernel"] + P[f"{pre}/gates/bias"])
        r, uu = g.chunk(2, 1)
        c = torch.tanh(torch.cat([cur, r * h], 1) @ P[f"{pre}/candidate/kernel"] + P[f"{pre}/candidate/bias"])
        return uu * h + (1 - uu) * c

    logits = []
    for s in range(S):
        e = F.embedding(xt[:, s], E) * (xt[:, s] > 0)[..., None]
        u = ste_bf16(torch.cat([e, st[:, None, :].expand(-1, L, -1)], -1) @ P["hier/gru/emb/kernel"])
        length = (u.detach().abs().amax(2) > 0).sum(1)          # model_gru.py:109-118: a count, used as a prefix length
        hs = torch.zeros((B, H * nl), dtype=torch.float64)
        outs = []
        for t in range(L):
            active = (t < length)[:, None]
            cur, new = u[:, t], []
            for l in range(nl):
                cur = cell("hier/gru/rnn", l, cur, hs[:, l * H:(l + 1) * H])
                new.append(cur)
            hs = torch.where(active, torch.cat(new, 1), hs)
            outs.append(torch.where(active, cur, torch.zeros_like(cur)))
        a = ste_bf16(torch.stack(outs, 1))
        logits.append(a @ P["hier/gru/dense/kernel"] + P["hier/gru/dense/bias"])
        ey = F.embedding(yt[:, s], E) * (yt[:, s] > 0)[..., None]
        cur = ey.sum(1) / (yt[:, s] > 0).sum(1, keepdim=True) + P["hier/emb/bias"]
        new = []
        for l in range(nl):
            cur = cell("hier", l, cur, st[:, l * H:(l + 1) * H])
            new.append(cur)
        st = torch.cat(new, 1) * mk[:, s:s + 1]
    z = torch.cat(logits, 1)
    yy = yt.reshape(B, S * L)
    m = (yy > 0).double()
    z = z * m[..., None]
    ce = F.cross_entropy(z.reshape(-1, N), yy.reshape(-1), reduction="none").reshape(B, -1) * m
    act = m.sum(1)
    users = (act > 0).double().sum()
    return ((ce.sum(1) / (act + 1e-6)).sum() / users), st


# the last shape has 640 sessions per step: the row-reduced weight-gradient GEMMs of the low-level cells (csrc/gemm.hip, gemm32_tn) then run in two
# row slices whose partial tiles are summed in slice order
@pytest.mark.parametrize("d,H,N,nl,shape", [(64, 64, 700, 2, (6, 3, 8)), (128, 64, 1500, 1, (6, 3, 8)), (64, 64, 700, 2, (160, 4, 6)),
                                            (128, 128, 900, 2, (9, 4, 7)), (128, 128, 900, 1, (40, 2, 5)), (256, 256, 600, 2, (5, 3, 6)),
                                            (256, 256, 600, 2, (80, 8, 5)), (128, 128, 600, 2, (150, 8, 4))],
                         ids=["d64_nl2", "d128_nl1", "d64_nl2_640_sessions", "d128_h128_nl2", "d128_h128_nl1", "d256_h256_nl2",
                              "d256_640_sessions_two_launches", "d128_1200_sessions"])
def test_hiergru_training_step_matches_autograd(d, H, N, nl, shape):
    from hiertcn_amd.engine import Engine, metrics_dict
    args = gru_args(N=N, d=d, H=H, nl=nl)
    p = make_params(args)
    eng = Engine(args, hier=True, train=True)
    eng.set_weights(p)
    rng = np.random.default_rng(d + 7)
    B, S, L = shape
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, H * nl))).astype(np.float32)
    st0[:2] = 0.0
    out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    grads = eng.export("grads")
    loss = metrics_dict(out["metrics8"])["loss"]
    pr = O.round_params_bf16(p, E=d)
    P = {k: torch.tensor(np.asarray(v, dtype=np.float64), requires_grad=True) for k, v in pr.items()}
    ref_loss, ref_state = torch_loss_gru(args, P, x, y, mask, st0)
    ref_loss.backward()
    assert abs(loss - float(ref_loss.detach())) < 2e-3
    assert np.abs(out["state"].cpu().numpy() - ref_state.detach().numpy()).max() < 1e-3
    assert set(grads) == set(P)
    bad, report = [], []
    for k in sorted(P):
        g, r = grads[k].astype(np.float64), P[k].grad.numpy()
        nr = np.linalg.norm(r)
        rel = np.linalg.norm(g - r) / max(nr, 1e-30)
        cos = float((g * r).sum() / max(np.linalg.norm(g) * nr, 1e-30))
        report.append(f"{k}: rel {rel:.3e} cos {cos:.5f} |ref| {nr:.3e}")
        if nr > 1e-9 and (rel > 2e-2 or cos < 0.999):
            bad.append(k)
    assert not bad, "\n".join(report)
    # three Adam steps on the batch lower its loss, and the masters come back under the reference's variable names
    l0 = loss
    for _ in range(3):
        eng.adam_step(1e-2)
        out = eng.train_forward_backward(dev(x), dev(y), dev(mask), dev(st0))
    assert metrics_dict(out["metrics8"])["loss"] < l0
    assert set(eng.export("weights")) == set(p)
    eng.close()


def test_split_operand_cells_agree_with_the_fp32_cells():
    """csrc/gru_x3.hip (a.w = a_hi w_hi + a_lo w_hi + a_hi w_lo on the 16-bit MFMA, fp32 accumulators) against the exact-fp32 cells
    (hiertcn_tuning_t.lgru_fp32_cells = 1) over 30 steps and two layers: the 16-bit output rows are identical except where a ~2^-16 relative
    difference of the recurrence straddles a rounding boundary of the storage type -- rare, and never more than one ulp."""
    import os
    from hiertcn_amd.engine import Engine
    N, d, H, B, S, L = 1500, 128, 128, 16, 4, 30
    args = gru_args(N=N, d=d, H=H, nl=2, L=L, S=S, B=B)
    eng = Engine(args, hier=True)
    eng.set_weights(make_params(args))
    rng = np.random.default_rng(9)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = dev((0.3 * rng.standard_normal((B, 2 * H))).astype(np.float32))
    pooled = eng.session_pool(dev(y.reshape(B * S, L))).view(B, S, -1)
    sproj, _, _ = eng.gru_scan(pooled, dev(mask), st0)
    outs = []
    for fp32 in (False, True):
        eng.set_tuning(lgru_fp32_cells=1 if fp32 else -1)
        try:
            outs.append(eng.tcn_forward(dev(x.reshape(B * S, L)), sproj.view(B * S, -1)).float().cpu().numpy())
        finally:
            eng.set_tuning(lgru_fp32_cells=-1)
    a, b = outs
    assert np.abs(b).max() > 0.05
    diff = np.abs(a - b)
    assert (diff > 0).mean() < 0.01                                     # almost every 16-bit output is bit-identical
    assert diff.max() <= 2.0 ** -8 * np.abs(b).max() + 1e-12             # ... and no output differs by more than one ulp of the storage type
    eng.close()


@pytest.mark.parametrize("d,nl,B,S,L", [(64, 2, 32, 10, 20), (128, 2, 32, 10, 20), (256, 2, 32, 10, 20), (128, 1, 5, 3, 7), (128, 3, 60, 10, 12)])
def test_persistent_low_level_matches_the_launch_chain_and_the_oracle(d, nl, B, S, L):
    """Small evaluation batches run the HierGRU low level as ONE persistent launch (csrc/gru_persist.hip, k_low_gru_persist: E == H, <= 1024
    sessions); HIERTCN_LGRU_CHAIN=1 keeps the chain of split-operand cell launches.  Same rows from both (a storage-type ulp where the cells'
    2^-16 products differ from the exact fp32 ones), the mirror oracle's zeros past sequence_length, and the fp32 mode's rows to 1e-5."""
    import os
    from hiertcn_amd.engine import Engine
    N = 900
    args = gru_args(N=N, d=d, H=d, nl=nl, L=L, S=S, B=B)
    p = make_params(args)
    rng = np.random.default_rng(d + nl)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, d * nl))).astype(np.float32)
    st0[:3] = 0.0
    xl, yl, ml = as_lists(x, y, mask)
    ref, _ = O.model_hier(args, xl, yl, ml, st0, O.round_params_bf16(p, E=d), O.MIRROR, return_hidden=True)
    ref = ref.reshape(B * S * L, d)
    ref64, _ = O.model_hier(args, xl, yl, ml, st0, p, O.EXACT64, return_hidden=True)
    ref64 = ref64.reshape(B * S * L, d)

    def rows(dtype):
        eng = Engine(args, hier=True, dtype=dtype)
        eng.set_weights(p)
        pooled = eng.session_pool(dev(y.reshape(B * S, L))).view(B, S, -1)
        sproj, _, _ = eng.gru_scan(pooled, dev(mask), dev(st0))
        hid = eng.tcn_forward(dev(x.reshape(B * S, L)), sproj.view(B * S, -1)).float().cpu().numpy()
        out = eng.forward_eval(dev(x), dev(y), dev(mask), dev(st0), want_rows=True)
        rk = out["rank"].cpu().numpy()
        assert not eng.persistent_timed_out()
        eng.close()
        return hid, rk

    assert os.environ.get("HIERTCN_LGRU_CHAIN") is None and os.environ.get("HIERTCN_LGRU_TILE") is None
    tile, rk_tile = rows("bf16")                 # default: the session-tile kernel (csrc/lgru_tile.hip) where it applies (E == H <= 128), else as below
    os.environ["HIERTCN_LGRU_TILE"] = "0"
    os.environ["HIERTCN_LGRU_CHAIN"] = "0"       # the persistent launch wherever it fits (by default not at H = 256, where it is no faster)
    try:
        one, rk_one = rows("bf16")
        one32, _ = rows("fp32")
        os.environ["HIERTCN_LGRU_CHAIN"] = "1"
        chain, rk_chain = rows("bf16")
        chain32, _ = rows("fp32")
    finally:
        os.environ.pop("HIERTCN_LGRU_CHAIN")
        os.environ.pop("HIERTCN_LGRU_TILE")
    assert np.array_equal(one == 0, ref == 0)                      # [TF-sem] exact zeros past the counted length, from the one launch too
    assert np.array_equal(tile == 0, ref == 0)
    assert (rk_tile == rk_chain).mean() > 0.97
    for got in (one, chain, tile):
        diff = np.abs(got - ref)
        assert diff.max() <= 2.0 ** -7 and diff.mean() <= 1e-4, (diff.max(), diff.mean())
    assert np.abs(one - chain).max() <= 2.0 ** -7
    assert (rk_one == rk_chain).mean() > 0.97                      # ranks move only where a storage-type ulp moved a near-tie
    assert np.abs(one32 - ref64).max() < 1e-5 and np.abs(chain32 - ref64).max() < 1e-5
    assert np.abs(one32 - chain32).max() < 1e-5
    assert not np.array_equal(one32, chain32)                      # two different kernels did run (fp32 dot products in another summation order)


@pytest.mark.parametrize("d,nl,B,S,L", [(64, 2, 13, 3, 9), (128, 1, 7, 3, 20), (128, 2, 32, 10, 20), (128, 2, 41, 5, 11),
                                        (256, 2, 32, 10, 20), (256, 1, 9, 3, 7), (256, 2, 11, 5, 9), (256, 2, 64, 10, 12), (256, 2, 140, 10, 6)])
def test_layer_pipelined_low_level_matches_the_session_tile_kernel(d, nl, B, S, L):
    """H <= 128: one workgroup per (32-session tile, LAYER) with the layer's weights resident in registers and the layers handing their state
    rows over through global slots + flags inside ONE launch (csrc/lgru_pipe.hip); HIERTCN_LGRU_PIPE=0 keeps the session-tile kernel (all layers in
    one workgroup, no hand-off).  Same arithmetic: the rows may differ by a storage-type ulp where the compiler contracted the blend differently
    (a handful per launch); zeros past sequence_length identical; 30 repeated launches of the pipelined kernel are bitwise identical to the first
    (a hand-off read too early or torn would show as a difference); the time-out word stays clear.  d = 256: csrc/lgru_wide.hip -- eight
    workgroups per (tile, layer), each keeping one unit block's weights with the reduction split over its four waves, two all-to-all hand-offs per
    cell (HIERTCN_LGRU_WIDE=0: the tile kernel); its sums over k are four partial sums, so a few more one-ulp differences are allowed.  640 and 1400
    sessions at d = 256 are more tiles than one launch keeps resident: two resp. three launches over consecutive tile ranges."""
    import os
    from hiertcn_amd.engine import Engine
    N = 900
    args = gru_args(N=N, d=d, H=d, nl=nl, L=L, S=S, B=B)
    p = make_params(args)
    rng = np.random.default_rng(7 * d + nl + B)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, d * nl))).astype(np.float32)
    eng = Engine(args, hier=True)
    eng.set_weights(p)
    pooled = eng.session_pool(dev(y.reshape(B * S, L))).view(B, S, -1)
    sproj, _, _ = eng.gru_scan(pooled, dev(mask), dev(st0))
    xs, sp = dev(x.reshape(B * S, L)), sproj.view(B * S, -1)

    def rows():
        h = eng.tcn_forward(xs, sp)
        torch.cuda.synchronize()
        return h.view(torch.int16).cpu().numpy().copy()

    assert os.environ.get("HIERTCN_LGRU_PIPE") is None and os.environ.get("HIERTCN_LGRU_WIDE") is None
    pipe = rows()
    for _ in range(30):
        assert np.array_equal(rows(), pipe)
    eng.set_tuning(lgru_pipe=0, lgru_wide=0)     # hiertcn_tuning_t: the session-tile kernel instead
    tile = rows()
    assert not eng.persistent_timed_out()
    eng.close()
    assert np.array_equal(pipe == 0, tile == 0)
    a = torch.from_numpy(pipe).view(torch.bfloat16).float().numpy()
    b = torch.from_numpy(tile).view(torch.bfloat16).float().numpy()
    diff = np.abs(a - b)
    assert (diff > 0).mean() < (2e-2 if d == 256 else 2e-3) and diff.max() <= 2.0 ** -8   # |h| < 1: one storage-type ulp at most


@pytest.mark.parametrize("d", [128, 256])
def test_in_launch_hand_offs_beside_a_long_kernel_on_another_stream(d):
    """csrc/lgru_pipe.hip (d = 128) and csrc/lgru_wide.hip (d = 256) hand rows between workgroups of ONE launch, so a consumer can be resident
    while a producer still waits for a CU that a long kernel on another stream holds.  Producers carry the lower block ids of their tile (they
    are dispatched first) and every wait is bounded: the rows are bitwise those of the undisturbed launch and the time-out word stays clear."""
    from hiertcn_amd.engine import Engine
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.params import init_params
    N, nl, B, S, L = 900, 2, 32, 10, 20
    args = gru_args(N=N, d=d, H=d, nl=nl, L=L, S=S, B=B)
    rng = np.random.default_rng(100 + d)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.3 * rng.standard_normal((B, d * nl))).astype(np.float32)
    eng = Engine(args, hier=True)
    eng.set_weights(make_params(args))
    pooled = eng.session_pool(dev(y.reshape(B * S, L))).view(B, S, -1)
    sproj, _, _ = eng.gru_scan(pooled, dev(mask), dev(st0))
    xs, sp = dev(x.reshape(B * S, L)), sproj.view(B * S, -1)
    want = eng.tcn_forward(xs, sp).clone()
    big = hier_tcn_args(item_num=120_000, emb_dim=256, hidden_dim=256, tcn_channel=[256, 256], num_layer=2)
    eb = Engine(big)
    eb.set_weights(init_params(big, 6))
    rows = 600_000
    hid = (torch.randn(rows, 256, device="cuda") * 0.5).to(torch.bfloat16)
    yy = torch.randint(1, 120_000, (rows,), device="cuda", dtype=torch.int32)
    eb.score_fused(hid, yy)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(side):
            eb.score_fused(hid, yy)               # ~30 ms of every CU
        got = eng.tcn_forward(xs, sp)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16), want.view(torch.int16))
    assert not eng.persistent_timed_out()
    eng.close()
    # the training step of the same batch: forward scan with records, wide backward cells (csrc/lgru_wide.hip), persistent BPTT
    from hiertcn_amd.engine import metrics_dict
    tr = Engine(args, hier=True, train=True)
    tr.set_weights(make_params(args))
    dx, dy, dm, ds = dev(x), dev(y), dev(mask), dev(st0)
    out = tr.train_forward_backward(dx, dy, dm, ds)
    want_loss, want_state = metrics_dict(out["metrics8"])["loss"], out["state"].clone()
    want_g = tr.export("grads")
    for _ in range(2):
        with torch.cuda.stream(side):
            eb.score_fused(hid, yy)
        out = tr.train_forward_backward(dx, dy, dm, ds)
        torch.cuda.synchronize()
        assert abs(metrics_dict(out["metrics8"])["loss"] - want_loss) < 1e-5
        assert torch.allclose(out["state"], want_state, atol=1e-6)
        got_g = tr.export("grads")
        for k in want_g:
            den = max(float(np.linalg.norm(want_g[k])), 1e-12)
            assert float(np.linalg.norm(got_g[k] - want_g[k])) / den < 1e-4, k    # fp32 atomics in a few reductions: order only
    assert not tr.persistent_timed_out()
    tr.close()
    eb.close()
