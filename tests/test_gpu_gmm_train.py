"""-m gpu: the TRAINING step of the mixture-output model (csrc/gmm_train.hip, C-ABI 12; model.py:135-141 on the lane-mean mixture loss of
model.py:111-117 + loss.py:247-257).  The reference cannot train these models on XING (its graph does not build: SURVEY.md 8(f) 4), so the
step is pinned the way the fp32 training mode is: every gradient against fp64 AUTOGRAD of an independent torch formulation that MATERIALISES
what the device never forms -- the one-hot inputs' row products as gathers, pred = h . W + b [B, T, M (1 + D)], the decode of
oracle/gmm_oracle.py, |y - mu_m|^2 over all D outputs -- whose forward is itself checked against the NumPy oracle; then an Adam trajectory
against an fp64 TF-form Adam, and the evaluation path of the same handle after hiertcn_gmm_train_sync."""
import numpy as np
import pytest

from hiertcn_amd.args import make_args
from oracle import gmm_oracle as G

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

dev = lambda v: torch.as_tensor(np.ascontiguousarray(v)).to("cuda:0")  # noqa: E731


def _args(N, M, channels, K, loss="gmm_prob", sigma=0.1):
    return make_args([], model_type="tcn_gmm", item_num=N, output_dim=N, num_mixture=M, sigma=sigma, loss=loss, tcn_channel=list(channels),
                     kernel_size=K, adjust=False)


def _batch(rng, N, B, T):
    ids = rng.integers(1, N, (B, T + 1))
    lens = rng.integers(2, T + 1, B)
    ids = ids * (np.arange(T + 1)[None, :] <= lens[:, None])           # right padding with the null id
    x, y = ids[:, :-1].astype(np.int32), ids[:, 1:].astype(np.int32)
    x[:, 0] = 0                                                          # a sequence starts from the null id (data_loader.py:246-261)
    y[B - 1] = 0                                                         # a lane without any target: no user of this batch (model.py:112-113)
    return x, y


def _params(a, seed):
    p = G.init_params(a, seed=seed)
    rng = np.random.default_rng(seed + 100)
    for k in p:   # non-zero biases so that a dropped or misplaced bias term shows in the gradients
        if k.endswith("bias") or k.endswith("biases"):
            p[k] = (0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    return p


def _mix_rows(a, pred, yt, mode):
    """The decode of oracle/gmm_oracle.py and the two likelihood losses on a MATERIALISED head output pred [B, T, n_out] (torch, fp64): fixed scale
    (args.sigma) or a learned one (args.sigma None: the third block of outputs as log-scales, sigma_d = exp(s_d)) -> loss rows [B, T] (unmasked)."""
    M, D = int(a.num_mixture), int(a.output_dim)
    B, T = yt.shape
    log_pi = torch.log_softmax(pred[..., :M], -1)
    mu = pred[..., M:M + M * D].reshape(B, T, M, D)
    yoh = torch.nn.functional.one_hot(yt, D).to(torch.float64) * (yt > 0)[..., None]
    if getattr(a, "sigma", 0.1):
        sigma = float(a.sigma)
        d2 = ((yoh[:, :, None, :] - mu) ** 2).sum(-1)
        mvn = -0.5 * d2 / sigma ** 2 - D * np.log(sigma) - 0.5 * D * np.log(2 * np.pi)
    else:
        sc = pred[..., M + M * D:M + 2 * M * D].reshape(B, T, M, D)
        mvn = -0.5 * (((yoh[:, :, None, :] - mu) * torch.exp(-sc)) ** 2).sum(-1) - sc.sum(-1) - 0.5 * D * np.log(2 * np.pi)
    return -torch.logsumexp(log_pi + mvn, -1) / D if mode == "gmm_prob" else -(log_pi + mvn / D).max(-1).values


def torch_loss(a, P, x, y, mode, scope="tcn_gmm"):
    """fp64, materialising: -> (batch loss, loss rows [B, T]).  P: {TF name: float64 tensor (requires_grad)}."""
    xt, yt = torch.as_tensor(x.astype(np.int64)), torch.as_tensor(y.astype(np.int64))
    B, T = xt.shape
    K = int(a.kernel_size)
    pre = f"{scope}/temporal_conv_net/tblock_0"
    W0, b0, Wr, br = P[f"{pre}/conv1/kernel"], P[f"{pre}/conv1/bias"], P[f"{pre}/dense/kernel"], P[f"{pre}/dense/bias"]
    live_x = (xt > 0).to(torch.float64)[..., None]
    c = b0.expand(B, T, -1).clone()
    for j in range(K):
        s = K - 1 - j
        rows = W0[j][xt] * live_x                                      # a one-hot row times the kernel = its row; the null id is an all-zero row
        c = c + (torch.nn.functional.pad(rows, (0, 0, s, 0))[:, :T] if s else rows)
    act = torch.relu(torch.relu(c) + Wr[xt] * live_x + br)
    for i in range(1, len(a.tcn_channel)):
        pre = f"{scope}/temporal_conv_net/tblock_{i}"
        W, b = P[f"{pre}/conv1/kernel"], P[f"{pre}/conv1/bias"]      # [K, Cin, Cout]
        dil = 2 ** i
        c = b.expand(B, T, -1).clone()
        for j in range(K):
            s = (K - 1 - j) * dil
            if s >= T:
                continue
            z = act @ W[j]
            c = c + (torch.nn.functional.pad(z, (0, 0, s, 0))[:, :T] if s else z)
        res = act @ P[f"{pre}/dense/kernel"] + P[f"{pre}/dense/bias"] if f"{pre}/dense/kernel" in P else act
        act = torch.relu(torch.relu(c) + res)
    pred = act @ P[f"{scope}/fully_connected/weights"] + P[f"{scope}/fully_connected/biases"]      # [B, T, M (1 + D)]: what the device never forms
    rows = _mix_rows(a, pred, yt, mode)
    mask = (yt > 0).to(torch.float64)
    cnt = mask.sum(1)
    users = (cnt > 0).sum()
    loss = ((rows * mask).sum(1) / (cnt + 1e-6)).sum() / users
    return loss, rows * mask


CASES = [
    # N (= D), M, channels, K, B, T
    (303, 3, (64, 64), 3, 5, 9),          # D not a multiple of 4 (pad rows / unaligned head columns), identity residual in block 1
    (500, 5, (64, 128), 5, 4, 12),        # Dense residual in block 1 (Cin != Cout)
    (1201, 2, (128,), 2, 6, 7),           # one block: the hidden rows are block 0's output
    (777, 4, (64, 64, 128), 3, 3, 20),    # three blocks, dilation 4 reaches past short sequences
    (20_778, 5, (256, 256), 5, 3, 11),    # the XING catalogue at the reference's widths
]


@pytest.mark.parametrize("mode", ["gmm_prob", "gmm_prob_max"])
@pytest.mark.parametrize("N,M,channels,K,B,T", CASES)
def test_every_gradient_against_fp64_autograd(N, M, channels, K, B, T, mode):
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    a = _args(N, M, channels, K, loss=mode)
    p = _params(a, seed=N % 97)
    rng = np.random.default_rng(N + M + len(channels))
    x, y = _batch(rng, N, B, T)
    # reference: fp64 autograd of the materialising formulation, whose forward agrees with the NumPy oracle
    P = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in p.items()}
    loss_ref, rows_ref = torch_loss(a, P, x, y, mode)
    loss_ref.backward()
    pred = G.model_tcn_gmm(a, x, {k: np.asarray(v, np.float64) for k, v in p.items()})
    want_rows = G.calc_loss_gmm_ids(pred, y, a) * (y > 0)
    assert np.abs(rows_ref.detach().numpy() - want_rows).max() < 1e-9 * max(1.0, np.abs(want_rows).max())
    # device
    model = TcnGmmModel(a, p)
    model.enable_training()
    out = model.train_forward_backward(dev(x), dev(y), mode, want_rows=True)
    sums = out["sums"].cpu().numpy()
    rows = out["loss_rows"].cpu().numpy()
    live = y > 0
    assert not rows[~live].any()
    assert np.abs(rows - want_rows)[live].max() < 2e-6 * max(1.0, np.abs(want_rows).max()), np.abs(rows - want_rows)[live].max()
    assert sums[1] == B - 1 and abs(sums[0] / sums[1] - float(loss_ref.detach())) < 2e-6 * max(1.0, abs(float(loss_ref.detach())))
    grads = model.export("grads")
    worst = {}
    for k, t in P.items():
        ref = t.grad.numpy()
        got = grads[k].astype(np.float64)
        assert got.shape == ref.shape, k
        err = np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30)
        worst[k] = err
        assert np.linalg.norm(ref) > 0, k                                                  # (every variable takes part)
        assert err < 2e-5, (k, err)
        assert np.abs(got - ref).max() < 1e-4 * max(np.abs(ref).max(), 1e-30), (k, np.abs(got - ref).max(), np.abs(ref).max())
    # rows of the gathered tables no id of the batch touched have an exactly zero gradient (scatter-adds only where the forward gathered)
    untouched = np.setdiff1d(np.arange(N), np.unique(x))
    assert not grads["tcn_gmm/temporal_conv_net/tblock_0/dense/kernel"][untouched].any()
    assert not grads["tcn_gmm/temporal_conv_net/tblock_0/conv1/kernel"][:, untouched].any()
    # the weights are what was handed in; a second call overwrites (does not accumulate) the gradients
    w = model.export("weights")
    for k in p:
        assert np.array_equal(w[k], np.asarray(p[k], np.float32)), k
    model.train_forward_backward(dev(x), dev(y), mode)
    again = model.export("grads")
    for k in grads:
        assert np.abs(again[k] - grads[k]).max() <= 1e-5 * max(np.abs(grads[k]).max(), 1e-30), k   # (fp32 atomics reorder sums)
    model.close()


def _adam_fp64(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)            # TF-1.x AdamOptimizer (model.py:139)
    for k in p:
        m[k] = b1 * m[k] + (1 - b1) * g[k]
        v[k] = b2 * v[k] + (1 - b2) * g[k] ** 2
        p[k] = p[k] - lr_t * m[k] / (np.sqrt(v[k]) + eps)


def test_adam_trajectory_against_fp64_and_evaluation_after_sync():
    """Six optimizer steps: the loss trajectory and the final weights follow an fp64 run (autograd + TF-form Adam) of the same model; the
    evaluation path of the same handle then scores the UPDATED weights (hiertcn_gmm_train_sync), exactly as a fresh handle created from the
    exported weights does; Adam moments and time step survive an export / restore round trip into a new handle."""
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    N, M, channels, K, B, T = 900, 3, (64, 64), 3, 6, 10
    a = _args(N, M, channels, K)
    p = _params(a, seed=5)
    rng = np.random.default_rng(11)
    batches = [_batch(rng, N, B, T) for _ in range(6)]
    lr = 2e-3
    ref = {k: np.asarray(v, np.float64).copy() for k, v in p.items()}
    m1 = {k: np.zeros_like(v) for k, v in ref.items()}
    m2 = {k: np.zeros_like(v) for k, v in ref.items()}
    want = []
    for t, (x, y) in enumerate(batches, 1):
        P = {k: torch.tensor(v, requires_grad=True) for k, v in ref.items()}
        loss, _ = torch_loss(a, P, x, y, "gmm_prob")
        loss.backward()
        want.append(float(loss.detach()))
        _adam_fp64(ref, {k: P[k].grad.numpy() for k in ref}, m1, m2, t, lr)
    model = TcnGmmModel(a, p)
    model.enable_training()
    before = model.eval_step(*batches[0])["sums"].cpu().numpy()
    got = []
    for x, y in batches:
        s = model.train_step(x, y, lr)["sums"].cpu().numpy()
        got.append(s[0] / s[1])
    assert np.abs(np.array(got) - np.array(want)).max() < 5e-6 * max(1.0, np.abs(want).max()), (got, want)
    assert model.adam_time() == 6
    w = model.export("weights")
    for k in ref:
        assert np.abs(w[k] - ref[k]).max() < 2e-5 * max(np.abs(ref[k]).max(), 1.0), (k, np.abs(w[k] - ref[k]).max())
    # evaluation of the stepped handle == a fresh handle on the exported weights; and it moved
    after = model.eval_step(*batches[0])["sums"].cpu().numpy()
    fresh = TcnGmmModel(a, w)
    want_after = fresh.eval_step(*batches[0])["sums"].cpu().numpy()
    assert np.array_equal(after, want_after)
    assert abs(after[0] - before[0]) > 1e-6
    # checkpoint round trip: weights + moments + time step into the fresh handle, one more step on both -> the same weights
    fresh.enable_training()
    for which in ("m", "v"):
        fresh.restore(which, model.export(which))
    fresh.adam_time(model.adam_time())
    x, y = _batch(rng, N, B, T)
    model.train_step(x, y, lr)
    fresh.train_step(x, y, lr)
    wa, wb = model.export("weights"), fresh.export("weights")
    for k in wa:
        assert np.abs(wa[k] - wb[k]).max() <= 1e-6 * max(np.abs(wa[k]).max(), 1e-30), k
    model.close()
    fresh.close()


def test_what_is_not_built_says_so():
    from hiertcn_amd._lib import HierTCNError
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    a = _args(300, 2, (64,), 3)
    m = TcnGmmModel(a, _params(a, 1))
    with pytest.raises(HierTCNError):
        m.train_forward_backward(dev(np.zeros((2, 3), np.int32)), dev(np.zeros((2, 3), np.int32)))      # enable_training() first
    m.enable_training()
    with pytest.raises(NotImplementedError):
        m.train_forward_backward(dev(np.zeros((2, 3), np.int32)), dev(np.zeros((2, 3), np.int32)), mode="gmm_prob_hinge")
    m.close()
    # the dense-row trunks (2, 3: the mixture low levels of model_hier) train only together with the high level's variables: the library says so
    import ctypes as C

    from hiertcn_amd import _lib
    lib = _lib.load()
    cfg = _lib.GmmConfig(item_num=300, output_dim=300, num_mixture=2, kernel_size=3, n_blocks=1, sigma=0.1, trunk=2, input_dim=128, hidden_dim=32, num_layer=2)
    cfg.channels[0] = 64
    h2 = C.c_void_p()
    assert lib.hiertcn_gmm_create(C.byref(h2), C.byref(cfg)) == 0
    w2 = _lib.GmmWeights()
    one = torch.zeros(8, device="cuda")
    w2.head_weights = w2.head_biases = one.data_ptr()
    assert lib.hiertcn_gmm_train_enable(h2, C.byref(w2), None) != 0 and b"hl_emb_kernel" in lib.hiertcn_gmm_last_error()
    lib.hiertcn_gmm_destroy(h2)
    cfg = _lib.GmmConfig(item_num=300, output_dim=300, num_mixture=2, sigma=0.1, trunk=3, input_dim=128, hidden_dim=64, num_layer=1)
    h = C.c_void_p()
    assert lib.hiertcn_gmm_create(C.byref(h), C.byref(cfg)) == 0
    w = _lib.GmmWeights()
    one = torch.zeros(8, device="cuda")
    w.head_weights = w.head_biases = one.data_ptr()
    assert lib.hiertcn_gmm_train_enable(h, C.byref(w), None) != 0 and b"hl_emb_kernel" in lib.hiertcn_gmm_last_error()
    lib.hiertcn_gmm_destroy(h)


def test_run_gmm_trains_on_the_synthetic_xing_file_and_the_checkpoint_resumes(tmp_path):
    """`run_xing --model_type tcn_gmm --train` = run_gmm in run_nn's cadence: epochs of optimizer steps at the scheduled learning rate, a
    checkpoint at save_epoch, evaluation at test_epoch.  The test loss falls; the checkpoint holds the trained weights under the reference's
    variable names and a second run restores it and evaluates to the same number."""
    import os

    from hiertcn_amd import checkpoint
    from hiertcn_amd.data_loader import Dataloader_single_level_model_xing
    from hiertcn_amd.data_prepare import load_xing
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    from hiertcn_amd.run_gmm_xing import evaluate_gmm, run_gmm
    data = load_xing(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "interactions_synth.csv"))
    N = int(data[1][:, 1].max()) + 1
    a = _args(N, 3, (64, 64), 3)
    a.batch_size = 6
    a.name = "gmm_train_test"
    a.log_path, a.save_path, a.load_path = str(tmp_path) + "/log/", str(tmp_path) + "/model/", str(tmp_path) + "/model/"
    os.makedirs(a.save_path)
    os.makedirs(a.log_path + "before/")
    a.train, a.validate, a.test, a.load, a.save = True, False, True, False, True
    a.epoch_max, a.epoch_batches_train, a.test_epoch, a.save_epoch, a.lr_schedule, a.learning_rate = 2, 12, 2, 2, False, 2e-3
    mk = lambda split: Dataloader_single_level_model_xing(a, split, data=data)  # noqa: E731
    p0 = G.init_params(a, seed=3)
    model = TcnGmmModel(a, p0)
    before, _, _ = evaluate_gmm(a, mk("test"), model, prefix="before")
    run_gmm(a, mk("train"), mk("validate"), mk("test"), sess=model)
    assert model.adam_time() == 3 * 12
    after, _, _ = evaluate_gmm(a, mk("test"), model, prefix="before")
    assert after["loss_epoch"] < before["loss_epoch"] - 1e-3, (before["loss_epoch"], after["loss_epoch"])
    saved = checkpoint.load(a.save_path + "2.ckpt")
    assert set(saved) == set(p0)
    w = model.export("weights")
    for k in p0:
        assert saved[k].shape == p0[k].shape and np.array_equal(saved[k], w[k]), k
        assert not np.array_equal(saved[k], p0[k]), k                                    # every variable moved
    scalars = open(a.log_path + "train/scalars.jsonl").read().splitlines()
    assert len(scalars) == 3
    model.close()
    # a second run: restore epoch 2 and evaluate only
    b = _args(N, 3, (64, 64), 3)
    b.batch_size, b.name = 6, "gmm_train_test"
    b.log_path, b.save_path, b.load_path = str(tmp_path) + "/log2/", a.save_path, a.save_path
    b.train, b.validate, b.test, b.load, b.load_epoch, b.save = False, False, True, True, 2, False
    m2 = run_gmm(b, None, None, mk("test"))
    again, _, _ = evaluate_gmm(a, mk("test"), m2, prefix="before")
    assert abs(again["loss_epoch"] - after["loss_epoch"]) < 1e-9 * max(1.0, abs(after["loss_epoch"]))
    m2.close()


@pytest.mark.parametrize("low", ["tcn_gmm", "gru_gmm"])
def test_run_hier_gmm_trains_on_the_synthetic_xing_file_and_the_checkpoint_resumes(tmp_path, low):
    """`run_xing --model_type hier --model_low_type tcn_gmm | gru_gmm --train` = run_hier_gmm in run_hier's cadence (run_hier_xing.py:210-350): the
    hierarchical loader, one carried state per loader, both levels stepping together.  The test loss falls, every variable of BOTH levels moves,
    the checkpoint holds them under the reference's names, and a second run restores it and evaluates to the same number."""
    import os

    from hiertcn_amd import checkpoint
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.data_loader import Dataloader_hier_model_xing
    from hiertcn_amd.data_prepare import load_xing
    from hiertcn_amd.model_hier_gmm import HierGmmModel
    from hiertcn_amd.run_gmm_xing import evaluate_hier_gmm, run_hier_gmm
    data = load_xing(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "interactions_synth.csv"))
    N = int(data[1][:, 1].max()) + 1

    def mkargs(log):
        a = hier_tcn_args(item_num=N, output_dim=N, emb_dim=64, hidden_dim=64, num_layer=2, tcn_channel=[64, 64], kernel_size=3, num_mixture=3,
                          model_low_type=low, loss="gmm_prob", sigma=0.1, batch_size=4, max_session_num=3, max_activity_len=6)
        a.name = "hier_gmm_train_test"
        a.log_path, a.save_path, a.load_path = str(tmp_path) + log, str(tmp_path) + "/model/", str(tmp_path) + "/model/"
        return a

    a = mkargs("/log/")
    os.makedirs(a.save_path)
    os.makedirs(a.log_path + "before/")
    a.train, a.validate, a.test, a.load, a.save = True, False, True, False, True
    a.epoch_max, a.epoch_batches_train, a.test_epoch, a.save_epoch, a.lr_schedule, a.learning_rate = 2, 10, 2, 2, False, 2e-3
    mk = lambda split: Dataloader_hier_model_xing(a, split, data=data)  # noqa: E731
    model = HierGmmModel(a, None, "hier")
    p0 = {k: np.array(v) for k, v in model.params.items()}
    zero = lambda: torch.zeros((4, 128), dtype=torch.float32, device="cuda")  # noqa: E731
    before, _, _, _ = evaluate_hier_gmm(a, mk("test"), zero(), model, prefix="before")
    run_hier_gmm(a, mk("train"), mk("validate"), mk("test"), sess=model)
    assert model.adam_time() == 3 * 10
    after, _, _, st_after = evaluate_hier_gmm(a, mk("test"), zero(), model, prefix="before")
    assert after["loss_epoch"] < before["loss_epoch"] - 1e-3, (before["loss_epoch"], after["loss_epoch"])
    assert float(st_after.abs().max()) > 0                                               # the state is carried, not re-zeroed, between batches
    saved = checkpoint.load(a.save_path + "2.ckpt")
    assert set(saved) == set(p0) and any(k.startswith("hier/multi_rnn_cell/") for k in saved) and "hier/emb/kernel" in saved
    w = model.export("weights")
    for k in p0:
        assert saved[k].shape == p0[k].shape and np.array_equal(saved[k], w[k]), k
        assert not np.array_equal(saved[k], p0[k]), k                                    # every variable of both levels moved
    assert len(open(a.log_path + "train/scalars.jsonl").read().splitlines()) == 3
    model.close()
    # a second run: restore epoch 2 and evaluate only
    b = mkargs("/log2/")
    b.train, b.validate, b.test, b.load, b.load_epoch, b.save = False, False, True, True, 2, False
    m2 = run_hier_gmm(b, None, None, mk("test"))
    again, _, _, _ = evaluate_hier_gmm(a, mk("test"), zero(), m2, prefix="before")
    assert abs(again["loss_epoch"] - after["loss_epoch"]) < 1e-9 * max(1.0, abs(after["loss_epoch"]))
    m2.close()


# ---- data parallel: two ranks on ONE GPU over gloo == one rank on the concatenated batch (model.py:135-141 on the global batch) ---------------------
def _dp_case():
    N, M, channels, K, B, T = 700, 3, (64, 64), 3, 7, 9
    a = _args(N, M, channels, K)
    p = _params(a, seed=9)
    x, y = _batch(np.random.default_rng(21), N, B, T)
    return a, p, x, y


def _dp_worker(rank, world, port, out_dir):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from hiertcn_amd.distributed import init_from_env
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    torch.cuda.set_device(0)
    init_from_env("gloo")
    a, p, x, y = _dp_case()
    sl = slice(0, 4) if rank == 0 else slice(4, 7)                      # uneven shards: 4 + 3 lanes (the last one has no target)
    m = TcnGmmModel(a, p)
    m.enable_training()
    out = m.train_step(x[sl], y[sl], 1e-2)
    torch.cuda.synchronize()
    if rank == 0:
        np.savez(os.path.join(out_dir, "dp.npz"), sums=out["sums"].cpu().numpy(), **{"g:" + k: v for k, v in m.export("grads").items()},
                 **{"w:" + k: v for k, v in m.export("weights").items()})
    m.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path):
    import socket

    import torch.multiprocessing as mp

    from hiertcn_amd.model_tcn_gmm import TcnGmmModel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    dp = np.load(str(tmp_path / "dp.npz"))
    a, p, x, y = _dp_case()
    m = TcnGmmModel(a, p)
    m.enable_training()
    out = m.train_step(x, y, 1e-2)
    sums = out["sums"].cpu().numpy()
    g, w = m.export("grads"), m.export("weights")
    assert dp["sums"][1] == sums[1] == 6                                                         # users of the GLOBAL batch
    assert abs(dp["sums"][0] - sums[0]) < 1e-6 * abs(sums[0])
    for k in g:
        assert np.linalg.norm(dp["g:" + k] - g[k]) <= 2e-6 * max(np.linalg.norm(g[k]), 1e-30), k   # the summed shard gradients = the batch gradient
        # ... and so is the Adam update: a first step moves every weight by ~lr (m / sqrt(v) = g / |g|), and where |g| is of the order of
        # Adam's epsilon (3e-7 behind the first step's bias correction) the quotient magnifies the last-bit differences of the summed gradients
        # (the scale of the gradients is what the line above checks -- Adam's first step does not see it): 1e-3 of the step
        assert np.abs(dp["w:" + k] - w[k]).max() <= 1e-3 * 1e-2, k
    m.close()


def _dp_hier_case(low):
    from hiertcn_amd import model_hier_gmm as HG
    from hiertcn_amd.args import hier_tcn_args
    from tests.gpu_common import make_batch
    N, B, S, L = 600, 6, 3, 7
    a = hier_tcn_args(item_num=N, output_dim=N, emb_dim=64, hidden_dim=64, num_layer=2, tcn_channel=[64, 64], kernel_size=3, num_mixture=3,
                      model_low_type=low, loss="gmm_prob", sigma=0.1)
    p = HG.init_params(a, seed=5)
    rng = np.random.default_rng(33)
    x, y, mask = make_batch(rng, B, S, L, N)     # (every session holds an event, as the loader's do: an empty one is the reference's 0 / 0, model_hier.py:84)
    st0 = (0.3 * rng.standard_normal((B, 128))).astype(np.float32)
    return a, p, x, y, mask, st0


def _dp_hier_worker(rank, world, port, out_dir, low):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch.distributed as dist

    from hiertcn_amd.distributed import init_from_env
    from hiertcn_amd.model_hier_gmm import HierGmmModel
    torch.cuda.set_device(0)
    init_from_env("gloo")
    a, p, x, y, mask, st0 = _dp_hier_case(low)
    sl = slice(0, 4) if rank == 0 else slice(4, 6)                      # uneven shards of the lanes: 4 + 2
    m = HierGmmModel(a, p, "hier")
    m.enable_training()
    out = m.train_step_hier(dev(x[sl]), dev(y[sl]), dev(mask[sl]), dev(st0[sl]), 1e-2)
    torch.cuda.synchronize()
    np.savez(os.path.join(out_dir, f"dp{rank}.npz"), sums=out["sums"].cpu().numpy(), state=out["state"].cpu().numpy(),
             **{"g:" + k: v for k, v in m.export("grads").items()}, **{"w:" + k: v for k, v in m.export("weights").items()})
    m.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("low", ["tcn_gmm", "gru_gmm"])
def test_hier_two_ranks_equal_one_rank_on_all_lanes(tmp_path, low):
    """HierGmmModel.train_step_hier under torch.distributed (gloo, two ranks on this one GPU): lanes sharded 4 + 2, the GLOBAL user count in the
    loss normaliser, one all-reduce of the flat gradient arena -> both ranks hold the gradient and the Adam update of the whole batch."""
    import socket

    import torch.multiprocessing as mp

    from hiertcn_amd.model_hier_gmm import HierGmmModel
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_dp_hier_worker, args=(2, port, str(tmp_path), low), nprocs=2, join=True)
    d0, d1 = np.load(str(tmp_path / "dp0.npz")), np.load(str(tmp_path / "dp1.npz"))
    a, p, x, y, mask, st0 = _dp_hier_case(low)
    m = HierGmmModel(a, p, "hier")
    m.enable_training()
    out = m.train_step_hier(dev(x), dev(y), dev(mask), dev(st0), 1e-2)
    sums = out["sums"].cpu().numpy()
    g, w = m.export("grads"), m.export("weights")
    assert d0["sums"][1] == d1["sums"][1] == sums[1] == 6                                         # users of the GLOBAL batch
    assert abs(d0["sums"][0] - sums[0]) < 1e-6 * abs(sums[0])
    np.testing.assert_allclose(np.concatenate([d0["state"], d1["state"]]), out["state"].cpu().numpy(), atol=1e-6, rtol=0)   # each rank carries ITS lanes
    for k in g:
        assert np.array_equal(d0["g:" + k], d1["g:" + k]) and np.array_equal(d0["w:" + k], d1["w:" + k]), (
            k, np.abs(d0["g:" + k] - d1["g:" + k]).max(), np.abs(d0["w:" + k] - d1["w:" + k]).max(), np.abs(d0["g:" + k]).max())   # replicas stay identical
        assert np.linalg.norm(d0["g:" + k] - g[k]) <= 5e-6 * max(np.linalg.norm(g[k]), 1e-30), (k, np.linalg.norm(d0["g:" + k] - g[k]) / np.linalg.norm(g[k]))
        # (Adam's first step divides g by |g| + eps: where |g| ~ eps the last bits of the summed gradients are magnified -- see the single-level test)
        assert np.abs(d0["w:" + k] - w[k]).max() <= 3e-3 * 1e-2, (k, np.abs(d0["w:" + k] - w[k]).max())
    m.close()


# ---- model_gru_gmm: the GRU stack on one-hot rows (model_gru_gmm.py:33-48) under the same head -----------------------------------------------------
def _gru_args(N, M, H, nl, loss="gmm_prob"):
    return make_args([], model_type="gru_gmm", item_num=N, output_dim=N, num_mixture=M, sigma=0.1, loss=loss, hidden_dim=H, num_layer=nl, adjust=False)


def torch_gru_loss(a, P, x, y, mode, scope="rnn_gmm"):
    """fp64: the cells of customed_gru_cell.py:323-337 on MATERIALISED one-hot rows under dynamic_rnn's sequence_length (the count of non-zero
    rows, a prefix length: past it the state is carried and the output row is zero), then the same head / decode / losses as torch_loss."""
    xt, yt = torch.as_tensor(x.astype(np.int64)), torch.as_tensor(y.astype(np.int64))
    B, T = xt.shape
    N, H, nl = int(a.item_num), int(a.hidden_dim), int(a.num_layer)
    onehot = torch.nn.functional.one_hot(xt, N).to(torch.float64) * (xt > 0)[..., None]
    length = (xt > 0).sum(1)
    state = [torch.zeros(B, H, dtype=torch.float64) for _ in range(nl)]
    outs = []
    for t in range(T):
        active = (t < length)[:, None]
        inp = onehot[:, t]
        for l in range(nl):
            pre = f"{scope}/rnn/multi_rnn_cell/cell_{l}/gru_cell"
            gates = torch.sigmoid(torch.cat([inp, state[l]], 1) @ P[f"{pre}/gates/kernel"] + P[f"{pre}/gates/bias"])
            r, u = gates[:, :H], gates[:, H:]
            c = torch.tanh(torch.cat([inp, r * state[l]], 1) @ P[f"{pre}/candidate/kernel"] + P[f"{pre}/candidate/bias"])
            new = u * state[l] + (1 - u) * c
            state[l] = torch.where(active, new, state[l])
            inp = new
        outs.append(torch.where(active, inp, torch.zeros_like(inp)))
    act = torch.stack(outs, 1)
    pred = act @ P[f"{scope}/fully_connected/weights"] + P[f"{scope}/fully_connected/biases"]
    rows = _mix_rows(a, pred, yt, mode)
    mask = (yt > 0).to(torch.float64)
    cnt = mask.sum(1)
    loss = ((rows * mask).sum(1) / (cnt + 1e-6)).sum() / (cnt > 0).sum()
    return loss, rows * mask


@pytest.mark.parametrize("mode", ["gmm_prob", "gmm_prob_max"])
@pytest.mark.parametrize("N,M,H,nl,B,T", [(303, 3, 64, 1, 5, 9), (500, 2, 64, 2, 6, 12), (900, 4, 128, 2, 4, 15), (20_778, 5, 256, 2, 3, 7),
                                         (300, 2, 64, 2, 4100, 3),    # (more lanes than the skinny products take: the five-launch cells of the backward)
                                         (350, 2, 64, 3, 4, 10), (280, 2, 64, 4, 3, 6)])   # three / four layers: anti-diagonals of the wavefront with an odd cell, two pairs
def test_gru_trunk_every_gradient_against_fp64_autograd(N, M, H, nl, B, T, mode):
    from hiertcn_amd.model_gru_gmm import GruGmmModel
    a = _gru_args(N, M, H, nl, loss=mode)
    p = G.init_params_gru(a, seed=N % 89)
    rng = np.random.default_rng(N + H)
    for k in p:
        if k.endswith("biases") or k.endswith("candidate/bias"):
            p[k] = (0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    x, y = _batch(rng, N, B, T)
    x[1, 3:] = 0                                     # a short sequence: steps past its length carry the state, emit zero rows, take no gradient
    x[2, 2] = 0                                      # a null id INSIDE a sequence: sequence_length is a count, used as a prefix length [TF-sem]
    P = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in p.items()}
    loss_ref, rows_ref = torch_gru_loss(a, P, x, y, mode)
    loss_ref.backward()
    pred = G.model_gru_gmm(a, x, {k: np.asarray(v, np.float64) for k, v in p.items()})
    want_rows = G.calc_loss_gmm_ids(pred, y, a) * (y > 0)
    assert np.abs(rows_ref.detach().numpy() - want_rows).max() < 1e-9 * max(1.0, np.abs(want_rows).max())
    model = GruGmmModel(a, p)
    model.enable_training()
    out = model.train_forward_backward(dev(x), dev(y), mode, want_rows=True)
    rows = out["loss_rows"].cpu().numpy()
    sums = out["sums"].cpu().numpy()
    live = y > 0
    assert np.abs(rows - want_rows)[live].max() < 3e-6 * max(1.0, np.abs(want_rows).max()), np.abs(rows - want_rows)[live].max()
    assert abs(sums[0] / sums[1] - float(loss_ref.detach())) < 3e-6 * max(1.0, abs(float(loss_ref.detach())))
    grads = model.export("grads")
    for k, t in P.items():
        ref = t.grad.numpy()
        got = grads[k].astype(np.float64)
        assert got.shape == ref.shape and np.linalg.norm(ref) > 0, k
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert err < 3e-5, (k, err)
    untouched = np.setdiff1d(np.arange(N), np.unique(x))
    assert not grads["rnn_gmm/rnn/multi_rnn_cell/cell_0/gru_cell/gates/kernel"][untouched].any()
    assert not grads["rnn_gmm/rnn/multi_rnn_cell/cell_0/gru_cell/candidate/kernel"][untouched].any()
    # one optimizer step, then the evaluation path of the same handle == a fresh handle on the exported weights
    model.adam_step(1e-3)
    w = model.export("weights")
    got_eval = model.eval_step(x, y)["sums"].cpu().numpy()
    fresh = GruGmmModel(a, w)
    assert np.array_equal(got_eval, fresh.eval_step(x, y)["sums"].cpu().numpy())
    model.close()
    fresh.close()


def test_training_step_replayed_as_a_hipgraph_equals_the_launch_chain():
    """GmmTrainGraph: forward + backward of one batch shape captured once and replayed -- same gradients, same Adam trajectory as the plain calls."""
    from hiertcn_amd.model_gru_gmm import GruGmmModel
    N, M, H, nl, B, T = 600, 3, 64, 2, 5, 14
    a = _gru_args(N, M, H, nl)
    p = G.init_params_gru(a, seed=4)
    rng = np.random.default_rng(8)
    batches = [_batch(rng, N, B, T) for _ in range(3)]
    plain, graph = GruGmmModel(a, p), GruGmmModel(a, p)
    plain.enable_training()
    gr = graph.capture_train_graph(B, T)
    for x, y in batches:
        s1 = plain.train_step(x, y, 1e-3)["sums"].cpu().numpy()
        s2 = gr.step(dev(x), dev(y), 1e-3)["sums"].cpu().numpy()
        assert np.array_equal(s1, s2)
    g1, g2 = plain.export("grads"), graph.export("grads")
    w1, w2 = plain.export("weights"), graph.export("weights")
    for k in g1:
        assert np.abs(g1[k] - g2[k]).max() <= 1e-5 * max(np.abs(g1[k]).max(), 1e-30), k     # (fp32 atomics reorder sums)
        assert np.abs(w1[k] - w2[k]).max() <= 1e-3 * 1e-3, k
    assert graph.adam_time() == 3
    plain.close()
    graph.close()


@pytest.mark.parametrize("seed", range(8))
def test_fuzzed_shapes_against_fp64_autograd(seed):
    """Random model / batch shapes for both trunks (widths 64 .. 256 per block, 1 .. 4 blocks or 1 .. 3 GRU layers, kernel sizes 1 .. 6, catalogues that
    are no multiple of anything, sequences shorter than the receptive field, lanes without targets): loss rows and every gradient vs fp64 autograd."""
    rng = np.random.default_rng(1000 + seed)
    N = int(rng.integers(130, 2600))
    M = int(rng.integers(1, 7))
    B, T = int(rng.integers(1, 9)), int(rng.integers(2, 40))
    mode = ["gmm_prob", "gmm_prob_max"][seed % 2]
    x, y = _batch(rng, N, max(B, 2), T)
    if seed % 2 == 0:
        from hiertcn_amd.model_tcn_gmm import TcnGmmModel as Model
        channels = tuple(int(c) for c in rng.choice([64, 128, 256], int(rng.integers(1, 5))))
        a = _args(N, M, channels, int(rng.integers(1, 7)), loss=mode)
        p = _params(a, seed)
        ref_loss = torch_loss
    else:
        from hiertcn_amd.model_gru_gmm import GruGmmModel as Model
        a = _gru_args(N, M, int(rng.choice([64, 128, 256])), int(rng.integers(1, 4)), loss=mode)
        p = G.init_params_gru(a, seed=seed)
        ref_loss = torch_gru_loss
    P = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in p.items()}
    loss_ref, rows_ref = ref_loss(a, P, x, y, mode)
    loss_ref.backward()
    model = Model(a, p)
    model.enable_training()
    out = model.train_forward_backward(dev(x), dev(y), mode, want_rows=True)
    rows = out["loss_rows"].cpu().numpy()
    want = rows_ref.detach().numpy()
    assert np.abs(rows - want).max() < 5e-6 * max(1.0, np.abs(want).max()), (a.tcn_channel if seed % 2 == 0 else (a.hidden_dim, a.num_layer), np.abs(rows - want).max())
    grads = model.export("grads")
    for k, t in P.items():
        ref = t.grad.numpy()
        nr = np.linalg.norm(ref)
        err = np.linalg.norm(grads[k].astype(np.float64) - ref) / max(nr, 1e-30)
        assert err < 5e-5 or nr < 1e-12, (k, err, nr)
    model.close()


@pytest.mark.parametrize("mode", ["gmm_prob", "gmm_prob_max"])
@pytest.mark.parametrize("trunk,N,M,width,depth,B,T", [("tcn", 303, 3, 64, 2, 5, 9), ("tcn", 1201, 2, 128, 1, 4, 13), ("gru", 500, 4, 64, 2, 4, 11),
                                                       ("tcn", 20_778, 5, 256, 2, 2, 6)])
def test_learned_scale_every_gradient_against_fp64_autograd(trunk, N, M, width, depth, B, T, mode):
    """args.sigma None (`--sigma none`): M (1 + 2 D) head outputs, sigma_d = exp(s_d).  sum_d mu_d^2 e^{-2 s_d} is no quadratic form of the hidden
    row: the device materialises mu and s with exact-fp32 products and differentiates them in place."""
    rng = np.random.default_rng(N + M)
    x, y = _batch(rng, N, B, T)
    if trunk == "tcn":
        from hiertcn_amd.model_tcn_gmm import TcnGmmModel as Model
        a = _args(N, M, (width,) * depth, 3, loss=mode)
        a.sigma = None
        p = _params(a, seed=N % 89)
        ref_loss, oracle_pred = torch_loss, G.model_tcn_gmm
    else:
        from hiertcn_amd.model_gru_gmm import GruGmmModel as Model
        a = _gru_args(N, M, width, depth, loss=mode)
        a.sigma = None
        p = G.init_params_gru(a, seed=N % 89)
        ref_loss, oracle_pred = torch_gru_loss, G.model_gru_gmm
    assert p[[k for k in p if k.endswith("fully_connected/weights")][0]].shape[1] == M * (1 + 2 * N)
    P = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in p.items()}
    loss_ref, rows_ref = ref_loss(a, P, x, y, mode)
    loss_ref.backward()
    pred = oracle_pred(a, x, {k: np.asarray(v, np.float64) for k, v in p.items()})
    want_rows = G.calc_loss_gmm_ids(pred, y, a) * (y > 0)
    assert np.abs(rows_ref.detach().numpy() - want_rows).max() < 1e-9 * max(1.0, np.abs(want_rows).max())
    model = Model(a, p)
    model.enable_training()
    out = model.train_forward_backward(dev(x), dev(y), mode, want_rows=True)
    rows = out["loss_rows"].cpu().numpy()
    live = y > 0
    assert not rows[~live].any()
    assert np.abs(rows - want_rows)[live].max() < 5e-6 * max(1.0, np.abs(want_rows).max()), np.abs(rows - want_rows)[live].max()
    sums = out["sums"].cpu().numpy()
    assert abs(sums[0] / sums[1] - float(loss_ref.detach())) < 5e-6 * max(1.0, abs(float(loss_ref.detach())))
    grads = model.export("grads")
    for k, t in P.items():
        ref = t.grad.numpy()
        err = np.linalg.norm(grads[k].astype(np.float64) - ref) / max(np.linalg.norm(ref), 1e-30)
        assert np.linalg.norm(ref) > 0 and err < 5e-5, (k, err)
    # a step, then the evaluation path of the same handle == a fresh handle on the exported weights
    model.adam_step(1e-3)
    w = model.export("weights")
    fresh = Model(a, w)
    assert np.array_equal(model.eval_step(x, y)["sums"].cpu().numpy(), fresh.eval_step(x, y)["sums"].cpu().numpy())
    model.close()
    fresh.close()


# ---- model_hier with the tcn_gmm low level (model_hier.py:39-93 around model_tcn_gmm): the whole model trains as one step ---------------------------
def torch_hier_loss(a, P, x, y, mask, st0, mode, scope="hier", kinks=None):
    """fp64: the reference's slot loop, literally -- per slot the [emb(x), carried state] rows into the TemporalConvNet (block 0 an ordinary block on
    them) and the head, then the high level (mean target embedding + bias -> MultiRNNCell -> state * mask) -- and the lane-mean mixture loss over the
    concatenated slots.  -> (loss, loss rows [B, S * L], final state).  `kinks` (a list) receives the smallest non-zero |argument| of every ReLU:
    a value below fp32's resolution of the sums sits on the other side of the kink on the device, and the two gradients then differ by a whole
    term for a reason that is neither's error -- the caller re-draws its inputs."""
    xt, yt = torch.as_tensor(x.astype(np.int64)), torch.as_tensor(y.astype(np.int64))
    mk = torch.as_tensor(mask.astype(np.float64))
    B, S, L = xt.shape
    E, H, nl, K = int(a.emb_dim), int(a.hidden_dim), int(a.num_layer), int(a.kernel_size)
    emb, eb = P[f"{scope}/emb/kernel"], P[f"{scope}/emb/bias"]
    state = torch.as_tensor(st0.astype(np.float64))
    preds = []
    gru_low = getattr(a, "model_low_type", "tcn_gmm") == "gru_gmm"
    root = f"{scope}/rnn_gmm" if gru_low else f"{scope}/tcn_gmm"
    for s in range(S):
        xe = emb[xt[:, s]] * (xt[:, s] > 0).to(torch.float64)[..., None]
        act = torch.cat([xe, state[:, None, :].expand(B, L, H * nl)], -1)
        if gru_low:   # model_gru_gmm.py:33-48 on the rows: dynamic_rnn from a zero state, sequence_length = the COUNT of rows that are not all-zero, a prefix length
            length = (act.detach().abs().amax(-1) > 0).sum(1)
            ls = [torch.zeros(B, H, dtype=torch.float64) for _ in range(nl)]
            outs = []
            for t in range(L):
                active = (t < length)[:, None]
                inp = act[:, t]
                for l in range(nl):
                    pre = f"{root}/rnn/multi_rnn_cell/cell_{l}/gru_cell"
                    gates = torch.sigmoid(torch.cat([inp, ls[l]], 1) @ P[f"{pre}/gates/kernel"] + P[f"{pre}/gates/bias"])
                    r, u = gates[:, :H], gates[:, H:]
                    cnd = torch.tanh(torch.cat([inp, r * ls[l]], 1) @ P[f"{pre}/candidate/kernel"] + P[f"{pre}/candidate/bias"])
                    hn = u * ls[l] + (1 - u) * cnd
                    ls[l] = torch.where(active, hn, ls[l])
                    inp = hn
                outs.append(torch.where(active, inp, torch.zeros_like(inp)))
            act = torch.stack(outs, 1)
        for i in range(0 if gru_low else len(a.tcn_channel)):
            pre = f"{root}/temporal_conv_net/tblock_{i}"
            Wc, bc = P[f"{pre}/conv1/kernel"], P[f"{pre}/conv1/bias"]
            dil = 2 ** i
            c = bc.expand(B, L, -1).clone()
            for j in range(K):
                sh = (K - 1 - j) * dil
                if sh >= L:
                    continue
                z = act @ Wc[j]
                c = c + (torch.nn.functional.pad(z, (0, 0, sh, 0))[:, :L] if sh else z)
            res = act @ P[f"{pre}/dense/kernel"] + P[f"{pre}/dense/bias"] if f"{pre}/dense/kernel" in P else act
            pre_out = torch.relu(c) + res
            if kinks is not None:
                for v in (c.detach().abs(), pre_out.detach().abs()):
                    kinks.append(float(v[v > 0].min()))
            act = torch.relu(pre_out)
        preds.append(act @ P[f"{root}/fully_connected/weights"] + P[f"{root}/fully_connected/biases"])
        ys = yt[:, s]
        valid = (ys > 0).to(torch.float64)
        inp = (emb[ys] * valid[..., None]).sum(1) / valid.sum(1, keepdim=True) + eb
        new = []
        for l in range(nl):
            pre = f"{scope}/multi_rnn_cell/cell_{l}/gru_cell"
            h = state[:, l * H:(l + 1) * H]
            gates = torch.sigmoid(torch.cat([inp, h], 1) @ P[f"{pre}/gates/kernel"] + P[f"{pre}/gates/bias"])
            r, u = gates[:, :H], gates[:, H:]
            cnd = torch.tanh(torch.cat([inp, r * h], 1) @ P[f"{pre}/candidate/kernel"] + P[f"{pre}/candidate/bias"])
            hn = u * h + (1 - u) * cnd
            new.append(hn)
            inp = hn
        state = torch.cat(new, 1) * mk[:, s:s + 1]
    pred = torch.cat(preds, 1)                                   # [B, S * L, n_out]
    yy = yt.reshape(B, S * L)
    rows = _mix_rows(a, pred, yy, mode)
    m = (yy > 0).to(torch.float64)
    cnt = m.sum(1)
    loss = ((rows * m).sum(1) / (cnt + 1e-6)).sum() / (cnt > 0).sum()
    return loss, rows * m, state


@pytest.mark.parametrize("mode", ["gmm_prob", "gmm_prob_max"])
@pytest.mark.parametrize("low,N,E,H,nl,M,channels,K,B,S,L", [("tcn_gmm", 900, 64, 32, 2, 3, (64, 128), 3, 5, 4, 7), ("tcn_gmm", 500, 64, 64, 1, 2, (128,), 2, 3, 5, 6),
                                                             ("tcn_gmm", 1500, 128, 64, 2, 4, (128, 128), 5, 6, 3, 12),
                                                             ("gru_gmm", 900, 64, 64, 2, 3, (64,), 3, 5, 4, 7), ("gru_gmm", 600, 64, 128, 1, 2, (64,), 3, 4, 3, 9),
                                                             # the reference's defaults (args.py: emb 128, hidden 128 x 2, [128, 128] K = 5, 5 mixtures, the XING catalogue)
                                                             ("tcn_gmm", 20_778, 128, 128, 2, 5, (128, 128), 5, 4, 3, 20),
                                                             ("gru_gmm", 20_778, 128, 128, 2, 5, (128,), 5, 4, 3, 20),
                                                             # three layers at both levels: anti-diagonals of the two wavefronts with an odd cell
                                                             ("gru_gmm", 450, 64, 64, 3, 2, (64,), 3, 3, 4, 5), ("tcn_gmm", 450, 64, 32, 3, 2, (64, 64), 3, 3, 4, 5)])
def test_hier_low_level_every_gradient_against_fp64_autograd(low, N, E, H, nl, M, channels, K, B, S, L, mode):
    _hier_gradient_case(low, N, E, H, nl, M, channels, K, B, S, L, mode, 0.1)


@pytest.mark.parametrize("mode", ["gmm_prob", "gmm_prob_max"])
@pytest.mark.parametrize("low,N,E,H,nl,M,channels,K,B,S,L", [("tcn_gmm", 700, 64, 32, 2, 3, (64, 64), 3, 4, 3, 6), ("gru_gmm", 400, 64, 64, 1, 2, (64,), 3, 3, 4, 5)])
def test_hier_low_level_with_a_learned_scale_every_gradient_against_fp64_autograd(low, N, E, H, nl, M, channels, K, B, S, L, mode):
    """--sigma none under model_hier: the materialising head (M (1 + 2 D) outputs) behind the [e, state] rows of either mixture low level."""
    _hier_gradient_case(low, N, E, H, nl, M, channels, K, B, S, L, mode, None)


def _hier_gradient_case(low, N, E, H, nl, M, channels, K, B, S, L, mode, sigma):
    from hiertcn_amd import model_hier_gmm as HG
    from hiertcn_amd.args import hier_tcn_args
    from tests.gpu_common import as_lists, make_batch
    a = hier_tcn_args(item_num=N, output_dim=N, emb_dim=E, hidden_dim=H, num_layer=nl, tcn_channel=list(channels), kernel_size=K, num_mixture=M,
                      model_low_type=low, loss=mode, sigma=0.1)
    a.sigma = sigma
    p = HG.init_params(a, seed=N % 83) if low == "gru_gmm" else G.init_params_hier(a, seed=N % 83)
    assert p[[k for k in p if k.endswith("fully_connected/weights")][0]].shape[1] == M * (1 + (1 if sigma else 2) * N)
    rng = np.random.default_rng(N + E)
    for k in p:
        if (k.endswith("/bias") and "gates" not in k) or k.endswith("biases"):
            p[k] = (p[k] + 0.05 * rng.standard_normal(p[k].shape)).astype(np.float32)
    P = {k: torch.tensor(np.asarray(v, np.float64), requires_grad=True) for k, v in p.items()}
    for _ in range(8):
        x, y, mask = make_batch(rng, B, S, L, N)
        mask[0, 1] = 0.0                                         # a user boundary inside the batch: the state restarts from zero behind it
        st0 = (0.3 * rng.standard_normal((B, H * nl))).astype(np.float32)
        st0[B - 1] = 0.0                                         # a fresh user: its first row (null id, zero state) is all-zero -- dynamic_rnn's length is one short [TF-sem]
        kinks = []
        loss_ref, rows_ref, st_ref = torch_hier_loss(a, P, x, y, mask, st0, mode, kinks=kinks)
        if not kinks or min(kinks) > 1e-7:                       # (500-64-64: the first draw puts one pre-activation at 1.1e-8, which fp32 rounds to the other side)
            break
    else:
        raise AssertionError("no draw keeps every ReLU argument clear of the kink")
    loss_ref.backward()
    # the torch formulation's forward against the NumPy oracle of the same model
    xl, yl, ml = as_lists(x, y, mask)
    oracle_model = G.model_hier_gru_gmm if low == "gru_gmm" else G.model_hier_tcn_gmm
    pred_o, st_o = oracle_model(a, xl, yl, ml, st0, {k: np.asarray(v, np.float64) for k, v in p.items()})
    yy = y.reshape(B, S * L)
    want_rows = G.calc_loss_gmm_ids(pred_o, yy, a) * (yy > 0)
    assert np.abs(rows_ref.detach().numpy() - want_rows).max() < 1e-9 * max(1.0, np.abs(want_rows).max())
    assert np.abs(st_ref.detach().numpy() - st_o).max() < 1e-12
    model = HG.HierGmmModel(a, p, "hier")
    model.enable_training()
    out = model.train_forward_backward_hier(dev(x), dev(y), dev(mask), dev(st0), mode, want_rows=True)
    rows = out["loss_rows"].cpu().numpy()
    live = yy > 0
    assert not rows[~live].any()
    assert np.abs(rows - want_rows)[live].max() < 5e-6 * max(1.0, np.abs(want_rows).max()), np.abs(rows - want_rows)[live].max()
    np.testing.assert_allclose(out["state"].cpu().numpy(), st_o, atol=2e-6, rtol=0)
    sums = out["sums"].cpu().numpy()
    assert abs(sums[0] / sums[1] - float(loss_ref.detach())) < 5e-6 * max(1.0, abs(float(loss_ref.detach())))
    grads = model.export("grads")
    assert set(grads) == set(P)
    for k, t in P.items():
        ref = t.grad.numpy()
        got = grads[k].astype(np.float64)
        assert got.shape == ref.shape and np.linalg.norm(ref) > 0, k
        err = np.linalg.norm(got - ref) / np.linalg.norm(ref)
        assert err < 5e-5, (k, err)
    # one optimizer step; evaluation (main handle: pool / scan / rows; mixture handle: trunk + head) then sees the updated weights of BOTH handles
    from hiertcn_amd.model_hier import model_hier  # noqa: F401
    before = model.forward_hier(dev(x), dev(y), dev(mask), dev(st0))[1].cpu().numpy()
    model.adam_step(1e-2)
    after_h, after_state = model.forward_hier(dev(x), dev(y), dev(mask), dev(st0))
    fresh = HG.HierGmmModel(a, model.export("weights"), "hier")
    fresh_h, fresh_state = fresh.forward_hier(dev(x), dev(y), dev(mask), dev(st0))
    assert torch.equal(after_h, fresh_h) and torch.equal(after_state, fresh_state)
    assert np.abs(after_state.cpu().numpy() - before).max() > 1e-5                       # the across-session cells moved
    model.close()
    fresh.close()


def test_hier_low_level_adam_trajectory():
    from hiertcn_amd import model_hier_gmm as HG
    from hiertcn_amd.args import hier_tcn_args
    from tests.gpu_common import make_batch
    N, E, H, nl, M, B, S, L = 700, 64, 32, 2, 3, 4, 3, 6
    a = hier_tcn_args(item_num=N, output_dim=N, emb_dim=E, hidden_dim=H, num_layer=nl, tcn_channel=[64, 64], kernel_size=3, num_mixture=M,
                      model_low_type="tcn_gmm", loss="gmm_prob", sigma=0.1)
    p = G.init_params_hier(a, seed=6)
    rng = np.random.default_rng(2)
    batches = [make_batch(rng, B, S, L, N) for _ in range(5)]
    lr = 2e-3
    ref = {k: np.asarray(v, np.float64).copy() for k, v in p.items()}
    m1 = {k: np.zeros_like(v) for k, v in ref.items()}
    m2 = {k: np.zeros_like(v) for k, v in ref.items()}
    st_ref = np.zeros((B, H * nl))
    model = HG.HierGmmModel(a, p, "hier")
    model.enable_training()
    st_dev = torch.zeros((B, H * nl), device="cuda")
    for t, (x, y, mask) in enumerate(batches, 1):
        P = {k: torch.tensor(v, requires_grad=True) for k, v in ref.items()}
        loss, _, st_new = torch_hier_loss(a, P, x, y, mask, st_ref, "gmm_prob")
        loss.backward()
        _adam_fp64(ref, {k: P[k].grad.numpy() for k in ref}, m1, m2, t, lr)
        st_ref = st_new.detach().numpy()                                                   # the carried state is re-fed as data (run_hier_xing.py:291,301)
        out = model.train_step_hier(dev(x), dev(y), dev(mask), st_dev, lr)
        st_dev = out["state"]
        s = out["sums"].cpu().numpy()
        assert abs(s[0] / s[1] - float(loss.detach())) < 1e-5 * max(1.0, abs(float(loss.detach()))), t
        np.testing.assert_allclose(st_dev.cpu().numpy(), st_ref, atol=5e-5, rtol=0)
    w = model.export("weights")
    for k in ref:
        assert np.abs(w[k] - ref[k]).max() < 5e-5 * max(np.abs(ref[k]).max(), 1.0), (k, np.abs(w[k] - ref[k]).max())
    model.close()
