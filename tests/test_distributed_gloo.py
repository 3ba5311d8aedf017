"""world_size-2 gloo test of the multi-GPU path's host logic: users sharded over ranks, each rank
reduces its lanes to the 8 metric accumulators, one sum all-reduce, and the result equals the
single-process metrics of the whole batch (model.py:116-117, loss.py:208-219)."""
import os
import socket

import numpy as np
import torch
import torch.multiprocessing as mp

from hiertcn_amd.distributed import user_shard
from oracle import hiertcn_oracle as O
from tests.gpu_common import as_lists, make_batch, make_params, small_args


def _metrics8(m, y_id):
    """lane-mean SUMS + user count + valid steps from oracle per-step outputs (the host analogue of
    hiertcn_metrics, include/hiertcn.h)."""
    mask = (y_id > 0).astype(np.float64)
    act = mask.sum(1)
    users = float(np.sign(act).sum())
    act = act + 1e-6
    ranks = m["ranks"]

    def lane_sum(v):
        return float(((v * mask).sum(1) / act).sum())

    return np.array([lane_sum(m["ce"]), lane_sum((ranks <= 0) * 1.0), lane_sum((ranks <= 4) * 1.0), lane_sum((ranks <= 9) * 1.0),
                     lane_sum(1.0 / (1.0 + ranks)), lane_sum(m["ranks_float"]), users, float(mask.sum())])


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hiertcn_amd.distributed import all_reduce_metrics, init_from_env, max_over_ranks
    import torch.distributed as dist

    r, w, _ = init_from_env("gloo")
    assert (r, w) == (rank, world)
    args = small_args(N=300, d=16, H=8, S=3, L=5)
    p = make_params(args)
    x, y, mask = make_batch(np.random.default_rng(5), 7, 3, 5, 300)   # 7 users: an uneven split (4 + 3)
    first, n = user_shard(7, rank, world)
    xs, ys, ms = x[first:first + n], y[first:first + n], mask[first:first + n]
    xl, yl, ml = as_lists(xs, ys, ms)
    pred, _ = O.model_hier(args, xl, yl, ml, np.zeros((n, 16)), p)
    m = O.masked_loss_and_metrics(pred, ys.reshape(n, -1), 300)
    m8 = torch.from_numpy(_metrics8(m, ys.reshape(n, -1)))
    all_reduce_metrics(m8)
    t = max_over_ranks(1.0 + rank)
    if rank == 0:
        np.save(os.path.join(out_dir, "m8.npy"), m8.numpy())
        np.save(os.path.join(out_dir, "t.npy"), np.array([t]))
    dist.barrier()
    dist.destroy_process_group()


def test_user_shard_is_a_partition():
    for n, w in ((200_000, 8), (7, 2), (5, 8), (0, 4)):
        spans = [user_shard(n, r, w) for r in range(w)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == n
        for (f0, c0), (f1, _) in zip(spans, spans[1:]):
            assert f0 + c0 == f1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def test_two_rank_metric_all_reduce_equals_single_process(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "m8.npy")
    args = small_args(N=300, d=16, H=8, S=3, L=5)
    p = make_params(args)
    x, y, mask = make_batch(np.random.default_rng(5), 7, 3, 5, 300)
    xl, yl, ml = as_lists(x, y, mask)
    pred, _ = O.model_hier(args, xl, yl, ml, np.zeros((7, 16)), p)
    m = O.masked_loss_and_metrics(pred, y.reshape(7, -1), 300)
    want = _metrics8(m, y.reshape(7, -1))
    np.testing.assert_allclose(got, want, rtol=1e-12)
    # and the reduced sums give the reference's batch metrics
    assert got[0] / got[6] == np.float64(m["loss"]) or abs(got[0] / got[6] - m["loss"]) < 1e-12
    assert abs(got[3] / got[6] - m["recall10"]) < 1e-12
    assert np.load(tmp_path / "t.npy")[0] == 2.0   # MAX over ranks of the per-rank time


def _grad_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hiertcn_amd.distributed import _sum_async_bucketed, init_from_env
    import torch.distributed as dist
    init_from_env("gloo")
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)      # stands for a dense segment of the gradient arena
    _sum_async_bucketed(g, bucket_bytes=4 * 96).wait()            # 96-float buckets: 11 async all-reduces in flight, a ragged last one
    if rank == 0:
        np.save(os.path.join(out_dir, "g.npy"), g.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_all_reduce(tmp_path):
    """HIERTCN_DP_COLLECTIVE=allreduce: the dense segment as several all-reduces in flight at once."""
    mp.spawn(_grad_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    np.testing.assert_array_equal(np.load(tmp_path / "g.npy"), np.arange(1000, dtype=np.float32) * 3)


class _HostEngine:
    """The five engine calls GradientExchange makes, on host tensors (test double: the host logic of the exchange -- which part goes
    through which collective, one all-gather capacity for all ranks, the rank-order rebuild of the row-sparse sum -- runs on CPU)."""

    def __init__(self, N, E, n_rest, rank):
        self.N, self.E, self.device = N, E, torch.device("cpu")
        self.seg = {0: (0, N * E), 1: (N * E, 2 * N * E + N), 2: (3 * N * E + N, n_rest)}   # item table | scoring kernel + bias | the rest
        g = torch.Generator().manual_seed(100 + rank)
        self.arena = torch.zeros(3 * N * E + N + n_rest)
        self.arena[N * E:] = torch.randn(self.arena.numel() - N * E, generator=g)

    def touch(self, x_ids, y_ids, rank):
        """a row-sparse item-table gradient: only ids of the batch have a row"""
        g = torch.Generator().manual_seed(200 + rank)
        tab = self.arena[:self.N * self.E].view(self.N, self.E)
        for ids in (x_ids, y_ids):
            for i in ids.reshape(-1).tolist():
                if i > 0:
                    tab[i] += torch.randn(self.E, generator=g)

    def grad_arena(self):
        return self.arena

    def grad_segment(self, which):
        return self.seg[which]

    def count_users(self, y):
        return (y.reshape(y.shape[0], -1) > 0).any(1).sum().float().reshape(1)

    def grad_rows_list(self, x_ids, y_ids, cap=None):
        ids = torch.unique(torch.cat([x_ids.reshape(-1), y_ids.reshape(-1)]))
        ids = ids[ids > 0].to(torch.int32)
        cap = max(int(cap or 0), min(self.N - 1, 2 * x_ids.numel()))
        out = torch.zeros(cap, dtype=torch.int32)
        out[:ids.numel()] = ids
        return out, torch.tensor([ids.numel()], dtype=torch.int32), cap

    def grad_rows_op(self, op, ids, count, rows=None):
        tab = self.arena[:self.N * self.E].view(self.N, self.E)
        n = int(count[0])
        idx = ids[:n].long()
        if op == 0:
            rows[:n] = tab[idx]
        elif op == 1:
            tab[idx] = 0
        else:
            tab.index_add_(0, idx, rows[:n])


def _exchange_worker(rank, world, port, out_dir, sparse_fraction, ids_bound=None, as_rccl=False, mode="rs_ag", N=97):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HIERTCN_DP_COLLECTIVE=mode)
    from hiertcn_amd import distributed as D
    from hiertcn_amd.distributed import GradientExchange, init_from_env
    import torch.distributed as dist
    init_from_env("gloo")
    calls = []
    if as_rccl:
        # RCCL cannot run here; its branches of the exchange (reduce_scatter_tensor + all_gather_into_tensor, the bucketed all-reduce, the
        # flat all_gather_into_tensor of the row lists, async handles) are backend-agnostic torch.distributed calls that gloo also
        # implements on host tensors: report "nccl" to the module and record which collectives it then issues
        D.dist.get_backend = lambda *a, **k: "nccl"
        for name in ("reduce_scatter_tensor", "all_gather_into_tensor", "all_reduce", "all_gather"):
            real = getattr(dist, name)

            def spy(*a, _real=real, _name=name, **k):
                calls.append(_name)
                h = _real(*a, **k)
                if h is not None:   # RCCL runs its collectives in issue order on one stream; gloo's async ops are unordered
                    h.wait()        # worker-thread jobs: complete each before the next is issued to keep that order
                return h
            setattr(D.dist, name, spy)
    E = 8
    eng = _HostEngine(N, E, 33, rank)
    rng = np.random.default_rng(rank)
    B = 3 + rank                                      # uneven shards: different numbers of ids per rank
    x = torch.from_numpy(rng.integers(0, N, (B, 2, 5)).astype(np.int32))
    y = torch.from_numpy(rng.integers(0, N, (B, 2, 5)).astype(np.int32))
    eng.touch(x, y, rank)
    np.save(os.path.join(out_dir, f"before{rank}.npy"), eng.arena.numpy().copy())
    ex = GradientExchange(eng, sparse_rows_max_fraction=sparse_fraction, ids_bound=ids_bound)
    users = ex.global_users(y)
    ex.after_scoring()
    ex.after_backward(x, y)
    np.save(os.path.join(out_dir, f"after{rank}.npy"), eng.arena.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "meta.npy"), np.array([float(users[0]), 1.0 if ex.last["item_table"] == "row-sparse" else 0.0]))
        with open(os.path.join(out_dir, "calls.txt"), "w") as f:
            f.write(" ".join(calls))
    dist.barrier()
    dist.destroy_process_group()


def _check_exchange(d, want_sparse):
    b0, b1 = np.load(d / "before0.npy"), np.load(d / "before1.npy")
    a0, a1 = np.load(d / "after0.npy"), np.load(d / "after1.npy")
    users, sparse = np.load(d / "meta.npy")
    assert sparse == want_sparse
    assert users == 3 + 4                                        # every lane of both shards has a target
    assert np.array_equal(a0, a1)                                 # replicas stay bit-identical
    np.testing.assert_allclose(a0, b0 + b1, rtol=1e-6, atol=1e-6)


def test_gradient_exchange_dense_and_row_sparse(tmp_path):
    """hiertcn_amd.distributed.GradientExchange on two ranks: every part of the arena ends as the SUM over ranks on BOTH ranks,
    bit-identical between them, whether the item table goes densely or as (row id, row) pairs; the user count is global.  With a
    host-known ids_bound (what the models pass) and without (the ranks agree per step)."""
    for bound in (None, 4 * 2 * 5):
        for frac, want_sparse in ((0.0, 0.0), (2.0, 1.0)):
            d = tmp_path / f"{frac}_{bound}"
            d.mkdir()
            mp.spawn(_exchange_worker, args=(2, _free_port(), str(d), frac, bound), nprocs=2, join=True)
            _check_exchange(d, want_sparse)


def test_gradient_exchange_on_four_ranks(tmp_path):
    """The same exchange at world 4 (the driver's scaling runs are 2, 4 and 8 ranks): dense and row-sparse item table, then the RCCL branches
    (reduce-scatter + all-gather: 1632 floats divide by 4) -- every rank ends with the sum over the four shards, bit-identical replicas, the
    global user count 3 + 4 + 5 + 6."""
    for tag, frac, as_rccl in (("dense", 0.0, False), ("sparse", 2.0, False), ("rccl_sparse", 2.0, True)):
        d = tmp_path / tag
        d.mkdir()
        mp.spawn(_exchange_worker, args=(4, _free_port(), str(d), frac, 6 * 2 * 5, as_rccl, "rs_ag", 96), nprocs=4, join=True)
        before = [np.load(d / f"before{r}.npy") for r in range(4)]
        after = [np.load(d / f"after{r}.npy") for r in range(4)]
        users, sparse = np.load(d / "meta.npy")
        assert sparse == (1.0 if frac > 1 else 0.0) and users == 3 + 4 + 5 + 6
        for r in range(1, 4):
            assert np.array_equal(after[0], after[r])
        np.testing.assert_allclose(after[0], sum(before), rtol=1e-6, atol=1e-6)
        if as_rccl:
            calls = open(d / "calls.txt").read().split()
            assert calls.count("reduce_scatter_tensor") == 1 and calls.count("all_gather_into_tensor") == 1 + 3 and "all_gather" not in calls


def test_gradient_exchange_rccl_branches_over_host_tensors(tmp_path):
    """The branches taken when dist.get_backend() == "nccl" (RCCL), run here over gloo on host tensors: reduce_scatter_tensor +
    all_gather_into_tensor on the scoring segment when its size divides by the world size (N = 96: 2 N E + N = 1632 floats), the plain
    async all-reduce when it does not (N = 97: 1649), the bucketed all-reduce of HIERTCN_DP_COLLECTIVE=allreduce, and the flat
    all_gather_into_tensor of the (row id, row) lists."""
    for tag, mode, frac, N in (("rs_ag", "rs_ag", 2.0, 96), ("rs_ag_odd", "rs_ag", 2.0, 97), ("allreduce", "allreduce", 0.0, 96)):
        d = tmp_path / tag
        d.mkdir()
        mp.spawn(_exchange_worker, args=(2, _free_port(), str(d), frac, 4 * 2 * 5, True, mode, N), nprocs=2, join=True)
        _check_exchange(d, 1.0 if frac > 1 else 0.0)
        calls = open(d / "calls.txt").read().split()
        if tag == "rs_ag":
            assert calls.count("reduce_scatter_tensor") == 1
            assert calls.count("all_gather_into_tensor") == 1 + 3          # the segment's all-gather + ids, counts, rows of the row-sparse exchange
        elif tag == "rs_ag_odd":
            assert "reduce_scatter_tensor" not in calls                     # the numel % world fallback
            assert calls.count("all_gather_into_tensor") == 3
        else:
            assert "reduce_scatter_tensor" not in calls and "all_gather_into_tensor" not in calls
            assert calls.count("all_reduce") >= 4                           # user count, scoring segment (bucketed), item table, the rest
        assert "all_gather" not in calls                                    # the gloo-only host staging is not what RCCL runs


def _ragged_steps_worker(rank, world, port, out_dir, ids_bound):
    """Several steps whose id tensors differ in size between the ranks AND from step to step, as run_xing produces them (pack_slots pads
    to the rank's own longest session).  Round 2 asked the ranks for their maximum only when the LOCAL size was new and cached the
    answer under it: rank 0 entered that all-reduce in a step where rank 1 hit its cache."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hiertcn_amd.distributed import GradientExchange, init_from_env
    import torch.distributed as dist
    import datetime
    init_from_env("gloo")
    N, E = 61, 4
    eng = _HostEngine(N, E, 10, rank)
    ex = GradientExchange(eng, sparse_rows_max_fraction=2.0, ids_bound=ids_bound)
    L_by_step = ([3, 5, 3, 7, 5], [5, 5, 4, 3, 7])[rank]     # rank 0 repeats a size where rank 1 sees a new one, and vice versa
    rng = np.random.default_rng(10 + rank)
    sums = []
    for step, L in enumerate(L_by_step):
        eng.arena.zero_()
        x = torch.from_numpy(rng.integers(0, N, (2, 2, L)).astype(np.int32))
        y = torch.from_numpy(rng.integers(0, N, (2, 2, L)).astype(np.int32))
        eng.touch(x, y, rank * 100 + step)
        before = eng.arena.clone()
        ex.global_users(y)
        ex.after_scoring()
        ex.after_backward(x, y)
        both = [torch.zeros_like(before) for _ in range(world)]
        dist.all_gather(both, before)
        sums.append(float((eng.arena - (both[0] + both[1])).abs().max()))
    if rank == 0:
        np.save(os.path.join(out_dir, "err.npy"), np.array(sums))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_exchange_with_sizes_that_differ_per_rank_and_step(tmp_path):
    for bound in (2 * 2 * 7, None):
        d = tmp_path / str(bound)
        d.mkdir()
        mp.spawn(_ragged_steps_worker, args=(2, _free_port(), str(d), bound), nprocs=2, join=True)
        err = np.load(d / "err.npy")
        assert err.shape == (5,) and err.max() < 1e-5


def _bound_violation_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hiertcn_amd.distributed import GradientExchange, init_from_env
    import torch.distributed as dist
    init_from_env("gloo")
    N, E = 61, 4
    eng = _HostEngine(N, E, 10, rank)
    ex = GradientExchange(eng, sparse_rows_max_fraction=2.0, ids_bound=3 * 2 * 5)      # holds on rank 0 (3 lanes), not on rank 1 (4 lanes)
    rng = np.random.default_rng(rank)
    B = 3 + rank
    x = torch.from_numpy(rng.integers(0, N, (B, 2, 5)).astype(np.int32))
    y = torch.from_numpy(rng.integers(0, N, (B, 2, 5)).astype(np.int32))
    ex.global_users(y)
    ex.after_scoring()
    try:
        ex.after_backward(x, y)
        msg = "no error"
    except ValueError as e:
        msg = str(e)
    with open(os.path.join(out_dir, f"msg{rank}.txt"), "w") as f:
        f.write(msg)
    dist.barrier()          # both ranks get here: neither is left waiting inside a collective the other never entered
    dist.destroy_process_group()


def test_ids_bound_violated_on_one_rank_raises_on_every_rank(tmp_path):
    """Round 3 raised only on the rank whose batch exceeded ids_bound, with the other rank already inside the next collective (a hang).  The
    overflow flag now rides on the user-count all-reduce of the same step and every rank raises."""
    mp.spawn(_bound_violation_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        msg = open(tmp_path / f"msg{r}.txt").read()
        assert "1 rank(s) feed more ids" in msg and "raised on every rank" in msg, msg


class _FakeGmm:
    """eval_step(x, y) -> {'sums': [sum of lane means, users]}: stands for TcnGmmModel in the run loop (no GPU here)."""

    def eval_step(self, x, y):
        x = np.asarray(x, np.float64)
        return dict(sums=torch.tensor([float(x.sum()), float(x.shape[0])], dtype=torch.float64))


class _ListLoader:
    def __init__(self, batches):
        self.batches, self.i, self.done = batches, 0, False

    def get_batch(self):
        b = self.batches[self.i]
        self.i += 1
        self.done = self.i == len(self.batches)
        return b, b, None


def _gmm_eval_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from types import SimpleNamespace

    from hiertcn_amd.distributed import init_from_env
    from hiertcn_amd.run_gmm_xing import evaluate_gmm
    init_from_env("gloo")
    rng = np.random.default_rng(10 + rank)
    n_batches = 3 if rank == 0 else 5          # shards of different lengths: a collective inside the batch loop would hang here
    batches = [rng.random((4, 6)) for _ in range(n_batches)]
    args = SimpleNamespace(log_path=os.path.join(out_dir, "log") + "/")
    os.makedirs(args.log_path + "test/", exist_ok=True)
    m, _, _ = evaluate_gmm(args, _ListLoader(batches), _FakeGmm(), prefix="test")
    np.save(os.path.join(out_dir, f"gmm_{rank}.npy"), np.array([m["loss_epoch"]] + [b.sum() / 4 for b in batches]))
    torch.distributed.destroy_process_group()


def test_evaluate_gmm_with_shards_of_different_lengths(tmp_path):
    """run_gmm_xing.evaluate_gmm under two ranks whose loaders yield 3 and 5 batches: it returns (no rank waits in a per-batch collective
    for a rank that is already done) and both ranks report the mean over all 8 batches."""
    mp.spawn(_gmm_eval_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = (np.load(tmp_path / f"gmm_{r}.npy") for r in range(2))
    want = np.concatenate([a[1:], b[1:]]).mean()
    assert abs(a[0] - want) < 1e-12 and abs(b[0] - want) < 1e-12
    assert open(tmp_path / "log" / "test" / "results.txt").read().startswith("loss_epoch ")
