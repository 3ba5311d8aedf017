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
    from hiertcn_amd.distributed import all_reduce_gradients, init_from_env
    import torch.distributed as dist
    init_from_env("gloo")
    g = torch.arange(1000, dtype=torch.float32) * (rank + 1)      # stands for the flat gradient arena
    w = all_reduce_gradients(g, bucket_bytes=4 * 96)              # 96-float buckets: 11 async all-reduces, a ragged last one
    if rank == 0:
        np.save(os.path.join(out_dir, "g.npy"), g.numpy())
        np.save(os.path.join(out_dir, "w.npy"), np.array([w]))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_gradient_all_reduce(tmp_path):
    mp.spawn(_grad_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert np.load(tmp_path / "w.npy")[0] == 2
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


def _exchange_worker(rank, world, port, out_dir, sparse_fraction):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from hiertcn_amd.distributed import GradientExchange, init_from_env
    import torch.distributed as dist
    init_from_env("gloo")
    N, E = 97, 8
    eng = _HostEngine(N, E, 33, rank)
    rng = np.random.default_rng(rank)
    B = 3 + rank                                      # uneven shards: different numbers of ids per rank
    x = torch.from_numpy(rng.integers(0, N, (B, 2, 5)).astype(np.int32))
    y = torch.from_numpy(rng.integers(0, N, (B, 2, 5)).astype(np.int32))
    eng.touch(x, y, rank)
    np.save(os.path.join(out_dir, f"before{rank}.npy"), eng.arena.numpy().copy())
    ex = GradientExchange(eng, sparse_rows_max_fraction=sparse_fraction)
    users = ex.global_users(y)
    ex.after_scoring()
    ex.after_backward(x, y)
    np.save(os.path.join(out_dir, f"after{rank}.npy"), eng.arena.numpy())
    if rank == 0:
        np.save(os.path.join(out_dir, "meta.npy"), np.array([float(users[0]), 1.0 if ex.last["item_table"] == "row-sparse" else 0.0]))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_exchange_dense_and_row_sparse(tmp_path):
    """hiertcn_amd.distributed.GradientExchange on two ranks: every part of the arena ends as the SUM over ranks on BOTH ranks,
    bit-identical between them, whether the item table goes densely or as (row id, row) pairs; the user count is global."""
    for frac, want_sparse in ((0.0, 0.0), (2.0, 1.0)):
        d = tmp_path / str(frac)
        d.mkdir()
        mp.spawn(_exchange_worker, args=(2, _free_port(), str(d), frac), nprocs=2, join=True)
        b0, b1 = np.load(d / "before0.npy"), np.load(d / "before1.npy")
        a0, a1 = np.load(d / "after0.npy"), np.load(d / "after1.npy")
        users, sparse = np.load(d / "meta.npy")
        assert sparse == want_sparse
        assert users == 3 + 4                                        # every lane of both shards has a target
        assert np.array_equal(a0, a1)                                 # replicas stay bit-identical
        np.testing.assert_allclose(a0, b0 + b1, rtol=1e-6, atol=1e-6)
