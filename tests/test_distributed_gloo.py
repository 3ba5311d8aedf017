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
