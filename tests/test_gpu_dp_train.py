"""-m gpu: data-parallel training (model.py:135-141 on the global batch).  Two ranks on ONE GPU, collectives over
gloo (RCCL refuses two ranks on one device; the code path above the backend is the same), each rank on its shard of the users --
an uneven split (5 + 4 lanes: normalising by the local user count would weigh the shards 1/5 : 1/4) -- must produce the same summed gradient and the same Adam
update as one rank on the whole batch: global user count in the loss normaliser, scoring-layer collective between the two
halves of the step, item table exchanged densely or as (row id, row) pairs."""
import os
import socket

import numpy as np
import pytest

from tests.gpu_common import make_batch, make_params, small_args

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

N, D, H, B, S, L = 3000, 64, 32, 9, 3, 11
LR = 1e-2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _batch():
    rng = np.random.default_rng(77)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.1 * rng.standard_normal((B, 2 * H))).astype(np.float32)
    return x, y, mask, st0


def _run(rank, world, first, count, out_dir, tag):
    from hiertcn_amd.engine import metrics_dict
    from hiertcn_amd.model import HierTCNModel
    args = small_args(N=N, d=D, H=H, S=S, L=L, B=B)
    model = HierTCNModel(args, make_params(args), train=True)
    x, y, mask, st0 = _batch()
    sl = slice(first, first + count)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")  # noqa: E731
    out = model.train_step_device(dev(x[sl]), dev(y[sl]), dev(mask[sl]), dev(st0[sl]), LR)
    torch.cuda.synchronize()
    if rank == 0:
        g = model.engine.export("grads")
        w = model.engine.export("weights")
        np.savez(os.path.join(out_dir, tag + ".npz"), loss=metrics_dict(out["metrics8"])["loss"],
                 users=metrics_dict(out["metrics8"])["user_count"], table=model._exchange.last.get("item_table", ""),
                 **{"g:" + k: v for k, v in g.items()}, **{"w:" + k: v for k, v in w.items()})
    model.engine.close()


def _worker(rank, world, port, out_dir, tag, sparse_fraction):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HIERTCN_DP_SPARSE_FRACTION=str(sparse_fraction))
    import torch.distributed as dist
    from hiertcn_amd.distributed import init_from_env
    torch.cuda.set_device(0)
    # the shared-GPU guard: two ranks on one device -> no launch that waits on co-resident workgroups, on a handle created BEFORE
    # init_from_env took the decision as well as on the ones created after it
    from hiertcn_amd import distributed as hd
    from hiertcn_amd.engine import Engine
    early = Engine(small_args(N=N, d=D, H=H, S=S, L=L, B=B), hier=True)
    assert early.get_tuning()["gru_scan_persistent"] == -1 and not hd.shared_device()
    init_from_env("gloo")
    assert hd.shared_device()
    late = Engine(small_args(N=N, d=D, H=H, S=S, L=L, B=B), hier=True)
    for e in (early, late):
        t = e.get_tuning()
        assert t["gru_scan_persistent"] == 0 and t["lgru_persistent"] == 0, t
        e.close()
    first, count = (0, 5) if rank == 0 else (5, 4)      # uneven shards: 5 + 4 lanes
    _run(rank, world, first, count, out_dir, tag)
    dist.barrier()
    dist.destroy_process_group()


def _single(out_dir):
    import torch.multiprocessing as mp
    mp.spawn(_single_worker, args=(out_dir,), nprocs=1, join=True)


def _single_worker(_i, out_dir):
    torch.cuda.set_device(0)
    _run(0, 1, 0, B, out_dir, "single")


@pytest.mark.parametrize("sparse_fraction,table", [(0.0, "dense"), (1.5, "row-sparse")])
def test_two_ranks_equal_one_rank_on_the_concatenated_batch(tmp_path, sparse_fraction, table):
    import torch.multiprocessing as mp
    _single(str(tmp_path))
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), "dp", sparse_fraction), nprocs=2, join=True)
    one, two = np.load(tmp_path / "single.npz"), np.load(tmp_path / "dp.npz")
    assert str(two["table"]) == table
    assert float(one["users"]) == float(two["users"]) == B              # summed over the ranks
    assert abs(float(one["loss"]) - float(two["loss"])) < 1e-5           # the global batch's loss (model.py:116-117)
    for k in one.files:
        if not k[:2] in ("g:", "w:"):
            continue
        a, b = one[k], two[k]
        if k.startswith("g:"):
            # the ranks' sums differ from the one-rank sums by fp32 summation order only
            assert np.abs(a - b).max() <= 2e-5 * max(np.abs(a).max(), 1e-12) + 1e-9, (k, np.abs(a - b).max(), np.abs(a).max())
        else:
            # first Adam step: update = lr * g / (|g| + eps) -- compare where the gradient is clear of the rounding noise
            g = one["g:" + k[2:]]
            clear = np.abs(g) > 1e-3 * np.abs(g).max()
            assert np.abs(a - b)[clear].max() < 1e-3 * LR if clear.any() else True, k
            assert np.abs(a - b).max() <= 2 * LR + 1e-7, k
