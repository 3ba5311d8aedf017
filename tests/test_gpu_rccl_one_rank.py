"""-m gpu: RCCL executed, with ONE rank, on the GPU box.  The driver's 8-GPU scaling run is the only place several GPUs exist;
everything short of a second GPU is proven here: librccl loads, `init_process_group("nccl", world_size=1, device_id=...)` builds a communicator,
every collective of hiertcn_amd/distributed.py runs through it ON DEVICE MEMORY with the dtypes the step uses -- the fp64 metric all-reduce,
the MAX all-reduce of the timing, the user-count all-reduce, the scoring gradient's reduce-scatter + all-gather (async on the backend's stream,
waited for by the compute stream), the bucketed all-reduce alternative, the small segment's all-reduce, and the item table's (row id, row)
all-gather -- and a training step through them leaves the gradient arena and the Adam update BIT FOR BIT what the same step leaves without a
process group (world 1: every sum is the identity, every gather returns the rank's own rows)."""
import os
import socket

import numpy as np
import pytest

from tests.gpu_common import make_batch, make_params, small_args

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

N, D, H, B, S, L = 3000, 64, 32, 8, 3, 11
LR = 1e-2


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _step(out_dir, tag):
    from hiertcn_amd.engine import metrics_dict
    from hiertcn_amd.model import HierTCNModel
    args = small_args(N=N, d=D, H=H, S=S, L=L, B=B)
    model = HierTCNModel(args, make_params(args), train=True)
    rng = np.random.default_rng(77)
    x, y, mask = make_batch(rng, B, S, L, N)
    st0 = (0.1 * rng.standard_normal((B, 2 * H))).astype(np.float32)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a)).to("cuda:0")  # noqa: E731
    out = model.train_step_device(dev(x), dev(y), dev(mask), dev(st0), LR)
    torch.cuda.synchronize()
    g, w = model.engine.export("grads"), model.engine.export("weights")
    ex = model._exchange
    np.savez(os.path.join(out_dir, tag + ".npz"), m8=out["metrics8"].cpu().numpy(), loss=metrics_dict(out["metrics8"])["loss"],
             table=ex.last.get("item_table", ""), dense_taken=str(ex.dense_taken), **{"g:" + k: v for k, v in g.items()}, **{"w:" + k: v for k, v in w.items()})
    model.engine.close()


def _plain(_i, out_dir):
    torch.cuda.set_device(0)
    _step(out_dir, "plain")


def _rccl(_i, port, out_dir, tag, env):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", **env)
    import torch.distributed as dist
    from hiertcn_amd import distributed as D_
    rank, world, _ = D_.init_from_env("nccl", device=torch.device("cuda:0"), force_group=True)
    assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl" and D_._active()
    # the evaluation path's one collective, on the dtype it uses (8 fp64 accumulators), and the bench's MAX over ranks
    m8 = torch.arange(8, dtype=torch.float64, device="cuda:0") * 1.25 + 0.5
    want = m8.clone()
    assert D_.all_reduce_metrics(m8) is m8 and torch.equal(m8, want)
    assert D_.max_over_ranks(1.5, device="cuda:0") == 1.5
    D_.barrier()
    # the primitives above the backend, on device tensors
    t = torch.randn(4096, device="cuda:0")
    ref = t.clone()
    D_._sum_async(t).wait()
    D_._sum_async_bucketed(t, bucket_bytes=4096).wait()      # four buckets in flight
    torch.cuda.synchronize()
    assert torch.equal(t, ref)
    rows = torch.randn((37, 64), device="cuda:0")
    ids = torch.arange(37, dtype=torch.int32, device="cuda:0")
    got_r, got_i = D_._gather_cat(rows), D_._gather_cat(ids)
    assert got_r.shape == (1, 37, 64) and torch.equal(got_r[0], rows) and torch.equal(got_i[0], ids)
    _step(out_dir, tag)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("tag,env,table,taken", [("rs_ag_dense", dict(HIERTCN_DP_SPARSE_FRACTION="0.0"), "dense", "rs_ag"),
                                                 ("rs_ag_rows", dict(HIERTCN_DP_SPARSE_FRACTION="1.5"), "row-sparse", "rs_ag"),
                                                 ("allreduce", dict(HIERTCN_DP_SPARSE_FRACTION="0.0", HIERTCN_DP_COLLECTIVE="allreduce"), "dense", "allreduce_bucketed")])
def test_training_step_through_a_one_rank_rccl_group_equals_the_step_without_a_group(tmp_path, tag, env, table, taken):
    import torch.multiprocessing as mp
    mp.spawn(_plain, args=(str(tmp_path),), nprocs=1, join=True)
    mp.spawn(_rccl, args=(_free_port(), str(tmp_path), tag, env), nprocs=1, join=True)
    one, two = np.load(tmp_path / "plain.npz"), np.load(tmp_path / (tag + ".npz"))
    assert str(two["table"]) == table and str(two["dense_taken"]) == taken       # the branch that was meant to run, ran -- over RCCL
    assert str(one["table"]) == "" and str(one["dense_taken"]) == "None"          # ... and the reference step entered no collective
    # The two steps run in two PROCESSES: a one-rank collective is the identity, so what may differ is what differs between any two runs of the
    # step -- fp32 atomics in a few reductions and the scoring kernels' item-walk start reorder sums (scripts/soak_refbatch.py: "gradients within
    # 1e-5").  Most runs are bitwise equal; the bound below is the soak's.
    np.testing.assert_allclose(one["m8"], two["m8"], rtol=1e-6, atol=0)
    n = 0
    for k in one.files:
        if k[:2] in ("g:", "w:"):
            d, ref = np.linalg.norm(one[k].astype(np.float64) - two[k]), np.linalg.norm(one[k].astype(np.float64))
            assert d <= 1e-5 * max(ref, 1e-30), (k, d, ref)
            n += 1
    assert n > 20
