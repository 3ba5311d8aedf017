"""Run by tests/test_gpu_run_loop.py in a process of its own with HIERTCN_LIB = libhiertcn_gfx950_testhooks.so: evaluate_hier over 60 batches, the
persistent scan of batch 3 made to time out (fault injection: a barrier that waits for a workgroup that does not exist).  The run loop's health
watch (run_hier_xing.py: an asynchronous read of the health word every 20 batches, where the reference writes results.txt) must raise well before
the epoch ends -- by batch 22 -- instead of reporting an epoch of metrics built on wrong carried states.  Prints HEALTH_WATCH_PROBE <batch>."""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.data_loader import Dataloader_hier_model_xing
from hiertcn_amd.data_prepare import load_xing
from hiertcn_amd.model import HierTCNModel
from hiertcn_amd.run_hier_xing import evaluate_hier
from tests.gpu_common import make_params

assert os.environ.get("HIERTCN_LIB", "").endswith("testhooks.so")
CSV = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "interactions_synth.csv")
data = load_xing(CSV)
n_items = int(data[1][:, 1].max()) + 1
args = hier_tcn_args(item_num=n_items, emb_dim=64, hidden_dim=64, num_layer=2, tcn_channel=[64, 64], kernel_size=5, batch_size=4, max_session_num=3,
                     max_activity_len=6)
log = sys.argv[1]
args.log_path = log + "/"
os.makedirs(args.log_path + "test/", exist_ok=True)


class Sixty:
    """60 packed batches: the synthetic loader, started over whenever it runs out (no prefetch thread: the batch count is the loop's)."""

    def __init__(self, n=60):
        self.n, self.i = n, 0
        self.inner = Dataloader_hier_model_xing(args, "test", data=data)
        self.viewer_table = self.inner.viewer_table

    def get_packed_batch(self):
        if self.inner.done:
            self.inner = Dataloader_hier_model_xing(args, "test", data=data)
        self.i += 1
        return self.inner.get_packed_batch()

    @property
    def done(self):
        return self.i >= self.n


if len(sys.argv) > 2 and sys.argv[2] == "gmm":
    # the same watch in run_gmm_xing.evaluate_hier_gmm (model_hier with a mixture low level: the high level's scan runs on the MAIN handle, persistent)
    import torch

    from hiertcn_amd.model_hier_gmm import HierGmmModel
    from hiertcn_amd.run_gmm_xing import evaluate_hier_gmm
    args.model_low_type, args.loss, args.num_mixture, args.sigma, args.output_dim, args.kernel_size = "tcn_gmm", "gmm_prob", 3, 0.1, n_items, 3
    gm = HierGmmModel(args, None, "hier")
    assert gm.engine.get_tuning()["gru_scan_persistent"] == -1
    calls = {"n": 0, "bad": 3}
    step = gm.eval_step_hier

    def faulty_step(*a, **k):
        calls["n"] += 1
        if calls["n"] == calls["bad"]:
            os.environ["HIERTCN_GRU_TEST_MISSING_WG"] = "1"
        try:
            return step(*a, **k)
        finally:
            os.environ.pop("HIERTCN_GRU_TEST_MISSING_WG", None)

    gm.eval_step_hier = faulty_step
    zero = lambda: torch.zeros((4, 128), dtype=torch.float32, device="cuda")  # noqa: E731
    ld = Sixty()
    try:
        evaluate_hier_gmm(args, ld, zero(), gm, prefix="test")
    except RuntimeError as e:
        m = re.search(r"by batch (\d+)", str(e))
        assert m and "timed out" in str(e), str(e)
        print("HEALTH_WATCH_PROBE", int(m.group(1)), ld.i)
        calls["bad"] = -1
        evaluate_hier_gmm(args, Sixty(n=25), zero(), gm, prefix="test")
        print("HEALTH_WATCH_PROBE recovered")
        sys.exit(0)
    print("HEALTH_WATCH_PROBE no exception after", ld.i, "batches")
    sys.exit(1)

model = HierTCNModel(args, make_params(args))
assert model.engine.get_tuning()["gru_scan_persistent"] == -1
# the fault switch of the test-hooks build is looked up when the scan is LAUNCHED: up for the launch of batch 3 alone
launches = {"n": 0, "bad": 3}
enqueue = model.eval_step_packed_async


def faulty(*a, **k):
    launches["n"] += 1
    if launches["n"] == launches["bad"]:
        os.environ["HIERTCN_GRU_TEST_MISSING_WG"] = "1"
    try:
        return enqueue(*a, **k)
    finally:
        os.environ.pop("HIERTCN_GRU_TEST_MISSING_WG", None)


model.eval_step_packed_async = faulty
ld = Sixty()
try:
    evaluate_hier(args, ld, np.zeros((4, 128)), model, prefix="test")
except RuntimeError as e:
    m = re.search(r"by batch (\d+)", str(e))
    assert m and "timed out" in str(e), str(e)
    print("HEALTH_WATCH_PROBE", int(m.group(1)), ld.i)
    # the handle was reset by the synchronising read behind the exception: a clean epoch now runs through
    os.environ.pop("HIERTCN_GRU_TEST_MISSING_WG", None)
    launches["bad"] = -1
    ld2 = Sixty(n=25)
    evaluate_hier(args, ld2, np.zeros((4, 128)), model, prefix="test")
    print("HEALTH_WATCH_PROBE recovered")
    sys.exit(0)
print("HEALTH_WATCH_PROBE no exception after", ld.i, "batches")
sys.exit(1)
