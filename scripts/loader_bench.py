"""Can the batch packer feed the GPU?  Measures `Dataloader_hier_model_xing.get_packed_batch` (the native packer that stands
where /root/reference/data_loader.py:219-272 stands) in sessions/s on a synthetic XING-format CSV, at the reference's batch width
and at the wide batches the HIP path is built for, with and without the background prefetch thread
(hiertcn_amd.data_loader.Prefetcher).  CPU only.

    python scripts/loader_bench.py [users=20000] [lanes ...]
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from hiertcn_amd.args import hier_tcn_args  # noqa: E402
from hiertcn_amd.data_loader import Dataloader_hier_model_xing, Prefetcher  # noqa: E402
from hiertcn_amd.data_prepare import load_xing  # noqa: E402

users = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
lanes_list = [int(v) for v in sys.argv[2:]] or [32, 512, 4096]
tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_synth_xing.py"), str(users), tmp])
t0 = time.perf_counter()
data = load_xing(os.path.join(tmp, "dataset", "interactions.csv"), max_activity=500)
print(f"load_xing: {data[0].shape[0]} users, {data[1].shape[0]} events in {time.perf_counter() - t0:.2f} s", flush=True)
n_items = int(data[1][:, 1].max()) + 1
for lanes in lanes_list:
    args = hier_tcn_args(item_num=n_items, batch_size=lanes, max_session_num=10, max_activity_len=20)
    for mode in ("direct", "prefetch"):
        ld = Dataloader_hier_model_xing(args, "all", data=data)
        src = Prefetcher(ld.get_packed_batch, depth=4) if mode == "prefetch" else None
        get = src.get if src else ld.get_packed_batch
        get()                                           # warm-up (first fill of the lane queues)
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 3.0:
            x, y, mask, widths = get()
            n += 1
        dt = time.perf_counter() - t0
        if src:
            src.close()
        sess = n * lanes * args.max_session_num
        print(f"lanes={lanes:5d} {mode:8s}: {n / dt:8.1f} batches/s = {sess / dt / 1e3:9.1f} k sessions/s "
              f"(batch ids {tuple(x.shape)}, {x.nbytes * 2 / 1e6:.2f} MB)", flush=True)
