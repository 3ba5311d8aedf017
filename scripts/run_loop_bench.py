"""End-to-end `evaluate_hier` (the reference's run_hier_xing.py:112-207 loop: loader -> model.eval_step -> meters) on a synthetic XING-format
split at the reference's batch (32 lanes x 10 slots x <= 20 steps): sessions/s of the WHOLE loop, with its split into loader time and
step time, next to the device-only figure of scripts/cfg2_bench.py.
usage: python scripts/run_loop_bench.py [users=6000] [lanes=32] [workers=0|1]"""
import os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.data_loader import Dataloader_hier_model_xing
from hiertcn_amd.data_prepare import load_xing
from hiertcn_amd.model import HierTCNModel
from hiertcn_amd.run_hier_xing import evaluate_hier

users = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 32
workers = int(sys.argv[3]) if len(sys.argv) > 3 else 0
tmp = tempfile.mkdtemp()
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_synth_xing.py"), str(users), tmp])
data = load_xing(os.path.join(tmp, "dataset", "interactions.csv"), max_activity=500)
n_items = int(data[1][:, 1].max()) + 1
d = int(os.environ.get("D", "256"))
args = hier_tcn_args(item_num=n_items, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, batch_size=lanes, max_session_num=10,
                     max_activity_len=20, model_low_type=os.environ.get("LOW", "tcn"), load_worker_num=workers)
args.log_path = None
model = HierTCNModel(args)
count = {"batches": 0, "sessions": 0}
_orig = model.eval_step_packed
def _counting(x, y, mask, widths, state_np, want_pred=False):
    count["batches"] += 1
    count["sessions"] += int((y.reshape(y.shape[0], y.shape[1], -1) > 0).any(-1).sum())
    return _orig(x, y, mask, widths, state_np, want_pred)
model.eval_step_packed = _counting
_orig_async = model.eval_step_packed_async
def _counting_async(x, y, mask, widths):
    count["batches"] += 1
    count["sessions"] += int((y.reshape(y.shape[0], y.shape[1], -1) > 0).any(-1).sum())
    return _orig_async(x, y, mask, widths)
model.eval_step_packed_async = _counting_async
for rep in range(2):   # (the second pass is the measured one: first-call costs out of the way)
    count["batches"] = count["sessions"] = 0
    loader = Dataloader_hier_model_xing(args, "all", data=data)
    state = np.zeros((lanes, args.hidden_dim * args.num_layer), np.float32)
    t0 = time.perf_counter()
    summary, _val, t_load, t_run, state = evaluate_hier(args, loader, state, model, prefix="test")
    dt = time.perf_counter() - t0
    if getattr(loader, "_prefetcher", None) is not None:
        loader._prefetcher.close()
print("evaluate_hier: %d users, lanes %d, workers %d: %d batches, %d sessions in %.3f s (loader %.3f s, steps %.3f s) = %.3f ms per batch, %.0f sessions/s" % (
    users, lanes, workers, count["batches"], count["sessions"], dt, t_load, t_run, dt / max(count["batches"], 1) * 1e3, count["sessions"] / dt), flush=True)
print({k: round(float(v), 5) for k, v in summary.items()})

if os.environ.get("TRAIN"):   # the training loop of run_hier (run_hier_xing.py:275-301), blocking calls vs the pipelined form
    tm = HierTCNModel(args, train=True)
    for form in ("blocking", "pipelined"):
        ld = Dataloader_hier_model_xing(args, "all", data=data)
        st = np.zeros((lanes, args.hidden_dim * args.num_layer), np.float32)
        n = int(os.environ["TRAIN"])
        t0 = time.perf_counter()
        if form == "blocking":
            for _ in range(n):
                x, y, mask, _w = ld.get_packed_batch()
                out = tm.train_step_packed(x, y, mask, st, 1e-3)
                st = out["state"]
        else:
            tm.train_begin(st)
            pend = None
            for _ in range(n):
                x, y, mask, _w = ld.get_packed_batch()
                nxt = tm.train_step_packed_async(x, y, mask, 1e-3)
                if pend is not None:
                    pend.result()
                pend = nxt
            pend.result()
            st = tm.train_end()
        dt = time.perf_counter() - t0
        print("training loop, %s: %d steps of %d lanes: %.3f ms per step end to end" % (form, n, lanes, dt / n * 1e3), flush=True)
