"""-m gpu: bench.py at N > 1 arms a watchdog over its secondary legs (the first multi-rank run of the gradient collectives happens on the
driver's node).  When it fires, rank 0 still prints the headline line -- with `secondary_legs.error` -- and every rank ends with a NON-ZERO
status: a run whose secondary legs never finished must not look like a clean one to a harness.  Two ranks on this box's one GPU
over gloo (HIERTCN_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device) with a limit the legs cannot meet."""
import json
import os
import socket
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_watchdog_prints_the_line_and_exits_non_zero():
    env = dict(os.environ, HIERTCN_BENCH_BACKEND="gloo", HIERTCN_BENCH_LEG_LIMIT_S="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--users-per-step", "256",
           "--train-users-per-step", "256", "--train-steps", "3", "--exact-users-per-step", "1024", "--exact-steps", "3", "--no-gmm"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode != 0, "the watchdog ended the run: the launcher must report a failure\n" + r.stdout[-1500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-3000:] + r.stderr[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["unit"] == "sessions/s"
    assert "not finished" in out["secondary_legs"]["error"] and "status 3" in out["secondary_legs"]["error"]
    assert out["roofline"]["launch_ms"] > 0 and out["multi_gpu"]["ranks"] == 2 and out["multi_gpu"]["backend"] == "gloo"


def test_two_rank_run_completes_every_leg_and_exits_zero():
    """The whole N > 1 path of bench.py on this box's one GPU (two ranks over gloo, small batches): the headline with the metric all-reduce,
    the mixture leg on rank 0 alone (it must enter no collective), the training leg with the gradient collectives, the split16 leg -- one
    JSON line, status 0, no `secondary_legs` marker."""
    env = dict(os.environ, HIERTCN_BENCH_BACKEND="gloo", HIERTCN_BENCH_LEG_LIMIT_S="600", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--users-per-step", "256",
           "--train-users-per-step", "128", "--train-steps", "1", "--train-warmup", "1", "--exact-users-per-step", "256", "--exact-steps", "1",
           "--exact-warmup", "1"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, r.stdout[-3000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and "secondary_legs" not in out
    assert out["multi_gpu"]["ranks"] == 2 and len(out["multi_gpu"]["per_rank"]) == 2
    assert out["config"]["parallelism"] == "user-batch dp2" and out["scaling"] == "weak"
    tr = out["train"]
    assert "error" not in tr, tr
    assert tr["sessions_per_step"] == 128 * 20 * 2 and tr["collectives"]["backend"] == "gloo" and tr["collectives"]["exposed_ms_per_step"] is not None
    ex = out["exact"]
    assert "error" not in ex and ex["n_gpus"] == 2 and ex["value"] > 0
    g = out["gmm"]
    assert "error" not in g and g["training"]["ms_per_step"] > 0 and g["training"]["ms_per_step_gru_gmm"] > 0
