#!/usr/bin/env python3
"""Headline benchmark: HierTCN forward + full-catalogue scoring, sessions/sec on XING-shaped
synthetic data (BASELINE.json configs[2]: 200k users x 20 sessions x 30 steps, 300k items, d=256,
bf16 storage / fp32 accumulate), user-batch data parallel over N GPUs.

    python bench.py --gpus 1 --steps 6 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one evaluation pass (session pool -> GRU scan -> fused TCN -> fused scoring with loss /
rank epilogue -> lane metrics [-> RCCL all-reduce of the 8 metric sums when N > 1]) over one batch
of `--users-per-step` users per GPU, ids already resident in HBM.  Weak scaling: every rank
walks its own shard of the 200k users.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0  # MI355X dense bf16 (MI355X_MICROARCH.md, chip-level parameters)
U_TOTAL, S_SESS, L_STEPS, N_ITEMS, DIM = 200_000, 20, 30, 300_000, 256


def zipf_sampler(n_items: int, a: float = 1.1):
    """Zipf(a) over a fixed random permutation of the items 1..N-1."""
    w = 1.0 / np.arange(1, n_items, dtype=np.float64) ** a
    cdf = np.cumsum(w / w.sum())
    perm = (np.random.default_rng(0).permutation(n_items - 1) + 1).astype(np.int32)

    def draw(rng, shape):
        k = np.searchsorted(cdf, rng.random(shape))
        return perm[np.minimum(k, n_items - 2)]

    return draw


def make_users(draw, first_user: int, n_users: int, ragged: bool):
    """ids for users [first_user, first_user+n_users): deterministic per 256-user block so the data
    of a user does not depend on the GPU count."""
    blocks = []
    u = first_user
    while u < first_user + n_users:
        blk = u // 256
        rng = np.random.default_rng([0, blk])
        ids = draw(rng, (256, S_SESS, L_STEPS))
        if ragged:
            lens = np.random.default_rng([1, blk]).integers(1, L_STEPS + 1, (256, S_SESS))
            ids = ids * (np.arange(L_STEPS)[None, None, :] < lens[..., None])
        lo = u - blk * 256
        hi = min(256, first_user + n_users - blk * 256)
        blocks.append(ids[lo:hi])
        u = blk * 256 + hi
    y = np.concatenate(blocks, 0).astype(np.int32)
    x = np.zeros_like(y)
    x[:, :, 1:] = y[:, :, :-1]  # the loader's shift (data_loader.py:260-261); x[:, :, 0] is the null id
    mask = np.ones((n_users, S_SESS), np.float32)
    mask[:, -1] = 0.0  # a user's last session resets the state (data_loader.py:186-187)
    return x, y, mask


def train_leg(a, args, params, rank, world, draw, shard_first, shard, dist, max_over_ranks):
    """-> the `train` object: ms per optimizer step, sessions/s over all ranks, the collectives' cost (exposed = step time minus the
    same step with the collectives switched off; standalone = the three gradient collectives alone, back to back) and their bus
    bandwidth against the node's 7 x 153 GB/s of xGMI per GPU."""
    import copy

    import torch

    from hiertcn_amd.model import HierTCNModel

    Bt = a.train_users_per_step
    n_total = a.train_warmup + a.train_steps
    lr = 1e-3
    # Phase 1, rank-local (no collective inside): everything that can fail on ONE rank -- the training handle's arenas, the batches, and one
    # forward + backward pass (kernel / workspace errors surface here).  Then the ranks AGREE on success with one
    # all-reduce that every rank reaches whatever happened before; only then does the collective part start.  (A rank that raised inside
    # the collective part while the others wait in RCCL would hang the run and the JSON line with it.)
    model, err = None, None
    try:
        targs = copy.copy(args)
        targs.train = True
        targs.batch_size = Bt           # the gradient exchange sizes its row-sparse all-gather from args.batch_size: the batch actually fed
        model = HierTCNModel(targs, params, train=True)
        eng = model.engine
        batches = []
        for i in range(n_total):
            first = shard_first + (i * Bt) % max(shard - Bt, 1)
            x, y, mask = make_users(draw, first, Bt, a.ragged)
            batches.append((torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(mask).cuda()))
        state0 = torch.zeros((Bt, eng.HT), dtype=torch.float32, device="cuda")
        state_out = torch.empty_like(state0)
        m8 = [torch.zeros(8, dtype=torch.float64, device="cuda") for _ in range(n_total)]
        if world > 1:   # forward + backward only: no collective, no Adam step (the replicas' optimizer state stays identical)
            eng.train_forward_backward(batches[0][0], batches[0][1], batches[0][2], state0, state_out=state_out, metrics8=m8[0])
            torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        import traceback
        err = {"error": repr(e)[:300], "where": traceback.format_exc()[-600:], "rank": rank}
    if world > 1:
        ok = torch.tensor([0.0 if err else 1.0], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0 and err is None:
            err = {"error": "the training leg failed on another rank before its first collective", "rank": rank}
    if err is not None:
        if model is not None:
            model.engine.close()
        return err

    def run(steps_from, steps_to):
        for i in range(steps_from, steps_to):
            xb, yb, mb = batches[i]
            model.train_step_device(xb, yb, mb, state0, lr, state_out=state_out, metrics8=m8[i])

    def timed(steps_from, steps_to):
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps_from, steps_to)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()
        return max_over_ranks(time.perf_counter() - t0, "cuda")

    run(0, a.train_warmup)
    dt = timed(a.train_warmup, n_total)
    losses = [float(v[0] / v[6]) for v in torch.stack(m8[a.train_warmup:]).cpu()]
    ms_step = dt / max(a.train_steps, 1) * 1e3
    out = {
        "what": "one optimizer step at cfg-3 (forward + masked lane-mean CE + backward of every variable + gradient collectives + "
                "TF-form Adam), users sharded over the ranks, global-batch loss normalisation",
        "users_per_step_per_gpu": Bt,
        "sessions_per_step": Bt * S_SESS * world,
        "steps": a.train_steps, "warmup": a.train_warmup,
        "ms_per_step": round(ms_step, 3),
        "sessions_per_s": round(Bt * S_SESS * world * a.train_steps / dt, 1),
        "loss_first_last": [round(losses[0], 5), round(losses[-1], 5)] if losses else None,
        "collectives": None,
    }
    ex = model._exchange
    seg = {k: eng.grad_segment(k) for k in (0, 1, 2)}
    coll = {"backend": (("rccl" if dist.get_backend() == "nccl" else dist.get_backend()) if world > 1 else None), "algorithm": ex.mode, "algorithm_taken": ex.dense_taken, "item_table_exchange": ex.last.get("item_table", "dense"),
            "bytes": {"item_table": seg[0][1] * 4, "scoring_layer": seg[1][1] * 4, "rest": seg[2][1] * 4}}
    if world > 1:
        # (a) the same steps with the collectives off (replicas drift apart from here on: timing only, the bench ends after it)
        os.environ["HIERTCN_DP_OFF"] = "1"
        dt_off = timed(a.train_warmup, n_total)
        os.environ.pop("HIERTCN_DP_OFF")
        # (b) the three collectives alone, back to back, on the gradient arena as it stands
        g = eng.grad_arena()
        xb, yb, _ = batches[-1]
        torch.cuda.synchronize()
        dist.barrier()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ex.after_scoring()
            ex.after_backward(xb, yb)
        torch.cuda.synchronize()
        dist.barrier()
        t_coll = max_over_ranks((time.perf_counter() - t0) / reps, "cuda")
        nbytes = (seg[0][1] + seg[1][1] + seg[2][1]) * 4
        bus = 2.0 * (world - 1) / world * nbytes / t_coll / 1e9
        coll.update({"standalone_ms": round(t_coll * 1e3, 3), "bus_GBps": round(bus, 1), "bus_peak_GBps": 7 * 153.0,
                     "bus_frac": round(bus / (7 * 153.0), 4),
                     "exposed_ms_per_step": round((dt - dt_off) / max(a.train_steps, 1) * 1e3, 3),
                     "ms_per_step_collectives_off": round(dt_off / max(a.train_steps, 1) * 1e3, 3)})
        del g
    out["collectives"] = coll
    eng.close()
    return out


def gmm_leg(rows=614_400, n_out=20_778, n_mix=5, dim=256, reps=5):
    """-> the `gmm` object: ms per hiertcn_gmm_loss call at BASELINE.json configs[4]'s shape, through the Gram form (csrc/gmm_gram.hip: |mu_m|^2
    as a C x C quadratic form of the hidden row, the default) and through the pass over the M [D, C] mean tables it replaces
    (HIERTCN_GMM_TABLE_PASS=1: the scoring kernel's main loop with a sum-of-squares epilogue), whose MFMA rate is `table_pass_tflops`."""
    import torch

    from hiertcn_amd.args import make_args
    from hiertcn_amd.model_tcn_gmm import TcnGmmModel, init_params
    ga = make_args([], model_type="tcn_gmm", item_num=n_out, output_dim=n_out, num_mixture=n_mix, sigma=0.1, loss="gmm_prob",
                   tcn_channel=[dim, dim], kernel_size=5, adjust=False)
    hid = (torch.randn(rows, dim, device="cuda") * 0.5).to(torch.bfloat16)
    y = torch.randint(1, n_out, (rows,), device="cuda", dtype=torch.int32)
    ms, losses = {}, {}
    saved = os.environ.pop("HIERTCN_GMM_TABLE_PASS", None)
    try:
        for name in ("gram", "table_pass"):
            if name == "table_pass":
                os.environ["HIERTCN_GMM_TABLE_PASS"] = "1"   # read at hiertcn_gmm_create
            model = TcnGmmModel(ga, init_params(ga, 3))
            losses[name] = model.loss_rows(hid, y).clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                model.loss_rows(hid, y)
            torch.cuda.synchronize()
            ms[name] = (time.perf_counter() - t0) / reps * 1e3
            model.close()
    finally:
        os.environ.pop("HIERTCN_GMM_TABLE_PASS", None)
        if saved is not None:
            os.environ["HIERTCN_GMM_TABLE_PASS"] = saved
    # a learned scale (args.sigma None: M (1 + 2 D) outputs): one paired pass per mixture over the interleaved means / log-scales table
    gl = make_args([], model_type="tcn_gmm", item_num=n_out, output_dim=n_out, num_mixture=n_mix, sigma=0.1, loss="gmm_prob",
                   tcn_channel=[dim, dim], kernel_size=5, adjust=False)
    gl.sigma = None
    model = TcnGmmModel(gl, init_params(gl, 3))
    model.loss_rows(hid, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        model.loss_rows(hid, y)
    torch.cuda.synchronize()
    ms["learned_scale"] = (time.perf_counter() - t0) / reps * 1e3
    model.close()
    # the TRAINING step of model_tcn_gmm (csrc/gmm_train.hip, C-ABI 12) at the reference's batch: 32 lanes (args.py:19) x 200 steps of the
    # single-level loader = 6,400 rows, the XING catalogue, d = 256, 5 mixtures: forward + backward of every variable + Adam, ids resident
    tB, tT = 32, 200
    rng = np.random.default_rng(5)
    ids = rng.integers(1, n_out, (tB, tT + 1))
    ids = ids * (np.arange(tT + 1)[None, :] <= rng.integers(2, tT + 1, tB)[:, None])
    tx = torch.as_tensor(np.ascontiguousarray(ids[:, :-1].astype(np.int32))).cuda()
    ty = torch.as_tensor(np.ascontiguousarray(ids[:, 1:].astype(np.int32))).cuda()
    tm = TcnGmmModel(ga, init_params(ga, 3))
    tm.enable_training()
    tl = []

    def step_local(model):   # forward + backward + Adam on THIS rank alone (this leg runs on rank 0 only: no collective may be entered here)
        out = model.train_forward_backward(tx, ty)
        model.adam_step(1e-3)
        return out["sums"]

    for _ in range(3):
        tl.append(step_local(tm))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        tl.append(step_local(tm))
    torch.cuda.synchronize()
    train_ms = (time.perf_counter() - t0) / 10 * 1e3
    tl = torch.stack(tl).cpu().numpy()
    tm.close()
    # ... and of model_gru_gmm (the GRU stack on the same one-hot rows: 2 layers, backward through the 200 steps)
    from hiertcn_amd.model_gru_gmm import GruGmmModel, init_params as gru_init
    gg = make_args([], model_type="gru_gmm", item_num=n_out, output_dim=n_out, num_mixture=n_mix, sigma=0.1, loss="gmm_prob", hidden_dim=dim, num_layer=2,
                   adjust=False)
    gm = GruGmmModel(gg, gru_init(gg, 3))
    gm.enable_training()
    for _ in range(2):
        step_local(gm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        step_local(gm)
    torch.cuda.synchronize()
    train_ms_gru = (time.perf_counter() - t0) / 5 * 1e3
    train_ms_gru_graph = None
    try:   # the same step with its forward + backward replayed as ONE hipGraph (the launch chain's host cost removed)
        gr = gm.capture_train_graph(tB, tT)
        for _ in range(2):
            gr.step(tx, ty, 1e-3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            gr.step(tx, ty, 1e-3)
        torch.cuda.synchronize()
        train_ms_gru_graph = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    except Exception as e:  # noqa: BLE001  (a secondary figure)
        train_ms_gru_graph = repr(e)[:200]
    gm.close()
    # ... and of model_hier with a mixture low level at the reference's batch (32 lanes x 10 sessions x 20 steps; emb 128, 2 x 128 across-session cells)
    hier_ms = {}
    try:
        from hiertcn_amd.args import hier_tcn_args
        from hiertcn_amd.model_hier_gmm import HierGmmModel
        hB, hS, hL = 32, 10, 20
        hg = torch.Generator(device="cpu").manual_seed(7)
        hy = torch.randint(1, n_out, (hB, hS, hL), generator=hg, dtype=torch.int32).cuda()
        hx = torch.zeros_like(hy)
        hx[:, :, 1:] = hy[:, :, :-1]
        hm = torch.ones((hB, hS), dtype=torch.float32, device="cuda")
        for low in ("tcn_gmm", "gru_gmm"):
            ha = hier_tcn_args(item_num=n_out, output_dim=n_out, emb_dim=128, hidden_dim=128, num_layer=2, tcn_channel=[128, 128], kernel_size=5,
                               num_mixture=n_mix, model_low_type=low, loss="gmm_prob", sigma=0.1)
            hmod = HierGmmModel(ha, None, "hier")
            hmod.enable_training()
            hst = torch.zeros((hB, 256), dtype=torch.float32, device="cuda")

            def hstep(hst):
                out = hmod.train_forward_backward_hier(hx, hy, hm, hst, "gmm_prob")     # (the local step: no collective on rank 0 alone)
                hmod.adam_step(1e-3)
                return out["state"]

            for _ in range(3):
                hst = hstep(hst)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                hst = hstep(hst)
            torch.cuda.synchronize()
            hier_ms[low] = round((time.perf_counter() - t0) / 10 * 1e3, 3)
            hmod.close()
    except Exception as e:  # noqa: BLE001  (a secondary figure)
        hier_ms["error"] = repr(e)[:200]
    diff = float((losses["gram"] - losses["table_pass"]).abs().max())
    flop = 2.0 * rows * dim * ((n_out + 63) // 64 * 64) * n_mix
    return {"training": {"what": "one optimizer step of model_tcn_gmm (fixed scale, gmm_prob): fp32 forward on the masters + backward of every variable + TF-form "
                                 "Adam (csrc/gmm_train.hip: the head's backward through G_m = W_m W_m^T, no [rows][M D] matrix; block 0's one-hot rows as "
                                 "scatter-adds), 32 lanes x 200 steps = 6,400 rows, ids resident; ms_per_step_gru_gmm: the same step of model_gru_gmm "
                                 "(2-layer GRU stack on the one-hot rows, forward and backward through the 200 steps as a (step, layer) wavefront of launches); ms_per_step_hier: model_hier with the mixture "
                                 "model as its low level (hiertcn_gmm_train_step_hier: both levels in one step) at 32 lanes x 10 sessions x 20 steps, emb 128, "
                                 "2 x 128 across-session cells, tcn_gmm = TemporalConvNet [128, 128] K = 5, gru_gmm = 2 GRU layers of 128",
                         "rows": tB * tT, "ms_per_step": round(train_ms, 3), "rows_per_s": round(tB * tT / train_ms * 1e3),
                         "ms_per_step_gru_gmm": round(train_ms_gru, 3), "ms_per_step_gru_gmm_hipgraph": train_ms_gru_graph,
                         "ms_per_step_hier": hier_ms,
                         "loss_first_last": [round(float(tl[0, 0] / tl[0, 1]), 5), round(float(tl[-1, 0] / tl[-1, 1]), 5)]},
            "what": "hiertcn_gmm_loss (gmm_prob) on random hidden rows.  ms: the Gram form (k_gmm_quad: h^T G_m h on the bf16 MFMA with G_m in three "
                    "bf16 parts, + one row-wise kernel for prior / linear terms / target mean + finish); ms_table_pass: M passes of "
                    "k_score_pipe<256,21,...> (sum-of-squares epilogue) over the [D, C] mean tables + prior + finish; ms_learned_scale: args.sigma None, M passes of "
                    "k_score_pipe<256,22,...> over the interleaved means / log-scales tables (twice the rows)",
            "rows": rows, "output_dim": n_out, "num_mixture": n_mix, "ms": round(ms["gram"], 3), "ms_table_pass": round(ms["table_pass"], 3),
            "rows_per_s": round(rows / ms["gram"] * 1e3), "max_abs_loss_difference": diff,
            "ms_learned_scale": round(ms["learned_scale"], 3),
            "learned_scale_tflops": round(2.0 * rows * dim * ((n_out + 31) // 32 * 64) * n_mix / ms["learned_scale"] / 1e9, 1),
            "table_pass_tflops": round(flop / ms["table_pass"] / 1e9, 1), "table_pass_frac_of_peak": round(flop / ms["table_pass"] / 1e9 / MFMA_PEAK_TFLOPS, 4)}


def dvfs_probe(eng, rows, n_items, dim, seconds=1.5):
    """The scoring kernel alone, back to back for `seconds`, on random-normal hidden rows and on all-zero hidden rows (the MFMA B
    operand: same instruction stream, same LDS / L2 traffic, every product zero), with the shader clock read from rocm-smi (sysfs)
    meanwhile: how much of the distance to the data-sheet peak is the clock the chip holds under this load, not the kernel."""
    import re
    import subprocess
    import threading

    import torch

    def sampler(stop, out):
        while not stop.is_set():
            try:
                r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
                ck = re.search(r"sclk clock level:?\s*\d*:?\s*\(?([0-9.]+)\s*Mhz", r, re.I)
                pw = re.search(r"Power \(W\):\s*([0-9.]+)", r)
                out.append((float(ck.group(1)) if ck else None, float(pw.group(1)) if pw else None))
            except Exception:  # noqa: BLE001  (no rocm-smi: the timings below still stand)
                return
            time.sleep(0.15)

    y = torch.randint(1, n_items, (rows,), device="cuda", dtype=torch.int32)
    res = {"what": "scoring kernel alone on random-normal vs all-zero hidden rows (same binary, same instruction stream); sclk / socket "
                   "power = median of rocm-smi samples during the loop"}
    for name, scale in (("random", 0.5), ("zero_hidden_rows", 0.0)):
        hid = (torch.randn(rows, dim, device="cuda") * scale).to(eng.act_dtype)
        eng.score_fused(hid, y)
        torch.cuda.synchronize()
        stop, smp = threading.Event(), []
        th = threading.Thread(target=sampler, args=(stop, smp))
        th.start()
        eng.timing_enable(True)
        eng.timing_read(reset=True)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            eng.score_fused(hid, y)
            torch.cuda.synchronize()
        ms, n = eng.timing_read(reset=True)
        eng.timing_enable(False)
        stop.set()
        th.join()
        ck = sorted(v[0] for v in smp if v[0] is not None)
        pw = sorted(v[1] for v in smp if v[1] is not None)
        tf = 2.0 * rows * dim * n_items / (ms * 1e-3) / 1e12
        res[name] = {"launch_ms": round(ms, 3), "launches": int(n), "tflops": round(tf, 1), "frac_of_peak": round(tf / MFMA_PEAK_TFLOPS, 4),
                     "sclk_mhz": ck[len(ck) // 2] if ck else None, "power_w": pw[len(pw) // 2] if pw else None}
        del hid
    return res


def reference_batch_leg(n_batches=100):
    """-> the `reference_batch` object: the reference's OWN operating point (args.py:19,40,51-53,126-129: 32 lanes x 10 slots x <= 20 steps per batch, two
    layers; BASELINE.json configs[1]'s XING catalogue of 20,778 items): ms per evaluation batch for HierTCN at d = 256 (configs[1]) and for HierGRU -- the
    reference's default low level -- at d = 128 (its default width) and 256, and ms per training step (forward + backward of every variable + Adam) of
    each.  Ragged batches, ids resident."""
    import torch

    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.engine import Engine
    from hiertcn_amd.params import init_params

    N, B, S, L = 20_778, 32, 10, 20
    rng = np.random.default_rng(0)
    ids = rng.integers(1, N, (8, B, S, L)) * (np.arange(L)[None, None, None, :] < rng.integers(1, L + 1, (8, B, S))[..., None])
    x = np.zeros_like(ids)
    x[..., 1:] = ids[..., :-1]
    mask = (rng.random((8, B, S)) > 0.15).astype(np.float32)
    xs, ys, ms = (torch.as_tensor(v).cuda().contiguous() for v in (x.astype(np.int32), ids.astype(np.int32), mask))
    out = {"what": "32 lanes x 10 slots x <= 20 steps per batch (the reference's batch_size / max_session_num / max_activity_len defaults), XING catalogue N = 20,778, "
                   "two layers, ragged sessions, ids resident; ms per batch = mean over %d back-to-back batches" % n_batches}
    # (hiertcn_d256_fp32: the same model in the fp32 TRAINING mode -- fp32 forward on the masters, 532 MB of materialised logits per step, every product of
    #  the backward an exact-fp32 GEMM: the configuration in which a TF training trajectory can be followed to fp32 grade; evaluation = the fp32 mode)
    for name, low, d, train, dtype in (("hiertcn_d256", "tcn", 256, True, None), ("hiergru_d128", "gru", 128, True, None), ("hiergru_d256", "gru", 256, True, None),
                                       ("hiertcn_d256_fp32", "tcn", 256, True, "fp32"), ("hiergru_d128_fp32", "gru", 128, True, "fp32")):
        args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, batch_size=B, model_low_type=low)
        eng = Engine(args, hier=True, train=train, dtype=dtype)
        eng.set_weights(init_params(args, seed=2))
        st = torch.zeros((B, eng.HT), dtype=torch.float32, device="cuda")
        st2 = torch.empty_like(st)
        m8 = torch.zeros(8, dtype=torch.float64, device="cuda")
        for i in range(10):
            eng.forward_eval(xs[i % 8], ys[i % 8], ms[i % 8], st, state_out=st2, metrics8=m8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_batches):
            eng.forward_eval(xs[i % 8], ys[i % 8], ms[i % 8], st, state_out=st2, metrics8=m8)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n_batches
        rec = {"eval_ms_per_batch": round(dt * 1e3, 4), "eval_sessions_per_s": round(B * S / dt)}
        if train:
            for i in range(5):
                eng.train_forward_backward(xs[i % 8], ys[i % 8], ms[i % 8], st)
                eng.adam_step(1e-3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(30):
                eng.train_forward_backward(xs[i % 8], ys[i % 8], ms[i % 8], st)
                eng.adam_step(1e-3)
            torch.cuda.synchronize()
            rec["train_ms_per_step"] = round((time.perf_counter() - t0) / 30 * 1e3, 4)
        rec["persistent_timed_out"] = bool(eng.persistent_timed_out())
        out[name] = rec
        eng.close()
    return out


def run_loop_leg(users=20000, lanes=32):
    """-> the `run_loop` object: the reference's evaluation LOOP end to end (run_hier_xing.py:112-207 -- loader, one step per batch, meters) over a
    synthetic XING-format file at the reference's batch, HierTCN d = 256: sessions/s of the whole loop, pipelined (the default: the device is left
    running) and blocking (one sess.run-shaped call per batch)."""
    import subprocess
    import tempfile

    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.data_loader import Dataloader_hier_model_xing
    from hiertcn_amd.data_prepare import load_xing
    from hiertcn_amd.model import HierTCNModel
    from hiertcn_amd.run_hier_xing import evaluate_hier

    root = os.path.dirname(os.path.abspath(__file__))
    tmp = tempfile.mkdtemp()
    subprocess.check_call([sys.executable, os.path.join(root, "scripts", "make_synth_xing.py"), str(users), tmp], stdout=subprocess.DEVNULL)
    data = load_xing(os.path.join(tmp, "dataset", "interactions.csv"), max_activity=500)
    n_items = int(data[1][:, 1].max()) + 1
    d = 256
    args = hier_tcn_args(item_num=n_items, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, batch_size=lanes, max_session_num=10,
                         max_activity_len=20)
    args.log_path = None
    model = HierTCNModel(args)
    n_sessions = int((np.diff(np.flatnonzero(np.r_[True, (data[1][1:, 4] != data[1][:-1, 4]) | (data[1][1:, 0] != data[1][:-1, 0]), True]))).shape[0])
    out = {"what": "evaluate_hier over a synthetic XING-format file (%d users, %d sessions, %d items), %d lanes x 10 slots x <= 20 steps per batch, HierTCN d = 256, "
                   "packer on the loop's thread; sessions/s of the whole loop" % (users, n_sessions, n_items, lanes)}
    for form in ("pipelined", "pipelined", "blocking"):   # (first pass: warm-up)
        if form == "blocking":
            os.environ["HIERTCN_EVAL_SERIAL"] = "1"
        try:
            loader = Dataloader_hier_model_xing(args, "all", data=data)
            state = np.zeros((lanes, 2 * d), np.float32)
            t0 = time.perf_counter()
            _s, _v, t_load, t_run, _st = evaluate_hier(args, loader, state, model, prefix="test")
            dt = time.perf_counter() - t0
        finally:
            os.environ.pop("HIERTCN_EVAL_SERIAL", None)
        out[form] = {"seconds": round(dt, 4), "loader_seconds": round(t_load, 4), "sessions_per_s": round(n_sessions / dt)}
    model.engine.close()
    from hiertcn_amd.data_loader import active_packer
    out["packer"] = active_packer()   # 'c' (csrc/packer.c) or 'python' (the NumPy form, ~10 x slower at 32 lanes): the figure depends on it
    return out


def _sha16(*names):
    import hashlib
    h = hashlib.sha256()
    for n in names:
        with open(os.path.join(ROOT, "hiertcn_amd", "csrc", n), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def traffic_from_profile(section, rows_per_launch, sources, ragged):
    """-> (bytes per launch | None, traffic_code, source).  profiles/score_traffic.json is written by scripts/profile_round.sh (rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of this same command; bench.py cannot read PMC counters itself) and names the code it was measured on: the
    sha-256 of the kernel's source files and the git head.  The figure is used only when the hash of today's sources equals the recorded one
    and the launch had the same rows; otherwise `traffic` is null and `traffic_code` says why."""
    try:
        with open(os.path.join(ROOT, "profiles", "score_traffic.json")) as f:
            tj = json.load(f)
        sec = tj[section]
    except (OSError, KeyError, ValueError):
        return None, "no profile of this kernel in profiles/score_traffic.json", None
    try:
        now = _sha16(*sources)
    except OSError:
        now = None
    rec = sec.get("source_sha16")
    code = {"profiled_git_head": tj.get("git_head"), "profiled_sources_sha16": rec, "current_sources_sha16": now, "sources": list(sources)}
    if rec is None or rec != now:
        code["stale"] = "the kernel's sources changed since the counters were read: traffic withheld"
        return None, code, sec.get("source")
    if int(sec["rows_per_launch"]) != int(rows_per_launch) or ragged:
        code["stale"] = "the profiled launch had %d rows, this one %d%s" % (int(sec["rows_per_launch"]), int(rows_per_launch), " (ragged)" if ragged else "")
        return None, code, sec.get("source")
    return sec["fetch_bytes_per_launch"] + sec["write_bytes_per_launch"], code, sec.get("source")


def exact_leg(a, args, params, rank, world, draw, shard_first, shard, dist, max_over_ranks):
    """-> the `exact` object: the SAME evaluation step in the parity-grade modes (tests/test_gpu_exact_mode.py: logits < 1e-3 of the exact
    oracle at |z| = 10 and 30, ranks equal on > 99 % of rows, top-10 equal on every clear row) -- 'split16' (HIERTCN_PRECISION_SPLIT16: fp32
    low level, fused scoring on hi + lo half operands, three MFMAs per product) timed like the headline (barrier, max over ranks, HIP events
    around the scoring launches), and the 'fp32' mode (fp32 MFMA, materialised logits) beside it on a smaller batch."""
    import torch

    from hiertcn_amd.engine import Engine

    Bx = a.exact_users_per_step
    eng = Engine(args, hier=True, dtype="split16")
    eng.set_precision("fp32")          # both parity modes on one handle: their weight copies are allocated before the weights are set
    eng.set_precision("split16")
    eng.set_weights(params)
    n_total = a.exact_warmup + a.exact_steps
    batches = []
    for i in range(n_total):
        first = shard_first + (i * Bx) % max(shard - Bx, 1)
        x, y, mask = make_users(draw, first, Bx, a.ragged)
        batches.append((torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(mask).cuda()))
    state0 = torch.zeros((Bx, eng.HT), dtype=torch.float32, device="cuda")
    m8 = torch.zeros(8, dtype=torch.float64, device="cuda")
    for i in range(a.exact_warmup):
        eng.forward_eval(*batches[i], state0, metrics8=m8)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    eng.timing_enable(True)
    eng.timing_read(reset=True)
    t0 = time.perf_counter()
    for i in range(a.exact_warmup, n_total):
        eng.forward_eval(*batches[i], state0, metrics8=m8)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = max_over_ranks(time.perf_counter() - t0, "cuda")
    score_ms, launches = eng.timing_read(reset=True)
    eng.timing_enable(False)
    rows = Bx * S_SESS * L_STEPS
    scored = float(m8.cpu()[7])
    flop = 2.0 * scored * DIM * N_ITEMS                      # algorithmic; the kernel executes three MFMAs per product
    alg = flop / (score_ms * 1e-3) / 1e12 if score_ms > 0 else 0.0
    out = {"mode": "split16",
           "what": "the evaluation step of the headline in HIERTCN_PRECISION_SPLIT16 (csrc/score_x3.hip): fp32 low level + the fused scoring kernel on hi + lo "
                   "IEEE-half operands, h.w = h_hi w_hi + h_lo w_hi + h_hi w_lo in one fp32 accumulator; the mode tests/test_gpu_exact_mode.py holds to "
                   "logits < 1e-3 / ranks / top-10 of the exact oracle at |z| = 10 and 30",
           "value": round(Bx * S_SESS * world * a.exact_steps / dt, 2), "unit": "sessions/s", "n_gpus": world,
           "users_per_step_per_gpu": Bx, "steps": a.exact_steps, "warmup": a.exact_warmup, "ms_per_step": round(dt / a.exact_steps * 1e3, 3),
           "kernel": "k_score_x3<256,0,0> (+ its fix-up launch)", "launch_ms": round(score_ms, 4), "launches_timed": int(launches),
           "scored_rows_per_launch": scored, "rows_per_launch": rows,
           "achieved_algorithmic_tflops": round(alg, 2), "achieved_executed_tflops": round(3 * alg, 2),
           "frac_of_peak_over_3": round(alg / (MFMA_PEAK_TFLOPS / 3), 4),
           "peak_over_3": round(MFMA_PEAK_TFLOPS / 3, 1)}
    tr_bytes, tr_code, tr_src = traffic_from_profile("exact", rows, ("score_x3.hip", "score_x3.h"), a.ragged)
    out["roofline"] = {"bound": "mfma", "kernel": out["kernel"], "launch_ms": round(score_ms, 4), "launches_timed": int(launches),
                       "flops_per_launch_algorithmic": flop, "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "achieved_algorithmic": round(alg, 2), "frac_algorithmic": round(alg / MFMA_PEAK_TFLOPS, 4),
                       "achieved_executed": round(3 * alg, 2), "frac_executed": round(3 * alg / MFMA_PEAK_TFLOPS, 4),
                       "traffic": tr_bytes, "traffic_code": tr_code, "traffic_source": tr_src,
                       "algorithmic_bytes_per_launch": rows * (4 * DIM + 4 + 8) + 4 * DIM * N_ITEMS + 4 * N_ITEMS}
    # the fp32 mode (exact.hip) on the same handle, a smaller batch: ~20x slower, bounded by the 157 TFLOP/s fp32 MFMA (single-GPU runs only:
    # every rank of an N-GPU run does the same work in the same order)
    if rank == 0 and world == 1:
        Bf = min(a.exact_fp32_users, Bx)
        eng.set_precision("fp32")
        xb, yb, mb = (t[:Bf].contiguous() for t in batches[0])
        st = state0[:Bf].contiguous()
        eng.forward_eval(xb, yb, mb, st, metrics8=m8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            eng.forward_eval(xb, yb, mb, st, metrics8=m8)
        torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / 2
        out["fp32_mode"] = {"value": round(Bf * S_SESS / tf, 2), "unit": "sessions/s", "n_gpus": 1, "users_per_step": Bf, "ms_per_step": round(tf * 1e3, 3),
                            "what": "HIERTCN_PRECISION_FP32 (csrc/exact.hip): fp32 MFMA, logits materialised per row chunk"}
    eng.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--users-per-step", type=int, default=16384, help="users per step PER GPU (16384 users = 19,200 scoring workgroups = exactly 75 rounds of 256 CUs: no "
                    "partly filled last round; 4096 users = 18.75 rounds measured 2.7 %% fewer sessions/s on the same box)")
    ap.add_argument("--ragged", action="store_true", help="session lengths ~ U{1..30} instead of full")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the evaluation steps alternate between (one handle + workspace each).  2: the pool / "
                    "GRU scan / low-level kernels of step i + 1 fill the tail round of step i's scoring launch, +0.7 %% sessions/s (scripts/two_stream_probe.py) "
                    "-- but the HIP events around a scoring launch then include its wait for the other stream's launch, so roofline.launch_ms is no longer the "
                    "kernel's own duration: the default stays 1 so that the roofline figures mean what they say")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gmm", action="store_true", help="skip the mixture-head figure (the `gmm` object of the JSON line)")
    ap.add_argument("--no-train", action="store_true", help="skip the data-parallel training leg (the `train` object of the JSON line)")
    ap.add_argument("--no-clock-probe", action="store_true", help="skip roofline.clock_probe (extra launches of the scoring kernel on random / zero operands; profiling runs skip it so that the kernel's average duration in the trace is the timed region's)")
    ap.add_argument("--train-users-per-step", type=int, default=4096, help="users per training step PER GPU (4096 users = 9600 workgroups of the scoring "
                    "backward = 37.5 rounds of 256 CUs; at 1024 users the 9.4 rounds end in a round that is 3/8 full: 58.7 k vs 61.4 k sessions/s on one box)")
    ap.add_argument("--train-steps", type=int, default=4)
    ap.add_argument("--train-warmup", type=int, default=2)
    ap.add_argument("--cpu-budget", type=float, default=40.0, help="seconds of CPU work for the cpu_baseline leg")
    ap.add_argument("--no-exact", action="store_true", help="skip the parity-grade modes (the `exact` object of the JSON line)")
    ap.add_argument("--no-reference-batch", action="store_true", help="skip the reference's own batch shape (the `reference_batch` object of the JSON line)")
    ap.add_argument("--exact-users-per-step", type=int, default=16384, help="users per step PER GPU of the split16 leg, measured like the headline: 16384 users = 38,400 "
                    "workgroups of 256 rows = exactly 150 rounds of 256 CUs")
    ap.add_argument("--exact-steps", type=int, default=10)
    ap.add_argument("--exact-warmup", type=int, default=2)
    ap.add_argument("--rccl-one-rank", action="store_true", help="N = 1 only: create a one-rank RCCL process group and run every collective of the step through it "
                    "(multi_gpu.rccl_ranks: 1): librccl loaded, device tensors accepted, stream ordering exercised -- everything short of a second GPU")
    ap.add_argument("--exact-fp32-users", type=int, default=256)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist

    from hiertcn_amd.distributed import all_reduce_metrics, init_from_env, max_over_ranks, user_shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    # HIERTCN_BENCH_BACKEND=gloo: smoke-test of the N > 1 code path on a box with fewer GPUs than ranks (ranks share devices,
    # the 8 metric sums go through the host); the measured configuration is always one rank per GPU over RCCL
    backend = os.environ.get("HIERTCN_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    rank, world, local_rank = init_from_env(backend, torch.device("cuda", dev_index), force_group=bool(a.rccl_one_rank and world == 1))  # "nccl" = RCCL on ROCm

    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.engine import Engine, metrics_dict
    from hiertcn_amd.params import init_params

    args = hier_tcn_args(item_num=N_ITEMS, emb_dim=DIM, hidden_dim=DIM, num_layer=2, tcn_channel=[DIM, DIM], kernel_size=5,
                         max_activity_len=L_STEPS, max_session_num=S_SESS, batch_size=a.users_per_step)
    t_setup = time.perf_counter()
    params = init_params(args, seed=2)  # glorot-uniform / GRU gate bias 1.0, default_rng(2)
    n_streams = max(1, int(a.streams))
    engs = [Engine(args, hier=True) for _ in range(n_streams)]
    for e in engs:
        e.set_weights(params)
    eng = engs[0]
    streams = [torch.cuda.Stream() for _ in range(n_streams)] if n_streams > 1 else [torch.cuda.current_stream()]

    B = a.users_per_step
    n_steps_total = a.warmup + a.steps
    shard_first, shard = user_shard(U_TOTAL, rank, world)  # this rank's users of the 200k
    draw = zipf_sampler(N_ITEMS)
    batches = []
    for i in range(n_steps_total):
        first = shard_first + (i * B) % max(shard - B, 1)
        x, y, mask = make_users(draw, first, B, a.ragged)
        batches.append((torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(mask).cuda()))
    state0 = torch.zeros((B, eng.HT), dtype=torch.float32, device="cuda")  # run_hier_xing.py:241
    state_outs = [torch.empty_like(state0) for _ in range(n_streams)]
    m8 = [torch.zeros(8, dtype=torch.float64, device="cuda") for _ in range(n_steps_total)]
    for e in engs:
        e.workspace(B, S_SESS, L_STEPS)
    torch.cuda.synchronize()
    t_setup = time.perf_counter() - t_setup

    def step(i):
        xb, yb, mb = batches[i]
        k = i % n_streams
        with torch.cuda.stream(streams[k]):
            engs[k].forward_eval(xb, yb, mb, state0, state_out=state_outs[k], metrics8=m8[i])
            all_reduce_metrics(m8[i])  # RCCL: the only collective of the evaluation path (8 doubles); no-op at N=1

    for i in range(a.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    for e in engs:
        e.timing_enable(True)
        e.timing_read(reset=True)
    t0 = time.perf_counter()
    for i in range(a.warmup, n_steps_total):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = max_over_ranks(dt, "cuda")
    tr = [e.timing_read(reset=True) for e in engs]      # (average ms, launches) per handle, HIP events on the handle's own stream
    launches = sum(n for _, n in tr)
    score_ms = sum(ms * n for ms, n in tr) / max(launches, 1)
    for e in engs:
        e.timing_enable(False)
    # health word of the persistent launches (the across-session scan of small batches; include/hiertcn.h): a bounded wait that ran out would
    # have produced wrong states in the timed region -- read once, behind it
    persistent_timed_out = any(e.persistent_timed_out() for e in engs)
    for e in engs[1:]:
        e.close()

    # what the driver needs to confirm an N-GPU run WAS one: the collective backend, how many ranks it joined, and how many DISTINCT
    # physical devices they ran on (uuid where the runtime reports one, PCI address otherwise), plus every rank's own scoring-launch
    # average so that a straggling GPU shows
    prop = torch.cuda.get_device_properties(dev_index)
    ident = str(getattr(prop, "uuid", "")) or ""
    pci = "%04x:%02x:%02x" % (getattr(prop, "pci_domain_id", 0), getattr(prop, "pci_bus_id", 0), getattr(prop, "pci_device_id", 0))
    mine = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "uuid": ident, "pci": pci, "name": prop.name,
            "score_launch_ms": round(score_ms, 4), "step_ms": round(dt / max(a.steps, 1) * 1e3, 3)}
    per_rank = [mine]
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    multi = {"backend": ("rccl" if backend == "nccl" else backend) if world > 1 else None,
             "rccl_ranks": (dist.get_world_size() if dist.is_initialized() and dist.get_backend() == "nccl" else (0 if world > 1 else None)),
             "ranks": world,
             "distinct_devices": len({(r["uuid"] or r["pci"]) for r in per_rank}),
             "per_rank": per_rank}

    sessions_per_step = B * S_SESS * world
    value = sessions_per_step * a.steps / dt
    rows_per_launch = B * S_SESS * L_STEPS
    # rows that carry a target (= all rows unless --ragged): padded steps are not scored at all (row compaction)
    scored_rows = float(torch.stack(m8[a.warmup:]).cpu()[:, 7].mean()) / world if a.steps > 0 else float(rows_per_launch)
    flops_per_launch = 2.0 * scored_rows * DIM * N_ITEMS  # algorithmic: scored rows x 2 C N
    achieved = flops_per_launch / (score_ms * 1e-3) / 1e12 if score_ms > 0 else 0.0
    last = metrics_dict(m8[-1])

    # HBM/fabric bytes per scoring launch come from a separate rocprofv3 --pmc run of this same command
    # (bench.py cannot read PMC counters itself); used only when that run had the same rows per launch.
    traffic, traffic_code, traffic_src = traffic_from_profile("headline", rows_per_launch, ("score.hip", "score_geom.h", "score_lists.h"), a.ragged)

    # a plain library GEMM on the same box, same data kind (random bf16), as a second yardstick beside the datasheet peak
    # ("the ratio against a measured hipBLASLt GEMM peak on the same box"); measurement only, not product path
    gemm_ref = None
    if rank == 0:
        try:
            n = 8192
            ga = torch.randn((n, n), device="cuda", dtype=torch.bfloat16)
            gb = torch.randn((n, n), device="cuda", dtype=torch.bfloat16)
            for _ in range(3):
                gc = ga @ gb
            torch.cuda.synchronize()
            tg = time.perf_counter()
            for _ in range(10):
                gc = ga @ gb
            torch.cuda.synchronize()
            tg = (time.perf_counter() - tg) / 10
            gemm_ref = {"what": "torch.mm bf16 8192^3, random normal operands (hipBLASLt/rocBLAS via PyTorch)",
                        "tflops": round(2.0 * n ** 3 / tg / 1e12, 1)}
            del ga, gb, gc
        except Exception as e:  # noqa: BLE001  (a yardstick, never fatal)
            gemm_ref = {"error": str(e)[:200]}

    dvfs = None
    if rank == 0 and world == 1 and not a.ragged and not a.no_clock_probe:
        try:
            dvfs = dvfs_probe(eng, rows_per_launch, N_ITEMS, DIM)
        except Exception as e:  # noqa: BLE001  (a diagnostic, never fatal)
            dvfs = {"error": repr(e)[:200]}

    gmm = train = exact = refb = rloop = cpu = None   # secondary legs fill these in; make_out() reads whatever is there

    def make_out():
        out = {
            "metric": "sessions/sec (XING-shape synth) at 1/2/4/8 MI355X; recall@10 parity",
            "value": round(value, 2),
            "unit": "sessions/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic",
            "config": {
                "workload": "cfg3-synth: 200k users x 20 sessions x 30 steps, 300k-item catalogue, d=256 (E=C=H), "
                            "2 TCN blocks K=5, 2-layer GRU; eval step = pool+GRU scan+TCN+fused scoring(loss+rank)+lane metrics",
                "users_per_step_per_gpu": B,
                "streams": n_streams,
                "sessions_per_step": sessions_per_step,
                "ragged": bool(a.ragged),
                "parallelism": f"user-batch dp{world}",
            },
            "roofline": {
                "bound": "mfma",
                "kernel": "k_score_pipe<256,14,2,3,pad> (fused scoring GEMM + loss/rank epilogue; + its fix-up launch)",
                "achieved": round(achieved, 2),
                "peak": MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                "library_gemm_reference": gemm_ref,
                "frac_of_library_gemm": (round(achieved / gemm_ref["tflops"], 4) if gemm_ref and gemm_ref.get("tflops") else None),
                "clock_probe": dvfs,
                # informational: the same achieved rate against the peak at the shader clock rocm-smi reported under this very load
                # (2.5 PFLOP/s assumes 2.4 GHz); `frac` above stays the data-sheet fraction
                "frac_at_held_clock": (round(achieved / (MFMA_PEAK_TFLOPS * dvfs["random"]["sclk_mhz"] / 2400.0), 4)
                                       if dvfs and dvfs.get("random") and dvfs["random"].get("sclk_mhz") else None),
                "traffic": traffic,
                "traffic_unit": "bytes per launch through the L2's memory-side port (Infinity-Cache hits included)",
                "traffic_code": traffic_code,
                "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": rows_per_launch * (2 * DIM + 4 + 8) + 2 * DIM * N_ITEMS + 4 * N_ITEMS,
                "launch_ms": round(score_ms, 4),
                "launch_ms_per_rank_min_max": [min(r["score_launch_ms"] for r in per_rank), max(r["score_launch_ms"] for r in per_rank)],
                "launches_timed": int(launches),
                "flops_per_launch": flops_per_launch,
                "scored_rows_per_launch": scored_rows,
            },
            "cpu_baseline": cpu,
            "exact": exact,
            "reference_batch": refb,
            "run_loop": rloop,
            "multi_gpu": multi,
            "gmm": gmm,
            "train": train,
            "persistent_timed_out": bool(persistent_timed_out),
            "batch_metrics_last_step": {k: round(v, 6) for k, v in last.items()},
            "setup_s": round(t_setup, 1),
        }
        return out

    # N > 1: the secondary legs below run collectives that no hardware run has exercised at more than one rank.  A rank stuck in one would take the
    # headline line with it, so a watchdog on EVERY rank prints the line as it stands (rank 0) and ends the process if they have not finished in
    # time -- with a NON-ZERO status (3): a run whose secondary legs never finished is not a clean run, whatever the line says (it says
    # `secondary_legs.error`).  The other ranks hold their exit back a little: under torch.distributed.run the first non-zero exit makes the
    # launcher end every rank, and rank 0 must have flushed its line by then.  (No re-exec, no retry from here: a process that has touched
    # the GPU is never replaced; a caller that wants another attempt starts a fresh one.)
    watchdog = None
    if world > 1:
        import threading
        limit = float(os.environ.get("HIERTCN_BENCH_LEG_LIMIT_S", "420"))

        def _expired():
            if rank == 0:
                o = make_out()
                o["secondary_legs"] = {"error": f"not finished after {limit:.0f} s: line printed by the watchdog, process ended with status 3"}
                print(json.dumps(o), flush=True)
                sys.stderr.write(f"bench.py: the secondary legs did not finish within {limit:.0f} s; exiting with status 3 after the partial line\n")
                sys.stderr.flush()
            else:
                time.sleep(10.0)
            os._exit(3)
        watchdog = threading.Timer(limit, _expired)
        watchdog.daemon = True
        watchdog.start()

    # ---- mixture-output head (BASELINE.json configs[4]: model_tcn_gmm / model_gru_gmm): hiertcn_gmm_loss = M passes of the scoring kernel's main
    # loop with the sum-of-squares epilogue over the [D, C] mean tables, at configs[4]'s shape (614,400 rows x 20,778 outputs x 5 mixtures)
    gmm = None
    if rank == 0 and not a.no_gmm:
        try:
            gmm = gmm_leg()
        except Exception as e:  # noqa: BLE001  (secondary figure, never fatal)
            gmm = {"error": repr(e)[:300]}

    # ---- training leg (north star: "RCCL all-reduce over xGMI on the shared item-embedding gradients"): the same cfg-3 model,
    # one optimizer step = forward + loss + backward + gradient collectives + TF-form Adam (model.py:135-141,
    # run_hier_xing.py:275-301), users sharded over the ranks, timed after the evaluation region under the same launch
    train = None
    if not a.no_train:
        if world > 1:
            train = train_leg(a, args, params, rank, world, draw, shard_first, shard, dist, max_over_ranks)   # (agrees on failure across ranks itself)
        else:
            try:
                train = train_leg(a, args, params, rank, world, draw, shard_first, shard, dist, max_over_ranks)
            except Exception as e:  # noqa: BLE001  (the headline line must survive a failure of the secondary leg; the error is reported)
                import traceback
                train = {"error": repr(e)[:300], "where": traceback.format_exc()[-600:]}

    exact = None
    if not a.no_exact:
        if world > 1:
            exact = exact_leg(a, args, params, rank, world, draw, shard_first, shard, dist, max_over_ranks)
        else:
            try:
                exact = exact_leg(a, args, params, rank, world, draw, shard_first, shard, dist, max_over_ranks)
            except Exception as e:  # noqa: BLE001  (the headline line must survive a failure of a secondary leg; the error is reported)
                import traceback
                exact = {"error": repr(e)[:300], "where": traceback.format_exc()[-600:]}

    refb = None
    if rank == 0 and world == 1 and not a.no_reference_batch:
        try:
            refb = reference_batch_leg()
        except Exception as e:  # noqa: BLE001  (secondary figure, never fatal)
            refb = {"error": repr(e)[:300]}

    rloop = None
    if rank == 0 and world == 1 and not a.no_reference_batch:
        try:
            rloop = run_loop_leg()
        except Exception as e:  # noqa: BLE001  (secondary figure, never fatal)
            rloop = {"error": repr(e)[:300]}

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        # the checker, timed on this host's cores on a bounded sample of the SAME workload
        from oracle.cpu_baseline import time_cpu_baseline
        xb, yb, mb = batches[a.warmup]
        n_cpu = min(B, 128)
        r = time_cpu_baseline(args, params, xb[:n_cpu].cpu().numpy(), yb[:n_cpu].cpu().numpy(), mb[:n_cpu].cpu().numpy(),
                              budget_s=a.cpu_budget)
        plan = 1024 * S_SESS          # BASELINE.md 3 planned 1,024 users x 20 sessions; the default budget times a fraction of that
        cpu = dict(value=round(r["value"], 3), unit="sessions/s", cores=r["cores"], kind="port",
                   sample=f"{r['sessions']} of the planned {plan:,} sessions (BASELINE.md 3) in the best variant: " + r["sample"],
                   sample_sessions=int(r["sessions"]), planned_sessions=plan,
                   seconds=round(r["seconds"], 2), variants=r["variants"])

    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        out = make_out()
        try:   # RCCL prints its version banner through C stdio, which flushes at exit: flush it NOW so that the JSON line is the last line of stdout
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:  # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
