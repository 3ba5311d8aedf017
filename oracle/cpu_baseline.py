"""Reference-literal CPU path, PyTorch-CPU fp32  --  TEST/BENCH INFRASTRUCTURE ONLY.

PARITY UNPINNED: same caveat as oracle/hiertcn_oracle.py (TensorFlow is unavailable; no golden
vectors exist for the model path).  tests/test_cpu_baseline.py pins this file to the NumPy oracle.

This is the "reference CPU path" timed beside the GPU numbers: the same op
sequence as the TF graph -- per-slot loop over sessions (model_hier.py:39-93), embedding
(:50), state broadcast + projection (:54-55, model_tcn.py:35), causal dilated conv blocks
(customized_tcn_cell.py:46-49,109-127), FULL [B,L,N] logits (model_tcn.py:41), mask
(model.py:105), cross entropy (loss.py:21), greater+sum ranks and recall@1/5/10, MRR
(loss.py:179-219), mean-pool + GRU + mask (model_hier.py:83-93).  The embedding is a gather;
the TF graph's dense one-hot matmul (model.py:59 + model_hier.py:50) would roughly double the
FLOPs and is not emulated.  The [B, sum L, N] concat (model_hier.py:79) is not materialised
(23 GB at the north-star shape): loss / rank partial sums are accumulated per slot, which is
arithmetically the same reduction.

Only bench.py's cpu_baseline leg and tests/ may import this module.
"""
from __future__ import annotations

import time

import numpy as np
import torch
import torch.nn.functional as F


class CpuHierTCN:
    def __init__(self, args, params, threads=None):
        if threads:
            torch.set_num_threads(int(threads))
        self.threads = torch.get_num_threads()
        self.args = args
        t = lambda k: torch.from_numpy(np.ascontiguousarray(params[k], dtype=np.float32))  # noqa: E731
        self.E = t("hier/emb/kernel").clone()
        self.E[0] = 0  # one_hot(id)*sign(id): the null id embeds to zero (model.py:59)
        self.b_emb = t("hier/emb/bias")
        self.Wp = t("hier/tcn/emb/kernel")
        self.blocks = []
        for i in range(len(args.tcn_channel)):
            pre = f"hier/tcn/temporal_conv_net/tblock_{i}"
            W = t(f"{pre}/conv1/kernel").permute(2, 1, 0).contiguous()  # [Cout, Cin, K]
            self.blocks.append((W, t(f"{pre}/conv1/bias"), 2 ** i))
        self.Wo = t("hier/tcn/dense/kernel")
        self.bo = t("hier/tcn/dense/bias")
        self.gru = []
        for l in range(args.num_layer):
            pre = f"hier/multi_rnn_cell/cell_{l}/gru_cell"
            self.gru.append((t(f"{pre}/gates/kernel"), t(f"{pre}/gates/bias"), t(f"{pre}/candidate/kernel"),
                             t(f"{pre}/candidate/bias")))

    @torch.no_grad()
    def eval_batch(self, x, y, mask, state):
        """x, y [B,S,L] int64; mask [B,S]; state [B,H*nl] -> (metrics dict, new state).  One call
        = one sess.run of the evaluation fetch list (run_hier_xing.py:145-149)."""
        B, S, L = x.shape
        H, N = self.args.hidden_dim, self.Wo.shape[1]
        K = self.args.kernel_size
        ce_sum = torch.zeros(B)
        sums = {k: torch.zeros(B) for k in ("r1", "r5", "r10", "rr", "rf")}
        act = torch.zeros(B)
        for i in range(S):
            xi, yi = x[:, i], y[:, i]
            e = F.embedding(xi, self.E)                                      # model_hier.py:50
            xs = torch.cat([e, state[:, None, :].expand(-1, L, -1)], -1)    # :54-55
            a = (xs @ self.Wp).transpose(1, 2)                               # model_tcn.py:35
            for W, b, d in self.blocks:                                      # customized_tcn_cell.py:109-127
                c = torch.relu(F.conv1d(F.pad(a, ((K - 1) * d, 0)), W, b, dilation=d))
                a = torch.relu(c + a)
            logits = a.transpose(1, 2) @ self.Wo + self.bo                   # model_tcn.py:41   [B,L,N]
            mk = (yi > 0).float()                                            # model.py:62
            logits = logits * mk[..., None]                                  # model.py:105
            tgt = torch.gather(logits, 2, yi[..., None])[..., 0] * mk        # loss.py:179 sum(score*y)
            ce = (torch.logsumexp(logits, -1) - tgt) * mk                    # loss.py:21
            rk = (logits > tgt[..., None]).float().sum(-1)                   # loss.py:179
            ce_sum += ce.sum(1)
            act += mk.sum(1)
            sums["r1"] += ((rk <= 0).float() * mk).sum(1)
            sums["r5"] += ((rk <= 4).float() * mk).sum(1)
            sums["r10"] += ((rk <= 9).float() * mk).sum(1)
            sums["rr"] += (mk / (1 + rk)).sum(1)
            sums["rf"] += (rk / N * mk).sum(1)
            ey = F.embedding(yi, self.E)                                     # model_hier.py:83-85
            cur = ey.sum(1) / mk.sum(1, keepdim=True) + self.b_emb
            new = []
            for l, (Wg, bg, Wc, bc) in enumerate(self.gru):                 # customed_gru_cell.py:323-336
                h = state[:, l * H:(l + 1) * H]
                g = torch.sigmoid(torch.cat([cur, h], 1) @ Wg + bg)
                r, u = g[:, :H], g[:, H:]
                c = torch.tanh(torch.cat([cur, r * h], 1) @ Wc + bc)
                cur = u * h + (1 - u) * c
                new.append(cur)
            state = torch.cat(new, 1) * mask[:, i:i + 1]                     # model_hier.py:93
        users = (act > 0).float().sum()
        den = act + 1e-6
        out = dict(loss=float((ce_sum / den).sum() / users), recall1=float((sums["r1"] / den).sum() / users),
                   recall5=float((sums["r5"] / den).sum() / users), recall10=float((sums["r10"] / den).sum() / users),
                   mrr=float((sums["rr"] / den).sum() / users), rank_frac=float((sums["rf"] / den).sum() / users))
        return out, state


def time_cpu_baseline(args, params, x, y, mask, variants=((32, 4), (128, 1)), repeats=3, budget_s=30.0, threads=None):
    """Times eval_batch on bounded samples of the workload: for every (lanes, slots) variant one untimed warm-up run and ALWAYS
    `repeats` (>= 3) timed runs (median) of the first `lanes` users x `slots` session slots of (x, y, mask) -- (32, 4)
    is the reference's own batch width (args.py:193), (128, 1) a wide batch (best case for the CPU's GEMMs; at N = 300,000 one slot of
    128 lanes is a 4.6 GB logits tensor, 1,024 lanes would be 37 GB).  The budget bounds the whole leg by SHRINKING THE SAMPLE, never
    the repeats: a calibration run (one slot of 32 lanes) gives the cost per session, and a variant whose 1 + repeats runs would not fit
    its share of the budget gets fewer slots (never fewer than one) or fewer lanes (in steps of 32, never fewer than 32).  Returns
    dict(value = the BEST variant's sessions/s, cores, sample, variants=[...])."""
    repeats = max(int(repeats), 3)
    model = CpuHierTCN(args, params, threads)
    xt, yt = torch.from_numpy(x.astype(np.int64)), torch.from_numpy(y.astype(np.int64))
    mt = torch.from_numpy(mask.astype(np.float32))
    B, S, L = x.shape
    HT = args.hidden_dim * args.num_layer
    t_leg = time.perf_counter()

    def run(lanes, slots):
        xs, ys, ms = xt[:lanes, :slots].contiguous(), yt[:lanes, :slots].contiguous(), mt[:lanes, :slots].contiguous()
        st = torch.zeros(lanes, HT)
        t0 = time.perf_counter()
        model.eval_batch(xs, ys, ms, st)
        return time.perf_counter() - t0

    cal_lanes = min(32, B)
    run(cal_lanes, 1)                                   # page-in, thread pool, allocator
    per_session = run(cal_lanes, 1) / cal_lanes         # calibration: seconds per session at the narrow width
    share = max(budget_s - (time.perf_counter() - t_leg), 1e-3) / len(variants)
    fit = max(int(share / ((repeats + 1) * per_session)), 1)      # sessions per run that keep 1 + repeats runs inside the share
    results = []
    for lanes, slots in variants:
        lanes, slots = min(lanes, B), min(slots, S)
        if lanes * slots > fit:
            if slots > 1:
                slots = max(1, min(slots, fit // lanes))
            if lanes * slots > fit:
                lanes = max(min(32, lanes), (fit // slots) // 32 * 32)
        times = [run(lanes, slots) for _ in range(repeats + 1)][1:]            # run 0 = this shape's warm-up
        med = float(np.median(times))
        results.append(dict(lanes=lanes, slots=slots, sessions=lanes * slots, seconds_median=round(med, 3), timed_runs=len(times),
                            sessions_per_s=round(lanes * slots / med, 3)))
    best = max(results, key=lambda r: r["sessions_per_s"])
    return dict(value=best["sessions_per_s"], sessions=best["sessions"], seconds=time.perf_counter() - t_leg, cores=model.threads,
                variants=results,
                sample="best of " + "; ".join(f"{r['lanes']} lanes x {r['slots']} slots ({r['timed_runs']} timed runs after 1 warm-up, "
                                               f"median {r['seconds_median']} s -> {r['sessions_per_s']} sessions/s)" for r in results)
                       + f"; {L} steps, N={args.item_num}, d={getattr(args, 'emb_dim', 128)}, fp32 torch-CPU {torch.__version__}, "
                         "full logits per slot")
