"""Soak of the reference-batch paths that hand data between workgroups inside a launch or size their grids on the device (round 4): the
layer-pipelined HierGRU low level (lgru_pipe.hip), the device-side catalogue split of the scoring kernels, the persistent scans.  Repeats the
evaluation of one ragged 32-lane batch and asserts that ranks / cross entropies / carried state are BITWISE those of the first run, then repeats a
training forward + backward and asserts the loss is bitwise and every gradient within 1e-5 relative of the first run's (atomics reorder sums).
usage: [LANES=260] python scripts/soak_refbatch.py [n_eval] [n_train]
LANES > 64 (the mid-size forms of DESIGN 4.9: split-K scan chain, the two-block kernel's device-side split, the training forward's split over several rounds, the
wide HierGRU kernels as several launches): ranks and carried state bitwise, cross entropies within 5e-5 and the loss within 1e-6 relative -- where the host-sized
split takes over, the scoring workgroups start their item walk wherever their XCD's walk stands (HIERTCN_FIXED_WALK), so sums are formed in another order run to run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

n_eval = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
n_train = int(sys.argv[2]) if len(sys.argv) > 2 else 200
N, B, S, L = 20778, int(os.environ.get("LANES", "32")), 10, 20
loose = B > 64
rng = np.random.default_rng(0)
ids = rng.integers(1, N, (B, S, L)) * (np.arange(L)[None, None, :] < rng.integers(1, L + 1, (B, S))[..., None])
x = np.zeros_like(ids); x[..., 1:] = ids[..., :-1]
mask = (rng.random((B, S)) > 0.15).astype(np.float32)
xs, ys, ms = (torch.as_tensor(v).cuda().contiguous() for v in (x.astype(np.int32), ids.astype(np.int32), mask))
bad = 0
for low, d, dtype in (("tcn", 256, None), ("gru", 128, None), ("gru", 64, None), ("gru", 256, None), ("tcn", 256, "split16")):
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, batch_size=B, model_low_type=low)
    eng = Engine(args, hier=True, train=dtype is None, **({"dtype": dtype} if dtype else {}))
    eng.set_weights(init_params(args, seed=2))
    st = (0.1 * torch.randn((B, eng.HT), device="cuda")).float()
    first = None
    for i in range(n_eval):
        out = eng.forward_eval(xs, ys, ms, st, want_rows=True)
        cur = (out["rank"].clone(), out["ce"].clone(), out["state"].clone())
        if first is None:
            first = cur
        elif loose and torch.equal(first[0], cur[0]) and torch.equal(first[2], cur[2]) and float((first[1] - cur[1]).abs().max()) <= 5e-5:
            pass
        elif not all(torch.equal(a, b) for a, b in zip(first, cur)):
            bad += 1
            print("MISMATCH eval", low, d, dtype, "iteration", i, flush=True)
    torch.cuda.synchronize()
    msg = "%s d=%d %s, %d lanes: %d evaluation launches %s" % (low, d, dtype or "16-bit", B, n_eval, "agree (ranks, state bitwise; ce 5e-5)" if loose else "bitwise identical")
    if dtype is None:
        g0, l0 = None, None
        for i in range(n_train):
            o = eng.train_forward_backward(xs, ys, ms, st)
            g = eng.export("grads")
            loss = o["metrics8"].clone()
            if g0 is None:
                g0, l0 = g, loss
            else:
                if loose and float((l0 - loss).abs().max()) <= 1e-6 * float(l0.abs().max()):
                    pass
                elif not torch.equal(l0, loss):
                    bad += 1; print("MISMATCH train loss", low, d, i, flush=True)
                for k in g0:
                    den = np.abs(g0[k]).max() + 1e-30
                    if np.abs(g[k] - g0[k]).max() / den > 1e-5:
                        bad += 1; print("MISMATCH grad", low, d, k, i, np.abs(g[k] - g0[k]).max() / den, flush=True)
        msg += "; %d training passes: loss %s, gradients within 1e-5" % (n_train, "within 1e-6" if loose else "bitwise")
    print(msg + "; timed_out %s" % eng.persistent_timed_out(), flush=True)
    bad += int(eng.persistent_timed_out())
    eng.close()
print("soak_refbatch:", "FAILED (%d)" % bad if bad else "ok")
sys.exit(1 if bad else 0)
