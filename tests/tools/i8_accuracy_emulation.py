"""CPU emulation (NumPy fp64) of the candidate forms of the parity-grade scoring product on the inputs of
tests/test_gpu_exact_mode.py::test_exact_mode_at_trained_logit_scale_xing_shape (XING catalogue 20,778, d = 256, 32 lanes x 10 x 20): logit error against
the exact oracle, share of rows whose rank (loss.py:179) equals the oracle's, share of rows the rank test is conclusive on.  The accuracy half of
VERDICT r4 item 3's gate (the rate half: scripts/i8_shape_bench.hip).  usage: python tests/tools/i8_accuracy_emulation.py [zscale]; results: profiles/r5_i8_shape_bench.txt"""
import sys, numpy as np, time
sys.path.insert(0, __import__('os').path.abspath(__import__('os').path.join(__import__('os').path.dirname(__file__), '..', '..')))
from oracle import hiertcn_oracle as O
from tests.gpu_common import as_lists, make_batch, make_params, small_args
from tests.test_gpu_exact_mode import _trained_like
zscale = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
N, d, B, S, L = 20778, 256, 32, 10, 20
args = small_args(N=N, d=d, H=d, nl=2, blocks=2, K=5, L=L, S=S, B=B)
p = _trained_like(make_params(args), args, "hier/tcn")
rng = np.random.default_rng(20778)
x, y, mask = make_batch(rng, B, S, L, N, ragged=True, zipf=True)
st0 = (0.2 * rng.standard_normal((B, 2 * d))).astype(np.float32)
xl, yl, ml = as_lists(x, y, mask)
yy = y.reshape(B, S * L).astype(np.int64); valid = yy > 0
hid64, _ = O.model_hier(args, xl, yl, ml, st0, p, O.EXACT64, return_hidden=True)
Wo, bo = p["hier/tcn/dense/kernel"].astype(np.float64), p["hier/tcn/dense/bias"].astype(np.float64)
z0 = hid64 @ Wo + bo
f = np.float32(zscale / np.abs(z0[valid]).max())
W = (p["hier/tcn/dense/kernel"] * f).astype(np.float32); b = (p["hier/tcn/dense/bias"] * f).astype(np.float32)
H = hid64.reshape(-1, d).astype(np.float32)
v = valid.reshape(-1); H = H[v]; yv = yy.reshape(-1)[v]
exact = H.astype(np.float64) @ W.astype(np.float64) + b.astype(np.float64)
print("rows", H.shape[0], "zmax", np.abs(exact).max())
def scale_exp(mx): return 14 - np.frexp(mx)[1]
kh = scale_exp(np.abs(H).max(1)); kw = scale_exp(np.abs(W).max())
Hs = np.ldexp(H.astype(np.float64), kh[:, None]); Ws = np.ldexp(W.astype(np.float64), kw)
f16 = lambda a: a.astype(np.float16).astype(np.float64)
hh, wh = f16(Hs), f16(Ws); hl, wl = Hs - hh, Ws - wh
hl16, wl16 = f16(hl), f16(wl)
inv = np.ldexp(1.0, -(kh + kw))[:, None]
def report(name, z):
    err = np.abs(z - exact); er = err.max(1)
    tgt = exact[np.arange(len(yv)), yv][:, None]
    rk_ref = (exact > tgt).sum(1)
    ztgt = z[np.arange(len(yv)), yv][:, None]
    rk = (z > ztgt).sum(1)
    gap = np.abs(exact - tgt); gap[np.arange(len(yv)), yv] = np.inf
    clear = gap.min(1) > 2 * er + 1e-7
    print("%-28s err max %.3e mean %.3e rms %.3e | rank equal %.4f | clear rows %.4f" % (name, err.max(), err.mean(), np.sqrt((err**2).mean()), (rk == rk_ref).mean(), clear.mean()))
report("fp32 inputs only (fp64 acc)", exact.astype(np.float32).astype(np.float64))
z = (hh @ wh + hh @ wl16 + hl16 @ wh) * inv + b; report("x3 f16 (fp64 acc)", z)
# fp32 accumulation model: torch fp32 matmul
import torch
zf = (torch.from_numpy(H) @ torch.from_numpy(W) + torch.from_numpy(b)).numpy().astype(np.float64); report("plain fp32 GEMM (torch cpu)", zf)
r8 = lambda a, k: np.clip(np.rint(np.ldexp(a, k)), -127, 127)
hh8, wh8 = r8(hh, -7), r8(wh, -7); hl8, wl8 = r8(hl, 5), r8(wl, 5)
z = (hh @ wh + 4.0 * (hh8 @ wl8 + hl8 @ wh8)) * inv + b; report("hybrid f16 + i8 cross", z)
# hybrid with 9-bit hi copies? (cross via two i8: not considered).  variant: cross operands rounded from the unsplit values
# pure i8 8+8 balanced digits: x = 2^8 d1 + d0 with |x| < 2^15
def digits(a, k):
    q = np.rint(np.ldexp(a, k)); d1 = np.rint(q / 256.0); d0 = q - 256 * d1; return np.clip(d1, -127, 127), d0
h1, h0 = digits(Hs, 0); w1, w0 = digits(Ws, 0)   # scaled max in [2^13, 2^14): 15 bits incl. sign; use k=1 for 16
z = (65536.0 * (h1 @ w1) + 256.0 * (h1 @ w0 + h0 @ w1)) * inv + b; report("i8 8+8 (k=0, 3 products)", z)
h1, h0 = digits(Hs, 1); w1, w0 = digits(Ws, 1)
z = (65536.0 * (h1 @ w1) + 256.0 * (h1 @ w0 + h0 @ w1)) * inv / 4 + b; report("i8 8+8 (k=1, 3 products)", z)
z = (65536.0 * (h1 @ w1) + 256.0 * (h1 @ w0 + h0 @ w1) + h0 @ w0) * inv / 4 + b; report("i8 8+8 (k=1, 4 products)", z)
