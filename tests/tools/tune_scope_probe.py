"""Run by tests/test_gpu_shims.py in a process of its own with HIERTCN_LIB = libhiertcn_gfx950_testhooks.so (the only build that exports
hiertcn_test_tune_in_scope): the thread-local "tuning of the call in progress" (csrc/common.h TuneScope) points into a handle ONLY inside an
API call.  Between calls it is null -- after a successful call, after a failed one, after hiertcn_destroy -- so the launch helpers that take
no model (the mixture model's GRU trunk reaches pool_gru.hip's launch_gemm) can never read another handle's thresholds or a freed handle
(ADVICE r5).  The mixture model's outputs are the same whether or not a retuned handle was used, or destroyed, on the thread before.
Exit code 0 = all of that held."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from hiertcn_amd import _lib
from hiertcn_amd.args import hier_tcn_args, make_args
from hiertcn_amd.engine import Engine, HierTCNError
from hiertcn_amd.model_gru_gmm import GruGmmModel, init_params as gmm_params
from hiertcn_amd.params import init_params

assert os.environ.get("HIERTCN_LIB", "").endswith("testhooks.so")
lib = _lib.load()
in_scope = lib.hiertcn_test_tune_in_scope
in_scope.restype, in_scope.argtypes = C.c_int, []
assert in_scope() == 0

N, d = 400, 64
ga = make_args([], model_type="gru_gmm", item_num=N, output_dim=N, num_mixture=3, sigma=0.1, loss="gmm_prob", hidden_dim=d, num_layer=2, adjust=False)
gp = gmm_params(ga, 3)
rng = np.random.default_rng(0)
x = torch.as_tensor(rng.integers(1, N, (6, 9)).astype(np.int32)).cuda()


def gmm_hidden():
    m = GruGmmModel(ga, gp)
    h = m.forward(x).float().cpu().numpy().copy()
    m.close()
    return h


want = gmm_hidden()                       # nothing else has run on this thread: the defaults
assert in_scope() == 0

args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, tcn_channel=[d, d], num_layer=2)
eng = Engine(args, tuning=dict(gemm_split_max_tiles=1, skinny_max_rows=1))   # thresholds that would change launch_gemm's choice
eng.set_weights(init_params(args, 5))
assert in_scope() == 0
B, S, L = 4, 3, 5
ids = torch.as_tensor(rng.integers(1, N, (B, S, L)).astype(np.int32)).cuda()
xs = torch.zeros_like(ids)
xs[..., 1:] = ids[..., :-1]
mask = torch.ones((B, S), device="cuda")
st = torch.zeros((B, eng.HT), device="cuda")
eng.forward_eval(xs, ids, mask, st)
assert in_scope() == 0, "a successful call left the thread-local tuning pointing into its handle"
try:
    eng.forward_eval(xs, ids, mask, st[:, :-1].contiguous())          # rejected on the Python side
except HierTCNError:
    pass
rc = lib.hiertcn_forward_eval(eng._h, None, None, None, None, B, S, L, None, 0, None, None, None, None)   # rejected inside the call
assert rc != 0 and in_scope() == 0, "a failed call left the thread-local tuning pointing into its handle"
got_live = gmm_hidden()                   # the mixture model with a retuned handle alive on the thread
assert in_scope() == 0
eng.close()                               # hiertcn_destroy
assert in_scope() == 0
got_dead = gmm_hidden()                   # ... and after that handle is gone
torch.cuda.synchronize()
assert np.array_equal(want, got_live) and np.array_equal(want, got_dead), "the mixture model's launches depend on another handle's tuning"
print("tune_scope_probe ok")
