"""Run by tests/test_gpu_gru_persist.py in a process of its own with HIERTCN_LIB = libhiertcn_gfx950_testhooks.so (the only build that reads
HIERTCN_GRU_TEST_MISSING_WG): every wait inside the persistent across-session scan is bounded -- with a workgroup missing (the barriers expect
one more arrival than there are workgroups) the launch ends within seconds, the handle's health word is set, nothing hangs, and the handle works
again afterwards.  Exit code 0 = all of that held."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params
from tests.test_gpu_gru_persist import _inputs, _scan

assert os.environ.get("HIERTCN_LIB", "").endswith("testhooks.so")
d, nl = 128, 1
args = hier_tcn_args(item_num=300, emb_dim=d, hidden_dim=d, tcn_channel=[d, d], num_layer=nl)
eng = Engine(args)
eng.set_weights(init_params(args, 5))
data = _inputs(np.random.default_rng(2), 8, 3, d, d * nl)
assert not eng.persistent_timed_out()
os.environ["HIERTCN_GRU_TEST_MISSING_WG"] = "1"
os.environ["HIERTCN_GRU_CHAIN"] = "0"
try:
    t0 = time.time()
    eng.gru_scan(*data)
    torch.cuda.synchronize()
    took = time.time() - t0
finally:
    os.environ.pop("HIERTCN_GRU_TEST_MISSING_WG", None)
    os.environ.pop("HIERTCN_GRU_CHAIN", None)
assert eng.persistent_timed_out()
assert took < 30.0, took
# the health read that reported the time-out also reset the handle's barrier counters (they were left at a value that is no multiple
# of the workgroup count, and every later barrier target derives from them): the handle works again, and reports healthy
want = _scan(eng, *data, chain=True)
got = _scan(eng, *data, chain=False)
for g, w in zip(got, want):
    np.testing.assert_allclose(g, w, rtol=2e-5, atol=2e-5)
assert not eng.persistent_timed_out()
eng.close()

# ---- the layer-pipelined HierGRU low level (csrc/lgru_pipe.hip): consumers wait for their producers' flags inside the launch, bounded the same way.
# HIERTCN_LGRU_PIPE_TEST_NO_FLAGS (test-hooks build only): the producers never signal.
from tests.gpu_common import make_batch
d, nl, B, S, L = 64, 2, 5, 3, 7
args = hier_tcn_args(item_num=300, emb_dim=d, hidden_dim=d, tcn_channel=[d, d], num_layer=nl, model_low_type="gru")
eng = Engine(args)
eng.set_weights(init_params(args, 5))
rng = np.random.default_rng(4)
x, y, mask = make_batch(rng, B, S, L, 300)
dv = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
sproj = (0.1 * torch.randn((B * S, d), device="cuda")).float()
xs = dv(x.reshape(B * S, L).astype(np.int32))
good = eng.tcn_forward(xs, sproj).float().cpu().numpy()
assert not eng.persistent_timed_out()
os.environ["HIERTCN_LGRU_PIPE_TEST_NO_FLAGS"] = "1"
try:
    t0 = time.time()
    eng.tcn_forward(xs, sproj)
    torch.cuda.synchronize()
    took2 = time.time() - t0
finally:
    os.environ.pop("HIERTCN_LGRU_PIPE_TEST_NO_FLAGS", None)
assert eng.persistent_timed_out()       # (the read also clears the word)
assert took2 < 60.0, took2
again = eng.tcn_forward(xs, sproj).float().cpu().numpy()
assert np.array_equal(again, good) and not eng.persistent_timed_out()
eng.close()
print("persist_timeout_probe ok: timed out in %.1f s, recovered; lgru_pipe: timed out in %.1f s, recovered" % (took, took2))
