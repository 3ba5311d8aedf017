"""INTEGRATION.md section 2 -- the ctypes stub a maintainer of the reference would paste -- EXECUTED as written against the built library
(a stub that asserts a stale ABI number or binds a wrong argument list must fail here, on the GPU box, not at a
maintainer's desk): the code block is extracted from the file, the library name replaced by the in-tree path, and its `make_handle` +
`eval_step` run one tiny evaluation batch whose metrics and carried state must equal the repo's own binding's (hiertcn_amd/engine.py) on
the same handle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from tests.gpu_common import make_batch, make_params

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def section2_stub():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## 2. The ctypes stub"):]
    block = re.search(r"```python\n(.*?)```", sec, flags=re.S).group(1)
    assert 'C.CDLL("libhiertcn_gfx950.so")' in block
    return block


@pytest.mark.parametrize("low", ["tcn", "gru"])
def test_the_stub_of_integration_md_runs_as_written(low):
    from hiertcn_amd import _lib
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.engine import Engine
    ns = {}
    exec(section2_stub().replace('C.CDLL("libhiertcn_gfx950.so")', "C.CDLL(%r)" % _lib.LIB_PATH), ns)   # its asserts (ABI number, struct sizes) run here
    N, d, B, S, L = 700, 128, 5, 3, 9   # (the stub fixes emb_dim = 128, the reference's embedding width)
    args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, num_layer=2, tcn_channel=[d, d], kernel_size=5, batch_size=B, model_low_type=low)
    h = ns["make_handle"](args)
    assert h.value
    # weights through the repo's binding ON THE STUB'S HANDLE (section 3 of the file describes the call; the stub itself stops at evaluation)
    eng = Engine(args, hier=True)
    own, eng._h = eng._h, h
    eng.set_weights(make_params(args))
    rng = np.random.default_rng(3)
    x, y, mask = make_batch(rng, B, S, L, N)
    dv = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    xd, yd, md = dv(x.astype(np.int32)), dv(y.astype(np.int32)), dv(mask.astype(np.float32))
    st = (0.2 * torch.randn((B, 2 * d), device="cuda")).float()
    ref = eng.forward_eval(xd, yd, md, st)
    ws_bytes = ns["lib"].hiertcn_workspace_bytes(h, B, S, L)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    st_out = torch.empty_like(st)
    m8 = torch.zeros(8, dtype=torch.float64, device="cuda")
    ns["eval_step"](h, xd.data_ptr(), yd.data_ptr(), md.data_ptr(), st.data_ptr(), B, S, L, ws.data_ptr(), ws_bytes, st_out.data_ptr(), m8.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(st_out, ref["state"])
    a, b = m8.cpu().numpy(), ref["metrics8"].cpu().numpy()
    assert np.array_equal(a[1:], b[1:])                  # recalls, reciprocal rank, rank, users, steps: the ranks are exact
    assert abs(a[0] - b[0]) <= 1e-6 * abs(b[0])          # (the loss sum's last bits follow where the walk over the table started: DESIGN 4.1)
    assert float(m8[6]) == B   # users counted
    # the stub's error path: a NULL handle comes back as a RuntimeError carrying hiertcn_last_error()
    with pytest.raises(RuntimeError):
        ns["eval_step"](None, xd.data_ptr(), yd.data_ptr(), md.data_ptr(), st.data_ptr(), B, S, L, ws.data_ptr(), ws_bytes, st_out.data_ptr(), m8.data_ptr())
    eng.close()            # destroys the stub's handle
    eng._h = own
    eng._weights_set = False
    eng.close()
