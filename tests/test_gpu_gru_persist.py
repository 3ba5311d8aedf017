"""The across-session GRU scan as one persistent launch (csrc/gru_persist.hip) against the chain of GEMM launches it replaces
(csrc/pool_gru.hip; reference: model_hier.py:91-93 -> customed_gru_cell.py:323-337) and against the oracle.

Both paths are fp32 dot products that differ in summation order only; the tolerance below is that of fp32 sums of <= 1024 terms carried
over S slots (test_gpu_parity.test_gru_scan holds the default path to the oracle with the same numbers).  The inter-workgroup hand-off
(sc1 stores / loads around one agent-scope counter) is soaked: many launches on one handle, every one compared."""
import os

import numpy as np
import pytest
import torch

from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params
from oracle import hiertcn_oracle as O

pytestmark = pytest.mark.gpu


def _scan(eng, pooled, mask, st0, chain):
    """chain: the scan as 2 (S + 1) GEMM launches (hiertcn_tuning_t.gru_scan_persistent = 0); else the library's choice (the persistent launch
    at these shapes)."""
    eng.set_tuning(gru_scan_persistent=0 if chain else -1)
    try:
        out = eng.gru_scan(pooled, mask, st0, want_state_before=True)
        torch.cuda.synchronize()
    finally:
        eng.set_tuning(gru_scan_persistent=-1)
    return [t.cpu().numpy() for t in out]


def _inputs(rng, B, S, E, HT, dead_lanes=True):
    pooled = torch.from_numpy(rng.standard_normal((B, S, E)).astype(np.float32) * 0.3).cuda()
    mask = (rng.random((B, S)) > 0.25).astype(np.float32)
    if dead_lanes and B > 2:
        mask[1] = 0.0                      # a lane without any session
    st0 = torch.from_numpy(rng.standard_normal((B, HT)).astype(np.float32) * 0.2).cuda()
    return pooled, torch.from_numpy(mask).cuda(), st0


@pytest.mark.parametrize("d,nl,B,S", [(256, 2, 32, 10), (256, 2, 3, 2), (256, 2, 7, 3), (128, 1, 50, 5), (64, 3, 128, 4), (256, 1, 33, 6),
                                      (256, 2, 100, 10)])
def test_persistent_scan_matches_the_gemm_chain_and_the_oracle(d, nl, B, S):
    args = hier_tcn_args(item_num=500, emb_dim=d, hidden_dim=d, tcn_channel=[d, d], num_layer=nl)
    p = init_params(args, 3)
    eng = Engine(args)
    eng.set_weights(p)
    rng = np.random.default_rng(B * 1000 + S)
    pooled, mask, st0 = _inputs(rng, B, S, d, d * nl)
    chain = _scan(eng, pooled, mask, st0, chain=True)
    pers = _scan(eng, pooled, mask, st0, chain=False)
    for name, a, b in zip(("sproj", "state_out", "state_before"), chain, pers):
        np.testing.assert_allclose(b, a, rtol=2e-5, atol=2e-5, err_msg=name)
    # the oracle's scan (fp64 accumulate) on the same inputs: carried state after the last slot
    st = st0.cpu().numpy().astype(np.float64)
    pn, mn = pooled.cpu().numpy(), mask.cpu().numpy()
    for s in range(S):
        _, st = O.multi_rnn_cell(pn[:, s].astype(np.float64), st, p, d, nl)
        st = st * mn[:, s:s + 1]
    np.testing.assert_allclose(pers[1], st, rtol=2e-4, atol=2e-5)
    eng.close()


def test_persistent_scan_soak_and_graph_replay():
    """200 launches on one handle (the barrier counter only grows), alternating batch shapes, each equal to the first result of its
    shape bit for bit: a stale or torn hand-off between workgroups would show as a mismatch."""
    d, nl = 256, 2
    args = hier_tcn_args(item_num=500, emb_dim=d, hidden_dim=d, tcn_channel=[d, d], num_layer=nl)
    eng = Engine(args)
    eng.set_weights(init_params(args, 5))
    rng = np.random.default_rng(11)
    shapes = [(32, 10), (5, 3), (64, 7)]
    data = [_inputs(rng, B, S, d, d * nl) for B, S in shapes]
    first = [_scan(eng, *dat, chain=False) for dat in data]
    ref = [_scan(eng, *dat, chain=True) for dat in data]
    for f, r in zip(first, ref):
        np.testing.assert_allclose(f[1], r[1], rtol=2e-5, atol=2e-5)
    assert eng.get_tuning()["gru_scan_persistent"] == -1
    try:
        for it in range(200):
            i = it % len(shapes)
            out = eng.gru_scan(*data[i], want_state_before=True)
            if it % 10 == 9:
                torch.cuda.synchronize()
            got = [t for t in out]
            for g, f in zip(got, first[i]):
                assert np.array_equal(g.cpu().numpy(), f), f"launch {it}: result differs from the first launch of its shape"
    finally:
        pass
    assert not eng.persistent_timed_out()   # no bounded wait ever ran out
    eng.close()


def test_a_wait_that_cannot_be_met_times_out_instead_of_hanging():
    """Every wait inside the persistent launch is bounded: with a workgroup missing the launch ends within seconds, the handle's health word
    is set, nothing hangs and the handle recovers.  The missing workgroup is a fault-injection switch that only the test-hooks build of the
    library contains (the shipped libraries read no environment variable on this path): tests/tools/persist_timeout_probe.py runs against it
    in a process of its own.  The shipped library ignores the switch (second half)."""
    import subprocess
    import sys
    from hiertcn_amd.build import LIB_TESTHOOKS
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    assert os.path.exists(LIB_TESTHOOKS), "python -m hiertcn_amd.build (or __graft_entry__.build()) builds the test-hooks library"
    env = dict(os.environ, HIERTCN_LIB=LIB_TESTHOOKS)
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "persist_timeout_probe.py")], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "persist_timeout_probe ok" in r.stdout
    # the shipped library: the same switch changes nothing
    d, nl = 128, 1
    args = hier_tcn_args(item_num=300, emb_dim=d, hidden_dim=d, tcn_channel=[d, d], num_layer=nl)
    eng = Engine(args)
    eng.set_weights(init_params(args, 5))
    data = _inputs(np.random.default_rng(2), 8, 3, d, d * nl)
    want = _scan(eng, *data, chain=True)
    os.environ["HIERTCN_GRU_TEST_MISSING_WG"] = "1"
    try:
        got = _scan(eng, *data, chain=False)
    finally:
        os.environ.pop("HIERTCN_GRU_TEST_MISSING_WG", None)
    for g, w in zip(got, want):
        np.testing.assert_allclose(g, w, rtol=2e-5, atol=2e-5)
    assert not eng.persistent_timed_out()
    eng.close()


def test_persistent_scan_beside_a_long_kernel_on_another_stream():
    """The launch's workgroups can only all be resident once a long kernel on another stream has let go of the CUs: the first barrier waits
    for that (long bound), the result is the same and no wait runs out."""
    d, nl = 256, 2
    args = hier_tcn_args(item_num=500, emb_dim=d, hidden_dim=d, tcn_channel=[d, d], num_layer=nl)
    eng = Engine(args)
    eng.set_weights(init_params(args, 5))
    data = _inputs(np.random.default_rng(4), 32, 10, d, d * nl)
    want = [t.clone() for t in eng.gru_scan(*data, want_state_before=True)]
    big = hier_tcn_args(item_num=120_000, emb_dim=d, hidden_dim=d, tcn_channel=[d, d], num_layer=nl)
    eb = Engine(big)
    eb.set_weights(init_params(big, 6))
    rows = 600_000
    hid = (torch.randn(rows, d, device="cuda") * 0.5).to(torch.bfloat16)
    y = torch.randint(1, 120_000, (rows,), device="cuda", dtype=torch.int32)
    eb.score_fused(hid, y)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    for _ in range(3):
        with torch.cuda.stream(side):
            eb.score_fused(hid, y)               # ~30 ms of every CU
        got = eng.gru_scan(*data, want_state_before=True)
        torch.cuda.synchronize()
        for g, w in zip(got, want):
            assert torch.equal(g, w)
    assert not eng.persistent_timed_out()
    eb.close()
    eng.close()
