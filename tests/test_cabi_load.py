"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/hiertcn.h
declares, reports the ABI version, and fails loudly (no fallback) without a GPU."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from hiertcn_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from hiertcn_amd.build import build_lib
        build_lib()
    return _lib.load()


def header_symbols():
    src = open(os.path.join(ROOT, "include", "hiertcn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(hiertcn_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree(lib):
    from hiertcn_amd import _lib
    names = header_symbols()
    assert names, "no prototypes parsed from include/hiertcn.h"
    assert set(names) == set(_lib.SYMBOLS), set(names) ^ set(_lib.SYMBOLS)
    for n in names:
        assert hasattr(lib, n), f"{n} not exported"


def test_abi_version(lib):
    assert lib.hiertcn_abi_version() == 6


def test_create_rejects_bad_config_without_touching_the_gpu(lib):
    from hiertcn_amd import _lib
    cfg = _lib.Config(item_num=100, emb_dim=100, hidden_dim=32, num_layer=2, kernel_size=5, n_blocks=2, hier=1)
    cfg.channels[0] = cfg.channels[1] = 100
    h = C.c_void_p()
    rc = lib.hiertcn_create(C.byref(h), C.byref(cfg))
    assert rc == -1 and b"emb_dim" in lib.hiertcn_last_error()
    assert lib.hiertcn_create(None, None) == -1


def test_no_silent_cpu_fallback():
    torch = pytest.importorskip("torch")
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from hiertcn_amd._lib import HierTCNError
    from hiertcn_amd.args import hier_tcn_args
    from hiertcn_amd.engine import Engine
    with pytest.raises(HierTCNError):
        Engine(hier_tcn_args(item_num=100, emb_dim=64, hidden_dim=32, tcn_channel=[64, 64]))


def test_both_storage_variants_load_and_report_their_dtype():
    """libhiertcn_gfx950.so stores bfloat16, libhiertcn_gfx950_f16.so IEEE half; same exports (no compute without a GPU)."""
    from hiertcn_amd import _lib
    a, b = _lib.load(), _lib.load(f16=True)
    assert a.hiertcn_storage_dtype() == 0 and b.hiertcn_storage_dtype() == 1
    assert a.hiertcn_abi_version() == b.hiertcn_abi_version()
