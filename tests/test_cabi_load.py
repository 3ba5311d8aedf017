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
    from hiertcn_amd import _lib
    src = open(os.path.join(ROOT, "include", "hiertcn.h")).read()
    header_version = int(re.search(r"#define\s+HIERTCN_ABI_VERSION\s+(\d+)", src).group(1))
    assert lib.hiertcn_abi_version() == header_version == _lib.ABI_VERSION
    # the documents that quote the version quote the current one
    integ = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    assert re.findall(r"==\s*(\d+)\s*\(HIERTCN_ABI_VERSION\)", integ) == [str(header_version)]
    # ... in EVERY statement that compares it (round 3's stub asserted the previous version under a comment that named the current one)
    asserted = re.findall(r"hiertcn_abi_version\(\)\s*==\s*(\d+)", integ)
    assert asserted and set(asserted) == {str(header_version)}, asserted
    assert not re.search(r"\)[a-z_]+\.hiertcn_", integ), "two statements fused on one line in a code block of INTEGRATION.md"
    design = open(os.path.join(ROOT, "DESIGN.md")).read()
    assert set(re.findall(r"C-ABI v(\d+)", design)) == {str(header_version)}


# ---- struct layouts: header (compiled as C99 by gcc) == hiertcn_amd/_lib.py == the built library == the stubs in INTEGRATION.md ----------
C_STRUCTS = ["hiertcn_config_t", "hiertcn_weights_t", "hiertcn_score_out_t", "hiertcn_gmm_config_t", "hiertcn_gmm_weights_t", "hiertcn_tuning_t"]


def _binding_structs():
    from hiertcn_amd import _lib
    return dict(zip(C_STRUCTS, [_lib.Config, _lib.Weights, _lib.ScoreOut, _lib.GmmConfig, _lib.GmmWeights, _lib.Tuning]))


def _header_fields():
    """{struct: [field names in declaration order]} parsed from the header text."""
    src = open(os.path.join(ROOT, "include", "hiertcn.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef\s+struct\s*\{(.*?)\}\s*(hiertcn_\w+_t)\s*;", src, flags=re.S):
        out[name] = [re.match(r".*?(\w+)\s*(\[[^\]]*\])?$", decl.strip(), flags=re.S).group(1)
                     for decl in body.split(";") if decl.strip()]
    return out


@pytest.fixture(scope="module")
def c_layouts(tmp_path_factory):
    """sizeof / offsetof of every field of the header's structs, from a C99 program compiled with gcc."""
    import shutil
    import subprocess
    if not shutil.which("gcc"):
        pytest.skip("gcc not available")
    fields = _header_fields()
    assert sorted(fields) == sorted(C_STRUCTS), fields.keys()
    d = tmp_path_factory.mktemp("layout")
    lines = ['#include <stddef.h>', '#include <stdio.h>', '#include "hiertcn.h"', "int main(void) {"]
    for st in C_STRUCTS:
        lines.append(f'  printf("{st} %zu\\n", sizeof({st}));')
        for f in fields[st]:
            lines.append(f'  printf("{st}.{f} %zu\\n", offsetof({st}, {f}));')
    lines += ["  return 0;", "}"]
    (d / "layout.c").write_text("\n".join(lines))
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(d / "layout.c"), "-o", str(d / "layout")],
                   check=True)
    out = subprocess.run([str(d / "layout")], check=True, capture_output=True, text=True).stdout
    return {k: int(v) for k, v in (ln.split() for ln in out.splitlines())}, fields


def _check_struct(cls, st, layouts, fields):
    assert [f[0] for f in cls._fields_] == fields[st], (st, [f[0] for f in cls._fields_], fields[st])
    assert C.sizeof(cls) == layouts[st], (st, C.sizeof(cls), layouts[st])
    for f in fields[st]:
        assert getattr(cls, f).offset == layouts[f"{st}.{f}"], (st, f)


def test_struct_layouts_header_equals_binding_equals_library(lib, c_layouts):
    from hiertcn_amd import _lib
    layouts, fields = c_layouts
    for st, cls in _binding_structs().items():
        _check_struct(cls, st, layouts, fields)
    sz = (C.c_size_t * _lib.N_STRUCTS)()
    assert lib.hiertcn_struct_sizes(sz) == 0
    assert list(sz) == [layouts[st] for st in C_STRUCTS]
    assert _lib.load(f16=True).hiertcn_struct_sizes(sz) == 0 and list(sz) == [layouts[st] for st in C_STRUCTS]


def integration_md_stubs(text):
    """The ctypes.Structure classes of INTEGRATION.md's python blocks, executed: {C struct name (from the trailing comment): class}."""
    found = {}
    for block in re.findall(r"```python\n(.*?)```", text, flags=re.S):
        for m in re.finditer(r"^class\s+(\w+)\(C\.Structure\):\s*#\s*(hiertcn_\w+_t)\n(?:[ \t]+.*\n)+", block, flags=re.M):
            ns = {"C": C}
            exec(m.group(0), ns)
            found[m.group(2)] = ns[m.group(1)]
    return found


def test_integration_md_stubs_match_the_header(c_layouts):
    layouts, fields = c_layouts
    stubs = integration_md_stubs(open(os.path.join(ROOT, "INTEGRATION.md")).read())
    assert {"hiertcn_config_t", "hiertcn_score_out_t", "hiertcn_gmm_config_t"} <= set(stubs), stubs.keys()
    for st, cls in stubs.items():
        _check_struct(cls, st, layouts, fields)


def test_a_stale_stub_is_caught(c_layouts):
    """The round-2 stub of hiertcn_gmm_config_t (it stopped at `sigma`: 56 bytes against the header's 68) must fail the check above."""
    layouts, fields = c_layouts
    stale = """```python
class GmmConfig(C.Structure):                  # hiertcn_gmm_config_t
    _fields_ = [("item_num", C.c_int32), ("output_dim", C.c_int32), ("num_mixture", C.c_int32), ("kernel_size", C.c_int32),
                ("n_blocks", C.c_int32), ("channels", C.c_int32 * 8), ("sigma", C.c_float)]
```"""
    stubs = integration_md_stubs(stale)
    assert list(stubs) == ["hiertcn_gmm_config_t"] and C.sizeof(stubs["hiertcn_gmm_config_t"]) == 56
    with pytest.raises(AssertionError):
        _check_struct(stubs["hiertcn_gmm_config_t"], "hiertcn_gmm_config_t", layouts, fields)


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


def test_the_shipped_libraries_hold_no_fault_injection_switch():
    """The switches the time-out tests use (a missing workgroup in the persistent scan, producers that never signal in the layer-pipelined
    HierGRU launch) exist only in libhiertcn_gfx950_testhooks.so: the shipped libraries do not even contain their names."""
    from hiertcn_amd.build import LIB, LIB_F16, LIB_TESTHOOKS
    names = [b"HIERTCN_GRU_TEST_MISSING_WG", b"HIERTCN_LGRU_PIPE_TEST_NO_FLAGS"]
    for path in (LIB, LIB_F16):
        blob = open(path, "rb").read()
        assert not any(n in blob for n in names), path
    hooks = open(LIB_TESTHOOKS, "rb").read()
    assert all(n in hooks for n in names)


def test_the_shipped_libraries_read_no_environment_variable():
    """hiertcn_tuning_t (ABI 10) replaced the HIERTCN_* switches the launch path used to look up: the shipped libraries import no getenv and hold at
    most a handful of HIERTCN_ strings (the precision names of two error messages); the A/B switches of superseded kernels live in the ablation
    build alone, which does hold them."""
    import subprocess
    from hiertcn_amd.build import LIB, LIB_ABLATION, LIB_F16
    for path in (LIB, LIB_F16):
        blob = open(path, "rb").read()
        names = set(re.findall(rb"HIERTCN_[A-Z0-9_]+", blob))
        assert len(names) <= 5, (path, sorted(names))
        assert not any(n.startswith((b"HIERTCN_GRU_", b"HIERTCN_LGRU_", b"HIERTCN_SCORE", b"HIERTCN_TOPK", b"HIERTCN_TCN_", b"HIERTCN_TRAIN_", b"HIERTCN_SGRAD")) for n in names), sorted(names)
        nm = subprocess.run(["nm", "-D", "--undefined-only", path], capture_output=True, text=True)
        if nm.returncode == 0:
            assert "getenv" not in nm.stdout.split(), path
    abl = open(LIB_ABLATION, "rb").read()
    assert b"HIERTCN_TRAIN_LEGACY" in abl and b"HIERTCN_TCN_LAYER_V1" in abl and b"HIERTCN_TOPK_V1" in abl


def test_tuning_defaults_round_trip_and_env_mapping(lib):
    """hiertcn_get_tuning / hiertcn_set_tuning need no GPU... but a handle does (hiertcn_create touches the device), so what is checked here is the
    host side: the binding's field list equals the header's, and the mapping of the old environment names covers exactly fields that exist."""
    from hiertcn_amd import _lib
    for name, (field, conv) in _lib.ENV_TUNING.items():
        assert field in _lib.TUNING_FIELDS, (name, field)
    env = {"HIERTCN_GRU_CHAIN": "1", "HIERTCN_LGRU_CHAIN": "0", "HIERTCN_LGRU_PIPE": "0", "HIERTCN_SGRAD_DYN_MAX_BLOCKS": "127", "HIERTCN_TCN_GS2_MAX": "0",
           "HIERTCN_GRU_CHAIN_UNRELATED": "1"}
    assert _lib.tuning_from_env(env) == {"gru_scan_persistent": 0, "lgru_persistent": 1, "lgru_pipe": 0, "sgrad_dyn_max_blocks": 127, "tcn_gs2_max_sessions": -1}
    assert _lib.tuning_from_env({"HIERTCN_GRU_CHAIN": "0"}) == {}       # "0" = the library's own choice
    assert lib.hiertcn_get_tuning(None, None) == -1 and lib.hiertcn_set_tuning(None, None) == -1
