"""Builds libhiertcn_gfx950.so in-tree with hipcc (gfx950 only).  Used by __graft_entry__.build()
and by `python -m hiertcn_amd.build`.  hipcc cross-compiles without a GPU."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libhiertcn_gfx950.so")
SOURCES = ["api.hip", "prep.hip", "pool_gru.hip", "gru_persist.hip", "tcn.hip", "tcn_layers.hip", "score.hip", "score_bwd.hip", "metrics.hip", "train.hip", "gmm.hip", "gmm_train.hip", "gmm_gram.hip", "exact.hip", "gemm.hip", "gru_x3.hip", "score_x3.hip", "score_x3w.hip", "lgru_tile.hip", "lgru_pipe.hip", "lgru_wide.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-pass-failed", "-Wno-shift-negative-value",
         "-fno-gpu-rdc", "-D__HIP_PLATFORM_AMD__"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


LIB_F16 = os.path.join(HERE, "libhiertcn_gfx950_f16.so")


LIB_TESTHOOKS = os.path.join(HERE, "libhiertcn_gfx950_testhooks.so")
TESTHOOK_SOURCES = ("gru_persist.hip", "lgru_pipe.hip", "api.hip")   # the sources that hold a fault-injection switch / a test probe


# libhiertcn_gfx950_ablation.so: the bfloat16 library PLUS every superseded kernel and A/B switch (-DHIERTCN_ABLATION=1): the previous forms of the
# scoring / top-k / TCN-layer kernels, the legacy scoring backward (score_grad.hip), the timing flavours.  Only there are the old HIERTCN_* environment
# switches of those alternatives looked up.  Loaded through HIERTCN_LIB / Engine(lib_variant="ablation") by the A/B tests and scripts/reproduce.sh;
# the shipped libraries hold one kernel per job.
LIB_ABLATION = os.path.join(HERE, "libhiertcn_gfx950_ablation.so")
ABLATION_EXTRA_SOURCES = ["score_grad.hip"]


def _mentions_ablation(src: str) -> bool:
    with open(os.path.join(CSRC, src)) as f:
        t = f.read()
    return "HIERTCN_ABLATION" in t or "ABL_ENV" in t


def build_ablation(force: bool = False) -> str:
    build_lib(False)
    hipcc = _hipcc()
    obj_dir = os.path.join(HERE, "build", "ablation")
    os.makedirs(obj_dir, exist_ok=True)
    headers = _headers()
    jobs, objs = [], []
    for src in SOURCES + ABLATION_EXTRA_SOURCES:
        if src in ABLATION_EXTRA_SOURCES or _mentions_ablation(src):
            s, o = os.path.join(CSRC, src), os.path.join(obj_dir, src.replace(".hip", ".o"))
            if force or _stale(o, [s] + headers):
                jobs.append((s, o))
            objs.append(o)
        else:   # no ablation content: the shipped library's object
            objs.append(os.path.join(HERE, "build", src.replace(".hip", ".o")))

    def cc(job):
        s, o = job
        per_file = ["-fno-honor-nans"] if s.endswith(NO_NAN_SOURCES) else []
        r = subprocess.run([hipcc, *FLAGS, *per_file, "-DHIERTCN_ABLATION=1", "-c", s, "-o", o], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (s, r.stderr))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    if force or jobs or _stale(LIB_ABLATION, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_ABLATION, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (r.stdout, r.stderr))
    return LIB_ABLATION


NO_NAN_SOURCES = ("score.hip", "score_x3.hip", "score_x3w.hip")


def _headers():
    return [os.path.join(CSRC, h) for h in ("common.h", "train.h", "exact.h", "gemm.h", "gmm.h", "score_geom.h", "score_lists.h", "score_x3.h")] + [
        os.path.join(HERE, "..", "include", "hiertcn.h")]


LIB_PACKER = os.path.join(HERE, "libhiertcn_packer.so")   # host-side batch packer (csrc/packer.c: plain C, gcc)


def build_packer(force: bool = False) -> str:
    src = os.path.join(CSRC, "packer.c")
    if force or _stale(LIB_PACKER, [src]):
        r = subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-std=c99", "-Wall", "-o", LIB_PACKER, src], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("gcc failed: %s\n%s" % (src, r.stderr))
    return LIB_PACKER


def build_all(force: bool = False, verbose: bool = False):
    """Both storage variants: bfloat16 (default, training + inference) and IEEE half (-DHIERTCN_F16, inference)."""
    libs = build_lib(force, verbose), build_lib(force, verbose, f16=True)
    build_testhooks(force)
    build_ablation(force)
    build_packer(force)
    return libs


def build_testhooks(force: bool = False) -> str:
    """libhiertcn_gfx950_testhooks.so: the bfloat16 library with the fault-injection switches of the tests compiled IN (-DHIERTCN_TEST_HOOKS; the
    shipped libraries have none).  Same objects as the shipped library except the few sources that hold a switch.  Loaded only by
    tests/tools/persist_timeout_probe.py through HIERTCN_LIB."""
    build_lib(False)
    hipcc = _hipcc()
    obj_dir = os.path.join(HERE, "build", "testhooks")
    os.makedirs(obj_dir, exist_ok=True)
    headers = _headers()
    objs = []
    rebuilt = False
    for src in SOURCES:
        if src in TESTHOOK_SOURCES:
            s, o = os.path.join(CSRC, src), os.path.join(obj_dir, src.replace(".hip", ".o"))
            if force or _stale(o, [s] + headers):
                r = subprocess.run([hipcc, *FLAGS, "-DHIERTCN_TEST_HOOKS=1", "-c", s, "-o", o], capture_output=True, text=True)
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed: %s\n%s" % (s, r.stderr))
                rebuilt = True
            objs.append(o)
        else:
            objs.append(os.path.join(HERE, "build", src.replace(".hip", ".o")))
    if force or rebuilt or _stale(LIB_TESTHOOKS, objs):
        r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_TESTHOOKS, *objs], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (r.stdout, r.stderr))
    return LIB_TESTHOOKS


def build_lib(force: bool = False, verbose: bool = False, extra_flags=(), f16: bool = False) -> str:
    LIB = LIB_F16 if f16 else globals()["LIB"]  # noqa: N806
    if f16:
        extra_flags = tuple(extra_flags) + ("-DHIERTCN_F16=1",)
    ablation = False   # (the ablation library is a target of its own: build_ablation)
    # objects live in a directory keyed on the flag set, and the library remembers which flag set it was linked from:
    # an ablation build can never leave objects (or a library) behind that a later plain build would ship as fresh
    import hashlib
    user_flags = [f for f in extra_flags if f not in ("-DHIERTCN_F16=1", "-DHIERTCN_ABLATION=1")]
    OBJ = os.path.join(HERE, "build", *(["ablation"] if ablation else []), *(["f16"] if f16 else []),  # noqa: N806
                       *(["x" + hashlib.sha1(" ".join(user_flags).encode()).hexdigest()[:8]] if user_flags else []))
    flagset = " ".join([*FLAGS, *extra_flags])
    stamp = LIB + ".flags"
    try:
        with open(stamp) as f:
            relink = f.read() != flagset
    except OSError:
        relink = True
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    headers = _headers()
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        # score.hip: logits are finite or -inf by construction; without NaN-honouring fmaxf needs no
        # canonicalising v_max on MFMA results (cdna_hip_programming.md, attention 'Pitfalls')
        per_file = ["-fno-honor-nans"] if s.endswith(NO_NAN_SOURCES) else []
        cmd = [hipcc, *FLAGS, *per_file, *extra_flags, "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(OBJ, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or relink or _stale(LIB, objs):
        # nothing but the HIP runtime: every kernel of the library, the plain GEMMs of the backward pass included (csrc/gemm.hip), is its own
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (r.stdout, r.stderr))
        with open(stamp, "w") as f:
            f.write(flagset)
    return LIB


if __name__ == "__main__":
    print(*build_all(force="--force" in sys.argv, verbose=True))
