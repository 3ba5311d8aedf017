"""Same-box A/B of ONE source file: links libhiertcn_gfx950_<tag>.so from the shipped library's objects with <source> recompiled under extra flags.
usage: python scripts/build_alt.py <tag> <source.hip> [-DFLAG=...]...   then   HIERTCN_LIB=$PWD/hiertcn_amd/libhiertcn_gfx950_<tag>.so python ..."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hiertcn_amd import build as B
tag, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_lib(False)
obj = os.path.join(B.HERE, "build", "alt_%s_%s.o" % (tag, src.replace(".hip", "")))
per_file = ["-fno-honor-nans"] if src.endswith(B.NO_NAN_SOURCES) else []
subprocess.run([B._hipcc(), *B.FLAGS, *per_file, *flags, "-c", os.path.join(B.CSRC, src), "-o", obj], check=True)
objs = [obj if s == src else os.path.join(B.HERE, "build", s.replace(".hip", ".o")) for s in B.SOURCES]
out = os.path.join(B.HERE, "libhiertcn_gfx950_%s.so" % tag)
subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs], check=True)
print(out)
