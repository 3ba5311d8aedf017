"""VGPRs / spills / scratch of every kernel of one source file, as hipcc reports them (-Rpass-analysis=kernel-resource-usage).
usage: python scripts/kernel_resources.py score_x3.hip [filter]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "hiertcn_amd", "csrc", sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-fno-gpu-rdc", "-D__HIP_PLATFORM_AMD__", "-Wno-shift-negative-value",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"]
if sys.argv[1].startswith("score"):
    cmd.insert(1, "-fno-honor-nans")
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, d = None, {}
for l in err.splitlines():
    m = re.search(r" Name: (\S+)", l)
    if m:
        cur = m.group(1); d[cur] = {}; continue
    m = re.search(r"(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill|SGPRs|LDS Size \[bytes/block\]): (\d+)", l)
    if m and cur:
        d[cur][m.group(1)] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(d), capture_output=True, text=True).stdout.splitlines()
for (k, v), n in zip(d.items(), names):
    n = n.split("(")[0]
    if flt in n:
        print(f"{n[:90]:90s} VGPR {v.get('VGPRs'):>3s} AGPR {v.get('AGPRs'):>3s} spill {v.get('VGPRs Spill'):>3s} scratch {v.get('ScratchSize [bytes/lane]'):>4s} waves/SIMD {v.get('Occupancy [waves/SIMD]')}")
