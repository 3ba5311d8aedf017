"""Is the scoring kernel bound by instruction issue or by the chip's power management?

The SAME kernel binary is launched back to back on (a) random-normal operands and (b) all-zero hidden rows (the MFMA B operand: every
product is zero, the instruction stream, the LDS / L2 traffic and the epilogue are unchanged), for a few seconds each, while a thread
samples `rocm-smi --showpower --showclocks` (sysfs reads, no HIP).  Prints one line per mode: ms per launch, TFLOP/s, median socket
power and median sclk.  usage: python scripts/power_probe.py [rows] [N] [seconds]"""
import os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from hiertcn_amd.args import hier_tcn_args
from hiertcn_amd.engine import Engine
from hiertcn_amd.params import init_params

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 2457600
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300000
secs = float(sys.argv[3]) if len(sys.argv) > 3 else 6.0
d = 256


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = re.search(r"Power \(W\):\s*([0-9.]+)", r)
            ck = re.search(r"sclk clock level:?\s*\d*:?\s*\(?([0-9.]+)\s*Mhz", r, re.I)
            out.append((float(pw.group(1)) if pw else float("nan"), float(ck.group(1)) if ck else float("nan")))
            if not out[-1][0] == out[-1][0] and len(out) == 1:
                print("rocm-smi output (no power field matched):\n" + r, flush=True)
        except Exception as e:   # a sampler failure must not take the timing down
            out.append((float("nan"), float("nan")))
            print("sampler:", e, flush=True)
        time.sleep(0.2)


args = hier_tcn_args(item_num=N, emb_dim=d, hidden_dim=d, tcn_channel=[d, d])
eng = Engine(args)
eng.set_weights(init_params(args, 2))
y = torch.randint(1, N, (rows,), device="cuda", dtype=torch.int32)
for mode, scale in (("random", 0.5), ("zero hidden rows", 0.0), ("random", 0.5), ("relu(random)", -1.0)):
    hid = torch.randn(rows, d, device="cuda")
    hid = (torch.relu(hid) if scale < 0 else hid * scale).to(eng.act_dtype)
    eng.score_fused(hid, y)
    torch.cuda.synchronize()
    stop, smp = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, smp))
    th.start()
    eng.timing_enable(True); eng.timing_read(True)
    t0, n = time.time(), 0
    while time.time() - t0 < secs:
        eng.score_fused(hid, y)
        torch.cuda.synchronize()
        n += 1
    ms, _ = eng.timing_read(True)
    stop.set(); th.join()
    smp = smp[1:-1] or smp   # first / last sample may straddle the loop's ends
    pw = float(np.nanmedian([s[0] for s in smp])) if smp else float("nan")
    ck = float(np.nanmedian([s[1] for s in smp])) if smp else float("nan")
    print(f"{mode:18s} launches={n} ms={ms:.2f} TFLOP/s={2.0 * rows * d * N / (ms * 1e-3) / 1e12:.1f} power_W={pw:.0f} sclk_MHz={ck:.0f} samples={len(smp)}", flush=True)
eng.close()
