# Kernel trace of the reference's default model (HierGRU, d = 128, batch 32 x 10 x 20, N = 20,778): 30 training steps + 30 evaluation batches.
# usage (on the GPU box): bash scripts/profile_hiergru.sh      -> gpurun_out/prof_hiergru/{d128,d256}_kernel_stats.csv, summary.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_hiergru
mkdir -p $OUT
export LOW=gru N=20778 EVAL=1
for D in 128 256; do
  export D
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_d$D -- python3 scripts/train_bench.py 32 30 ragged > $OUT/d$D.log 2>&1
  tail -2 $OUT/d$D.log
  cp $(find $OUT/trace_d$D -name '*kernel_stats.csv' | head -1) $OUT/d${D}_kernel_stats.csv
done
python3 - <<'PY' > gpurun_out/prof_hiergru/summary.txt
import csv
for D in (128, 256):
    rows = list(csv.DictReader(open("gpurun_out/prof_hiergru/d%d_kernel_stats.csv" % D)))
    print("== HierGRU d = %d: 31 training steps + 30 evaluation batches (calls / 30 ~ launches per step or batch)" % D)
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:24]:
        print("%-110s calls %5d  avg %9.1f us  %5.1f %%" % (r["Name"][:110], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
    print("kernels: %d distinct, %d launches, %.2f ms of kernel time in total" % (len(rows), sum(int(r["Calls"]) for r in rows), tot / 1e6))
PY
cat gpurun_out/prof_hiergru/summary.txt
