# Kernel trace of evaluation at the reference's own batch (cfg-2: HierTCN d = 256, 32 lanes x 10 x 20, N = 20,778): 210 + 210 batches.
# usage (on the GPU box): bash scripts/profile_cfg2.sh      -> gpurun_out/prof_cfg2/kernel_stats.csv, summary.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_cfg2
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 scripts/cfg2_bench.py 32 200 > $OUT/run.log 2>&1
grep cfg2 $OUT/run.log
cp $(find $OUT/trace -name '*kernel_stats.csv' | head -1) $OUT/kernel_stats.csv
python3 - <<'PY' > gpurun_out/prof_cfg2/summary.txt
import csv
rows = list(csv.DictReader(open("gpurun_out/prof_cfg2/kernel_stats.csv")))
print("== cfg-2 evaluation, 210 direct + 210 graph-replayed batches")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:20]:
    print("%-110s calls %5d  avg %9.1f us  %5.1f %%" % (r["Name"][:110], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("kernels: %d distinct, %d launches, %.2f ms of kernel time in total = %.1f us per batch" % (len(rows), sum(int(r["Calls"]) for r in rows), tot / 1e6, tot / 420e3))
PY
cat gpurun_out/prof_cfg2/summary.txt
