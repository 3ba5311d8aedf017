# Kernel trace of training + evaluation at the reference's batch (32 lanes x 10 slots x 20 steps, N = 20,778): 31 training steps + 30 evaluation batches.
# usage (on the GPU box): LOW=tcn|gru D=256 bash scripts/profile_refbatch_train.sh   -> gpurun_out/prof_refbatch/${LOW}_d${D}_kernel_stats.csv + _summary.txt
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_refbatch
mkdir -p $OUT
export LOW=${LOW:-tcn} D=${D:-256} N=20778 EVAL=1 S=10 L=20
T=${LOW}_d${D}
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$T -- python3 scripts/train_bench.py 32 30 ragged > $OUT/$T.log 2>&1
grep "step" $OUT/$T.log | cut -c1-120
cp $(find $OUT/trace_$T -name '*kernel_stats.csv' | head -1) $OUT/${T}_kernel_stats.csv
python3 - <<PY > $OUT/${T}_summary.txt
import csv
rows = list(csv.DictReader(open("$OUT/${T}_kernel_stats.csv")))
print("== $LOW d = $D: 31 training steps + 30 evaluation batches (calls / 31 ~ launches per step)")
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-100s calls %5d  avg %9.1f us  %5.1f %%" % (r["Name"][:100], int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
print("kernels: %d distinct, %d launches, %.2f ms of kernel time in total" % (len(rows), sum(int(r["Calls"]) for r in rows), tot / 1e6))
PY
cat $OUT/${T}_summary.txt
