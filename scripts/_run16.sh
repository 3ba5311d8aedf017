cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
export DTYPE=split16
mkdir -p gpurun_out/prof_s16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s16/trace -- python3 scripts/cfg2_bench.py 32 100 > gpurun_out/prof_s16/run.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/prof_s16/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:16]:
    print("%-100s calls %5d avg %8.1f us" % (r["Name"][:100], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
PY
