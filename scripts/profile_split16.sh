# The split-operand scoring kernels (csrc/score_x3.hip, csrc/score_x3w.hip) under rocprofv3: kernel trace + stats, then two PMC passes
# (MFMA-pipe busy / stalls / LDS; instruction mix), then one more for the HBM-side traffic.  Summaries -> gpurun_out/<tag>_*.
# usage: bash scripts/profile_split16.sh [tag] [program ...]
#   default           r4_split16       python3 scripts/split16_bench.py 1024                           (d = 256, 614,400 x 300,000)
#   d = 512 (round 5) r5_split16_d512  python3 scripts/split16_kernel_bench.py 512 1000000 153600 0 3
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r4_split16}
export TAG
[ $# -gt 0 ] && shift
B="${*:-python3 scripts/split16_bench.py 1024}"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp16_trace -- $B > gpurun_out/sp16_trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/sp16_pmc1 -- $B > gpurun_out/sp16_pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY --output-format csv -d gpurun_out/sp16_pmc2 -- $B > gpurun_out/sp16_pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/sp16_pmc3 -- $B > gpurun_out/sp16_pmc3.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json, os
TAG = os.environ["TAG"]
res = {}
for f in glob.glob("gpurun_out/sp16_trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [{"name": r["Name"][:100], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])} for r in rows[:14]]
    open("gpurun_out/%s_kernel_stats.csv" % TAG, "w").write(open(f).read())
for d in ("sp16_pmc1", "sp16_pmc2", "sp16_pmc3"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-44:]
            if "k_score_x3" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        n = len(disp[k])
        e = res.setdefault("pmc_per_launch", {}).setdefault(k, {"launches": n})
        e.update({c: x / n for c, x in v.items()})
for k, v in res.get("pmc_per_launch", {}).items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and v.get("GRBM_GUI_ACTIVE", 0) > 0:
        # MFMA-pipe busy: SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (MI355X_MICROARCH.md)
        v["mfma_pipe_busy"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)
    if v.get("SQ_INSTS_MFMA", 0) > 0:
        v["valu_per_mfma"] = (v["SQ_INSTS_VALU"] - v["SQ_INSTS_MFMA"]) / v["SQ_INSTS_MFMA"]
        v["lds_per_mfma"] = v["SQ_INSTS_LDS"] / v["SQ_INSTS_MFMA"]
    if v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0) > 0:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
        v["fetch_bytes"] = (v["TCC_EA0_RDREQ_sum"] - v["TCC_EA0_RDREQ_32B_sum"]) * 64 + v["TCC_EA0_RDREQ_32B_sum"] * 32
json.dump(res, open("gpurun_out/%s_summary.json" % TAG, "w"), indent=1)
print(json.dumps(res, indent=1)[:6000])
PY
tail -5 gpurun_out/sp16_trace.log
