# The fp32 mode's GEMM (k_xgemm, csrc/exact.hip) under counters: python3 scripts/exact_mode_bench.py 64 = logits chunks of 768 rows x 300,000 items x 256.
# usage (GPU box): bash scripts/profile_xgemm.sh   -> gpurun_out/xgemm_pmc.json
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 scripts/exact_mode_bench.py 64"
O=gpurun_out/xgemm_prof; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d $O/p1 -- $B > $O/p1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $O/p2 -- $B > $O/p2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $O/p3 -- $B > $O/p3.log 2>&1
python3 - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for d in ("p1", "p2", "p3"):
    agg = collections.defaultdict(float); n = set(); dur = []
    for f in glob.glob(f"gpurun_out/xgemm_prof/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if not r["Kernel_Name"].startswith("k_xgemm") or int(r["Grid_Size"]) < 256 * 10000: continue   # the logits chunks only
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); n.add(r["Dispatch_Id"])
    for f in glob.glob(f"gpurun_out/xgemm_prof/{d}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith("k_xgemm") and int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) >= 256 * 10000:
                dur.append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    k = max(len(n), 1)
    res[d] = {c: v / k for c, v in agg.items()}
    res[d]["launches"] = len(n); res[d]["avg_us"] = sum(dur) / max(len(dur), 1) / 1e3
p1 = res["p1"]
if p1.get("GRBM_GUI_ACTIVE"):
    p1["mfma_busy"] = p1.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / (p1["GRBM_GUI_ACTIVE"] / 8)
    p1["clock_GHz"] = p1["GRBM_GUI_ACTIVE"] / 8 / (p1["avg_us"] * 1e3)
json.dump(res, open("gpurun_out/xgemm_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
