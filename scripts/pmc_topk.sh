# PMC readings of the ranked top-k flavour beside the plain loss / rank kernel (scripts/topk_bench.py launches both)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="python3 scripts/topk_bench.py"
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmc_tk1 -- $B > gpurun_out/pmc_tk1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_tk2 -- $B > gpurun_out/pmc_tk2.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_tk1", "pmc_tk2"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"][:48]
            if "k_score_pipe" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        n = len(disp[k])
        print(d, k, "launches", n, {c: f"{x / n:.4g}" for c, x in v.items()})
PY
