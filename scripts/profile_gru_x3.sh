# PMC passes over the HierGRU evaluation step, k_gru_x3 rows only.  usage (GPU box): bash scripts/profile_gru_x3.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_gx_pmc
mkdir -p $OUT
export LOW=gru EVAL=1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 scripts/train_bench.py 1024 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAIT_ANY --output-format csv -d $OUT/lds -- python3 scripts/train_bench.py 1024 1 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/tcc -- python3 scripts/train_bench.py 1024 1 > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("sq", "lds", "tcc"):
    agg = collections.defaultdict(float); disp = set()
    for f in glob.glob("$OUT/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gru_x3" not in r["Kernel_Name"]: continue
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
    n = max(len(disp), 1)
    print(d, "dispatches", n, {k: round(v / n) for k, v in agg.items()})
PY
