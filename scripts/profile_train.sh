# training-step profile only (the k_sgrad passes of scripts/profile_misc.sh): kernel stats + three PMC passes.  usage: bash scripts/profile_train.sh <tag>
TAG=${1:-r3}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_train_$TAG
mkdir -p $OUT
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/train -- python3 scripts/train_bench.py 1024 3 > $OUT/train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d $OUT/train_pmc_fetch -- python3 scripts/train_bench.py 1024 3 > $OUT/train_pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/train_pmc_tcc -- python3 scripts/train_bench.py 1024 3 > $OUT/train_pmc_tcc.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/train_pmc_sq -- python3 scripts/train_bench.py 1024 3 > $OUT/train_pmc_sq.log 2>&1
tail -1 $OUT/train.log | cut -c1-200; f=$(find $OUT/train -name '*kernel_stats.csv' | head -1); head -6 "$f" | cut -c1-160
python3 - <<PY
import csv, glob, collections, json
out = "$OUT"
res = {}
for d in ("train_pmc_fetch", "train_pmc_tcc", "train_pmc_sq"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for f in glob.glob(out + "/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-50:]
            if "sgrad<" not in k: continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        n = max(len(disp[k]), 1)
        res.setdefault(k, {}).update({c: val / n for c, val in v.items()})
for k, v in res.items():
    if "FETCH_SIZE" in v: v["fetch_bytes_corrected"] = v["FETCH_SIZE"] * 1024 * 2
    if "TCC_HIT_sum" in v: v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v: v["mfma_busy"] = v["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)
json.dump(res, open(out + "/sgrad_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1)[:3000])
PY
