"""Condense one scripts/profile_round.sh output directory into a small JSON + CSV for profiles/."""
import collections
import csv
import glob
import json
import os
import sys

out = sys.argv[1]
res = {}
# kernel stats
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    res["kernel_stats"] = [{"name": r["Name"][:90], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "pct": float(r["Percentage"])}
                           for r in rows[:12]]
# PMC: per kernel, per counter: sum over dispatches / number of dispatches
for d in ("pmc_fetch", "pmc_write", "pmc_sq"):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for f in glob.glob(os.path.join(out, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-60:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    for k, v in agg.items():
        if "score" in k or "tcn" in k or "sgrad" in k:
            n = max(len(disp[k]), 1)
            res.setdefault("pmc_per_launch", {}).setdefault(k, {}).update({c: val / n for c, val in v.items()})
            res["pmc_per_launch"][k]["launches"] = n
for k, v in res.get("pmc_per_launch", {}).items():
    if "FETCH_SIZE" in v:   # KB; gfx950: x2 for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM)
        v["fetch_bytes_corrected"] = v["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in v:
        v["write_bytes"] = v["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in v:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
try:
    res["bench"] = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    res["bench_error"] = str(e)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(res.get("pmc_per_launch", {}), indent=1)[:3000])
print(json.dumps(res.get("kernel_stats", []), indent=0)[:2500])
