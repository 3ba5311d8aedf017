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
        if any(w in k for w in ("score", "tcn", "sgrad", "pool", "metrics", "gru_scan", "embed")):   # + the HBM-bound sub-kernels (reported as GB/s)
            n = max(len(disp[k]), 1)
            res.setdefault("pmc_per_launch", {}).setdefault(k, {}).update({c: val / n for c, val in v.items()})
            res["pmc_per_launch"][k]["launches"] = n
# durations of the SAME launches the counters were read on: the FETCH_SIZE pass's own kernel trace (the counter passes skip the training / mixture /
# reference-batch legs, so a name's average over the full trace run would mix in launches of other sizes)
dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(out, "pmc_fetch", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"].split("(")[0][-60:]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
avg_ns_of = {k: sum(v) / len(v) for k, v in dur.items() if v}
for k, v in res.get("pmc_per_launch", {}).items():
    if "FETCH_SIZE" in v:   # KB; gfx950: x2 for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM)
        v["fetch_bytes_corrected"] = v["FETCH_SIZE"] * 1024 * 2
    if "WRITE_SIZE" in v:
        v["write_bytes"] = v["WRITE_SIZE"] * 1024
    if "TCC_HIT_sum" in v:
        v["l2_hit_rate"] = v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])
    if "fetch_bytes_corrected" in v and "write_bytes" in v and k in avg_ns_of:
        # memory-side GB/s of the kernel: counter bytes per launch / the same launches' average duration (FETCH_SIZE pass), against ~8 TB/s
        v["avg_ns_counter_run"] = avg_ns_of[k]
        v["memory_side_GBps"] = (v["fetch_bytes_corrected"] + v["write_bytes"]) / avg_ns_of[k]
        v["memory_side_frac_of_8TBps"] = v["memory_side_GBps"] / 8000.0
try:
    res["bench"] = json.loads(open(os.path.join(out, "bench.json")).read().strip().splitlines()[-1])
except Exception as e:  # noqa: BLE001
    res["bench_error"] = str(e)
json.dump(res, open(os.path.join(out, "summary.json"), "w"), indent=1)

# profiles/score_traffic.json (read by bench.py: roofline.traffic of the headline kernel and of the split16 kernel), stamped with the code it was
# measured on: the sha-256 of the kernels' source files and the git head (passed in by the caller: the GPU box has no .git)
import hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sha16(*names):
    h = hashlib.sha256()
    for n in names:
        h.update(open(os.path.join(ROOT, "hiertcn_amd", "csrc", n), "rb").read())
    return h.hexdigest()[:16]


tag = sys.argv[2] if len(sys.argv) > 2 else "r?"
head = sys.argv[3] if len(sys.argv) > 3 else None
bench = res.get("bench") or {}
sections = {"headline": ("k_score_pipe<256, 14, 2, 3, true", ("score.hip", "score_geom.h", "score_lists.h"), (bench.get("roofline") or {}).get("scored_rows_per_launch")),
            "exact": ("k_score_x3<256, 0, 0>", ("score_x3.hip", "score_x3.h"), ((bench.get("exact") or {}).get("rows_per_launch")))}
traffic = {"git_head": head, "tag": tag}
for sec, (needle, sources, rows) in sections.items():
    hit = [(k, v) for k, v in res.get("pmc_per_launch", {}).items() if needle in k and "fetch_bytes_corrected" in v and "write_bytes" in v]
    if not hit or not rows:
        continue
    k, v = max(hit, key=lambda kv: kv[1]["fetch_bytes_corrected"])
    traffic[sec] = {"kernel": k, "rows_per_launch": int(rows), "fetch_bytes_per_launch": v["fetch_bytes_corrected"], "write_bytes_per_launch": v["write_bytes"],
                    "l2_hit_rate": v.get("l2_hit_rate"), "source_sha16": sha16(*sources), "sources": list(sources),
                    "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, scripts/profile_round.sh %s) on `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline "
                              "--no-clock-probe` at git %s; FETCH_SIZE (KB) x 1024 x 2 (gfx950 correction, MI355X_MICROARCH.md HBM)" % (tag, head)}
json.dump(traffic, open(os.path.join(out, "score_traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1)[:1500])
print(json.dumps(res.get("pmc_per_launch", {}), indent=1)[:3000])
print(json.dumps(res.get("kernel_stats", []), indent=0)[:2500])
