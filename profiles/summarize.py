#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (profiles/collect.sh) into the committed evidence:
   profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.json and profiles/traffic_<workload>.json (read by bench.py).
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide streaming reads, so the read side is reported both raw and doubled (upper bound)."""
import csv
import glob
import json
import os
import shutil
import sys

tag, workload = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)


def pmc(sub):
    f = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if "pathTraceKernel" in r["Kernel_Name"] and "<true" not in r["Kernel_Name"]:
            per.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
            per[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in per.items()}     # mean per launch


stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
kavg = None
for r in csv.DictReader(open(stats)):
    if "pathTraceKernel" in r["Name"] and "<true" not in r["Name"]:
        kavg = float(r["AverageNs"]) * 1e-6
f, w = pmc("pmc_fetch"), pmc("pmc_write")
fetch_kb, write_kb = f.get("FETCH_SIZE", 0.0), w.get("WRITE_SIZE", 0.0)
bench = json.loads(open(os.path.join(src, "bench_line.json")).read())
out = {"tag": tag, "workload": workload, "bench": bench, "rocprof_kernel_avg_ms": kavg,
       "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "hbm_bytes_per_launch": fetch_kb * 2 * 1024 + write_kb * 1024,
       "hbm_bytes_per_launch_raw_fetch": fetch_kb * 1024 + write_kb * 1024,
       "TCC_HIT_sum": w.get("TCC_HIT_sum"), "TCC_MISS_sum": w.get("TCC_MISS_sum")}
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_summary.json"), "w"), indent=1)
json.dump({"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "from": f"profiles/{tag}_summary.json",
           "paths_per_launch": bench["config"]["paths_per_step"]}, open(os.path.join(root, "profiles", f"traffic_{workload}.json"), "w"))
print(json.dumps(out, indent=1))
