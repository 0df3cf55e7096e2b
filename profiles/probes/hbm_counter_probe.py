"""Reads the rocprofv3 counter CSV of hbm_counter_probe (gpurun_out/hbmprobe) and prints, per launch in order, FETCH_SIZE (KB -> bytes) against
the known byte count the binary printed (gpurun_out/hbmprobe.log)."""
import csv, glob, json, re, sys
log = [l for l in open("gpurun_out/hbmprobe.log") if l.startswith("KNOWN")]
per = {}
for f in glob.glob("gpurun_out/hbmprobe*/*/*counter_collection.csv"):      # one directory per counter pass, the same launches in the same order
    for r in csv.DictReader(open(f)):
        per.setdefault(int(r["Dispatch_Id"]), {"name": r["Kernel_Name"].split("(")[0]})[r["Counter_Name"]] = float(r["Counter_Value"])
ks = [v for k, v in sorted(per.items()) if v["name"] in ("walk64", "walk16", "stream16")]
out = []
for line, k in zip(log, ks):
    m = re.match(r"KNOWN (\w+) (\w+) rep (\d) bytes (\d+)", line)
    known = float(m.group(4)); lines = known * (4 if m.group(1) == "walk16" else 1)
    fetch = k.get("FETCH_SIZE", 0.0) * 1024.0
    row = {"kernel": m.group(1), "table": m.group(2), "rep": int(m.group(3)), "bytes_used": known, "bytes_of_lines_touched": lines, "FETCH_SIZE_bytes": fetch,
           "FETCH_over_lines": round(fetch / lines, 4), "RDREQ": k.get("TCC_EA0_RDREQ_sum"), "RDREQ_32B": k.get("TCC_EA0_RDREQ_32B_sum"),
           "l2_hit_rate": round(k["TCC_HIT_sum"] / max(k["TCC_HIT_sum"] + k["TCC_MISS_sum"], 1.0), 4) if "TCC_HIT_sum" in k else None}
    out.append(row); print(row)
json.dump(out, open("gpurun_out/hbmprobe_summary.json", "w"), indent=1)
