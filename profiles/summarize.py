#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (profiles/collect.sh) into the committed evidence:
   profiles/<tag>_kernel_stats.csv, profiles/<tag>_summary.json and profiles/traffic_<workload>.json (read by bench.py).
HBM bytes follow MI355X_MICROARCH.md (HBM section): FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide streaming reads, so the read side is reported both raw and doubled (upper bound)."""
import csv
import glob as _glob
class glob:                     # gpurun merges every call's files into the same local directory (names carry the process id): always take the newest
    @staticmethod
    def glob(pattern):
        return sorted(_glob.glob(pattern), key=os.path.getmtime, reverse=True)
import json
import os
import shutil
import sys

tag, workload = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", tag)


CALLS = 3     # profiles/collect.sh runs bench.py --steps 2 --warmup 1: three PathTraceBlock calls


def product_kernel(name):
    """Kernels of one PathTraceBlock call: the persistent megakernel, or the wavefront schedule's shade / trace / init kernels.
    The instrumented build (first template argument true / third for the trace kernel) is bench.py's counting probe, never timed."""
    if workload == "spectral_interior":           # the wavefront schedule under spectral mode: its own shade kernel in front of the shared trace kernel
        return "wfTraceKernel" in name or "wfShadeSpecKernel" in name or "wfInitKernel" in name
    if workload == "spectral":
        return "SpectralKernel" in name                 # pathTraceSpectralKernel, or pathTraceBlockSpectralKernel under the block-local schedule
    if workload == "film":
        return "pathTraceKernel<false, false, 4" in name
    if workload.startswith("dr"):                 # bench.py renders the target image with the forward kernel first: not part of a PathTraceDR call
        return "pathTraceKernel<false, true" in name or "pathTraceBlockKernel<true" in name or ("wfShadeKernel<true" in name) or (workload == "dr_interior" and ("wfTraceKernel" in name or "wfInitKernel" in name))
    if "pathTraceBlockKernel" in name:
        return "pathTraceBlockKernel<false" in name
    if "pathTraceKernel" in name:
        return "<true" not in name
    return "wfTraceKernel" in name or "wfShadeKernel" in name or "wfInitKernel" in name


def trace_ms(sub):
    """Sum of the product kernels' durations (ms) in the kernel trace of the SAME run the counters of `sub` come from: the time base of
    every PMC-derived rate. (bench.py's HIP-event time of a call run under --pmc also holds the profiler's per-dispatch serialisation and
    counter read-out - for the wavefront schedule's thousands of short kernels that inflated the time and read as an 'effective clock' of
    1.8 GHz in round 2; per kernel, GRBM_GUI_ACTIVE / 8 / duration gives 2.35 - 2.5 GHz.)"""
    f = glob.glob(os.path.join(src, sub, "*", "*kernel_trace.csv"))
    if not f:
        return None
    tot = 0.0
    for r in csv.DictReader(open(f[0])):
        if product_kernel(r["Kernel_Name"]):
            tot += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    return tot


def pmc(sub):
    f = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if product_kernel(r["Kernel_Name"]):
            per[r["Counter_Name"]] = per.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return {k: v / CALLS for k, v in per.items()}     # mean per PathTraceBlock call


def fetch_calibrated_kb(sub):
    """FETCH_SIZE per call under the rule CALIBRATED for this code's access patterns (profiles/probes/hbm_counter_probe.hip, r3_hbm_counter_probe.json):
    divergent 64-byte record fetches - BVH nodes, triangle and shading records: what the trace kernels and the megakernels read - are counted
    at their exact bytes (FETCH / lines touched = 1.00, also when the Infinity Cache serves them); coalesced 16-byte-per-lane streams - the path
    pool the wavefront SHADE kernel reads - at half. So: trace kernels and megakernels x 1, shade / init kernels x 2 (an upper bound for them:
    their material, texture and shading-record fetches are divergent too)."""
    f = glob.glob(os.path.join(src, sub, "*", "*counter_collection.csv"))[0]
    tot = 0.0
    for r in csv.DictReader(open(f)):
        if product_kernel(r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE":
            stream = ("wfShadeKernel" in r["Kernel_Name"]) or ("wfShadeSpecKernel" in r["Kernel_Name"]) or ("wfInitKernel" in r["Kernel_Name"])
            tot += float(r["Counter_Value"]) * (2.0 if stream else 1.0)
    return tot / CALLS


stats = glob.glob(os.path.join(src, "stats", "*", "*kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(root, "profiles", f"{tag}_kernel_stats.csv"))
kavg, per_kernel = 0.0, {}
for r in csv.DictReader(open(stats)):
    if product_kernel(r["Name"]):
        kavg += float(r["TotalDurationNs"]) * 1e-6 / CALLS
        per_kernel[r["Name"].split("(")[0]] = {"calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) * 1e-6, "avg_us": float(r["AverageNs"]) * 1e-3}
f, w = pmc("pmc_fetch"), pmc("pmc_write")
fetch_kb, write_kb = f.get("FETCH_SIZE", 0.0), w.get("WRITE_SIZE", 0.0)
bench = json.loads(open(os.path.join(src, "bench_line.json")).read())
out = {"tag": tag, "workload": workload, "bench": bench, "rocprof_kernel_ms_per_call": kavg, "rocprof_kernels": per_kernel,
       "FETCH_SIZE_KB_per_launch": fetch_kb, "WRITE_SIZE_KB_per_launch": write_kb,
       "hbm_bytes_per_launch": fetch_calibrated_kb("pmc_fetch") * 1024 + write_kb * 1024,       # the calibrated rule (fetch_calibrated_kb)
       "hbm_bytes_per_launch_doubled_fetch": fetch_kb * 2 * 1024 + write_kb * 1024,             # the guide's streaming rule applied to everything (round 2's figure)
       "hbm_bytes_per_launch_raw_fetch": fetch_kb * 1024 + write_kb * 1024,
       "TCC_HIT_sum": w.get("TCC_HIT_sum"), "TCC_MISS_sum": w.get("TCC_MISS_sum")}
import subprocess
commit = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or "unknown"
json.dump({"hbm_bytes_per_launch": out["hbm_bytes_per_launch"], "hbm_bytes_per_launch_doubled_fetch": out["hbm_bytes_per_launch_doubled_fetch"],
           "hbm_bytes_per_launch_raw_fetch": out["hbm_bytes_per_launch_raw_fetch"], "rule": "FETCH_SIZE x 1 for divergent 64-B record fetches (trace kernels, megakernels), x 2 for the wavefront shade kernel's coalesced streams, + WRITE_SIZE (profiles/r3_hbm_counter_probe.json)",
           "from": f"profiles/{tag}_summary.json", "commit": commit,
           "paths_per_launch": bench["config"]["paths_per_step"]}, open(os.path.join(root, "profiles", f"traffic_{workload}.json"), "w"))
# SQ passes (profiles/collect.sh step 4): one PathTraceBlock call at fewer passes; summed over the product kernels of that call
if glob.glob(os.path.join(src, "pmc_sq1", "*", "*counter_collection.csv")):
    CALLS = 1
    sq = dict(pmc("pmc_sq1")); sq.update(pmc("pmc_sq2"))
    bsq = json.loads(open(os.path.join(src, "bench_line_sq.json")).read())
    paths = float(bsq["config"]["paths_per_step"])
    k_ms_events = float(bsq["roofline"]["kernel_ms"])
    k_ms = trace_ms("pmc_sq1") or k_ms_events                          # kernel durations of the profiled run itself (see trace_ms)
    k_ms2 = trace_ms("pmc_sq2") or k_ms
    lane = sq["SQ_THREAD_CYCLES_VALU"] / (64.0 * sq["SQ_ACTIVE_INST_VALU"])
    clk_ghz = sq.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 / (k_ms2 * 1e6)      # effective shader clock (MI355X_MICROARCH.md, DVFS): sum over 8 XCDs, over the kernels' own durations
    pj = {"workload": workload, "commit": commit, "from": f"gpurun_out/{tag}/pmc_sq1, pmc_sq2 (profiles/collect.sh)", "paths": paths, "kernel_ms": k_ms, "kernel_ms_hip_events_under_pmc": k_ms_events,
          "counters": sq, "valu_insts_per_path": sq["SQ_INSTS_VALU"] / paths, "lane_utilisation": round(lane, 4),
          # issue-slot occupancy at the spec rate (2 cycles per wave64 instruction: MI355X_MICROARCH.md) and at the 4 cycles VERDICT r1 priced it with
          "valu_issue_frac_2cyc_at_2p4GHz": round(2.0 * sq["SQ_INSTS_VALU"] / (1024 * 2.4e9 * k_ms * 1e-3), 4),
          "valu_busy_4cyc_at_2p4GHz": round(4.0 * sq["SQ_INSTS_VALU"] / (1024 * 2.4e9 * k_ms * 1e-3), 4),
          "cycles_per_valu_inst_per_simd_at_2p4GHz": round(1024 * 2.4e9 * k_ms * 1e-3 / sq["SQ_INSTS_VALU"], 3),
          "effective_clock_GHz": round(clk_ghz, 3) if clk_ghz else None,
          "wait_any_frac": round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 4), "active_inst_frac": round(sq["SQ_ACTIVE_INST_ANY"] / sq["SQ_WAVE_CYCLES"], 4)}
    json.dump(pj, open(os.path.join(root, "profiles", f"pmc_{workload}.json"), "w"), indent=1)
    out["pmc"] = pj
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
