#!/bin/bash
# Instruction-cache and L1 (TCP) counters of the megakernel on the Cornell box (each pass its own run, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-pmci}
CMD="python3 bench.py --workload cornell --steps 1 --warmup 0 --spp 64 --no-cpu-baseline --no-also --no-build"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ SQ_INSTS_BRANCH SQ_WAVE_CYCLES --output-format csv -d gpurun_out/${TAG}_a -- $CMD > gpurun_out/${TAG}_a.log 2>&1 && \
timeout -k 5 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d gpurun_out/${TAG}_b -- $CMD > gpurun_out/${TAG}_b.log 2>&1 && \
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_SCA --output-format csv -d gpurun_out/${TAG}_c -- $CMD > gpurun_out/${TAG}_c.log 2>&1
for p in a b c; do f=$(ls gpurun_out/${TAG}_$p/*/*counter_collection.csv | head -1); echo "== $p"; python3 - "$f" <<'PY'
import csv, sys
per = {}
for r in csv.DictReader(open(sys.argv[1])):
    if "pathTraceKernel" in r["Kernel_Name"] and "<true" not in r["Kernel_Name"]:
        per[r["Counter_Name"]] = per.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
for k, v in sorted(per.items()): print(f"{k:40s} {v:.4e}")
PY
done
