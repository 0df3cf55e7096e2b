#!/bin/bash
# Round-3 evidence on the final kernels, one gpurun call: rocprofv3 --kernel-trace --stats + the PMC passes of profiles/collect.sh per workload.
# (run from the repo root on the GPU box:  profiles/r3_collect_all.sh)
set -o pipefail
SPECS=("r3c cornell 1024 64" "r3i interior 64 16" "r3d dr 256 64" "r3di dr_interior 64 16" "r3s spectral 64 64" "r3f film 64 64" "r3si spectral_interior 64 16")
[ "$1" = "a" ] && SPECS=("${SPECS[@]:0:3}")
[ "$1" = "b" ] && SPECS=("${SPECS[@]:3:3}")
[ "$1" = "c" ] && SPECS=("${SPECS[@]:6:1}")
for spec in "${SPECS[@]}"; do
  set -- $spec
  echo "== $1 $2 $(date +%T)" | tee -a gpurun_out/r3_collect.log
  timeout -k 10 900 profiles/collect.sh $1 $2 $3 $4 >> gpurun_out/r3_collect.log 2>&1 || { echo "collect $1 failed" | tee -a gpurun_out/r3_collect.log; }
done
echo "== done $(date +%T)" | tee -a gpurun_out/r3_collect.log
