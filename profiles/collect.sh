#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on a GPU box (run through gpurun from the repo root):
#   profiles/collect.sh <tag> <workload> <spp>
# 1. --kernel-trace --stats of the exact bench command  -> gpurun_out/<tag>/stats
# 2. PMC pass FETCH_SIZE (3 of 4 TCC slots)             -> gpurun_out/<tag>/pmc_fetch
# 3. PMC pass WRITE_SIZE + TCC_HIT_sum/TCC_MISS_sum     -> gpurun_out/<tag>/pmc_write
# (counters are collected in their own runs, never together with a trace domain other than --kernel-trace)
set -o pipefail
TAG=${1:-r1}; WL=${2:-cornell}; SPP=${3:-1024}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
mkdir -p $OUT
CMD="python3 bench.py --workload $WL --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_line.json
echo "done $TAG"
