#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on a GPU box (run through gpurun from the repo root):
#   profiles/collect.sh <tag> <workload> <spp>
# 0. build OUTSIDE the profiler (a compiler child under the profiler's preload would be a GPU-initialised process that execs)
# 1. --kernel-trace --stats of the bench command             -> gpurun_out/<tag>/stats
# 2. PMC pass FETCH_SIZE (3 of 4 TCC slots)                  -> gpurun_out/<tag>/pmc_fetch
# 3. PMC pass WRITE_SIZE + TCC_HIT_sum/TCC_MISS_sum          -> gpurun_out/<tag>/pmc_write
# 4. SQ passes: VALU instructions, lane utilisation, waits   -> gpurun_out/<tag>/pmc_sq1, pmc_sq2
# (counters are collected in their own runs, never together with a trace domain other than --kernel-trace)
set -o pipefail
TAG=${1:-r2}; WL=${2:-cornell}; SPP=${3:-1024}; SQSPP=${4:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/$TAG
rm -rf $OUT && mkdir -p $OUT        # a re-collection under the same tag must not mix with the previous one
python3 __graft_entry__.py > $OUT/build.log 2>&1 || { echo "build failed"; exit 1; }
CMD="python3 bench.py --workload $WL --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline --no-also --no-build"
SQ="python3 bench.py --workload $WL --steps 1 --warmup 0 --spp $SQSPP --no-cpu-baseline --no-also --no-build"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $CMD > $OUT/stats.log 2>&1 && \
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 && \
rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 && \
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d $OUT/pmc_sq1 -- $SQ > $OUT/pmc_sq1.log 2>&1 && \
timeout -k 5 300 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_sq2 -- $SQ > $OUT/pmc_sq2.log 2>&1
grep -h '"metric"' $OUT/stats.log | tail -1 > $OUT/bench_line.json
grep -h '"metric"' $OUT/pmc_sq1.log | tail -1 > $OUT/bench_line_sq.json
echo "done $TAG"
