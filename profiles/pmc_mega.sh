#!/bin/bash
# SQ counters of the persistent megakernel on the Cornell box (each pass its own run, --kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CMD="python3 bench.py --workload cornell --steps 1 --warmup 0 --spp 64 --no-cpu-baseline"
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/pmcm1 -- $CMD > gpurun_out/pmcm1.log 2>&1 && \
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_FLAT --output-format csv -d gpurun_out/pmcm2 -- $CMD > gpurun_out/pmcm2.log 2>&1 && \
timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d gpurun_out/pmcm3 -- $CMD > gpurun_out/pmcm3.log 2>&1
echo done
