#!/bin/bash
# megakernel vs wavefront, with / without the voted node-loop exit, over scene sizes (one gpurun call)
run() { echo -n "$1 subdiv=$2 sched=$3 nodeMin=$4: "; HYDRA_BENCH_SUBDIV=$2 HPT_NODE_MIN=$4 python bench.py --workload $1 --steps 2 --warmup 1 --spp $5 --no-cpu-baseline --schedule $3 2>&1 | grep -o "\"value\": [0-9.]*" || echo failed; }
for sd in 1 2 3; do for sc in 1 2; do for nm in 0 16; do run interior $sd $sc $nm 64; done; done; done
for sc in 1 2; do for nm in 0 16; do run cornell 4 $sc $nm 128; done; done
