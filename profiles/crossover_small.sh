#!/bin/bash
# the small end of the schedule / vote crossover: the interior generator at 4 850 and 17 090 triangles (one gpurun call)
run() { echo -n "interior subdiv=$1 sched=$2 nodeMin=$3: "; HYDRA_BENCH_SUBDIV=$1 HPT_NODE_MIN=$3 python bench.py --workload interior --steps 2 --warmup 1 --spp 32 --no-cpu-baseline --schedule $2 2>&1 | grep -o "\"value\": [0-9.]*" || echo failed; }
for sd in 0 1; do for sc in 1 2; do for nm in 0 16; do run $sd $sc $nm; done; done; echo -n "interior subdiv=$sd automatic: "; HYDRA_BENCH_SUBDIV=$sd python bench.py --workload interior --steps 2 --warmup 1 --spp 32 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*"; done
