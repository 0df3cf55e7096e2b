#!/bin/bash
# schedule A/B inside ONE gpurun call: profiles/ab_sched.sh <workload> <spp> "<arg set 1>" "<arg set 2>" ...
WL=$1; SPP=$2; shift 2
for r in 1 2; do for a in "$@"; do
  echo -n "$WL [$a]: "; python bench.py --workload $WL --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline $a 2>&1 | grep -o "\"value\": [0-9.]*" || echo failed
done; done
