#!/bin/bash
# env-var A/B inside ONE gpurun call: profiles/ab_env.sh <workload> <spp> "<bench args>" VAR v1 v2 ...
WL=$1; SPP=$2; ARGS=$3; VAR=$4; shift 4
for r in 1 2; do for v in "$@"; do
  echo -n "$WL $VAR=$v: "; env $VAR=$v python bench.py --workload $WL --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline $ARGS 2>&1 | grep -o "\"value\": [0-9.]*" || echo failed
done; done
