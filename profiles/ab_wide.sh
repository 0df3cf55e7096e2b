#!/bin/bash
# Heavy-scene trace kernel on the 4-wide compressed tree (default) against the BVH2 (HPT_WIDE_NODES=0), same binary, one gpurun call.
SPP=${1:-32}
for r in 1 2; do
  echo -n "interior wide nodes: "; python bench.py --workload interior --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline --no-also --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
  echo -n "interior bvh2:       "; HPT_WIDE_NODES=0 python bench.py --workload interior --steps 2 --warmup 1 --spp $SPP --no-cpu-baseline --no-also --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done
