#!/bin/bash
# heavy scene on the MEGAKERNEL (what a call below the wavefront's pixel threshold gets): 4-wide tree (default) vs BVH2 (HPT_WIDE_NODES=0)
B="python bench.py --workload interior --steps 2 --warmup 1 --spp 16 --schedule 1 --no-cpu-baseline --no-also --no-build"
for r in 1 2; do echo -n "megakernel wide: "; $B 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; echo -n "megakernel bvh2: "; HPT_WIDE_NODES=0 $B 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
