#!/bin/bash
# --workload dr (test_228 class, PathTraceDR + Adam): schedule and node-loop vote
B="python bench.py --workload dr --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-build"
echo -n "megakernel (default): "; $B 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
for nm in 8 16 24; do echo -n "megakernel, node_min $nm: "; HPT_NODE_MIN=$nm $B 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
echo -n "wavefront: "; $B --schedule 2 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
for nm in 16 32; do echo -n "wavefront, node_min $nm: "; HPT_NODE_MIN=$nm $B --schedule 2 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
echo -n "megakernel, 3 blocks per CU: "; $B --blocks-per-cu 3 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
echo -n "megakernel, two-level: "; $B --accel-layout 1 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
