#!/bin/bash
# Knobs of the trace kernel on the 4-wide tree (interior, 32 spp, one gpurun call): voted node-loop exit, trace waves per SIMD, refill threshold
B="python bench.py --workload interior --steps 2 --warmup 1 --spp 32 --no-cpu-baseline --no-also --no-build"
for nm in 0 8 16 24 32 48; do echo -n "node_min $nm: "; HPT_NODE_MIN=$nm $B 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
for w in 3 4 5 6; do echo -n "trace blocks per CU $w: "; $B --trace-blocks-per-cu $w 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
for rb in 40 48 56 60; do echo -n "refill below $rb: "; $B --refill-below $rb 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
