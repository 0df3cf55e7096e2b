#!/bin/bash
# tail handling and refill threshold of the trace kernel on the 4-wide tree (interior, 64 spp)
B="python bench.py --workload interior --steps 2 --warmup 1 --spp 64 --no-cpu-baseline --no-also --no-build"
for g in 0 4 8 16 32 64; do echo -n "grace $g: "; HPT_WF_GRACE=$g $B 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
for rb in 48 56 60 62; do echo -n "refill below $rb: "; $B --refill-below $rb 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
for gr in 1 2; do echo -n "pixel groups $gr: "; $B --groups $gr 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
