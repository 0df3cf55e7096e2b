#!/bin/bash
# trace-pass knobs re-swept after the folded node step: grace trips before suspension, refill threshold, vote threshold of the 4-wide node loop
run() { python bench.py --workload interior --spp 128 --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also "$@" 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; }
for g in 0 4 8 16 32 64; do echo -n "grace $g: "; HPT_WF_GRACE=$g run; done
for r in 32 40 48 56 60 64; do echo -n "refill-below $r: "; run --refill-below $r; done
for n in 0 8 16 24 32 40 48; do echo -n "node-min $n: "; HPT_NODE_MIN=$n run; done
