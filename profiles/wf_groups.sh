#!/bin/bash
# two pixel groups on two streams (one group's shade pass beside the other's trace pass) with smaller trace grids, after the folded node step
run() { python bench.py --workload interior --spp 128 --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also "$@" 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; }
echo -n "groups 1 trace-blocks 5: "; run
for g in 2 3; do for tb in 2 3 4 5; do echo -n "groups $g trace-blocks $tb: "; run --groups $g --trace-blocks-per-cu $tb; done; done
