#!/bin/bash
# the headline frame (16 384 tiles of 64 pixels) with 8 (default) / 7 / 6 / 5 resident blocks per CU: more rounds of the work queue against fewer waves per SIMD
for b in 0 7 6 5; do echo -n "cornell 1024 spp blocks-per-cu $b: "; python bench.py --workload cornell --steps 2 --warmup 1 --spp 1024 --no-cpu-baseline --no-also --no-build --blocks-per-cu $b 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
