#!/bin/bash
# one round of resident waves against a work queue with something left to hand out: PathTraceDR at 512^2, the C1-sized Cornell frame, an eighth of the 1024^2 frame
for b in 0 3 2; do echo -n "dr blocks-per-cu $b: "; python bench.py --workload dr --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-build --blocks-per-cu $b 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
for b in 0 6 3 2; do echo -n "cornell 512x512 blocks-per-cu $b: "; python bench.py --workload cornell --width 512 --height 512 --spp 256 --steps 2 --warmup 1 --no-cpu-baseline --no-also --no-build --blocks-per-cu $b 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
for b in 0 1; do echo -n "cornell share 1/8 blocks-per-cu $b: "; python bench.py --workload cornell --steps 1 --warmup 1 --spp 1024 --no-cpu-baseline --no-also --no-build --scaling strong --shard pixels --emulate-share 8 --blocks-per-cu $b 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
