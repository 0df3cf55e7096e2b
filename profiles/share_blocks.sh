#!/bin/bash
# small shares of the Cornell frame (multi-GPU pixel split) with fewer resident waves, so that the work queue has a second round to balance with
for k in 2 4; do for b in 0 6 5 4 3 2; do
  echo -n "cornell share 1/$k blocks-per-cu $b: "; python bench.py --workload cornell --steps 1 --warmup 1 --spp 1024 --no-cpu-baseline --no-also --no-build --scaling strong --shard pixels --emulate-share $k --blocks-per-cu $b 2>&1 | grep -o "\"value\": [0-9.]*" || echo failed
done; done
