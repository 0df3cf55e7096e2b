#!/bin/bash
# projected strong scaling: one GPU renders rank 0's interleaved share of a K-rank job (no reduce); value / value(K=1) / K = efficiency bound
for wl in "cornell 256" "interior 32"; do set -- $wl; for k in 1 2 4 8; do
  echo -n "$1 share 1/$k: "; python bench.py --workload $1 --steps 2 --warmup 1 --spp $2 --no-cpu-baseline --no-also --no-build --scaling strong --emulate-share $k 2>&1 | grep -o "\"value\": [0-9.]*" || echo failed
done; done
