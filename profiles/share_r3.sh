#!/bin/bash
# Projected strong scaling of the PIXEL split (the default, bit-identical to one GPU): one GPU renders rank 0's interleaved share of a K-rank job at the
# config's spp (no reduce: 33 MB over xGMI is < 1 ms); value = K x its rate = what K GPUs would deliver; value / value(K = 1) = projected speed-up.
for wl in "cornell 1024" "interior 1024"; do set -- $wl; for k in 1 2 4 8; do
  echo -n "$1 @ $2 spp, pixel split, share 1/$k: "; python bench.py --workload $1 --steps 1 --warmup 1 --spp $2 --no-cpu-baseline --no-also --no-build --scaling strong --shard pixels --emulate-share $k 2>&1 | grep -o "\"value\": [0-9.]*" || echo failed
done; done
