#!/bin/bash
# A/B inside ONE gpurun call: two-level (TLAS/BLAS) vs flat world-space BVH on the small scenes (the heavy ones already pick flat).
for r in 1 2; do for l in 1 2; do
  echo -n "cornell layout $l: "; python bench.py --steps 3 --warmup 1 --spp 256 --accel-layout $l --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*"
done; done
for l in 1 2; do echo "== fixtures, layout $l"; LAYOUT=$l python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"; done
