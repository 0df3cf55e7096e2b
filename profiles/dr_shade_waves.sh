#!/bin/bash
# wavefront DR shade kernel built for 3 / 4 (product) / 5 waves per SIMD: dr_interior at 64 spp
for v in hydra_hip hydra_hip_drw3 hydra_hip_drw5; do
  echo -n "dr_interior 64 spp lib$v: "; HYDRA_HIP_LIB=$PWD/hydracore3_amd/lib$v.so python bench.py --workload dr_interior --spp 64 --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done
