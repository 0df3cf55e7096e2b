#!/bin/bash
# the lean forward shade kernel built for 4 / 5 (product) / 6 waves per SIMD, after its pool traffic went non-temporal
for lib in hydra_hip hydra_hip_wfs4 hydra_hip_wfs6 hydra_hip; do
  echo -n "interior 64 spp lib$lib: "; HYDRA_HIP_LIB=$PWD/hydracore3_amd/lib$lib.so python bench.py --workload interior --spp 64 --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done
