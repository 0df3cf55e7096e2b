#!/bin/bash
# trace kernel built for 5 (product) / 6 waves per SIMD after the folded node step; trace blocks per CU 5 / 6
for cfg in "hydra_hip 5" "hydra_hip 6" "hydra_hip_wf6 5" "hydra_hip_wf6 6"; do set -- $cfg
  echo -n "interior 64 spp lib$1 trace-blocks-per-cu $2: "; HYDRA_HIP_LIB=$PWD/hydracore3_amd/lib$1.so python bench.py --workload interior --spp 64 --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also --trace-blocks-per-cu $2 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done
