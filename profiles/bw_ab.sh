#!/bin/bash
# --workload dr under the block-local schedule, product build vs variants (HYDRA_HIP_LIB); arguments: variant tags
for v in base "$@"; do
  [ "$v" = "base" ] && lib=hydracore3_amd/libhydra_hip.so || lib=hydracore3_amd/libhydra_hip_$v.so
  for bpc in 0 3; do
    echo -n "dr schedule 3 $v blocks-per-cu $bpc: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload dr --schedule 3 --blocks-per-cu $bpc --steps 2 --warmup 1 --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
  done
done
