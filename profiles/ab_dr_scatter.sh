#!/bin/bash
# Diagnostic A/B inside ONE gpurun call: PathTraceDR (config 4 class) with the gradient scatter and without it (the diagnostic build writes no
# usable gradient). Build the variant first: python __graft_entry__.py variants noat
for r in 1 2; do for v in base noat; do
  [ "$v" = "base" ] && lib=hydracore3_amd/libhydra_hip.so || lib=hydracore3_amd/libhydra_hip_$v.so
  echo -n "dr $v: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload dr --steps 2 --warmup 1 --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done; done
