#!/bin/bash
# Diagnostic A/B inside ONE gpurun call: PathTraceDR (config 4 class) with the whole gradient scatter, without the camera-visible vertex's
# part of it, and without any of it (the two diagnostic builds write no usable gradient).
for r in 1 2; do for v in base skip1 noat; do
  [ "$v" = "base" ] && lib=hydracore3_amd/libhydra_hip.so || lib=hydracore3_amd/libhydra_hip_$v.so
  echo -n "dr $v: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload dr --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*"
done; done
