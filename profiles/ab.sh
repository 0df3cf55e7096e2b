#!/bin/bash
# A/B of kernel builds inside ONE gpurun call (box-to-box variance makes cross-call comparisons unreliable):
#   profiles/ab.sh "<suffix1> <suffix2> ..." <workload> <spp> [extra bench args]   (suffix "" = libhydra_hip.so)
WL=${2:-cornell}; SPP=${3:-64}; shift 3
for r in 1 2; do for v in $VARIANTS; do
  [ "$v" = "base" ] && lib=hydracore3_amd/libhydra_hip.so || lib=hydracore3_amd/libhydra_hip_$v.so
  echo -n "$WL $v: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload $WL --steps 3 --warmup 1 --spp $SPP --no-cpu-baseline "$@" 2>&1 | grep -o "\"value\": [0-9.]*"
done; done
