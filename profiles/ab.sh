#!/bin/bash
# A/B of kernel builds inside ONE gpurun call (box-to-box variance makes cross-call comparisons unreliable):
#   profiles/ab.sh "<variant1> <variant2> ..." <workload> <spp> [extra bench args]
# variant "base" = hydracore3_amd/libhydra_hip.so, anything else = hydracore3_amd/libhydra_hip_<variant>.so
# (built beforehand by `python __graft_entry__.py variants <name>...` or `python profiles/mkvariant.py <name> -D...`).
VARIANTS=${1:-base}; WL=${2:-cornell}; SPP=${3:-64}
shift; shift; shift
for r in 1 2; do for v in $VARIANTS; do
  [ "$v" = "base" ] && lib=hydracore3_amd/libhydra_hip.so || lib=hydracore3_amd/libhydra_hip_$v.so
  echo -n "$WL $v: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload $WL --steps 3 --warmup 1 --spp $SPP --no-cpu-baseline --no-also --no-build "$@" 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done; done
