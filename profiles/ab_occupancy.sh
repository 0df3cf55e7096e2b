#!/bin/bash
# Waves per SIMD of the lean megakernel on the Cornell box: libhydra_hip_mw{3,5,6}.so = pt1 built with -DHPT_MIN_WAVES=N
# (python profiles/mkvariant.py mwN --units=pt1 -DHPT_MIN_WAVES=N), each run with a grid of N blocks per CU.
SPP=${1:-256}
for r in 1 2; do for v in 4 3 5 6; do
  [ "$v" = "4" ] && lib=hydracore3_amd/libhydra_hip.so || lib=hydracore3_amd/libhydra_hip_mw$v.so
  echo -n "cornell $v waves/SIMD: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload cornell --steps 3 --warmup 1 --spp $SPP --blocks-per-cu $v --no-cpu-baseline --no-also --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done; done
