#!/bin/bash
# pool state as non-temporal traffic (hpt_decl.h: ldP / stP) in the wavefront kernels: product build against the build before it (libhydra_hip_base.so)
for v in base new; do
  [ "$v" = "base" ] && lib=hydracore3_amd/libhydra_hip_base.so || lib=hydracore3_amd/libhydra_hip.so
  for w in "interior --spp 64" "dr_interior --spp 64" "spectral_interior"; do
    echo -n "$w $v: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload $w --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
  done
done
