#!/bin/bash
# gradient scatter deferred to the end of the reverse sweep (hpt_shade.h: drReverseSweep) against the build before it (libhydra_hip_base.so)
for v in base new; do
  [ "$v" = "base" ] && lib=hydracore3_amd/libhydra_hip_base.so || lib=hydracore3_amd/libhydra_hip.so
  for w in "dr" "dr --schedule 1" "dr --schedule 2" "dr_interior --spp 64"; do
    echo -n "$w $v: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload $w --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
  done
done
