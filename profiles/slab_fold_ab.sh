#!/bin/bash
# folded BVH2 slab test (hpt_device.h: nodeSlabs + slabRay) against the build before it (libhydra_hip_wide.so = folded 4-wide step only)
for v in wide new; do
  [ "$v" = "wide" ] && lib=hydracore3_amd/libhydra_hip_wide.so || lib=hydracore3_amd/libhydra_hip.so
  for w in "dr --schedule 3" "dr --schedule 1" "spectral" "film" "cornell --spp 256"; do
    echo -n "$w $v: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload $w --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
  done
done
