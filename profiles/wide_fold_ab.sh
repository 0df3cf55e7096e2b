#!/bin/bash
# folded 4-wide node step (hpt_device.h: wideNodeStep) against the previous build (libhydra_hip_base.so): heavy scene, all three schedules
for v in base new; do
  [ "$v" = "base" ] && lib=hydracore3_amd/libhydra_hip_base.so || lib=hydracore3_amd/libhydra_hip.so
  for sched in 2 3; do
    echo -n "interior 64 spp schedule $sched $v: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload interior --spp 64 --schedule $sched --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
  done
  echo -n "dr_interior 64 spp $v: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload dr_interior --spp 64 --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done
