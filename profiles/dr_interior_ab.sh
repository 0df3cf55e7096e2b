#!/bin/bash
# --workload dr_interior (wavefront DR): the DR shade pass with parts compiled out (variants' gradients are NOT usable), and the parameter texture's size
for tex in 4096 1024; do for v in base "$@"; do
  [ "$v" = "base" ] && lib=hydracore3_amd/libhydra_hip.so || lib=hydracore3_amd/libhydra_hip_$v.so
  echo -n "dr_interior tex $tex $v: "; HYDRA_BENCH_TEX=$tex HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload dr_interior --spp 32 --steps 2 --warmup 1 --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done; done
