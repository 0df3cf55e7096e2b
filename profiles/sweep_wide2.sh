#!/bin/bash
# 4-wide tree: trace kernel built for 5 (base) / 6 (libhydra_hip_wf6.so: mkvariant.py wf6 --units=wf_trace -DHPT_WF_WAVES=6) waves per SIMD x grid x vote
B="python bench.py --workload interior --steps 2 --warmup 1 --spp 32 --no-cpu-baseline --no-also --no-build"
for lib in base wf6; do for w in 5 6 7; do for nm in 28 32 40; do
  [ "$lib" = "base" ] && L=hydracore3_amd/libhydra_hip.so || L=hydracore3_amd/libhydra_hip_$lib.so
  echo -n "$lib, trace blocks per CU $w, node_min $nm: "; HYDRA_HIP_LIB=$PWD/$L HPT_NODE_MIN=$nm $B --trace-blocks-per-cu $w 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done; done; done
