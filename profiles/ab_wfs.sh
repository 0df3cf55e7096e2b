#!/bin/bash
# A/B inside ONE gpurun call: the wavefront shade kernel with every BSDF branch at 3 (base) vs 4 waves per SIMD, on the 1M-triangle interior forced
# onto the full-material kernels; plus the lean baseline. Build the variant first: python __graft_entry__.py variants wfs4
B="--steps 2 --warmup 1 --spp 32 --no-cpu-baseline --no-also --no-build"
for r in 1 2; do
  echo -n "interior lean:            "; python bench.py --workload interior $B 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
  echo -n "interior full, shade 3w:  "; HYDRA_BENCH_FORCE_FULL=1 python bench.py --workload interior $B 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
  echo -n "interior full, shade 4w:  "; HYDRA_BENCH_FORCE_FULL=1 HYDRA_HIP_LIB=$PWD/hydracore3_amd/libhydra_hip_wfs4.so python bench.py --workload interior $B 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done
