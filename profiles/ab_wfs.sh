#!/bin/bash
# A/B inside ONE gpurun call: the wavefront shade kernel with every BSDF branch at 4 vs 3 waves per SIMD, on the 1M-triangle interior forced
# onto the full-material kernels; plus the lean baseline and the full megakernel (now 3 waves) on the fixtures.
for r in 1 2; do
  echo -n "interior lean:            "; python bench.py --workload interior --steps 2 --warmup 1 --spp 32 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*"
  echo -n "interior full, shade 4w:  "; HYDRA_BENCH_FORCE_FULL=1 python bench.py --workload interior --steps 2 --warmup 1 --spp 32 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*"
  echo -n "interior full, shade 3w:  "; HYDRA_BENCH_FORCE_FULL=1 HYDRA_HIP_LIB=$PWD/hydracore3_amd/libhydra_hip_wfs3.so python bench.py --workload interior --steps 2 --warmup 1 --spp 32 --no-cpu-baseline 2>&1 | grep -o "\"value\": [0-9.]*"
done
python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"
