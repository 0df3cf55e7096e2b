#!/bin/bash
# A/B of the full-material megakernel compiled for 3 (base) / 2 / 4 waves per SIMD (HPT_FULL_WAVES), inside ONE gpurun call.
# Build the variants first: python __graft_entry__.py variants full2 full4
for r in 1 2; do
  echo "== 3 waves (base)";  BPC=3 FORCE_FULL=1 python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"
  echo "== 2 waves";         HYDRA_HIP_LIB=$PWD/hydracore3_amd/libhydra_hip_full2.so BPC=2 FORCE_FULL=1 python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"
  echo "== 4 waves";         HYDRA_HIP_LIB=$PWD/hydracore3_amd/libhydra_hip_full4.so BPC=4 FORCE_FULL=1 python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"
done
