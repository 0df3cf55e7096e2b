#!/bin/bash
# A/B of the full-material megakernel compiled for 4 / 3 / 2 waves per SIMD (HPT_FULL_WAVES), inside ONE gpurun call.
for r in 1 2; do
  echo "== 4 waves (base)";           python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"
  echo "== 4 waves, forced full";     FORCE_FULL=1 python profiles/full_kernel.py 1024 64 2>&1 | grep "test_035.*schedule 1"
  echo "== 3 waves";                  HYDRA_HIP_LIB=$PWD/hydracore3_amd/libhydra_hip_full3.so BPC=3 FORCE_FULL=1 python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"
  echo "== 2 waves";                  HYDRA_HIP_LIB=$PWD/hydracore3_amd/libhydra_hip_full2.so BPC=2 FORCE_FULL=1 python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"
done
