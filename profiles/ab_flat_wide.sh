#!/bin/bash
# The megakernel's single-level traversal on the 4-wide compressed tree (libhydra_hip_fw.so: mkvariant.py fw -DHPT_FLAT_WIDE=1) against the BVH2 (base)
for r in 1 2; do for v in base fw; do
  [ "$v" = "base" ] && L=hydracore3_amd/libhydra_hip.so || L=hydracore3_amd/libhydra_hip_$v.so
  echo -n "dr $v: "; HYDRA_HIP_LIB=$PWD/$L python bench.py --workload dr --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
  echo "fixtures $v:"; HYDRA_HIP_LIB=$PWD/$L LAYOUT=2 SCENES=test_228,typed_materials,env_map python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"
done; done
