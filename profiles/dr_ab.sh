#!/bin/bash
# Diagnostic A/B inside ONE gpurun call: PathTraceDR (--workload dr) with parts of the adjoint machinery compiled out (the variants' gradients are
# NOT usable). Build first:  python __graft_entry__.py unit <tag> pt6 -DHPT_DBG_DR_...   (tags: see the loop)
for v in base "$@"; do
  [ "$v" = "base" ] && lib=hydracore3_amd/libhydra_hip.so || lib=hydracore3_amd/libhydra_hip_$v.so
  echo -n "dr $v: "; HYDRA_HIP_LIB=$PWD/$lib python bench.py --workload dr --steps 2 --warmup 1 --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done
