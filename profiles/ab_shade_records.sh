#!/bin/bash
# 64-byte shading records per triangle (default) against the index chain (HPT_SHADE_RECORDS=0), same binary, one gpurun call
for r in 1 2; do for wl in "interior 32" "dr 256" "dr_interior 32"; do set -- $wl
  echo -n "$1 records:     "; python bench.py --workload $1 --steps 2 --warmup 1 --spp $2 --no-cpu-baseline --no-also --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
  echo -n "$1 index chain: "; HPT_SHADE_RECORDS=0 python bench.py --workload $1 --steps 2 --warmup 1 --spp $2 --no-cpu-baseline --no-also --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done; done
echo "test_228 forward (megakernel, lean):"; SCENES=test_228 python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"; HPT_SHADE_RECORDS=0 SCENES=test_228 python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"
