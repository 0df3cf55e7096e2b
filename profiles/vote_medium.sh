#!/bin/bash
# the voted node-loop exit on scenes between "light" and "heavy" (test_228: SAH estimate 10.9; own fixtures 4 ... 5): megakernel, forward and DR
for nm in 0 4 8 12; do
  echo "== node_min $nm"
  HPT_NODE_MIN=$nm SCENES=test_228,typed_materials,env_map python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"
  echo -n "dr: "; HPT_NODE_MIN=$nm python bench.py --workload dr --steps 3 --warmup 1 --no-cpu-baseline --no-also --no-build 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done
