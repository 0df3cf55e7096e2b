#!/bin/bash
# Schedule crossover again, now that the wavefront trace kernel (and the megakernel on heavy scenes) walks the 4-wide tree:
# interior generator at 4 850 ... 1 M triangles, both schedules, automatic choice; and the light fixtures under the forced wavefront schedule
B="python bench.py --workload interior --steps 2 --warmup 1 --spp 32 --no-cpu-baseline --no-also --no-build"
for sd in 0 1 2 3; do
  for sc in 1 2; do echo -n "interior subdiv=$sd schedule=$sc: "; HYDRA_BENCH_SUBDIV=$sd $B --schedule $sc 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
  echo -n "interior subdiv=$sd automatic:  "; HYDRA_BENCH_SUBDIV=$sd $B 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done
SCENES=test_228,typed_materials,env_map python profiles/full_kernel.py 1024 64 2>&1 | grep schedule
