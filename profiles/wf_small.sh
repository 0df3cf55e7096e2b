#!/bin/bash
# heavy scene, small calls (a multi-GPU pixel share): wavefront with fewer trace blocks per CU vs block-local
for wh in "640 360" "960 540"; do set -- $wh
  for tb in 5 3 2 1; do echo -n "interior $1x$2 wavefront trace-blocks-per-cu $tb: "; python bench.py --workload interior --width $1 --height $2 --spp 64 --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also --schedule 2 --trace-blocks-per-cu $tb 2>&1 | grep -o "\"value\": [0-9.]*" | head -1; done
  echo -n "interior $1x$2 block-local: "; python bench.py --workload interior --width $1 --height $2 --spp 64 --steps 2 --warmup 1 --no-build --no-cpu-baseline --no-also --schedule 3 2>&1 | grep -o "\"value\": [0-9.]*" | head -1
done
