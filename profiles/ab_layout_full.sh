#!/bin/bash
# Full-material megakernel (every BSDF branch) on the own fixtures under the three layouts; optional library variant in $1
[ -n "$1" ] && export HYDRA_HIP_LIB=$PWD/hydracore3_amd/libhydra_hip_$1.so
for L in 0 2 1; do echo "== layout $L ${1:-base}"; LAYOUT=$L BPC=3 FORCE_FULL=1 SCENES=${SCENES:-legacy_materials,typed_materials,env_map} python profiles/full_kernel.py 1024 64 2>&1 | grep "schedule 1"; done
