#!/bin/bash
# build_variant_src.sh <suffix> <source stem: bsp_gemm | bsp_kc | bsp_trunk | ...> <extra hipcc flags>:
# tools/ablate/libsnerf_hip_<suffix>.so = the product objects with <stem>.hip rebuilt with the extra flags
set -e
cd "$(dirname "$0")/../../semantic-nerf-for-satellite-data_amd/csrc"
SUF=$1; STEM=$2; shift; shift
make -j6 ARCH=gfx950 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I. -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize "$@" -c $STEM.hip -o /tmp/${STEM}_$SUF.o
OBJS=""
for o in profile bsp_kc bsp_trunk bsp_gemm bsp_aux bsp_pass aux_kernels composite loss optim api; do
  if [ $o = $STEM ]; then OBJS="$OBJS /tmp/${STEM}_$SUF.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libsnerf_hip_$SUF.so $OBJS
