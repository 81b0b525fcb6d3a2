#!/bin/bash
# build_variant.sh <suffix> <extra hipcc flags for bsp_kc.hip>: tools/ablate/libsnerf_hip_<suffix>.so = the product objects with bsp_kc.hip rebuilt
set -e
cd "$(dirname "$0")/../../semantic-nerf-for-satellite-data_amd/csrc"
SUF=$1; shift
make -j6 ARCH=gfx950 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize "$@" -c bsp_kc.hip -o /tmp/bsp_kc_$SUF.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libsnerf_hip_$SUF.so profile.o /tmp/bsp_kc_$SUF.o bsp_trunk.o bsp_gemm.o bsp_aux.o bsp_pass.o aux_kernels.o composite.o loss.o optim.o api.o
