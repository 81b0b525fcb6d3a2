#!/bin/bash
# one ablation build: build_one.sh <name> [-DMACRO ...]  ->  tools/ablate/libsnerf_hip_<name>.so (product objects + a re-compiled bsp_gemm)
set -e
cd "$(dirname "$0")/../../semantic-nerf-for-satellite-data_amd/csrc"
make -j6 ARCH=gfx950 >/dev/null
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize "$@" -c bsp_gemm.hip -o /tmp/bsp_gemm_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libsnerf_hip_$name.so gemm.o gemm_x6.o /tmp/bsp_gemm_$name.o bsp_aux.o bsp_pass.o aux_kernels.o composite.o loss.o optim.o api.o
