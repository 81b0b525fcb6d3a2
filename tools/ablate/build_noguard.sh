#!/bin/bash
# tools/ablate/libsnerf_hip_noguard.so: the product objects + bsp_kc.hip WITHOUT store_data_guard (csrc/bsp_dev.h) -- the reproducer of
# the round-4 plane corruption (python tools/ablate/fin_big.py 262144 under SNERF_LIB_PATH=...) and the guard's cost in a same-box A/B.
set -e
cd "$(dirname "$0")/../../semantic-nerf-for-satellite-data_amd/csrc"
make -j6 ARCH=gfx950 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize -DKC_NO_STORE_GUARD -c bsp_kc.hip -o /tmp/bsp_kc_noguard.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libsnerf_hip_noguard.so profile.o /tmp/bsp_kc_noguard.o bsp_gemm.o bsp_aux.o bsp_pass.o aux_kernels.o composite.o loss.o optim.o api.o
