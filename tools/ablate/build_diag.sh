#!/bin/bash
# Diagnostic build of the K-contiguous kernel: tools/ablate/libsnerf_hip_diag.so (product objects + bsp_kc.hip with -DKC_DIAG_BUILD).
# SNERF_KC_DBG bits then remove operand traffic (zero-size descriptors): 1 A, 2 W, 4 stores, 8 A always L2-resident.  Timing only.
set -e
cd "$(dirname "$0")/../../semantic-nerf-for-satellite-data_amd/csrc"
make -j6 ARCH=gfx950 >/dev/null
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize -DKC_DIAG_BUILD -c bsp_kc.hip -o /tmp/bsp_kc_diag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize -DTRUNK_DIAG_BUILD -c bsp_trunk.hip -o /tmp/bsp_trunk_diag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/ablate/libsnerf_hip_diag.so profile.o /tmp/bsp_kc_diag.o /tmp/bsp_trunk_diag.o bsp_gemm.o bsp_aux.o bsp_pass.o aux_kernels.o composite.o loss.o optim.o api.o
