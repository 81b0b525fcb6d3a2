#!/bin/bash
# Ablation builds of the block-scaled plane GEMM (diagnostics): one libsnerf_hip_<variant>.so per macro set, linked against
# the product objects.  Usage: tools/ablate/build_bsp_variants.sh ; then SNERF_LIB_PATH=tools/ablate/libsnerf_hip_<v>.so ...
set -e
cd "$(dirname "$0")/../../semantic-nerf-for-satellite-data_amd/csrc"
make -j6 ARCH=gfx950 >/dev/null
OUT=../../tools/ablate
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -fno-slp-vectorize"
build() {  # name, macros...
  name=$1; shift
  /opt/rocm/bin/hipcc $FLAGS "$@" -c bsp_gemm.hip -o /tmp/bsp_gemm_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/libsnerf_hip_$name.so gemm.o gemm_x6.o /tmp/bsp_gemm_$name.o bsp_aux.o bsp_pass.o aux_kernels.o composite.o loss.o optim.o api.o
}
build o_noepi -DBSP_ABL_ONEWG -DBSP_ABL_NOEPI &
build o_mfmaonly -DBSP_ABL_ONEWG -DBSP_ABL_NOEPI -DBSP_ABL_NOLDSREAD -DBSP_ABL_NODMA -DBSP_ABL_NOBLOAD &
build o_nobload -DBSP_ABL_ONEWG -DBSP_ABL_NOEPI -DBSP_ABL_NOBLOAD &
build o_nolds -DBSP_ABL_ONEWG -DBSP_ABL_NOEPI -DBSP_ABL_NOLDSREAD &
build o_nodma -DBSP_ABL_ONEWG -DBSP_ABL_NOEPI -DBSP_ABL_NODMA &
wait
ls -la $OUT/*.so
