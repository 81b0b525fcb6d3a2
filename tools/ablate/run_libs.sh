#!/bin/bash
# kernel durations of tools/bsp_kernel_bench.py <mode> under several builds of the library: run_libs.sh <tag> <mode> <lib suffix | base>...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; MODE=$2; shift; shift
for v in "$@"; do
  if [ $v = base ]; then unset SNERF_LIB_PATH; else export SNERF_LIB_PATH=$GRAFT_REPO_ROOT/tools/ablate/libsnerf_hip_$v.so; fi
  O=gpurun_out/r3a/$TAG/$v; mkdir -p $O
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/bsp_kernel_bench.py 6 $MODE > $O.log 2>&1 || exit 1
  echo "lib=$v"; python tools/ablate/summarize.py $O | grep "gemm_kc\|gemm_dw"
done
