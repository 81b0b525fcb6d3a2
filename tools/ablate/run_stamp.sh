#!/bin/bash
# per-workgroup phase stamps of the K-contiguous GEMM: run_stamp.sh <tag> <stamp build> [...]   (builds made with -DBSP_ABL_STAMP)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; shift
O=gpurun_out/r2b/$TAG; mkdir -p $O
for v in "$@"; do
  export SNERF_LIB_PATH=$GRAFT_REPO_ROOT/tools/ablate/libsnerf_hip_$v.so STAMP_OUT=$O/$v.npy
  timeout -k 10 120 python3 tools/bsp_kernel_bench.py 3 ${STAMP_MODE:-stamp} > $O/$v.log 2>&1 || { tail $O/$v.log; exit 1; }
  echo "== $v"; python3 tools/ablate/stamp_summary.py $O/$v.npy
done
