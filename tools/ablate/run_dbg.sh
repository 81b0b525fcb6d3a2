#!/bin/bash
# timing-only runs of the K-contiguous kernel with operand traffic removed (SNERF_KC_DBG bits): usage run_dbg.sh <tag> <mode> <dbg values...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; MODE=$2; shift; shift
for v in "$@"; do
  O=gpurun_out/r3a/$TAG/dbg$v; mkdir -p $O
  SNERF_LIB_PATH=$GRAFT_REPO_ROOT/tools/ablate/libsnerf_hip_diag.so SNERF_KC_DBG=$v timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/bsp_kernel_bench.py 6 $MODE > $O.log 2>&1 || exit 1
  echo "dbg=$v"; python tools/ablate/summarize.py $O | grep gemm_kc
done
