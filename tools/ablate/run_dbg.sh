#!/bin/bash
# timing-only runs of the K-contiguous kernel with operand traffic / parts of the tile removed (SNERF_KC_DBG bits, csrc/bsp_kc.hip DIAG):
# usage run_dbg.sh <tag> <mode> <planes> <dbg values...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; MODE=$2; PL=$3; shift; shift; shift
for v in "$@"; do
  O=gpurun_out/r5ab/$TAG/dbg${v}_pl$PL; mkdir -p $O
  SNERF_LIB_PATH=$GRAFT_REPO_ROOT/tools/ablate/libsnerf_hip_diag.so SNERF_KC_DBG=$v timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/bsp_kernel_bench.py 6 $MODE $PL > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
  echo "dbg=$v planes=$PL"; python tools/ablate/summarize.py $O | grep gemm_kc
done
