#!/bin/bash
# run_trunk_dbg.sh <tag> <samples> <dbg values...>: duration of trunk_kernel under the diagnostic build's SNERF_TRUNK_DBG bits
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; S=$2; shift; shift
for v in "$@"; do
  O=gpurun_out/r5ab/$TAG/tdbg$v; mkdir -p $O
  SNERF_LIB_PATH=$GRAFT_REPO_ROOT/tools/ablate/libsnerf_hip_diag.so SNERF_TRUNK_DBG=$v timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/ablate/trunk_dbg.py $S > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
  echo "trunk dbg=$v S=$S"; python tools/ablate/summarize.py $O | grep "trunk_kernel"
done
