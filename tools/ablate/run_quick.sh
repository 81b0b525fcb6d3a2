#!/bin/bash
# usage: run_quick.sh <tag> <mode> [variant...]   -- kernel durations of the product build (and ablation builds)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; MODE=$2; shift; shift
O=gpurun_out/r2b/$TAG; mkdir -p $O
timeout -k 10 200 python -m pytest tests/test_gpu_bsp.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for v in base "$@"; do
  if [ $v = base ]; then unset SNERF_LIB_PATH; else export SNERF_LIB_PATH=$GRAFT_REPO_ROOT/tools/ablate/libsnerf_hip_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/$v -o t -- python3 tools/bsp_kernel_bench.py 8 $MODE > $O/$v.log 2>&1 || exit 1
done
