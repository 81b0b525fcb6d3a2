#!/bin/bash
# Same-box A/B of isolated launches: run_ab.sh <tag> <mode> <planes> <lib suffix | base>...   (base = the product library)
# kernel durations of tools/bsp_kernel_bench.py <mode> <planes> (262,144 x 512 x 512) under rocprofv3 --kernel-trace, per library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; MODE=$2; PL=$3; shift; shift; shift
for v in "$@"; do
  if [ $v = base ]; then unset SNERF_LIB_PATH; else export SNERF_LIB_PATH=$GRAFT_REPO_ROOT/tools/ablate/libsnerf_hip_$v.so; fi
  O=gpurun_out/r5ab/$TAG/${v}_pl$PL; mkdir -p $O
  timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/bsp_kernel_bench.py 8 $MODE $PL > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
  echo "lib=$v planes=$PL"; python tools/ablate/summarize.py $O | grep "gemm_kc\|gemm_dw"
done
