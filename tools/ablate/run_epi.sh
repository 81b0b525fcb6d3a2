#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b/epi; mkdir -p $O
export SNERF_KC_STAGGER=0
for v in base nostore; do
  if [ $v = base ]; then unset SNERF_LIB_PATH; else export SNERF_LIB_PATH=$GRAFT_REPO_ROOT/tools/ablate/libsnerf_hip_$v.so; fi
  for m in fwd plain; do
    timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/${v}_$m -o t -- python3 tools/bsp_kernel_bench.py 8 $m > $O/${v}_$m.log 2>&1 || exit 1
  done
done
