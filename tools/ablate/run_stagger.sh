#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b/stag2; mkdir -p $O
for v in 0 2 3 4 6; do
  export SNERF_KC_STAGGER=$v
  timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/s$v -o t -- python3 tools/bsp_kernel_bench.py 8 kc > $O/s$v.log 2>&1 || exit 1
done
