#!/bin/bash
# kernel tests and isolated timing of the 256 x 256 K-contiguous kernel (SNERF_KC=8) next to the 128 x 256 one: run_kc8.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k/$1; mkdir -p $O
SNERF_KC=8 timeout -k 10 300 python -m pytest tests/test_gpu_bsp.py -m gpu -q -x -p no:cacheprovider > $O/tests8.log 2>&1 || { tail -40 $O/tests8.log; exit 1; }
tail -1 $O/tests8.log
for v in 8 4; do
  SNERF_KC=$v timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/kc$v -o t -- python3 tools/bsp_kernel_bench.py 6 kc > $O/kc$v.log 2>&1 || { tail -5 $O/kc$v.log; exit 1; }
  python tools/ablate/summarize.py $O/kc$v | grep gemm_kc
done
