#!/bin/bash
# kernel tests and isolated timing of the staggered-halves kernel (SNERF_KC=9, SIREN forward launches): run_kc9.sh <tag>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k/$1; mkdir -p $O
for grid in 0 2 6; do
  if [ $grid = 0 ]; then unset SNERF_KC_GRID; else export SNERF_KC_GRID=$grid; fi
  SNERF_KC=9 timeout -k 10 200 python -m pytest tests/test_gpu_bsp.py -m gpu -q -x -p no:cacheprovider -k "test_kc_ and not narrow and not child" > $O/tests9_$grid.log 2>&1 || { tail -40 $O/tests9_$grid.log; exit 1; }
  tail -1 $O/tests9_$grid.log
done
unset SNERF_KC_GRID
for v in 9 4; do
  SNERF_KC=$v timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d $O/kc$v -o t -- python3 tools/bsp_kernel_bench.py 6 fwd > $O/kc$v.log 2>&1 || { tail -5 $O/kc$v.log; exit 1; }
  python tools/ablate/summarize.py $O/kc$v | grep gemm_kc
done
