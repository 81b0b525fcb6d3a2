#!/bin/bash
# Per-row cost of the isolated GEMM launches at 262,144 and 524,288 rows (what ONE launch per layer over both passes would buy):
# run_rows.sh <tag> <planes>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=$1; PL=$2
for P in 262144 524288 262144 524288; do
  for MODE in fwd dx dw; do
    O=gpurun_out/r5rows/$TAG/${MODE}_${P}_pl$PL; mkdir -p $O
    SNERF_BENCH_P=$P timeout -k 10 150 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 tools/bsp_kernel_bench.py 8 $MODE $PL > $O.log 2>&1 || { tail -5 $O.log; exit 1; }
    echo "rows=$P mode=$MODE planes=$PL"; python tools/ablate/summarize.py $O | grep "gemm_kc\|gemm_dw"
  done
done
