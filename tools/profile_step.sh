#!/bin/bash
# rocprofv3 kernel statistics of the bench command (profiles/rNN/*): tools/profile_step.sh <out dir under gpurun_out>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o step -- python3 bench.py --steps 20 --warmup 5 --serial-passes --no-cpu-baseline --no-eager-gpu-baseline --no-inference --no-profile --no-reduced > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || { tail $O/bench_under_rocprof.err; exit 1; }
ls $O; head -40 $O/step_kernel_stats.csv | cut -c1-200
