#!/bin/bash
# rocprofv3 kernel trace of the DEFAULT bench step (main and solar-correction pass on two streams): tools/profile_two_streams.sh <out dir under gpurun_out> [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --output-format csv -d $O -o t -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-eager-gpu-baseline --no-inference --no-profile --no-reduced "$@" > $O/bench.json 2> $O/bench.err || { tail $O/bench.err; exit 1; }
python3 tools/step_gaps.py $O
