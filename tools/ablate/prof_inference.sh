#!/bin/bash
# kernel statistics of one-plane lean inference with the fused trunk: prof_inference.sh <tag> <samples>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5ab/$1; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o inf -- python3 tools/ablate/trunk_ab.py $2 > $O/run.log 2>&1 || { tail $O/run.log; exit 1; }
grep fused $O/run.log
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/inf_kernel_stats.csv")))
tot=sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print("%-100s calls %5s avg %9.1f us  %5.1f %%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"])/1e3, 100.0*int(r["TotalDurationNs"])/tot))
PY
