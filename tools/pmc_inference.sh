#!/bin/bash
# HBM-side bytes of a full-frame forward-only run (tools/inference_rate.py: 409,600 rays x 64 samples, batched + lean, each twice):
# FETCH_SIZE / WRITE_SIZE in passes of their own; summed per pass over all kernels.  usage: pmc_inference.sh <out dir under gpurun_out>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/$c -o t -- python3 tools/inference_rate.py > $O/$c.log 2>&1 || { tail -5 $O/$c.log; exit 1; }
done
python3 - <<PY
import csv, collections
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open("$O/%s/t_counter_collection.csv" % c)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("snerf::bsp::", "").replace("snerf::", "")
        agg[k] += float(r["Counter_Value"]); n[k] += 1
    tot[c] = agg
    print("==", c, "(raw counter units summed over the run; per kernel)")
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:12]:
        print("   %-70s launches %5d  sum %.4g" % (k[:70], n[k], v))
    print("   total", sum(agg.values()))
PY
