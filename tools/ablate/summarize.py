#!/usr/bin/env python3
"""kernel-duration summary of rocprofv3 --kernel-trace CSVs: summarize.py <dir> [<dir> ...]"""
import csv, sys, collections, os
for d in sys.argv[1:]:
    f = os.path.join(d, "t_kernel_trace.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "snerf" not in n: continue
        n = n.replace("void ", "").replace("snerf::bsp::", "").replace("snerf::", "")
        n = n.split("(")[0]
        agg[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(d)
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if max(v) < 50: continue
        print(f"   {n[:70]:70s} n={len(v):3d} avg={sum(v)/len(v):8.1f} us min={min(v):8.1f}")
