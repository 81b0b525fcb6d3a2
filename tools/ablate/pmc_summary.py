#!/usr/bin/env python3
import csv, sys, collections
for d in sys.argv[1:]:
    for pm in ("pmc1", "pmc2"):
        rows = list(csv.DictReader(open(f"{d}/{pm}/t_counter_collection.csv")))
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in rows:
            k = r["Kernel_Name"]
            if "gemm_" not in k: continue
            k = k.replace("void snerf::bsp::", "").split("(")[0]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, c in agg.items():
            print(pm, k[:40], {n: "%.4g" % (sum(v) / len(v)) for n, v in c.items()})
