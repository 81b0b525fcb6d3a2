#!/usr/bin/env python3
"""Summarise the PMC passes of tools/pmc_traffic.sh: per kernel, average counter values per launch; HBM-side bytes calibrated on
the to_planes launches of the isolated run (536,870,912 B read and written each).  Writes <dir>/pmc_hbm_traffic.json
(bench.py reads `kc_bytes_per_launch` from the copy committed under profiles/rNN/)."""
import collections
import csv
import json
import os
import sys

d = sys.argv[1]


def per_kernel(sub):
    f = os.path.join(d, sub, "t_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(f):
        return agg
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "snerf" not in k:
            continue
        k = k.replace("void ", "").replace("snerf::bsp::", "").replace("snerf::", "").split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def avg(v):
    return sum(v) / len(v)


KNOWN = 262144 * 512 * 4   # bytes one to_planes launch reads (fp32) and writes (two fp16 planes)
iso_f, iso_w = per_kernel("iso_fetch"), per_kernel("iso_write")
cal = {}
if "to_planes_kernel" in iso_f:
    cal["fetch_bytes_per_unit"] = KNOWN / avg(iso_f["to_planes_kernel"]["FETCH_SIZE"])
    cal["write_bytes_per_unit"] = KNOWN / avg(iso_w["to_planes_kernel"]["WRITE_SIZE"])
print("calibration (bytes per counter unit, from to_planes: %d B each way):" % KNOWN, cal)
out = {"calibration": cal, "known_bytes_to_planes": KNOWN, "kernels": {}}
for tag, ff, ww in (("isolated 262144x512x512", iso_f, iso_w), ("isolated 262144x512x512, one plane", per_kernel("iso1_fetch"), per_kernel("iso1_write")),
                    ("bench step", per_kernel("step_fetch"), per_kernel("step_write"))):
    print("==", tag)
    for k in sorted(ff, key=lambda k: -sum(ff[k]["FETCH_SIZE"])):
        f = avg(ff[k]["FETCH_SIZE"]) * cal.get("fetch_bytes_per_unit", 0.0)
        w = avg(ww[k]["WRITE_SIZE"]) * cal.get("write_bytes_per_unit", 0.0) if k in ww else 0.0
        n = len(ff[k]["FETCH_SIZE"])
        if f + w < 1e6:
            continue
        print(f"   {k[:60]:60s} n={n:4d}  read {f / 1e6:9.1f} MB  written {w / 1e6:9.1f} MB per launch")
        out["kernels"].setdefault(tag, {})[k] = {"launches": n, "read_bytes": f, "written_bytes": w}
step = out["kernels"].get("bench step", {})
kc = [(v["launches"], v["read_bytes"] + v["written_bytes"]) for k, v in step.items() if k.startswith("gemm_kc_kernel")]
if kc:
    out["kc_bytes_per_launch"] = sum(n * b for n, b in kc) / sum(n for n, _ in kc)
    print("gemm_kc_kernel, bench step: %.1f MB per launch (launch-weighted over its variants)" % (out["kc_bytes_per_launch"] / 1e6))
out["bench_steps_profiled"] = 5      # `bench.py --steps 3 --warmup 2`: every launch of the process is in the trace
tot = sum((v["read_bytes"] + v["written_bytes"]) * v["launches"] for v in step.values()) / out["bench_steps_profiled"] if step else 0.0
if tot:
    out["step_hbm_bytes"] = tot
    print("whole bench step: %.1f GB of HBM traffic (read + written, all kernels)" % (tot / 1e9))
for sub in ("iso_sq", "iso_sq2", "iso1_sq", "iso1_sq2", "step_sq"):
    a = per_kernel(sub)
    print("==", sub)
    for k, c in a.items():
        if not k.startswith("gemm_"):
            continue
        vals = {n: avg(v) for n, v in c.items()}
        print("  ", k[:44], {n: "%.4g" % v for n, v in vals.items()})
        out.setdefault(sub, {})[k] = vals
try:
    import subprocess
    out["commit"] = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or os.environ.get("SNERF_COMMIT")
except Exception:
    out["commit"] = os.environ.get("SNERF_COMMIT")
out["provenance"] = ("tools/pmc_traffic.sh on one MI355X (gpurun box): rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in passes of their own over "
                     "`bench.py --steps 3 --warmup 2 --serial-passes --no-reduced` and over isolated launches (both plane counts); units calibrated on to_planes (known bytes)")
json.dump(out, open(os.path.join(d, "pmc_hbm_traffic.json"), "w"), indent=1)
