#!/usr/bin/env python3
"""Copy the judged summaries of a gpurun_out/<dir> (tools/profile_step.sh + tools/pmc_traffic.sh + a default bench.py run)
into profiles/rNN/:  make_profile_artifacts.py <gpurun_out dir> <profiles dir> [<default bench json>]"""
import csv
import json
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
rows = list(csv.DictReader(open(os.path.join(src, "prof", "step_kernel_stats.csv"))))
shutil.copy(os.path.join(src, "prof", "step_kernel_stats.csv"), os.path.join(dst, "bench_n1_kernel_stats.csv"))
under = open(os.path.join(src, "prof", "bench_under_rocprof.json")).read().strip().splitlines()[-1]
open(os.path.join(dst, "bench_n1_under_rocprof.json"), "w").write(under + "\n")
u = json.loads(under)
steps = u["steps"] + u["warmup"]
tot = sum(int(r["TotalDurationNs"]) for r in rows)
calls = sum(int(r["Calls"]) for r in rows)
out = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps %d --warmup %d --serial-passes --no-cpu-baseline --no-eager-gpu-baseline --no-inference --no-profile --no-reduced" % (u["steps"], u["warmup"]),
       "",
       "Round 5, 1x MI355X, %s, default arithmetic (block-scaled fp16 planes, three MFMA products)." % u["metric"],
       "%d steps in the trace; `--serial-passes` keeps the main and solar-correction passes on one stream so per-kernel durations" % steps,
       "are not stretched by overlap.  bench.py under the profiler: %.0f rays/s, %.2f ms/step.  Full table: bench_n1_kernel_stats.csv." % (u["value"], u["ms_per_step"]),
       "", "| kernel | calls / step | total ms | avg us | % of device time |", "|---|---|---|---|---|"]
for r in rows[:22]:
    out.append("| `%s` | %.1f | %.1f | %.1f | %.2f |" % (r["Name"][:110], int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                  100.0 * int(r["TotalDurationNs"]) / tot))
out += ["", "Device time in kernels: %.2f ms per step, %.1f launches per step." % (tot / steps / 1e6, calls / steps)]
open(os.path.join(dst, "bench_n1_summary.md"), "w").write("\n".join(out) + "\n")
for f in ("summary.txt", "pmc_hbm_traffic.json"):
    shutil.copy(os.path.join(src, "pmc", f), os.path.join(dst, "pmc_summary.txt" if f == "summary.txt" else f))
if len(sys.argv) > 3:
    line = open(sys.argv[3]).read().strip().splitlines()[-1]
    json.loads(line)
    open(os.path.join(dst, "bench_n1_default_run.json"), "w").write(line + "\n")
print("\n".join(out[:14]))
