#!/bin/bash
# rocprofv3 kernel statistics of the one-plane (REDUCED) step at BASELINE configs[2]'s per-GPU shape: tools/profile_f16x1.sh <out dir under gpurun_out> [samples]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; S=${2:-96}; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o step -- python3 bench.py --mfma f16x1 --samples $S --steps 10 --warmup 3 --serial-passes --no-cpu-baseline --no-eager-gpu-baseline --no-inference --no-profile --no-reduced > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err || { tail $O/bench_under_rocprof.err; exit 1; }
python3 - <<PY
import csv, json
rows = list(csv.DictReader(open("$O/step_kernel_stats.csv")))
u = json.loads(open("$O/bench_under_rocprof.json").read().strip().splitlines()[-1])
steps = u["steps"] + u["warmup"]
tot = sum(int(r["TotalDurationNs"]) for r in rows); calls = sum(int(r["Calls"]) for r in rows)
out = ["# rocprofv3 --kernel-trace --stats -- python3 bench.py --mfma f16x1 --samples $S --steps 10 --warmup 3 --serial-passes --no-cpu-baseline --no-eager-gpu-baseline --no-inference --no-profile --no-reduced", "",
       "Round 5, 1x MI355X: the REDUCED-precision mode (SNERF_FLAG_F16X1, one fp16 plane) at 4096 rays x $S samples, trunk forward as ONE persistent launch (bsp_trunk.hip).",
       "%d steps in the trace; bench.py under the profiler: %.0f rays/s, %.2f ms/step (dtype %s)." % (steps, u["value"], u["ms_per_step"], u["dtype"]), "",
       "| kernel | calls / step | total ms | avg us | % of device time |", "|---|---|---|---|---|"]
for r in rows[:18]:
    out.append("| \`%s\` | %.1f | %.1f | %.1f | %.2f |" % (r["Name"][:110], int(r["Calls"]) / steps, int(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100.0 * int(r["TotalDurationNs"]) / tot))
out += ["", "Device time in kernels: %.2f ms per step, %.1f launches per step." % (tot / steps / 1e6, calls / steps)]
open("$O/f16x1_s${S}_summary.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
