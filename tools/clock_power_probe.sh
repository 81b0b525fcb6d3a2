#!/bin/bash
# Clock and power of the GPU while the bench's timed region runs (read-only rocm-smi samples, 4 per second):
# tools/clock_power_probe.sh <out dir under gpurun_out> [bench args]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; shift; mkdir -p $O
( while true; do echo "t $(date +%s.%N)"; rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (junction|edge|memory)"; sleep 0.25; done ) > $O/smi.log 2>&1 &
SMI=$!
python3 bench.py --steps ${PROBE_STEPS:-200} --warmup 20 --no-cpu-baseline --no-eager-gpu-baseline --no-inference --no-profile --no-reduced "$@" > $O/bench.json 2> $O/bench.err
kill $SMI
python3 - <<PY
import re, statistics
blocks, cur = [], {}
for l in open("$O/smi.log"):
    if l.startswith("t "):
        if cur: blocks.append(cur)
        cur = {"t": float(l.split()[1])}
    else:
        m = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", l)
        if m: cur["sclk"] = int(m.group(1))
        m = re.search(r"mclk clock level: \d+: \((\d+)Mhz\)", l)
        if m: cur["mclk"] = int(m.group(1))
        m = re.search(r"Power \(W\): ([\d.]+)", l)
        if m: cur["power"] = float(m.group(1))
        m = re.search(r"junction\) \(C\): ([\d.]+)", l)
        if m: cur["tj"] = float(m.group(1))
if cur: blocks.append(cur)
print(len(blocks), "samples")
for k in ("sclk", "mclk", "power", "tj"):
    v = [b[k] for b in blocks if k in b]
    if v: print(k, "min %.0f median %.0f max %.0f" % (min(v), statistics.median(v), max(v)), "| last 12:", v[-12:])
PY
python3 -c "import json;d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]);print(d['metric'],round(d['value']),'rays/s',round(d['ms_per_step'],2),'ms',d['dtype'])"
