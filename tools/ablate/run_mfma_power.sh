#!/bin/bash
# tools/ablate/mfma_power (built in the container) with rocm-smi sampled beside it: run_mfma_power.sh <out dir under gpurun_out> [seconds per operand kind]
cd $GRAFT_REPO_ROOT
O=gpurun_out/$1; mkdir -p $O
( while true; do echo "t $(date +%s.%N) $(rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Power' | sed -e 's/.*sclk clock level: [^(]*(\([0-9]*\)Mhz).*/sclk \1/' -e 's/.*Power (W): \([0-9.]*\).*/W \1/' | tr '\n' ' ')"; sleep 0.2; done ) > $O/smi.log 2>&1 &
SMI=$!
( while true; do echo "mark $(date +%s.%N)"; sleep 0.5; done ) > /dev/null &
MK=$!
timeout -k 10 120 stdbuf -oL tools/ablate/mfma_power ${2:-4} | while IFS= read -r l; do echo "$(date +%s.%N) $l"; done > $O/run.log
kill $SMI $MK
python3 - <<PY
import re
smi = []
for l in open("$O/smi.log"):
    m = re.match(r"t (\S+) .*sclk (\d+).*W ([\d.]+)", l)
    if m: smi.append((float(m.group(1)), int(m.group(2)), float(m.group(3))))
kind, rows = None, []
for l in open("$O/run.log"):
    t, rest = l.split(" ", 1)
    if rest.startswith("=="): kind = rest[3:].strip(); continue
    m = re.search(r"t = ([\d.]+) s: (\d+) TFLOP/s", rest)
    if m:
        near = min(smi, key=lambda s: abs(s[0] - (float(t) - 0.25))) if smi else (0, 0, 0)
        rows.append((kind, float(m.group(1)), int(m.group(2)), near[1], near[2]))
last = None
for k, t, f, c, w in rows:
    if k != last: print(k); last = k
    print(f"   t = {t:4.1f} s  {f:5d} TFLOP/s   sclk {c} MHz   {w:.0f} W")
PY
