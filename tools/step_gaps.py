#!/usr/bin/env python3
"""Where a step's wall time is NOT covered by any kernel: step_gaps.py <dir with t_kernel_trace.csv> [steps to skip]
Union of the kernel intervals of a rocprofv3 --kernel-trace of bench.py (two streams), cut into steps at the optimiser's adam_kernel:
per step the wall time, the time some kernel was running, the time two kernels overlapped, and the largest idle gaps with the kernels on
either side."""
import csv, sys, os, statistics
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rows = []
for r in csv.DictReader(open(os.path.join(d, "t_kernel_trace.csv"))):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("snerf::bsp::", "").replace("snerf::", "")[:48]))
rows.sort()
ends = [e for s, e, n in rows if n.startswith("adam_kernel")]
steps = []
for a, b in zip(ends[skip:-1], ends[skip + 1:]):
    ks = [(s, e, n) for s, e, n in rows if s >= a and e <= b + 1]
    busy, both, cur_e, idle = 0, 0, a, []
    prev = "adam_kernel"
    for s, e, n in ks:
        if s > cur_e:
            idle.append((s - cur_e, prev, n))
            busy += e - s
            cur_e, prev = e, n
        else:
            both += min(e, cur_e) - s
            if e > cur_e:
                busy += e - cur_e
                cur_e, prev = e, n
    steps.append((b - a, busy, both, sum(k[1] - k[0] for k in ks), idle))
print(f"{len(steps)} steps: wall {statistics.mean(s[0] for s in steps) / 1e6:.3f} ms, some kernel running {statistics.mean(s[1] for s in steps) / 1e6:.3f} ms, "
      f"two at once {statistics.mean(s[2] for s in steps) / 1e6:.3f} ms, sum of kernel durations {statistics.mean(s[3] for s in steps) / 1e6:.3f} ms, "
      f"idle {statistics.mean(s[0] - s[1] for s in steps) / 1e6:.3f} ms in {statistics.mean(len(s[4]) for s in steps):.0f} gaps")
w = steps[len(steps) // 2]
print("largest gaps of one step (us, after -> before):")
for g, p, n in sorted(w[4], reverse=True)[:14]:
    print(f"   {g / 1e3:7.1f}  {p} -> {n}")
