#!/usr/bin/env python3
"""Print the headline numbers of a bench.py JSON line (development aid)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f'{d["value"]:.0f} rays/s  {d["ms_per_step"]:.2f} ms/step  kc avg {r["avg_launch_ms"]*1e3:.1f} us  frac {r["frac"]:.4f}')
for k, v in r["per_variant"].items():
    print("  ", k[:70], f'{v["launches"]} x {v["avg_ms"]*1e3:.1f} us')
print("   inference", d.get("inference"))
