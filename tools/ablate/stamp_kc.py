#!/usr/bin/env python3
"""Per-tile phase stamps of the persistent K-contiguous kernel (diagnostic build): shader-clock cycles of
tile start (waits + barrier + first fragment reads) | k-loop | drain + barrier | next-tile requests | epilogue, and the
time spent in the top-of-sub-step waits (all / sub-steps 2 and 3)."""
import sys
import numpy as np
raw = np.load(sys.argv[1]).astype(np.int64).reshape(-1)
d = raw[: 8 * 4096].reshape(4096, 8).copy()
wg = raw[8 * 4096:].reshape(-1, 4)
d = d[d[:, 0] != 0]
d[:, 7] &= 0xFFFFFFFF
names = ["start", "loop", "drain", "head", "epilogue", "waits_all", "strip_flush"]
print("tiles stamped:", len(d))
for i, n in enumerate(names):
    c = d[:, i + 1]
    print(f"{n:12s} median {np.median(c):9.0f}  mean {c.mean():9.0f}  p10 {np.percentile(c, 10):9.0f}  p90 {np.percentile(c, 90):9.0f}")
tot = d[:, 1:6].sum(1)
print(f"{'tile total':12s} median {np.median(tot):9.0f}  mean {tot.mean():9.0f}")
wg = wg[wg[:, 1] != 0]
clk = wg[:, 1] / np.maximum(wg[:, 2], 1) * 100.0    # MHz
print(f"workgroups {len(wg)}: cycles min {wg[:,1].min()} median {np.median(wg[:,1]):.0f} max {wg[:,1].max()};  tiles per workgroup min {wg[:,3].min()} max {wg[:,3].max()}")
print(f"clock (cycles / 100 MHz ticks): median {np.median(clk):.0f} MHz  min {clk.min():.0f}  max {clk.max():.0f};  span of the launch {(wg[:,0]+wg[:,1]).max() - wg[:,0].min()} cycles")
