#!/usr/bin/env python3
"""phase summary of a BSP_ABL_STAMP run: stamp_summary.py <stamps.npy>  (rows: workgroups; see bsp_gemm.hip for the columns)"""
import sys
import numpy as np
d = np.load(sys.argv[1]).astype(np.int64)
d = d[d[:, 0] > 0]
t0 = d[:, 0].min()
loop_us = (d[:, 1] - d[:, 0]) / 100.0
epi_us = (d[:, 2] - d[:, 1]) / 100.0
span = (d[:, 2].max() - t0) / 100.0
cyc = lambda x: f"{x.mean():9.0f} cyc (p10 {np.percentile(x, 10):7.0f}, p90 {np.percentile(x, 90):7.0f})"
print(f"workgroups {len(d)}, launch span {span:.1f} us; per workgroup: loop {loop_us.mean():.2f} us, epilogue {epi_us.mean():.2f} us")
print("k-loop            ", cyc(d[:, 3]), f" -> {d[:,3].mean() / loop_us.mean():.0f} cyc/us")
print("  in vmcnt waits  ", cyc(d[:, 4] >> 32))
print("  in barriers     ", cyc(d[:, 4] & 0xFFFFFFFF))
print("epilogue phase A  ", cyc(d[:, 5]))
print("max exchange      ", cyc(d[:, 6]))
print("split+store issue ", cyc(d[:, 7] >> 32))
print("store drain       ", cyc(d[:, 7] & 0xFFFFFFFF))
