#!/usr/bin/env python3
"""Is the step bound by the 1400 W package power cap?  The SAME launch sequence timed twice: with the bench's random-init weights, and
with every parameter zeroed and the learning rate at 0 (all activations, gradients and weight packs are then zeros: the same instructions
on operands that toggle nothing), rocm-smi sampled four times a second throughout.

    python tools/power_bound_probe.py [f16x2|f16x1] [samples]"""
import os
import re
import statistics
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import snerf_amd  # noqa: E402,F401
from snerf_amd.framework.pipelines import load_pipeline, TrainLoop  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
cfgs = bench.make_cfgs(4096, S, 1, mode)
pipe = load_pipeline(cfgs)
pipe.log_metrics = False
loop = TrainLoop(pipe, cfgs, dev)

samples, stop = [], threading.Event()


def smi():
    while not stop.is_set():
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", o)
            p = re.search(r"Power \(W\): ([\d.]+)", o)
            if c and p:
                samples.append((time.perf_counter(), int(c.group(1)), float(p.group(1))))
        except Exception:
            pass
        time.sleep(0.2)


th = threading.Thread(target=smi, daemon=True)
th.start()


def timed(n, step0):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        loop.step(step0 + i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    mid = [(c, p) for t, c, p in samples if t0 + 0.3 * (t1 - t0) <= t <= t1]
    return (t1 - t0) / n * 1e3, (statistics.median([c for c, _ in mid]) if mid else None), (statistics.median([p for _, p in mid]) if mid else None), len(mid)


for i in range(15):
    loop.step(i)
N = 150
a = timed(N, 15)
with torch.no_grad():
    for p in pipe.parameters():
        p.zero_()
for g in loop.optimizer.param_groups:
    g["lr"] = 0.0
if hasattr(loop.optimizer, "lr"):
    loop.optimizer.lr = 0.0
for i in range(10):
    loop.step(15 + N + i)
b = timed(N, 25 + N)
nz = sum(float(p.abs().sum()) for p in pipe.parameters())
stop.set()
print(f"{mode} 4096 x {S}: random-init weights {a[0]:.2f} ms/step at sclk {a[1]} MHz, {a[2]} W ({a[3]} samples) | "
      f"all-zero weights, lr 0 (sum |w| = {nz}) {b[0]:.2f} ms/step at sclk {b[1]} MHz, {b[2]} W ({b[3]} samples)")
