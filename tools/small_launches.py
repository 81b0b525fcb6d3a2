"""Which Python line issues each small torch-side kernel of a training step?  torch.profiler (with_stack) over two steady-state steps:
for every aten op that launched a kernel, the first stack frame inside this repository.  Usage: python tools/small_launches.py"""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from bench import make_cfgs
from snerf_amd.framework.pipelines import load_pipeline, TrainLoop

dev = torch.device("cuda:0")
cfgs = make_cfgs(4096, 64, 1, "f16x2")
pipe = load_pipeline(cfgs); pipe.log_metrics = False
loop = TrainLoop(pipe, cfgs, dev)
for s in range(6):
    loop.step(s)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for s in range(6, 6 + N):
        loop.step(s)
    torch.cuda.synchronize()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cnt = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.kernels:
        continue
    frame = next((f for f in (ev.stack or []) if root in f and "tools/small_launches" not in f), "(no repo frame)")
    frame = frame.replace(root + "/", "")
    cnt[(ev.name, frame, tuple(k.name.split("(")[0][-40:] for k in ev.kernels))] += 1
for (name, frame, kern), c in sorted(cnt.items(), key=lambda x: x[0][1]):
    print(f"{c / N:5.1f}/step  {name:32s} {frame[:110]:110s} {kern}")
