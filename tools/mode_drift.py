"""Same seeded training run (headline config, synthetic rays) under two arithmetic modes; prints the loss every 100 steps.
Usage: python tools/mode_drift.py <steps> <modeA> <modeB>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_cfgs
from snerf_amd.framework.pipelines import load_pipeline, TrainLoop

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
modes = sys.argv[2:4] if len(sys.argv) > 3 else ["f16x1", "f16x2"]
dev = torch.device("cuda:0")
curves = {}
for m in modes:
    torch.manual_seed(0)
    cfgs = make_cfgs(4096, 64, 1, m)
    pipe = load_pipeline(cfgs)
    pipe.log_metrics = False
    loop = TrainLoop(pipe, cfgs, dev)
    acc, out = [], []
    for s in range(steps):
        torch.manual_seed(1000 + s)          # same jitter in both runs
        o = loop.step(s)
        acc.append(o["loss"].detach())
        if (s + 1) % 100 == 0:
            out.append(float(torch.stack(acc).mean())); acc = []
    curves[m] = out
    w = torch.cat([p.detach().reshape(-1) for p in pipe.parameters()])
    curves[m + "_w"] = w
    del loop, pipe
    torch.cuda.empty_cache()
a, b = modes
print("mean loss per 100 steps")
for i, (x, y) in enumerate(zip(curves[a], curves[b])):
    print(f"  steps {100*i:5d}-{100*i+99:5d}: {a} {x:.6f}   {b} {y:.6f}   diff {y - x:+.2e}")
wa, wb = curves[a + "_w"], curves[b + "_w"]
print(f"final weights: relative L2 difference {float((wa - wb).norm() / wa.norm()):.3e}")
