"""Where does the first timed step after a device synchronize lose its ~3 ms?  Events inside the step (after the batch / forward+loss /
backward / optimiser) for the steps following a synchronize, with and without a short busy kernel queued before the first step.
Usage: python tools/first_step_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_cfgs
from snerf_amd import ops
from snerf_amd.framework.pipelines import load_pipeline, TrainLoop

dev = torch.device("cuda:0")
cfgs = make_cfgs(4096, 64, 1, "f16x2")
pipe = load_pipeline(cfgs); pipe.log_metrics = False
loop = TrainLoop(pipe, cfgs, dev)
for s in range(6):
    loop.step(s)
torch.cuda.synchronize()

def ev():
    e = torch.cuda.Event(enable_timing=True); e.record(); return e

def run(n, s0, label, prequeue=False, sleep=0.0):
    torch.cuda.synchronize()
    if sleep: time.sleep(sleep)
    if prequeue:   # ~2 ms of device work queued first: the host gets ahead before the step's first kernel is due
        x = torch.empty(1 << 28, device=dev)
        for _ in range(4): x.add_(1.0)
    marks = []
    for s in range(s0, s0 + n):
        pl = loop.pipeline
        m = [ev()]
        t0 = time.perf_counter()
        pl.current_epoch = s // loop.steps_per_epoch
        batch = {"rgb": loop.bank.batch(s, loop.global_batch, loop.rank, loop.world, shuffle=loop.shuffle)}
        loop.optimizer.zero_grad()
        m.append(ev())
        out = pl.training_step(batch, s)
        m.append(ev())
        with ops.accumulate_into_sinks():
            out["loss"].backward()
        m.append(ev())
        loop.optimizer._collect_foreign_grads()
        loop.optimizer.step()
        m.append(ev())
        marks.append((m, time.perf_counter() - t0))
    torch.cuda.synchronize()
    print(label)
    for i, (m, host) in enumerate(marks):
        d = [m[j].elapsed_time(m[j + 1]) for j in range(4)]
        gap = marks[i][0][4].elapsed_time(marks[i + 1][0][0]) if i + 1 < len(marks) else 0.0
        print(f"  step {i}: batch {d[0]:6.2f}  forward+loss {d[1]:6.2f}  backward {d[2]:6.2f}  optimiser {d[3]:6.2f}  total {sum(d):6.2f} ms   host {host*1e3:6.2f} ms   gap to next {gap:5.2f}")

run(4, 6, "after synchronize")
run(4, 10, "after synchronize + 50 ms idle", sleep=0.05)
run(4, 14, "after synchronize, 2 ms of device work queued first", prequeue=True)
run(4, 18, "after synchronize (again)")
