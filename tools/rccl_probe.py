#!/usr/bin/env python3
"""RCCL on this pool, as far as one GPU can show it: the backend initialises (world size 1) and the per-call cost of the step's two
collectives -- the 16-float loss-count all-reduce and the flat gradient bucket (2,826,766 floats) -- issued back to back and between
kernels of the compute stream.  A one-rank all-reduce moves nothing: what is measured is RCCL's launch path on this box."""
import os
import time

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
for n in (16, 2826766):
    x = torch.ones(n, device="cuda")
    for _ in range(5):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    host = (time.perf_counter() - t0) / 50 * 1e3
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    y = torch.zeros(1 << 20, device="cuda")
    dev = []
    for _ in range(20):
        y.add_(1.0)
        ev[0].record()
        dist.all_reduce(x)
        ev[1].record()
        y.add_(1.0)
        torch.cuda.synchronize()
        dev.append(ev[0].elapsed_time(ev[1]))
    dev.sort()
    print(f"all_reduce of {n} floats, world 1: {host:.3f} ms per call back to back (host clock); on the compute stream between two kernels: "
          f"median {dev[len(dev) // 2]:.3f} ms, min {dev[0]:.3f} (HIP events)")
print("nccl / rccl version", torch.cuda.nccl.version())
dist.barrier()
dist.destroy_process_group()
