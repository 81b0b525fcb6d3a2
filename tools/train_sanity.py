#!/usr/bin/env python3
"""Development probe: train the default semantic pipeline on the synthetic ray bank for a few hundred steps and
print the loss curve (the fused HIP path must reduce the loss like any NeRF on fixed targets)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from snerf_amd.framework.configs import MainConfig
from snerf_amd.framework.pipelines import load_pipeline, TrainLoop

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cfgs = MainConfig(run={"max_train_steps": steps, "synthetic_rays": 16384, "synthetic_images": 8},
                  pipeline={"pipeline": "snerf_amd.semantic.pipelines.rs_semantic.RSSemanticPipeline", "batch_size": 2048,
                            "n_samples": 64, "fc_units": 256, "ignore_car_index": True, "depth_enabled": True,
                            "use_car_reg_loss": True, "lambda_c": 0.1, "render_chunk_size": 1 << 20})
torch.manual_seed(0)
pipe = load_pipeline(cfgs)
loop = TrainLoop(pipe, cfgs, torch.device("cuda:0"))
hist = []
for s in range(steps):
    out = loop.step(s)
    if s % 20 == 0 or s == steps - 1:
        terms = {k.split("/")[1]: round(float(v), 5) for k, v in pipe.logged.items() if k.startswith("train/coarse_")}
        print(f"step {s:4d} epoch {pipe.current_epoch} loss {float(out['loss']):.5f} psnr {float(pipe.logged.get('train/psnr', 0)):.2f} {terms}", flush=True)
        hist.append(float(out["loss"]))
assert all(map(lambda x: x == x, hist)), "NaN in the loss"
print("first", hist[0], "last", hist[-1])
