"""lean inference rate in the one-plane mode with the fused trunk (csrc/bsp_trunk.hip) on and off, same process, same box"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_cfgs
from snerf_amd import _lib
from snerf_amd.framework.pipelines import load_pipeline
from snerf_amd.eval.utils.util import lean_inference
from oracle import snerf_oracle as O

dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfgs = make_cfgs(4096, S, 1, "f16x1")
cfgs.pipeline.render_chunk_size = 40960
pipe = load_pipeline(cfgs).to(dev)
R = 409600
b = O.batch_to_torch(O.synthetic_batch(R, S, seed=1))
rays, extras = b["rays"].to(dev), b["extras"].to(dev)
L = _lib.lib()
for rep in range(2):
    for on in (0, 1):
        L.snerf_test_set_trunk_fusion(on)
        lean_inference(cfgs, pipe.renderer, pipe.models, rays[:81920], extras[:81920]); torch.cuda.synchronize()
        t0 = time.perf_counter(); r = lean_inference(cfgs, pipe.renderer, pipe.models, rays, extras); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"fused trunk {on}: {R} rays x {S}: {dt*1e3:.1f} ms = {R/dt/1e3:.1f} k rays/s", flush=True)
