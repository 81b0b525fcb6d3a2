"""timing of the fused trunk launch alone (one chunk of 40,960 rays x S) under the diagnostic library's SNERF_TRUNK_DBG bits"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_cfgs
from snerf_amd.framework.pipelines import load_pipeline
from snerf_amd.eval.utils.util import lean_inference
from oracle import snerf_oracle as O
dev = torch.device("cuda:0")
S = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfgs = make_cfgs(4096, S, 1, "f16x1")
cfgs.pipeline.render_chunk_size = 40960
pipe = load_pipeline(cfgs).to(dev)
b = O.batch_to_torch(O.synthetic_batch(81920, S, seed=1))
rays, extras = b["rays"].to(dev), b["extras"].to(dev)
for _ in range(3):
    lean_inference(cfgs, pipe.renderer, pipe.models, rays, extras)
torch.cuda.synchronize()
print("ok")
