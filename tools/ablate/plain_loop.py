"""TrainLoop steps with nothing else on the stream (no bench marks): for kernel traces of the bare loop.  python tools/ablate/plain_loop.py [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from bench import make_cfgs
from snerf_amd.framework.pipelines import load_pipeline, TrainLoop
dev = torch.device("cuda:0")
cfgs = make_cfgs(4096, 64, 1, "f16x2")
pipe = load_pipeline(cfgs); pipe.log_metrics = False
loop = TrainLoop(pipe, cfgs, dev)
for s in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    loop.step(s)
torch.cuda.synchronize()
