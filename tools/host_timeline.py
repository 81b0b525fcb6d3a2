"""Where does the host spend a steady-state training step?  cProfile over a few steps (cumulative time per function, C calls of the
library included): a host call that BLOCKS on the device shows up with ~a step's time.  Usage: python tools/host_timeline.py"""
import sys, os, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_cfgs
from snerf_amd.framework.pipelines import load_pipeline, TrainLoop

dev = torch.device("cuda:0")
cfgs = make_cfgs(4096, 64, 1, "f16x2")
pipe = load_pipeline(cfgs); pipe.log_metrics = False
loop = TrainLoop(pipe, cfgs, dev)
for s in range(6):
    loop.step(s)
pr = cProfile.Profile()
pr.enable()
for s in range(6, 16):
    loop.step(s)
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(25)
