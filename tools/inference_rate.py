"""Full-frame inference rate (rays/s) of the reference-shaped batched_inference vs lean_inference on synthetic rays."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench import make_cfgs
from snerf_amd.framework.pipelines import load_pipeline
from snerf_amd.eval.utils.util import batched_inference, lean_inference
from oracle import snerf_oracle as O

dev = torch.device("cuda:0")
which = sys.argv[2] if len(sys.argv) > 2 else "both"      # both | lean | batched
cfgs = make_cfgs(4096, 64, 1, sys.argv[3] if len(sys.argv) > 3 else "f16x2")
cfgs.pipeline.render_chunk_size = int(sys.argv[1]) if len(sys.argv) > 1 else 40960
pipe = load_pipeline(cfgs).to(dev)
R = 640 * 640
b = O.batch_to_torch(O.synthetic_batch(R, 64, seed=1))
rays, extras = b["rays"].to(dev), b["extras"].to(dev)
for name, fn in (("batched_inference (all results, sc pass)", lambda: batched_inference(cfgs, pipe.renderer, pipe.models, rays, extras)),
                 ("lean_inference (rgb, depth, label)", lambda: lean_inference(cfgs, pipe.renderer, pipe.models, rays, extras))):
    if which != "both" and not name.startswith(which):
        continue
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"{name}: {R} rays x 64 samples, chunk {cfgs.pipeline.render_chunk_size}: {dt*1e3:.1f} ms = {R/dt/1e3:.1f} k rays/s, peak mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
    del r; torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
