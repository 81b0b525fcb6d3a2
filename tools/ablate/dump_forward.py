"""Forward outputs of the headline shape (seeded) saved to a file -- run once per library (SNERF_LIB_PATH) and compare:
python tools/ablate/dump_forward.py out.pt [N S]; python tools/ablate/dump_forward.py --cmp a.pt b.pt"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
if sys.argv[1] == "--cmp":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        if a[k].dtype.is_floating_point:
            d = (a[k] - b[k]).abs()
            flat = d.reshape(d.shape[0], -1).max(1).values if d.dim() > 1 else d
            bad = (flat > 1e-5).nonzero().flatten()
            print(f"{k:28s} max {float(d.max()):.3e}  rows > 1e-5: {bad.numel()}  first {bad[:12].tolist()}")
        else:
            print(f"{k:28s} differ: {int((a[k] != b[k]).sum())}")
    sys.exit(0)
from oracle import snerf_oracle as O            # (inputs only: the seeded synthetic batch and initial weights)
from tests.test_gpu_kernels import _gpu_params, _hip_render
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
S = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dev = torch.device("cuda:0")
cfg = O.OracleCfg(n_samples=S)
pn = O.init_params_numpy(cfg, 22); emb = O.init_embedding_numpy(cfg, 22)
b = O.batch_to_torch(O.synthetic_batch(N, S, seed=122, car_prob=0.03, n_images=19))
gp = _gpu_params(pn, dev, requires_grad=bool(int(os.environ.get("TRAIN", "0"))))
out = _hip_render(cfg, gp, torch.from_numpy(emb).to(dev), b, dev)
torch.cuda.synchronize()
torch.save({k: v.detach().cpu() for k, v in out.items()}, sys.argv[1])
print("saved", sys.argv[1], {k: tuple(v.shape) for k, v in out.items()})
