import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_gpu_bsp as T
I = int(sys.argv[1]); rows = (3, 5, 1, 0); J = 1024; K = 528
g = torch.Generator().manual_seed(1)
X = (torch.rand(I, K, generator=g) * 2 - 1).to(T.DEV)
W = (torch.randn(J, K, generator=g) * (0.3 / K ** 0.5)).to(T.DEV)
b = (torch.randn(J, generator=g) * 0.1).to(T.DEV)
nw = torch.randn(sum(rows), J, generator=g).to(T.DEV)
for signs in (False, True):
    for planes in (2, 1):
        H0, s0, _ = T._kc(X, W, b, act=T.ACT_SIN, want_sign=signs, planes=planes)
        for rep in range(3):
            H1, s1, parts = T._kc(X, W, b, act=T.ACT_SIN, want_sign=signs, planes=planes, nd_w=nw, nd_rows=rows)
            d = (H0 != H1)
            print(f"signs {signs} planes {planes} rep {rep}: differing elements {int(d.sum())}, rows {int(d.any(1).sum())}, cols by tile {[int(d[:, 256*t:256*t+256].sum()) for t in range(4)]}", flush=True)
        H2, s2, p1 = T._kc(X, W[:256], b[:256], act=T.ACT_SIN, want_sign=signs, planes=planes, nd_w=nw[0, :256].contiguous())
        print("   1-wide fold on the first tile: differing", int((H2 != H0[:, :256]).sum()))
# column-sum (backward, no activation) launches: run-to-run determinism of the output planes
for planes in (2, 1):
    outs = [T._kc(X, W, None, act=T.ACT_NONE, want_colsum=True, planes=planes)[0] for _ in range(4)]
    print(f"colsum launch planes {planes}: elements differing from run 0:", [int((o != outs[0]).sum()) for o in outs[1:]], flush=True)
