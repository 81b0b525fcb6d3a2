#!/usr/bin/env python3
"""Diagnostic: the K-contiguous kernel at the headline layer shape, twice -- bitwise equal? close to fp64 on sampled rows?"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snerf_amd import _lib
L = _lib.lib(); dev = "cuda:0"
P = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
J = int(sys.argv[2]) if len(sys.argv) > 2 else 512
K = int(sys.argv[3]) if len(sys.argv) > 3 else 512
act = int(sys.argv[4]) if len(sys.argv) > 4 else 1
g = torch.Generator().manual_seed(0)
X = (torch.rand(P, K, generator=g) * 2 - 1).to(dev)
Wm = (torch.randn(J, K, generator=g) * 0.06).to(dev)
b = (torch.randn(J, generator=g) * 0.1).to(dev)
p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
outs = []
for r in range(3):
    H = torch.empty(P, J, device=dev)
    _lib.check(L.snerf_test_bsp_kc(p(X), None, K, p(Wm), p(b), P, J, K, 0, 0, act, 1.0, 0, None, None, p(H), None, None, None, None, None, 0, 2, st), "fwd")
    outs.append(H)
torch.cuda.synchronize()
print("run0 == run1:", torch.equal(outs[0], outs[1]), " run0 == run2:", torch.equal(outs[0], outs[2]))
ref = X.double() @ Wm.double().T + b.double()
if act == 1: ref = torch.sin(ref)
err = (outs[0].double() - ref).abs()
nanm = torch.isnan(outs[0])
print("NaN elements:", int(nanm.sum()), " rows with NaN:", int(nanm.any(1).sum()), " first:", nanm.any(1).nonzero().flatten()[:12].tolist(), " cols of first:", nanm[nanm.any(1).nonzero().flatten()[0]].nonzero().flatten()[:12].tolist() if nanm.any() else [])
err = torch.nan_to_num(err, nan=0.0)
rows = err.max(dim=1).values
bad = (rows > 1e-4).nonzero().flatten()
print("max err", float(err.max()), " bad rows:", bad.numel(), " first bad rows:", bad[:16].tolist())
if bad.numel():
    t = (bad // 128).unique()
    print("bad 128-row tiles:", t.numel(), t[:32].tolist())
    cols = (err[bad[0]] > 1e-4).nonzero().flatten()
    print("bad cols in first bad row:", cols.numel(), cols[:8].tolist(), cols[-8:].tolist())
    for r in bad[:6].tolist():
        cc = (err[r] > 1e-4).nonzero().flatten()[:4].tolist()
        print(" row", r, [(c, float(outs[0][r, c]), float(outs[1][r, c]), float(ref[r, c])) for c in cc])
