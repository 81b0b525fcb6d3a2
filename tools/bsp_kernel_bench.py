#!/usr/bin/env python3
"""Isolated launches of the block-scaled plane GEMMs at the headline layer shape (262,144 points x 512 x 512) through the
test hooks -- run under `rocprofv3 --kernel-trace --stats` to read per-kernel durations (the hooks themselves allocate
and convert, so wall time here means nothing).

    python tools/bsp_kernel_bench.py <reps> <all|kc|fwd|fwdnd|plain|dx|dw|narrow>  (fwdnd: the SIREN forward launch with the folded 1-wide projection) [planes = 2 | 1]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snerf_amd import _lib  # noqa: E402

L = _lib.lib()
if os.environ.get("SNERF_BENCH_KC_GRID"):      # persistent grid of the K-contiguous launches (default: two workgroups per CU)
    L.snerf_test_set_kc_grid(int(os.environ["SNERF_BENCH_KC_GRID"]))
dev = "cuda:0"
P, W = int(os.environ.get("SNERF_BENCH_P", "262144")), 512      # SNERF_BENCH_P: rows (points) of the launch
g = torch.Generator().manual_seed(0)
X = (torch.rand(P, W, generator=g) * 2 - 1).to(dev)
Wm = (torch.randn(W, W, generator=g) * 0.06).to(dev)
b = torch.zeros(W, device=dev)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
which = sys.argv[2] if len(sys.argv) > 2 else "all"
planes = int(sys.argv[3]) if len(sys.argv) > 3 else 2


def p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
H = torch.empty(P, W, device=dev)
sign = torch.zeros((P // 32) * (W // 64) * 64, dtype=torch.int32, device=dev)
cs = torch.zeros(P // 32, W, device=dev)
nw = torch.randn(W, generator=g).to(dev)
ndo = torch.empty(8, P, device=dev)
for _ in range(reps if which in ("all", "kc", "fwd", "fwdnd") else 1):   # forward SIREN layer (always once: the dX launches need H and the sign words)
    _lib.check(L.snerf_test_bsp_kc(p(X), None, W, p(Wm), p(b), P, W, W, 0, 0, 1, 1.0, 0, None, None, p(H), p(sign), None, p(nw) if which == "fwdnd" else None, p(ndo) if which == "fwdnd" else None, None, 0, planes, st), "fwd")
for _ in range(reps if which in ("plain",) else 0):   # forward layer without activation
    _lib.check(L.snerf_test_bsp_kc(p(X), None, W, p(Wm), p(b), P, W, W, 0, 0, 0, 1.0, 0, None, None, p(H), None, None, None, None, None, 0, planes, st), "plain")
G = torch.randn(P, W, generator=g).to(dev) * 1e-3
D = torch.empty(P, W, device=dev)
for _ in range(reps if which in ("all", "kc", "dx") else 0):   # dX with the derivative epilogue + bias-gradient column sums
    _lib.check(L.snerf_test_bsp_kc(p(G), None, W, p(Wm), None, P, W, W, 0, 0, 0, 1.0, 3, p(H), p(sign), p(D), None, p(cs), None, None, None, 0, planes, st), "dx")
Cw = torch.empty(W, W, device=dev)
for _ in range(reps if which in ("all", "dw") else 0):   # dW, 64 splits
    _lib.check(L.snerf_test_bsp_dw(p(G), W, p(X), W, P, W, W, 0, 0, P // 64, 0, p(Cw), planes, st), "dw")
S32 = torch.empty(P, 32, device=dev)
W32 = (torch.randn(32, W, generator=g) * 0.05).to(dev)
for _ in range(reps if which in ("all", "narrow") else 0):   # 32-wide head
    _lib.check(L.snerf_test_bsp_kc(p(X), None, W, p(W32), None, P, 32, W, 0, 0, 0, 1.0, 0, None, None, p(S32), None, None, None, None, None, 1, planes, st), "narrow")
torch.cuda.synchronize()
print("ok")
