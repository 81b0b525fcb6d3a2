"""dW launch time against the points per workgroup: python tools/ablate/dw_ksplit.py <P> <k_split> [reps]  (under rocprofv3 --kernel-trace --stats)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from snerf_amd import _lib
L = _lib.lib()
P, ks = int(sys.argv[1]), int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
W = 512
g = torch.Generator().manual_seed(0)
X = (torch.rand(P, W, generator=g) * 2 - 1).to("cuda:0")
G = (torch.randn(P, W, generator=g) * 1e-3).to("cuda:0")
Cw = torch.empty(W, W, device="cuda:0")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(reps):
    _lib.check(L.snerf_test_bsp_dw(C.c_void_p(G.data_ptr()), W, C.c_void_p(X.data_ptr()), W, P, W, W, 0, 0, ks, 0, C.c_void_p(Cw.data_ptr()), 2, st), "dw")
torch.cuda.synchronize()
print("ok")
