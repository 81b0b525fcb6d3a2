import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tests import test_gpu_bsp as T
I = int(sys.argv[1]); rows = (3, 5, 1, 0); J = 1024; K = 528
g = torch.Generator().manual_seed(1)
X = (torch.rand(I, K, generator=g) * 2 - 1).to(T.DEV)
W = (torch.randn(J, K, generator=g) * (0.3 / K ** 0.5)).to(T.DEV)
b = (torch.randn(J, generator=g) * 0.1).to(T.DEV)
nw = torch.randn(sum(rows), J, generator=g).to(T.DEV)
H0, s0, _ = T._kc(X, W, b, act=T.ACT_SIN, want_sign=True, planes=2)
H1, s1, parts = T._kc(X, W, b, act=T.ACT_SIN, want_sign=True, planes=2, nd_w=nw, nd_rows=rows)
d = (H0 != H1)
idx = d.nonzero().cpu()
print("differing", idx.shape[0], "signs differ", int((s0 != s1).sum()))
r, c = idx[:, 0], idx[:, 1]
print("row % 128 // 32 (mi):", collections.Counter((r % 128 // 32).tolist()))
print("row % 32 (pt) top:", collections.Counter((r % 32).tolist()).most_common(8))
print("col % 256 // 64 (wave):", collections.Counter((c % 256 // 64).tolist()))
print("col % 64 // 32 (nj):", collections.Counter((c % 64 // 32).tolist()))
print("col % 32 // 16 (gg):", collections.Counter((c % 32 // 16).tolist()))
print("col % 16 // 8 (lh):", collections.Counter((c % 16 // 8).tolist()))
print("tile_i % 8:", collections.Counter((r // 128 % 8).tolist()))
err = (H0 - H1).abs()[d]
print("error magnitudes: min %.3e median %.3e max %.3e" % (float(err.min()), float(err.median()), float(err.max())))
# are the wrong values other rows' values?  (compare the wrong 8-column piece with the same columns of the other rows of the tile)
k = 0
for (ri, ci) in idx[:: max(1, idx.shape[0] // 12)].tolist():
    c8 = ci // 8 * 8
    piece = H1[ri, c8:c8 + 8]
    t0 = ri // 128 * 128
    blk = H0[t0:t0 + 128, :]
    m = (blk.unfold(1, 8, 8) == piece).all(-1).nonzero()
    print(f"row {ri} (in tile {ri % 128}) cols {c8}: H1 piece equals H0 at (row in tile, col8 index) {m[:4].tolist()}  |  max err {float((H0[ri, c8:c8+8]-piece).abs().max()):.3e}")
torch.set_printoptions(precision=5, linewidth=200)
z = torch.sin(X[:, :].double() @ W.double().T + b.double()).float()
for (ri, ci) in idx[:: max(1, idx.shape[0] // 6)].tolist():
    c8 = ci // 8 * 8
    print("row", ri, "cols", c8, "\n  H0", H0[ri, c8:c8+8].cpu(), "\n  H1", H1[ri, c8:c8+8].cpu(), "\n  true", z[ri, c8:c8+8].cpu())
    for o in range(nw.shape[0]):
        print("   nw row", o, nw[o, c8:c8+8].cpu())
    print("   bias", b[c8:c8+8].cpu())
    # the same point's other pieces
    print("   H1 of the point, gg=0 of the same half:", H1[ri, c8-16:c8-8].cpu(), " other lane half:", H1[ri, (c8 ^ 8):(c8 ^ 8)+8].cpu())
