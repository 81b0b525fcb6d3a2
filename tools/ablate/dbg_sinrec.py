import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from tests.test_gpu_bsp import _kc, ACT_SIN, AUX_SINREC, DEV
for (I, J, K, w0) in [(300, 512, 64, 30.0), (1000, 1024, 544, 1.0)]:
  for planes in (2, 1):
    g = torch.Generator().manual_seed(I + K)
    X = (torch.rand(I, K, generator=g) * 2 - 1).to(DEV)
    W = (torch.randn(J, K, generator=g) * (0.3 / K ** 0.5)).to(DEV)
    b = (torch.randn(J, generator=g) * 0.1).to(DEV)
    H, sign, _ = _kc(X, W, b, act=ACT_SIN, w0=w0, want_sign=True, c_col0=128, planes=planes)
    z = (X.double() @ W.double().T + b.double()) * w0
    Kg = 256
    G = torch.randn(I, Kg, generator=g).to(DEV)
    G[:128] *= 1e-7
    W2 = (torch.randn(J, Kg, generator=g) * 0.05).to(DEV)
    D, _, cs = _kc(G, W2, None, aux=AUX_SINREC, Hact=H, Hsign=sign, w0=w0, want_colsum=True, c_col0=128, planes=planes)
    ref = (G.double() @ W2.double().T) * (w0 * torch.cos(z))
    rows = (D.double() - ref).norm(dim=1) / ref.norm(dim=1)
    print(f"case {(I, J, K, w0)} planes {planes}: worst row err {float(rows.max()):.3e} at row {int(rows.argmax())}; rows>1e-3: {int((rows > 1e-3).sum())}")
    bad = (rows > 1e-3).nonzero().flatten().tolist()
    print("   bad rows:", bad[:40])
    if bad:
        r = bad[0]
        d = (D[r].double() - ref[r]).abs()
        c = int(d.argmax())
        print(f"   row {r}: worst col {c}: D {float(D[r, c]):.6e} ref {float(ref[r, c]):.6e} ratio {float(D[r, c] / ref[r, c]):.4f}; cols with rel err > 1e-2: {int((d > 1e-2 * ref[r].abs()).sum())}")
        ratio = (D[r].double() / ref[r])
        print("   ratios first 16 cols:", [round(float(x), 3) for x in ratio[:16]])
        print("   ratios cols 256..272:", [round(float(x), 3) for x in ratio[256:272]])
    want = torch.stack([ref[r:r + 128].sum(0) for r in range(0, I, 128)])
    print("   colsum relerr", float((cs.double().cpu() - want.cpu()).norm() / want.norm()))
