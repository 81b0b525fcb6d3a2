#!/usr/bin/env python3
"""Development probe: time one full training step (main+sc forward, losses, backward) at a given size
straight through snerf_amd.ops; used with rocprofv3 to see the per-kernel split."""
import argparse
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from snerf_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--samples", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    spec = ops.ModelSpec()
    g = torch.Generator().manual_seed(0)
    import math
    params = {}
    shapes = {}
    # SIREN-range init straight from the spec names
    W, H, E, tau, C = 512, 256, 60, 4, 5
    def shp(n):
        if n.startswith("fc_net."):
            i = int(n.split(".")[1]) // 2
            k = E if i == 0 else (W + E if i == 4 else W)
            return (W, k) if n.endswith("weight") else (W,)
        table = {"sigma_from_xyz.0": (1, W), "feats_from_xyz": (W, W), "rgb_from_xyzdir.0": (H, W), "rgb_from_xyzdir.2": (3, H),
                 "semantic_prediction.0": (H, W), "semantic_prediction.2": (C, H), "sun_v_net.0": (H, W + 3), "sun_v_net.2": (H, H),
                 "sun_v_net.4": (H, H), "sun_v_net.6": (1, H), "sky_color.0": (H, 3), "sky_color.2": (3, H),
                 "beta_from_xyz.0": (H, W + tau), "beta_from_xyz.2": (1, H)}
        base = n.rsplit(".", 1)[0]
        s = table[base]
        return s if n.endswith("weight") else (s[0],)
    for n in spec.param_names():
        s = shp(n)
        fan = s[1] if len(s) == 2 else shp(n[:-4] + "weight")[1]
        b = 1 / math.sqrt(fan)
        if n.endswith("weight") and (n.startswith("fc_net.") or n.startswith("sun_v_net.")):
            b = 1 / fan if n in ("fc_net.0.weight", "sun_v_net.0.weight") else math.sqrt(6 / fan)
        params[n] = ((torch.rand(s, generator=g) * 2 - 1) * b).to(dev).requires_grad_(True)
    emb = torch.randn(50, 4, generator=g).to(dev).requires_grad_(True)
    N, S = a.rays, a.samples
    o = torch.rand(N, 3, generator=g) * 2 - 1
    d = torch.nn.functional.normalize(torch.randn(N, 3, generator=g), dim=1)
    rays = torch.cat([o, d, torch.zeros(N, 1), torch.rand(N, 1, generator=g) + 0.5], 1).to(dev)
    sun = torch.nn.functional.normalize(torch.rand(N, 3, generator=g), dim=1)
    extras = torch.cat([sun, torch.randint(0, 19, (N, 1), generator=g).float()], 1).to(dev)
    gt = torch.rand(N, 3, generator=g).to(dev)
    lab = torch.randint(0, 5, (N,), generator=g).to(dev)
    zs = torch.linspace(0, 1, S).to(dev)
    opt = torch.optim.Adam(list(params.values()) + [emb], lr=5e-4)

    def step():
        opt.zero_grad(set_to_none=True)
        u = torch.rand(N, S, device=dev)
        t = emb[extras[:, 3].long()]
        packed = ops.pack_params(spec, params)
        r = ops.render_pass(spec, params, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_steps=zs, u=u), t, packed=packed)
        sc = ops.render_pass(spec, params, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_vals=r["z_vals"]), t, sc_pass=True, packed=packed)
        beta = (r["weights"].unsqueeze(-1) * r["beta"]).sum(-2) + 0.05
        loss = ((r["rgb"] - gt) ** 2 / (2 * beta ** 2)).mean() + (3 + torch.log(beta).mean()) / 2
        s_ = sc["sun"].squeeze(-1)
        loss = loss + 0.05 / 3 * ((sc["transparency"].detach() - s_) ** 2).sum(-1).mean() + 0.05 / 3 * (1 - (sc["weights"].detach() * s_).sum(-1)).mean()
        loss = loss + 0.04 * torch.nn.functional.cross_entropy(r["semantic_logits"], lab, ignore_index=4)
        loss.backward()
        opt.step()
        return loss

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    import time
    t0 = time.time()
    for _ in range(a.steps):
        l = step()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / a.steps
    flops = 31453696.0 * N * S
    print(f"rays={N} S={S} ms/step={dt*1e3:.2f} rays/s={N/dt:.0f} TFLOP/s={flops/dt/1e12:.1f} loss={float(l):.4f} mem={torch.cuda.max_memory_allocated()/2**30:.1f}GiB")


if __name__ == "__main__":
    main()
