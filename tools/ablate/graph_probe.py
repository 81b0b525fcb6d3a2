#!/usr/bin/env python3
"""Diagnostic: which part of a pass survives capture in a HIP graph?  (SNERF_WS_POISON=1 fills every buffer the library
writes with 0xFF first.)  usage: graph_probe.py <stage>   stage = pack | fwd | fwdbwd"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import snerf_oracle as O
from tests.test_gpu_kernels import _gpu_params, _spec
from snerf_amd import ops
stage = sys.argv[1] if len(sys.argv) > 1 else "fwd"
dev = torch.device("cuda:0")
cfg = O.OracleCfg(fc_units=64, n_samples=16)
pn = O.init_params_numpy(cfg, 3)
b = {k: v.to(dev) for k, v in O.batch_to_torch(O.synthetic_batch(256, 16, seed=5)).items()}
emb = torch.from_numpy(O.init_embedding_numpy(cfg, 3)).to(dev)
spec = _spec(cfg)
rays, extras, u = b["rays"], b["extras"], b["u"]
t = emb[extras[:, 3].long()]
zs = torch.linspace(0, 1, cfg.n_samples).to(dev)
gp = _gpu_params(pn, dev, requires_grad=(stage == "fwdbwd"))
packed_static = ops.pack_params(spec, gp)

def run():
    if stage == "pack":
        return {"packed": ops.pack_params(spec, gp).clone()}
    packed = ops.pack_params(spec, gp) if stage != "fwd_static" else packed_static
    ctx = torch.enable_grad() if stage == "fwdbwd" else torch.no_grad()
    with ctx:
        res = ops.render_pass(spec, gp, ops.PassInputs(sun_d=extras[:, :3], rays=rays, z_steps=zs, u=u), t, None, packed=packed)
        out = {k: v.detach().clone() for k, v in res.items() if v.is_floating_point()}
        if stage == "fwdbwd":
            for p in gp.values():
                p.grad = None
            (res["rgb"].square().sum() + res["depth"].sum()).backward()
            out.update({"g_" + k: p.grad.clone() for k, p in gp.items() if p.grad is not None})
    return out

ref = run(); torch.cuda.synchronize()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    got = run()
for rep in range(4):
    g.replay(); torch.cuda.synchronize()
    bad = [(k, float((got[k] - ref[k]).abs().max())) for k in ref if not torch.equal(got[k], ref[k])]
    print(stage, "replay", rep, "mismatching:", bad[:6] if bad else "none")
