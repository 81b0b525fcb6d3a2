"""Semantic renderer -- mirror of semantic/components/rendering.py:12-80 (RSSemanticRendering).

ts -> int64 happens on the device (the reference round-trips through a CPU LongTensor, :35-40; same
values), the embedding lookup stays a torch op (its scatter-add backward gives the nn.Embedding
gradient), and both passes -- main and, if sc_lambda > 0, the solar-correction pass on o + sun_d*z --
run as fused HIP passes that share one packed copy of the weights."""
import torch

from ... import ops
from ...framework.components.rendering import BaseRenderer, z_steps_on
from ...framework.components.rays import extras_component_fn
from ..models.rs_semantic import inference as rs_semantic_inference


def fused_model_rendering(renderer, models, typ, rays, extras, render_options, inference_func, default_inference):
    if inference_func is not default_inference:
        raise NotImplementedError(
            "only the library's own inference() can be injected: sampling, encoding, MLP and compositing are one "
            "fused HIP pass (snerf_forward); a foreign per-sample inference callable has nothing to plug into")
    cfgs = renderer.cfgs
    opts = render_options or {}
    sun_d = extras_component_fn(extras, "sun_d")
    ts = extras_component_fn(extras, "ts").squeeze(-1).long()
    rays_t = models["t"](ts)
    rays_t_s = models["t_s"](ts) if "t_s" in models else None
    model = models[typ]
    params = dict(model.named_parameters())
    packed = opts.get("packed_params")
    if packed is None:
        packed = ops.pack_params(model.spec, params)
    pin = ops.PassInputs(sun_d=sun_d, rays=rays, z_vals=opts.get("given_z_vals"),
                         z_steps=z_steps_on(rays.device, renderer.N_samples), u=opts.get("perturb_rand"))
    result = ops.render_pass(model.spec, params, pin, rays_t, rays_t_s, packed=packed)
    z_vals = result.pop("z_vals")
    if cfgs.pipeline.sc_lambda > 0:
        sc = ops.render_pass(model.spec, params, ops.PassInputs(sun_d=sun_d, rays=rays, z_vals=z_vals), rays_t, rays_t_s,
                             sc_pass=True, packed=packed)
        result["weights_sc"] = sc["weights"]
        result["transparency_sc"] = sc["transparency"]
        result["sun_sc"] = sc["sun"]
    if opts.get("return_z_vals"):
        result["z_vals"] = z_vals
    return result


class RSSemanticRendering(BaseRenderer):
    def __init__(self, cfgs, inference=rs_semantic_inference):
        super().__init__(cfgs)
        self.inference_func = inference

    def _model_rendering(self, models, typ, cfgs, rays, extras, xyz, z_vals, rays_d, epoch=None, progress=1.0,
                         render_options={}) -> dict:
        return fused_model_rendering(self, models, typ, rays, extras, render_options, self.inference_func,
                                     rs_semantic_inference)
