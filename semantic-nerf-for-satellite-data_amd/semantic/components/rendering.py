"""Semantic renderer -- mirror of semantic/components/rendering.py:12-80 (RSSemanticRendering).

ts -> int64 happens on the device (the reference round-trips through a CPU LongTensor, :35-40; same
values), the embedding lookup is ops.embed_rows (one launch forward, one deterministic launch backward
for the nn.Embedding gradient), and both passes -- main and, if sc_lambda > 0, the solar-correction pass on o + sun_d*z --
run as fused HIP passes that share one packed copy of the weights."""
import os

import torch

from ... import ops
from ...framework.components.rendering import BaseRenderer, z_steps_on
from ...framework.components.rays import extras_component_fn
from ..models.rs_semantic import inference as rs_semantic_inference


# main and solar-correction passes on two HIP streams (set SNERF_OVERLAP_SC=0 to serialise them)
OVERLAP_SC_PASS = os.environ.get("SNERF_OVERLAP_SC", "1") != "0"
if OVERLAP_SC_PASS and hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
    # the two passes deliberately run (and back-propagate) on two streams; autograd synchronises the accumulation itself
    torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
_SIDE_STREAMS = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


def fused_model_rendering(renderer, models, typ, rays, extras, render_options, inference_func, default_inference):
    if inference_func is not default_inference:
        raise NotImplementedError(
            "only the library's own inference() can be injected: sampling, encoding, MLP and compositing are one "
            "fused HIP pass (snerf_forward); a foreign per-sample inference callable has nothing to plug into")
    cfgs = renderer.cfgs
    opts = render_options or {}
    sun_d = extras_component_fn(extras, "sun_d")
    ts = extras_component_fn(extras, "ts").squeeze(-1).long()
    rays_t = ops.embed_rows(models["t"], ts)
    rays_t_s = ops.embed_rows(models["t_s"], ts) if "t_s" in models else None
    model = models[typ]
    params = dict(model.named_parameters())
    packed = opts.get("packed_params")
    if packed is None:
        packed = ops.pack_params(model.spec, params)
    pin = ops.PassInputs(sun_d=sun_d, rays=rays, z_vals=opts.get("given_z_vals"),
                         z_steps=z_steps_on(rays.device, renderer.N_samples), u=opts.get("perturb_rand"))
    do_sc = cfgs.pipeline.sc_lambda > 0
    z_given = opts.get("given_z_vals")
    side = _side_stream(rays.device) if (do_sc and OVERLAP_SC_PASS) else None
    if side is not None:
        # The solar-correction pass only shares z_vals with the main pass.  Sample z first (tiny kernel), then run the
        # two independent MLP passes on two HIP streams: their GEMM launches interleave on the CUs and fill each
        # other's ramp-up / tail.  autograd replays each pass's backward on the stream of its forward.
        if z_given is None:
            z_given = ops.sample_z(rays, z_steps_on(rays.device, renderer.N_samples), opts.get("perturb_rand"))
            pin = ops.PassInputs(sun_d=sun_d, rays=rays, z_vals=z_given)
        main = torch.cuda.current_stream(rays.device)
        side.wait_stream(main)      # (before the main pass is issued: the side stream waits for the weights pack and z, not for that pass)
        # The main pass is CREATED first: autograd runs the younger node first, so in the backward the solar-correction pass -- the
        # shorter one, done ~2 ms before the main pass -- adds its gradients into the optimiser's bucket first and the main pass's
        # unpack only waits for an event that completed long ago.  The other way round the short pass queued behind the long one's
        # unpack at the very end of the step: two cross-stream hand-overs (~15-25 us each) and its unpack kernels, all exposed.
        result = ops.render_pass(model.spec, params, pin, rays_t, rays_t_s, packed=packed)
        with torch.cuda.stream(side):
            sc = ops.render_pass(model.spec, params, ops.PassInputs(sun_d=sun_d, rays=rays, z_vals=z_given), rays_t,
                                 rays_t_s, sc_pass=True, packed=packed)
        main.wait_stream(side)
        for v in sc.values():
            v.record_stream(main)
    else:
        result = ops.render_pass(model.spec, params, pin, rays_t, rays_t_s, packed=packed)
        if do_sc:
            sc = ops.render_pass(model.spec, params, ops.PassInputs(sun_d=sun_d, rays=rays, z_vals=result["z_vals"]),
                                 rays_t, rays_t_s, sc_pass=True, packed=packed)
    z_vals = result.pop("z_vals")
    if do_sc:
        result["weights_sc"] = sc["weights"]
        result["transparency_sc"] = sc["transparency"]
        result["sun_sc"] = sc["sun"]
    if opts.get("return_z_vals"):
        result["z_vals"] = z_vals
    return result


_SC_KEYS = {"weights_sc": "weights", "transparency_sc": "transparency", "sun_sc": "sun"}


@torch.no_grad()
def fused_model_rendering_into(renderer, models, typ, rays, extras, render_options, out: dict):
    """Inference-only twin of fused_model_rendering: writes just the results named in `out` (un-suffixed keys) into
    the given tensors.  The solar-correction pass runs only if one of its three results is asked for -- a point-cloud
    or image render (rgb / depth / label) does half the work of render_rays."""
    opts = render_options or {}
    sun_d = extras_component_fn(extras, "sun_d")
    ts = extras_component_fn(extras, "ts").squeeze(-1).long()
    rays_t = ops.embed_rows(models["t"], ts)
    rays_t_s = ops.embed_rows(models["t_s"], ts) if "t_s" in models else None
    model = models[typ]
    params = dict(model.named_parameters())
    packed = opts.get("packed_params")
    if packed is None:
        packed = ops.pack_params(model.spec, params)
    main_out = {k: v for k, v in out.items() if k not in _SC_KEYS}
    sc_out = {_SC_KEYS[k]: v for k, v in out.items() if k in _SC_KEYS}
    ws = opts.get("workspace")
    z = opts.get("given_z_vals")
    if sc_out and z is None:   # both passes must share the depths
        z = ops.sample_z(rays, z_steps_on(rays.device, renderer.N_samples), opts.get("perturb_rand"))
    pin = ops.PassInputs(sun_d=sun_d, rays=rays, z_vals=z, z_steps=z_steps_on(rays.device, renderer.N_samples),
                         u=opts.get("perturb_rand"))
    if main_out:
        ws = ops.render_pass_into(model.spec, params, pin, rays_t, rays_t_s, main_out, packed=packed, workspace=ws)
    if sc_out:
        ws = ops.render_pass_into(model.spec, params, ops.PassInputs(sun_d=sun_d, rays=rays, z_vals=z), rays_t, rays_t_s,
                                  sc_out, sc_pass=True, packed=packed, workspace=ws)
    return ws


class RSSemanticRendering(BaseRenderer):
    def __init__(self, cfgs, inference=rs_semantic_inference):
        super().__init__(cfgs)
        self.inference_func = inference

    def _model_rendering(self, models, typ, cfgs, rays, extras, xyz, z_vals, rays_d, epoch=None, progress=1.0,
                         render_options={}) -> dict:
        return fused_model_rendering(self, models, typ, rays, extras, render_options, self.inference_func,
                                     rs_semantic_inference)
