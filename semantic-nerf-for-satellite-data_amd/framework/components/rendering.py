"""Renderer base -- mirror of framework/components/rendering.py:119-174 (BaseRenderer).

Stratified sampling (reference sample_rays, :84-116) is fused into the HIP pass: render_rays only
draws the jitter tensor (the reference calls torch.rand_like with perturb=1.0 on every call, train
and eval alike) and hands it down; z_vals / xyz never materialise on the host side."""
import abc
import functools

import torch

from .rays import ray_component_fn


@functools.lru_cache(maxsize=16)
def _host_linspace(n: int):
    # computed on the host like the CPU reference does (torch.linspace(0, 1, S)); GPU linspace may
    # differ in the last bit, which would break bit-exact z_vals.
    return torch.linspace(0, 1, n)


_Z_STEPS_DEV: dict = {}


def z_steps_on(device, n: int) -> torch.Tensor:
    """The host's linspace on `device`, copied ONCE per (device, n).  (Copied per call it was the step's one blocking call: a
    host-to-device copy from pageable memory returns when the stream has reached it, i.e. the host waited a whole step behind
    the device here every step and the device then idled ~0.13 ms while the host caught up -- tools/host_timeline.py.)"""
    device = torch.device(device)
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    key = (device.type, device.index, n)
    t = _Z_STEPS_DEV.get(key)
    if t is None:
        t = _Z_STEPS_DEV[key] = _host_linspace(n).to(device)
    return t


class BaseRenderer:
    def __init__(self, cfgs) -> None:
        super().__init__()
        self.cfgs = cfgs
        self.N_samples = cfgs.pipeline.n_samples

    def render_rays(self, models: dict, rays: torch.Tensor, extras: torch.Tensor, epoch=None, progress=1.0,
                    render_options={}):
        rays_d = ray_component_fn(rays, "directions")
        opts = dict(render_options) if render_options else {}
        if "perturb_rand" not in opts and opts.get("perturb", 1.0) > 0:
            opts["perturb_rand"] = torch.rand(rays.shape[0], self.N_samples, device=rays.device, dtype=torch.float32)
        model_results = self._model_rendering(models, "coarse", self.cfgs, rays, extras, None, None, rays_d,
                                              epoch=epoch, progress=progress, render_options=opts)
        # "_coarse" postfix as in the reference (no fine network is ever built)
        return {f"{k}_coarse": v for k, v in model_results.items()}

    @torch.no_grad()
    def render_rays_into(self, models: dict, rays: torch.Tensor, extras: torch.Tensor, out: dict, render_options={}):
        """Inference-only render_rays that fills the preallocated tensors of `out` (keys as render_rays returns them,
        e.g. 'rgb_coarse', 'depth_coarse', 'semantic_label_coarse') and computes nothing else.  Returns the pass's
        workspace buffer so that a caller can hand it back through render_options['workspace'] for the next chunk."""
        from ...semantic.components.rendering import fused_model_rendering_into
        opts = dict(render_options) if render_options else {}
        if "perturb_rand" not in opts and opts.get("perturb", 1.0) > 0:
            opts["perturb_rand"] = torch.rand(rays.shape[0], self.N_samples, device=rays.device, dtype=torch.float32)
        bare = {}
        for k, v in out.items():
            if not k.endswith("_coarse"):
                raise KeyError(f"render_rays_into: result keys end in '_coarse', got '{k}'")
            bare[k[:-len("_coarse")]] = v
        return fused_model_rendering_into(self, models, "coarse", rays, extras, opts, bare)

    @abc.abstractmethod
    def _model_rendering(self, models: dict, typ: str, cfgs, rays, extras, xyz, z_vals, rays_d, epoch=None,
                         progress=1.0, render_options=None) -> dict:
        pass
