"""Baseline Sat-NeRF renderer -- mirror of baseline/components/rendering.py:12-67 (SatNeRFRendering)."""
from ...framework.components.rendering import BaseRenderer
from ...semantic.components.rendering import fused_model_rendering
from ..models.satnerf import inference as satnerf_inference


class SatNeRFRendering(BaseRenderer):
    def _model_rendering(self, models, typ, cfgs, rays, extras, xyz, z_vals, rays_d, epoch=None, progress=1.0,
                         render_options={}) -> dict:
        return fused_model_rendering(self, models, typ, rays, extras, render_options, satnerf_inference,
                                     satnerf_inference)
