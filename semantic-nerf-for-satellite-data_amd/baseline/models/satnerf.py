"""Sat-NeRF model container + inference() -- mirror of baseline/models/satnerf.py:8-255.

The nn.Module owns the parameters under the reference's state_dict names; all arithmetic happens in
libsnerf_hip.so through snerf_amd.ops.render_pass."""
import torch

from ... import ops
from .commons import Siren, sine_init, first_layer_sine_init


def _check_external_inputs(rays_d):
    if rays_d is not None:
        raise NotImplementedError("view-direction input is unused by the reference pipelines (input_sizes=[3,0])")


def inference(model, cfgs, rays_xyz, z_vals, rays_d=None, sun_d=None, rays_t=None, epoch=None):
    """Explicit-position seam (baseline/models/satnerf.py:8-98). Returns the reference's result dict."""
    _check_external_inputs(rays_d)
    res = ops.render_pass(model.spec, dict(model.named_parameters()),
                          ops.PassInputs(sun_d=sun_d, xyz=rays_xyz, z_vals=z_vals), rays_t)
    res.pop("z_vals")
    return res


class _NerfBase(torch.nn.Module):
    """Shared construction of fc_net / sigma / feats / rgb / sun_v / sky / beta heads."""

    def _build_common(self, in_xyz, feat, feat_last, layers, skips, siren, t_dims, rgb_extra=0):
        nl = (lambda: Siren()) if siren else (lambda: torch.nn.ReLU())
        fc = [torch.nn.Linear(in_xyz, feat), Siren(w0=30.0) if siren else torch.nn.ReLU()]
        for i in range(1, layers):
            fc += [torch.nn.Linear(feat + in_xyz if i in skips else feat, feat), nl()]
        self.fc_net = torch.nn.Sequential(*fc)
        self.sigma_from_xyz = torch.nn.Sequential(torch.nn.Linear(feat, 1), torch.nn.Softplus())
        self.feats_from_xyz = torch.nn.Linear(feat, feat)
        self.rgb_from_xyzdir = torch.nn.Sequential(torch.nn.Linear(feat + rgb_extra, feat_last), nl(),
                                                   torch.nn.Linear(feat_last, 3), torch.nn.Sigmoid())
        return nl

    def _build_shadow_heads(self, feat, feat_last, siren, t_dims, nl):
        sun = [torch.nn.Linear(feat + 3, feat_last), nl()]
        for _ in range(2):
            sun += [torch.nn.Linear(feat_last, feat_last), nl()]
        sun += [torch.nn.Linear(feat_last, 1), torch.nn.Sigmoid()]
        self.sun_v_net = torch.nn.Sequential(*sun)
        self.sky_color = torch.nn.Sequential(torch.nn.Linear(3, feat_last), torch.nn.ReLU(),
                                             torch.nn.Linear(feat_last, 3), torch.nn.Sigmoid())
        if siren:  # rs_semantic.py:239-243 / satnerf.py:189-193
            self.fc_net.apply(sine_init)
            self.fc_net[0].apply(first_layer_sine_init)
            self.sun_v_net.apply(sine_init)
            self.sun_v_net[0].apply(first_layer_sine_init)
        self.beta_from_xyz = torch.nn.Sequential(torch.nn.Linear(t_dims + feat, feat_last), nl(),
                                                 torch.nn.Linear(feat_last, 1), torch.nn.Softplus())

    def forward(self, *a, **k):
        raise NotImplementedError("the per-point MLP is fused into libsnerf_hip.so: use inference() or the renderer")


class SatNeRF(_NerfBase):
    def __init__(self, cfgs, layers=8, feat=256, mapping=False, mapping_sizes=[10, 4], skips=[4], siren=True,
                 t_embedding_dims=16):
        super().__init__()
        self.layers, self.skips, self.t_embedding_dims = layers, skips, t_embedding_dims
        self.input_sizes = [3, 0]
        self.rgb_padding = 0.001
        self.number_of_outputs = 9
        self.feat_last = feat if cfgs.pipeline.fc_use_full_features else feat // 2
        n_freq = mapping_sizes[0] if mapping else 0
        in_xyz = 2 * n_freq * 3 if mapping else 3
        nl = self._build_common(in_xyz, feat, self.feat_last, layers, skips, siren, t_embedding_dims)
        self._build_shadow_heads(feat, self.feat_last, siren, t_embedding_dims, nl)
        self.spec = ops.ModelSpec(fc_units=feat, fc_layers=layers, feat_last=self.feat_last, fc_skips=tuple(skips),
                                  n_freq=n_freq, siren=bool(siren), t_dim=t_embedding_dims, n_classes=0,
                                  sem_sigmoid=False, mfma=ops.mfma_mode(cfgs.pipeline, getattr(cfgs, "run", None)))
