"""Semantic Sat-NeRF model container + inference() -- mirror of semantic/models/rs_semantic.py:8-340."""
import torch

from ... import ops
from ...baseline.models.satnerf import _NerfBase, _check_external_inputs


def inference(model, cfgs, rays_xyz, z_vals, rays_d=None, sun_d=None, rays_t=None, rays_t_s=None, epoch=None,
              render_options={}):
    """Explicit-position seam (semantic/models/rs_semantic.py:8-128): rays_xyz (N,S,3), z_vals (N,S),
    sun_d (N,3), rays_t (N,tau) -> the reference's result dict (rgb, depth, weights, transparency, albedo,
    sun, sky, beta, sigmas, semantic_logits, semantic_label[, beta_semantic])."""
    _check_external_inputs(rays_d)
    sc_only = bool(render_options.get("sc_pass", False)) if render_options else False
    res = ops.render_pass(model.spec, dict(model.named_parameters()),
                          ops.PassInputs(sun_d=sun_d, xyz=rays_xyz, z_vals=z_vals), rays_t, rays_t_s, sc_pass=sc_only)
    res.pop("z_vals")
    return res


class RSSemanticNeRF(_NerfBase):
    def __init__(self, cfgs, dataset_semantic):
        super().__init__()
        pc = cfgs.pipeline
        self.cfg = pc
        self.layers, self.feat = pc.fc_layers, pc.fc_units
        self.feat_last = self.feat if pc.fc_use_full_features else self.feat // 2
        self.skips = pc.fc_skips
        self.siren = pc.activation_function == "siren"
        self.t_embedding_dims = pc.t_embedding_tau
        self.input_sizes = [3, 0]
        self.rgb_padding = 0.001
        self.semantic_n_classes = dataset_semantic.semantic_n_classes
        tau = self.t_embedding_dims
        in_xyz = 2 * pc.mapping_pos_n_freq * 3
        nl = self._build_common(in_xyz, self.feat, self.feat_last, self.layers, self.skips, self.siren, tau,
                                rgb_extra=tau if pc.use_tj_instead_of_beta else 0)
        s_in = self.feat + (tau if pc.use_tj_for_s else 0)
        sem = [torch.nn.Linear(s_in, self.feat_last), nl(), torch.nn.Linear(self.feat_last, self.semantic_n_classes)]
        if pc.semantic_activation_function == "sigmoid":
            sem.append(torch.nn.Sigmoid())
        self.semantic_prediction = torch.nn.Sequential(*sem)
        self._build_shadow_heads(self.feat, self.feat_last, self.siren, tau, nl)
        if pc.use_separate_beta_for_s:
            self.semantic_beta_from_xyz = torch.nn.Sequential(
                torch.nn.Linear(tau + self.feat, self.feat_last), nl(), torch.nn.Linear(self.feat_last, 1),
                torch.nn.Softplus())
        self.spec = ops.ModelSpec.from_pipeline_cfg(pc, self.semantic_n_classes, model="semantic",
                                                    run_cfg=getattr(cfgs, "run", None))
