"""Colour / uncertainty / solar-correction / depth losses -- mirror of baseline/components/loss.py:4-94.

Same constructor arguments, same forward(inputs, targets) -> (loss, loss_dict) contract and the same
loss_dict keys; the arithmetic (forward values and gradients) runs in the fused HIP loss kernels."""
import torch

from ... import _lib
from ...loss_ops import LossSpec, fused_loss, run_plans

_T = {k: i for i, k in enumerate(_lib.LOSS_TERMS)}


def _pick(terms, keys):
    return {k: terms[_T[k]] for k in keys}


def _sc_keys(on):
    return ["coarse_sc_term2", "coarse_sc_term3"] if on else []


class SNerfLoss(torch.nn.Module):
    """plain MSE (+ solar correction), used while epoch < first_beta_epoch (loss.py:71-94)"""

    def __init__(self, lambda_sc=0.05, solar_correction_enabled=True):
        super().__init__()
        self.lambda_sc = lambda_sc
        self.solar_correction_enabled = solar_correction_enabled

    def plan(self, inputs, targets):
        """(LossSpec, aux, loss_dict keys) of this module's call: loss_ops.run_plans merges it with other modules' plans"""
        sc = self.lambda_sc > 0 and self.solar_correction_enabled
        return LossSpec(color_mode=1, has_sc=sc, sc_lambda=float(self.lambda_sc)), {"gt_rgb": targets}, ["coarse_color"] + _sc_keys(sc)

    def forward(self, inputs, targets):
        return run_plans([self.plan(inputs, targets)], inputs)


class SatNerfLoss(torch.nn.Module):
    """beta-weighted colour loss + log beta (+ solar correction) (loss.py:16-27,50-68)"""

    def __init__(self, lambda_sc=0.0, solar_correction_enabled=True):
        super().__init__()
        self.lambda_sc = lambda_sc
        self.solar_correction_enabled = solar_correction_enabled

    def plan(self, inputs, targets):
        sc = self.lambda_sc > 0 and self.solar_correction_enabled
        return (LossSpec(color_mode=2, has_sc=sc, sc_lambda=float(self.lambda_sc)), {"gt_rgb": targets},
                ["coarse_color", "coarse_logbeta"] + _sc_keys(sc))

    def forward(self, inputs, targets):
        return run_plans([self.plan(inputs, targets)], inputs)


class DepthLoss(torch.nn.Module):
    """(lambda_ds / 3) * mean(weights * (depth - target)^2) (loss.py:30-47)"""

    def __init__(self, lambda_ds=1.0):
        super().__init__()
        self.lambda_ds = lambda_ds / 3.0

    def forward(self, inputs, targets, weights=1.0):
        aux = {"gt_depth": targets}
        if torch.is_tensor(weights):
            aux["depth_weights"] = weights
        elif float(weights) != 1.0:
            aux["depth_weights"] = torch.full_like(targets, float(weights))
        spec = LossSpec(has_depth=True, ds_lambda=float(self.lambda_ds * 3.0))
        loss, terms = fused_loss(spec, inputs, aux)
        return loss, _pick(terms, ["coarse_ds"])
