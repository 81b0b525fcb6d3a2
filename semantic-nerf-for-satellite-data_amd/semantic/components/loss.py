"""Semantic losses -- mirror of semantic/components/loss.py:6-157 (same ctor args / forward contract /
loss_dict keys; NaN for an empty car set or an all-ignored CE batch, as in the reference)."""
import torch

from ...loss_ops import LossSpec, run_plans


def _n_classes(inputs):
    return int(inputs["semantic_logits_coarse"].shape[-1])


class SemanticLoss(torch.nn.Module):
    def __init__(self, lambda_s, car_index, ignore_car_index=False):
        super().__init__()
        self.lambda_s = lambda_s
        self.ignore_index = car_index if ignore_car_index else -100

    def plan(self, inputs, targets, ignore_mask=None):
        spec = LossSpec(sem_mode=1, ignore_index=int(self.ignore_index), lambda_s=float(self.lambda_s),
                        n_classes=_n_classes(inputs))
        return spec, {"labels": targets, "mask": ignore_mask}, ["coarse_semantic"]

    def forward(self, inputs, targets, ignore_mask=None):
        return run_plans([self.plan(inputs, targets, ignore_mask)], inputs)


class SemanticUncertaintyLoss(torch.nn.Module):
    def __init__(self, lambda_s, car_index, detach_beta_for_s=False, ignore_car_index=False):
        super().__init__()
        self.lambda_s = lambda_s
        self.ignore_index = car_index if ignore_car_index else -100
        self.detach_beta_for_s = detach_beta_for_s

    def plan(self, inputs, targets, ignore_mask=None):
        sbeta = "beta_semantic_coarse" in inputs
        spec = LossSpec(sem_mode=2, ignore_index=int(self.ignore_index), lambda_s=float(self.lambda_s),
                        use_sbeta=sbeta, detach_beta_for_s=bool(self.detach_beta_for_s), n_classes=_n_classes(inputs))
        return spec, {"labels": targets, "mask": ignore_mask}, ["coarse_semantic"] + (["coarse_semantic_logbeta"] if sbeta else [])

    def forward(self, inputs, targets, ignore_mask=None):
        return run_plans([self.plan(inputs, targets, ignore_mask)], inputs)


class SemanticCarRegLoss(torch.nn.Module):
    """L_t: lambda_c * MSE(1, sum_j w_j beta_j) over non-ignored rays labelled car (loss.py:117-157)"""

    def __init__(self, lambda_c, car_label):
        super().__init__()
        self.lambda_c = lambda_c
        self.car_label = car_label

    def plan(self, inputs, targets, ignore_mask=None):
        return (LossSpec(car_reg=True, car_label=int(self.car_label), lambda_c=float(self.lambda_c)),
                {"labels": targets, "mask": ignore_mask}, ["coarse_car_reg_loss"])

    def forward(self, inputs, targets, ignore_mask=None):
        return run_plans([self.plan(inputs, targets, ignore_mask)], inputs)
