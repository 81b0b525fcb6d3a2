"""Sat-NeRF training step -- mirror of baseline/components/training_step.py:19-59 (loss gating)."""
import torch

from ...framework.components.training_step import BaseTrainingStep


def color_and_depth_losses(pipeline, batch, results, more_plans=()):
    """colour loss by epoch (SNerfLoss before `first_beta_epoch`, SatNerfLoss after) + the depth-supervision branch while
    `train_steps < ds_drop`; the log keys and the gate order are the reference's (they are what its TensorBoard shows).
    `more_plans`: plans of further loss modules on the SAME rendered tensors (the semantic step's: loss_ops.run_plans evaluates them
    with the colour loss in one fused call); their terms come back in the same dict, their sum in the same total."""
    from ...loss_ops import run_plans
    pc = pipeline.cfgs.pipeline
    with_beta = pipeline.get_current_epoch() >= pc.first_beta_epoch
    color = pipeline.loss if with_beta else pipeline.loss_without_beta
    if hasattr(color, "plan"):
        total, terms = run_plans([color.plan(results, batch["rgb"]["rgbs"])] + list(more_plans), results)
    else:      # a foreign colour loss module: called as the reference calls it, the other modules' plans on their own
        total, terms = color(results, batch["rgb"]["rgbs"])
        if more_plans:
            t2, d2 = run_plans(list(more_plans), results)
            total = total + t2
            terms.update(d2)
    pipeline.log("train/beta_loss_activated", 1.0 if with_beta else 0.0)
    if not pc.depth_enabled:
        return total, terms
    depth_on = pipeline.train_steps < pipeline.ds_drop
    pipeline.log("train/depth_loss_activated", 1.0 if depth_on else 0.0)
    if depth_on:
        d = batch["depth"]
        rendered = pipeline({"rays": d["rays"], "extras": d["extras"]})      # a second full forward on the depth rays
        targets = d["depths"][:, 0].reshape(-1)
        ray_weights = 1.0 if pc.ds_noweights else d["weights"].reshape(-1)
        depth_total, depth_terms = pipeline.depth_loss(rendered, targets, ray_weights)
        total = total + depth_total
        terms.update(depth_terms)
    return total, terms


class SatNeRFTrainingStep(BaseTrainingStep):
    def training_step(self, pipeline, batch, batch_idx):
        results = pipeline({"rays": batch["rgb"]["rays"], "extras": batch["rgb"]["extras"]})
        loss, loss_dict = color_and_depth_losses(pipeline, batch, results)
        return results, loss, loss_dict
