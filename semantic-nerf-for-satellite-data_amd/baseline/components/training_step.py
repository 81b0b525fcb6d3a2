"""Sat-NeRF training step -- mirror of baseline/components/training_step.py:19-59 (loss gating)."""
import torch

from ...framework.components.training_step import BaseTrainingStep


def color_and_depth_losses(pipeline, batch, results):
    """colour loss by epoch (SNerfLoss before `first_beta_epoch`, SatNerfLoss after) + the depth-supervision branch while
    `train_steps < ds_drop`; the log keys and the gate order are the reference's (they are what its TensorBoard shows)."""
    pc = pipeline.cfgs.pipeline
    with_beta = pipeline.get_current_epoch() >= pc.first_beta_epoch
    total, terms = (pipeline.loss if with_beta else pipeline.loss_without_beta)(results, batch["rgb"]["rgbs"])
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
