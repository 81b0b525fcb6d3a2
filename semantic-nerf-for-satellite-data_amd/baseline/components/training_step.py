"""Sat-NeRF training step -- mirror of baseline/components/training_step.py:19-59 (loss gating)."""
import torch

from ...framework.components.training_step import BaseTrainingStep


def color_and_depth_losses(pipeline, batch, results):
    rgbs = batch["rgb"]["rgbs"]
    if pipeline.get_current_epoch() < pipeline.cfgs.pipeline.first_beta_epoch:
        loss, loss_dict = pipeline.loss_without_beta(results, rgbs)
        pipeline.log("train/beta_loss_activated", 0.0)
    else:
        loss, loss_dict = pipeline.loss(results, rgbs)
        pipeline.log("train/beta_loss_activated", 1.0)
    if pipeline.cfgs.pipeline.depth_enabled:
        if pipeline.train_steps < pipeline.ds_drop:
            tmp = pipeline({"rays": batch["depth"]["rays"], "extras": batch["depth"]["extras"]})
            kp_depths = torch.flatten(batch["depth"]["depths"][:, 0])
            kp_weights = 1.0 if pipeline.cfgs.pipeline.ds_noweights else torch.flatten(batch["depth"]["weights"])
            loss_depth, loss_dict_depth = pipeline.depth_loss(tmp, kp_depths, kp_weights)
            loss = loss + loss_depth
            loss_dict.update(loss_dict_depth)
            pipeline.log("train/depth_loss_activated", 1.0)
        else:
            pipeline.log("train/depth_loss_activated", 0.0)
    return loss, loss_dict


class SatNeRFTrainingStep(BaseTrainingStep):
    def training_step(self, pipeline, batch, batch_idx):
        results = pipeline({"rays": batch["rgb"]["rays"], "extras": batch["rgb"]["extras"]})
        loss, loss_dict = color_and_depth_losses(pipeline, batch, results)
        return results, loss, loss_dict
