"""Semantic training step -- mirror of semantic/components/training_step.py:10-99: which loss fires when
(epoch < first_beta_epoch, train_steps < ds_drop, use_beta_for_s, use_car_reg_loss & epoch >= car_reg_loss_start)."""
from ...baseline.components.training_step import color_and_depth_losses
from ...framework.components.training_step import BaseTrainingStep
from .metrics import semantic_accuracy


class RSSemanticTrainingStep(BaseTrainingStep):
    def training_step(self, pipeline, batch, batch_idx):
        pc = pipeline.cfgs.pipeline
        results = pipeline({"rays": batch["rgb"]["rays"], "extras": batch["rgb"]["extras"]})
        loss, loss_dict = color_and_depth_losses(pipeline, batch, results)
        labels = batch["rgb"]["semantic"]
        mask = batch["rgb"].get("semantic_sparsity_mask")
        if pipeline.get_current_epoch() < pc.first_beta_epoch or not pc.use_beta_for_s:
            semantic_loss, semantic_loss_dict = pipeline.semantic_loss(results, labels, mask)
            pipeline.log("train/semantic_beta_loss_activated", 0.0)
        else:
            semantic_loss, semantic_loss_dict = pipeline.uncertainty_semantic_loss(results, labels, mask)
            pipeline.log("train/semantic_beta_loss_activated", 1.0)
        loss = loss + semantic_loss
        loss_dict.update(semantic_loss_dict)
        if pc.use_car_reg_loss and pipeline.get_current_epoch() >= pc.car_reg_loss_start:
            car_reg_loss, car_reg_loss_dict = pipeline.car_reg_loss(results, labels, mask)
            loss = loss + car_reg_loss
            loss_dict.update(car_reg_loss_dict)
            pipeline.log("train/car_reg_loss_activated", 1.0)
        if pipeline.log_metrics:
            pipeline.log("train/semantic_accuracy", semantic_accuracy(results, labels))
        return results, loss, loss_dict
