"""Semantic training step -- mirror of semantic/components/training_step.py:10-99: which loss fires when
(epoch < first_beta_epoch, train_steps < ds_drop, use_beta_for_s, use_car_reg_loss & epoch >= car_reg_loss_start)."""
from ...baseline.components.training_step import color_and_depth_losses
from ...framework.components.training_step import BaseTrainingStep
from .metrics import semantic_accuracy


class RSSemanticTrainingStep(BaseTrainingStep):
    def training_step(self, pipeline, batch, batch_idx):
        pc = pipeline.cfgs.pipeline
        results = pipeline({"rays": batch["rgb"]["rays"], "extras": batch["rgb"]["extras"]})
        labels = batch["rgb"]["semantic"]
        mask = batch["rgb"].get("semantic_sparsity_mask")
        # the reference's gates and log keys; the gated modules hand over their PLANS and are evaluated with the colour loss in
        # one fused call (loss_ops.run_plans): loss = colour + semantic (+ L_t), loss_dict = the union of their terms, as before
        if pipeline.get_current_epoch() < pc.first_beta_epoch or not pc.use_beta_for_s:
            plans = [pipeline.semantic_loss.plan(results, labels, mask)]
            pipeline.log("train/semantic_beta_loss_activated", 0.0)
        else:
            plans = [pipeline.uncertainty_semantic_loss.plan(results, labels, mask)]
            pipeline.log("train/semantic_beta_loss_activated", 1.0)
        if pc.use_car_reg_loss and pipeline.get_current_epoch() >= pc.car_reg_loss_start:
            plans.append(pipeline.car_reg_loss.plan(results, labels, mask))
            pipeline.log("train/car_reg_loss_activated", 1.0)
        loss, loss_dict = color_and_depth_losses(pipeline, batch, results, more_plans=plans)
        if pipeline.log_metrics:
            pipeline.log("train/semantic_accuracy", semantic_accuracy(results, labels))
        return results, loss, loss_dict
