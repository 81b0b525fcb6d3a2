"""Semantic pipeline + config -- mirror of semantic/pipelines/rs_semantic.py:26-175."""
from typing import Literal, Union

import torch

from ...baseline.pipelines.satnerf import SatNeRFPipeline, SatNeRFConfig
from ..components.loss import SemanticLoss, SemanticUncertaintyLoss, SemanticCarRegLoss
from ..components.rendering import RSSemanticRendering
from ..components.training_step import RSSemanticTrainingStep
from ..models.rs_semantic import RSSemanticNeRF, inference


class RSSemanticConfig(SatNeRFConfig):
    lambda_s: float = 0.04
    semantic_dataset_type: Literal["own", "us3d", "own_corrupted"] = "own"
    sparsity_n_images: int = -1
    ignore_car_index: Union[bool, int] = False
    semantic_activation_function: Literal["none", "sigmoid"] = "sigmoid"
    use_tj_for_s: Union[bool, int] = False
    use_beta_for_s: Union[bool, int] = False
    use_tj_instead_of_beta: Union[bool, int] = False
    use_separate_beta_for_s: Union[bool, int] = False
    use_separate_tj_for_semantic: Union[bool, int] = False
    detach_beta_for_s: Union[bool, int] = False
    use_car_reg_loss: Union[bool, int] = False
    lambda_c: float = 1.0
    car_reg_loss_start: int = 3


class RSSemanticPipeline(SatNeRFPipeline):
    def __init__(self, cfgs, ckpt_info=None):
        super().__init__(cfgs, ckpt_info)
        if cfgs.pipeline.use_tj_instead_of_beta:
            cfgs.pipeline.first_beta_epoch = 10000000  # rs_semantic.py:30-32: disables the beta loss

    def _init_loss(self):
        super()._init_loss()
        pc, car = self.cfgs.pipeline, self.datasets["rgb"].car_cls_idx
        self.semantic_loss = SemanticLoss(pc.lambda_s, car, ignore_car_index=pc.ignore_car_index)
        self.uncertainty_semantic_loss = SemanticUncertaintyLoss(pc.lambda_s, car, detach_beta_for_s=pc.detach_beta_for_s,
                                                                 ignore_car_index=pc.ignore_car_index)
        if pc.use_car_reg_loss:
            self.car_reg_loss = SemanticCarRegLoss(pc.lambda_c, car)

    def _init_renderer(self):
        return RSSemanticRendering(self.cfgs, inference=inference)

    def _init_models(self) -> dict:
        pc = self.cfgs.pipeline
        d = {"coarse": RSSemanticNeRF(self.cfgs, self.datasets["rgb"]),
             "t": torch.nn.Embedding(pc.t_embedding_vocab, pc.t_embedding_tau)}
        if pc.use_separate_tj_for_semantic:
            d["t_s"] = torch.nn.Embedding(pc.t_embedding_vocab, pc.t_embedding_tau)
        return d

    def _init_training_step(self):
        return RSSemanticTrainingStep()

    # ---- validation: the base step plus the semantic metrics of eval/eval_semantic.py:66-100, on the device -------
    def _val_result_keys(self):
        return super()._val_result_keys() + ["semantic_label_coarse"]

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """adds semantic_accuracy, the (C, C) confusion COUNTS of this image (rows = ground truth; the caller sums them
        over the split -- eval_semantic.py:63,75-77 -- and over ranks), mIoU of this image and, when the image has car
        rays, the composited uncertainty at the transient class."""
        from ..components import metrics as M
        out = super().validation_step(batch, batch_idx)
        res = out["results"]
        gt = batch["semantic"].reshape(-1, 1)
        n_cls = self.datasets["rgb"].semantic_n_classes
        out["semantic_accuracy"] = M.semantic_accuracy(res, gt)
        out["confusion_counts"] = M.confusion_matrix_values(res, gt, n_cls, normalize=None)
        out["mIoU"] = M.semantic_mIoU(out["confusion_counts"])
        out["uncertainty_at_transient"] = M.uncertainty_at_transient(res, gt, self.datasets["rgb"].car_cls_idx)
        if batch.get("split", "test") == "test":
            self.log("test/semantic_accuracy", out["semantic_accuracy"], batch_size=1)
            self.log("test/mIoU", out["mIoU"], batch_size=1)
        return out

    @classmethod
    def init_config(cls, cfg_information):
        return RSSemanticConfig(**cfg_information)
