"""Sat-NeRF pipeline + config -- mirror of baseline/pipelines/satnerf.py:23-132 and the config chain
NeRFConfig (baseline/pipelines/nerf.py:63-88) <- SNeRFConfig (snerf.py:67-68) <- SatNeRFConfig."""
from typing import List, Literal, Optional, Union

import numpy as np
import torch
from pydantic import BaseModel

from ...framework.datasets import GpuRayBank
from ..components.loss import SNerfLoss, SatNerfLoss, DepthLoss
from ..components.rendering import SatNeRFRendering
from ..components.training_step import SatNeRFTrainingStep
from ..models.satnerf import SatNeRF
from .base_ray_pipeline import BaseRayPipeline


class NeRFConfig(BaseModel):
    pipeline: Optional[str] = None
    precision: int = 32
    mfma_precision: Literal["f16x2", "f16x1", "bf16", "auto"] = "f16x2"  # build extension: ops.mfma_mode
    use_utm_coordinate_system: Union[bool, int] = False
    version: int = 1
    n_samples: int = 64
    use_fine_network: Union[bool, int] = False
    n_importance: int = 0
    render_chunk_size: int = 5120
    batch_size: int = 1024
    learnrate: float = 5e-4
    noise_std: float = 0.0
    activation_function: Literal["siren", "relu"] = "siren"
    mapping_pos_n_freq: int = 10
    mapping_dir_n_freq: int = 4
    fc_units: int = 512
    fc_layers: int = 8
    fc_skips: List[int] = [4]
    ray_subsampling_activated: Union[bool, int] = False
    ray_subsampling_amount: float = 1.0


class SNeRFConfig(NeRFConfig):
    sc_lambda: float = 0.05


class SatNeRFConfig(SNeRFConfig):
    fc_use_full_features: Union[bool, int] = False
    depth_enabled: Union[bool, int] = True
    depth_supervision_drop: float = 0.25
    ds_lambda: int = 1000
    first_beta_epoch: int = 2
    t_embedding_vocab: int = 50
    t_embedding_tau: int = 4
    ds_noweights: Union[bool, int] = False


class SatNeRFPipeline(BaseRayPipeline):
    def __init__(self, cfgs, ckpt_info=None) -> None:
        super().__init__(cfgs, ckpt_info)
        if self.cfgs.pipeline.depth_enabled:
            self.ds_drop = np.round(self.cfgs.pipeline.depth_supervision_drop * self.cfgs.run.max_train_steps)

    def _n_classes(self):
        return 5

    def _init_datasets(self) -> dict:
        r = self.cfgs.run
        d = {"rgb": GpuRayBank.synthetic(r.synthetic_rays, r.synthetic_images, self._n_classes(), r.synthetic_seed),
             "rgb_test": GpuRayBank.synthetic(4096, r.synthetic_images, self._n_classes(), r.synthetic_seed + 1)}
        if self.cfgs.pipeline.depth_enabled:
            d["depth"] = GpuRayBank.synthetic(max(r.synthetic_rays // 8, self.cfgs.pipeline.batch_size),
                                              r.synthetic_images, self._n_classes(), r.synthetic_seed + 2, depth=True)
        return d

    def _init_loss(self):
        self.loss = SatNerfLoss(lambda_sc=self.cfgs.pipeline.sc_lambda)
        self.loss_without_beta = SNerfLoss(lambda_sc=self.cfgs.pipeline.sc_lambda)
        if self.cfgs.pipeline.depth_enabled:
            self.depth_loss = DepthLoss(lambda_ds=self.cfgs.pipeline.ds_lambda)

    def _init_models(self) -> dict:
        pc = self.cfgs.pipeline
        return {"coarse": SatNeRF(self.cfgs, layers=pc.fc_layers, feat=pc.fc_units, skips=pc.fc_skips,
                                  siren=pc.activation_function == "siren", t_embedding_dims=pc.t_embedding_tau),
                "t": torch.nn.Embedding(pc.t_embedding_vocab, pc.t_embedding_tau)}

    def _init_renderer(self):
        return SatNeRFRendering(self.cfgs)

    def _init_training_step(self):
        return SatNeRFTrainingStep()

    @classmethod
    def init_config(cls, cfg_information):
        return SatNeRFConfig(**cfg_information)
