"""Ray pipeline -- mirror of baseline/pipelines/base_ray_pipeline.py:14-269: forward = ray-chunk loop over
render_chunk_size rays with key-wise concatenation, training_step wrapper, Adam + StepLR."""
import time
from collections import defaultdict

import torch

from ...eval.utils.metrics import psnr
from ...framework.pipelines import Pipeline


class BaseRayPipeline(Pipeline):
    def forward(self, data: dict, render_options: dict = None):
        rays, extras = data["rays"], data["extras"]
        epoch = data.get("epoch", self.get_current_epoch())
        progress = data.get("progress", self.get_current_progress())
        chunk = self.cfgs.pipeline.render_chunk_size
        n = rays.shape[0]
        parts = defaultdict(list)
        for i in range(0, n, chunk):
            r = self.renderer.render_rays(self.models, rays[i:i + chunk], extras[i:i + chunk] if extras is not None else None,
                                          epoch=epoch, progress=progress, render_options=render_options or {})
            for k, v in r.items():
                if v is not None:
                    parts[k].append(v)
        # one concatenation per key (the reference re-concatenates the growing tensor per chunk: O(chunks^2))
        return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in parts.items()}

    def training_step(self, batch, batch_idx):
        if self.log_metrics and self.optimizer is not None:
            self.log("lr", self.optimizer.param_groups[0]["lr"])
        self.train_steps += 1
        batch_size = batch["rgb"]["rays"].shape[0]
        results, loss, loss_dict = self._training_step.training_step(self, batch, batch_idx)
        self.log("train/loss", loss, batch_size=batch_size)
        for k in loss_dict.keys():
            self.log("train/{}".format(k), loss_dict[k], batch_size=batch_size)
        if self.log_metrics:
            with torch.no_grad():
                self.log("train/psnr", psnr(results["rgb_coarse"], batch["rgb"]["rgbs"]))
            now = time.time()
            if self._time_of_last_step is not None:
                self.log("train/time_since_last_step", now - self._time_of_last_step)
            self._time_of_last_step = now
        return {"loss": loss}

    def configure_optimizers(self):
        # same optimiser and schedule as the reference (:246-269), as ONE fused HIP launch over a flat parameter
        # buffer whose flat gradient twin doubles as the all-reduce bucket (optim.py)
        from ...optim import FlatAdam, StepLR
        params = [p for m in self.models.values() for p in m.parameters()]
        import os
        if os.environ.get("SNERF_TORCH_ADAM") == "1":  # diagnostics only: stock multi-tensor Adam for A/B timing
            self.optimizer = torch.optim.Adam(params, lr=self.cfgs.pipeline.learnrate, weight_decay=0)
            return {"optimizer": self.optimizer, "lr_scheduler": {"scheduler": torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=1, gamma=0.9), "interval": "epoch"}}
        self.optimizer = FlatAdam(params, lr=self.cfgs.pipeline.learnrate, weight_decay=0)
        scheduler = StepLR(self.optimizer, step_size=1, gamma=0.9)
        return {"optimizer": self.optimizer, "lr_scheduler": {"scheduler": scheduler, "interval": "epoch"}}
