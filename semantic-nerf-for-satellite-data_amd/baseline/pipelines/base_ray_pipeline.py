"""Ray pipeline -- mirror of baseline/pipelines/base_ray_pipeline.py:14-269: forward = ray-chunk loop over
render_chunk_size rays with key-wise concatenation, training_step wrapper, validation_step (full-image render under
no_grad incl. the solar-correction pass -> loss -> PSNR; visualisers / SSIM / DSM-MAE are out of scope), Adam + StepLR."""
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

    # ---- validation (base_ray_pipeline.py:101-193) -----------------------------------------------------------
    def _val_render_options(self, split):
        return {}

    def _val_result_keys(self):
        """results the validation loss and metrics read (the reference renders all fifteen and keeps them)"""
        keys = ["rgb_coarse", "depth_coarse", "weights_coarse", "beta_coarse"]
        if self.cfgs.pipeline.sc_lambda > 0:
            keys += ["weights_sc_coarse", "transparency_sc_coarse", "sun_sc_coarse"]
        return keys

    @torch.no_grad()
    def validation_step(self, batch, batch_idx):
        """One image: full-frame forward without gradients (lean: only the results the loss and the metrics read, written
        in place chunk by chunk; the solar-correction pass runs because the loss has its terms), `self.loss`, PSNR.
        Returns {"loss", "psnr", "sse", "count", <loss_dict>} as 0-d device tensors (no host sync); only the test split
        contributes to the logged test/loss and test/psnr (:160-163).  Under data parallelism `batch` holds this
        rank's slice of the image: the loss kernels all-reduce their sums and counts, and the caller forms the
        image PSNR from the summed (sse, count)."""
        from ...eval.utils.util import lean_inference
        from ...eval.utils.metrics import sum_squared_error
        split = batch.get("split", "test")
        rays, rgbs, extras = batch["rays"], batch["rgbs"], batch["extras"]
        rays, rgbs, extras = rays.reshape(-1, rays.shape[-1]), rgbs.reshape(-1, 3), extras.reshape(-1, extras.shape[-1])
        assert rays.shape[0] == rgbs.shape[0], "Rays&RGBs shape dont match (validation step)"
        opts = dict(self._val_render_options(split))
        results = lean_inference(self.cfgs, self.renderer, self.models, rays, extras, keys=self._val_result_keys(),
                                 render_options=opts)
        loss, loss_dict = self.loss(results, rgbs)
        sse, count = sum_squared_error(results["rgb_coarse"], rgbs)
        out = {"loss": loss, "sse": sse, "count": count, "psnr": -10.0 * torch.log10(sse / count), "results": results}
        out.update(loss_dict)
        if split == "test":
            self.log("test/loss", loss, batch_size=1)
            self.log("test/psnr", out["psnr"], batch_size=1)
        return out

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
