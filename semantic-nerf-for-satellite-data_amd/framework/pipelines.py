"""Pipeline base, plugin loader and training loop -- mirror of framework/pipelines.py:22-352.

Lightning is not part of this build: `Pipeline` is a plain nn.Module exposing the hooks the reference's
LightningModule subclass defines (_init_datasets/_init_loss/_init_models/_init_renderer/_init_visualizers/
_init_training_step, models registered as attributes model_<key> so state_dict keys keep the
`model_<key>.` prefix that load_ckpoint.py:94-129 filters on), and `run_pipeline` is a small
single-process-per-GPU loop (Adam + StepLR per epoch, gradient all-reduce under torch.distributed)."""
import abc
import collections
import importlib
import os
import time

import torch

from .. import ops, parallel


class Pipeline(torch.nn.Module):
    def __init__(self, cfgs, ckpt_info=None) -> None:
        super().__init__()
        self.optimizer = None
        self.cfgs = cfgs
        self.train_steps = 0
        self.current_epoch = 0
        self.epoch_from_ckpt = None
        if ckpt_info is not None:
            self.epoch_from_ckpt, self.train_steps = ckpt_info
        self.logged = {}
        self.log_metrics = True
        self.datasets = self._init_datasets()
        assert "rgb" in self.datasets and "rgb_test" in self.datasets, "need both rgb and rgb_test datasets in pipeline"
        self.models = self.__init_models()
        self.renderer = self._init_renderer()
        self.visualizers = self._init_visualizers()
        self._training_step = self._init_training_step()
        self._time_of_last_step = None
        self._init_loss()

    # ---- Lightning-compatible helpers -------------------------------------------------------------
    def log(self, name, value, **kwargs):
        self.logged[name] = value

    def get_current_epoch(self):
        if self.epoch_from_ckpt is not None:
            return self.epoch_from_ckpt
        return self.current_epoch

    def get_current_progress(self, tstep=None):
        if tstep is None:
            tstep = self.train_steps
        return tstep / self.cfgs.run.max_train_steps

    def load_datasets(self):
        dev = next(self.parameters()).device
        for ds in self.datasets.values():
            ds.to(dev)

    def configure_optimizers(self):
        assert False, "needs to be implemented by subclass"

    def forward(self, data: dict):
        assert False, "needs to be implemented by sub class"

    def training_step(self, batch, batch_idx):
        assert False, "needs to be implemented by sub class"

    # ---- hooks ------------------------------------------------------------------------------------
    @abc.abstractmethod
    def _init_datasets(self) -> dict:
        pass

    @abc.abstractmethod
    def _init_loss(self):
        pass

    def __init_models(self) -> dict:
        models = self._init_models()
        for key in models.keys():
            setattr(self, f"model_{key}", models[key])
            models[key] = getattr(self, f"model_{key}")
        return models

    @abc.abstractmethod
    def _init_models(self) -> dict:
        pass

    @abc.abstractmethod
    def _init_renderer(self):
        pass

    def _init_visualizers(self) -> list:
        return []  # visualisers are out of scope

    @abc.abstractmethod
    def _init_training_step(self):
        pass


def load_pipeline(cfgs, ckpt_info=None) -> Pipeline:
    """Instantiate the class named by `pipeline = "pkg.mod.Class"` (framework/pipelines.py:341-352)."""
    name = cfgs.pipeline.pipeline.split(".")
    module = importlib.import_module(".".join(name[:-1]))
    return getattr(module, name[-1])(cfgs, ckpt_info=ckpt_info)


class TrainLoop:
    """One optimiser step = sample batch (GPU ray bank) -> training_step (main + sc forward, losses) ->
    backward -> one flat gradient all-reduce (RCCL) -> Adam; StepLR(0.9) at epoch boundaries
    (base_ray_pipeline.py:246-269, framework/util/train_util.py:45-60)."""

    def __init__(self, pipeline, cfgs, device=None):
        self.rank, self.world = parallel.world()
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device())
        self.pipeline, self.cfgs, self.device = pipeline, cfgs, device
        pipeline.to(device)
        pipeline.load_datasets()
        opt = pipeline.configure_optimizers()
        self.optimizer, self.scheduler = opt["optimizer"], opt["lr_scheduler"]["scheduler"]
        self.global_batch = cfgs.pipeline.batch_size
        self.bank = pipeline.datasets["rgb"]
        self.steps_per_epoch = self.bank.steps_per_epoch(self.global_batch)
        self.params = [p for p in pipeline.parameters() if p.requires_grad]
        self.bucket = None
        self.shuffle = bool(cfgs.run.shuffle_dataset)
        self.exchange_events = None      # bench: a list -> (start, end) HIP events around the gradient all-reduce of every step
        # The host issues a step in ~2 ms, the device runs it in ~26: unchecked, the host runs ahead until the HIP queue is full, and
        # every step it is ahead keeps that step's result and gradient tensors alive in the caching allocator (more device memory,
        # fresh hipMallocs deep into a run).  The loop therefore waits, before issuing step n, for the END of step n - MAX_LEAD:
        # the device always has a whole step queued behind the running one, the host is never further ahead than that.
        self.max_lead = int(os.environ.get("SNERF_MAX_LEAD", "2"))
        self._in_flight = collections.deque()
        # The reference's DataLoader workers have the next batch ready when a step ends.  Here the batch is five row gathers from
        # the HBM-resident ray bank -- tiny kernels that depend on nothing the optimiser writes, but on the compute stream they sat
        # between Adam and the first kernel of the next forward.  They are issued for step n + 1 on a data stream of their own when
        # step n has been issued, run beside its kernels, and the next step only waits for their event.  (SNERF_PREFETCH=0: in line.)
        self.prefetch = os.environ.get("SNERF_PREFETCH", "1") != "0" and self.device.type == "cuda"
        self._data_stream = torch.cuda.Stream(device=self.device) if self.prefetch else None
        self._next = None                # (step, batch, event) issued ahead on the data stream

    def _make_batch(self, step: int):
        batch = {"rgb": self.bank.batch(step, self.global_batch, self.rank, self.world, shuffle=self.shuffle)}
        if "depth" in self.pipeline.datasets:
            batch["depth"] = self.pipeline.datasets["depth"].batch(step, self.global_batch, self.rank, self.world, shuffle=self.shuffle)
        return batch

    def _issue_batch(self, step: int):
        """the batch of `step` on the data stream; the event that marks it ready"""
        if self._next is None or self._next[0] != step - 1:
            # first use, or the caller jumped (resume, evaluation in between): behind whatever built, moved or touched the bank
            # on the compute stream since.  The steady state (step n + 1 issued right after step n) needs no such wait.
            self._data_stream.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(self._data_stream):
            batch = self._make_batch(step)
            ev = torch.cuda.Event()
            ev.record(self._data_stream)
        return step, batch, ev

    def _take_batch(self, step: int):
        if not self.prefetch:
            return self._make_batch(step)
        if self._next is None or self._next[0] != step:     # first step, or the caller jumped: issue it now
            self._next = self._issue_batch(step)
        _, batch, ev = self._next
        main = torch.cuda.current_stream(self.device)
        # The batch was issued a whole step ago: normally its event has long completed and the HOST can see that -- then no
        # device-side wait is queued (a cross-stream wait in front of the step's first kernels costs ~25 us of queue bubble even when
        # it is already satisfied); only if the data stream is really behind does the compute stream wait for it.
        if not ev.query():
            main.wait_event(ev)
        for part in batch.values():
            for t in part.values():
                if torch.is_tensor(t):
                    t.record_stream(main)      # allocated on the data stream, consumed (and freed) on the compute stream
        return batch

    def step(self, step: int):
        pl = self.pipeline
        while self.max_lead > 0 and len(self._in_flight) >= self.max_lead:
            self._in_flight.popleft().synchronize()
        pl.current_epoch = step // self.steps_per_epoch
        batch = self._take_batch(step)
        self.optimizer.zero_grad()
        pl.logged = {k: v for k, v in pl.logged.items() if not k.startswith("train/")}   # a step reports only what IT logged (no stale terms of dropped losses)
        out = pl.training_step(batch, step)
        with ops.accumulate_into_sinks():   # a plain accumulate-into-.grad backward: the passes may write the optimiser's bucket directly
            out["loss"].backward()
        if hasattr(self.optimizer, "flat_g"):
            # every .grad is a view of the optimiser's flat gradient buffer: it is the all-reduce bucket
            self.optimizer._collect_foreign_grads()
            if self.exchange_events is not None and self.world > 1:
                ops.wait_grad_sinks(self.device)      # what the collective waits for anyway: keep it out of the measured interval
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
                self.bucket = parallel.allreduce_sum_(self.optimizer.flat_g)
                ev[1].record()
                self.exchange_events.append(ev)
            else:
                self.bucket = parallel.allreduce_sum_(self.optimizer.flat_g)
        else:
            self.bucket = parallel.allreduce_gradients(self.params, self.bucket)
        self.optimizer.step()
        if (step + 1) % self.steps_per_epoch == 0:
            self.scheduler.step()
        if self.max_lead > 0 and self.device.type == "cuda":
            ev = torch.cuda.Event()
            ev.record()
            self._in_flight.append(ev)
        if self.prefetch:
            self._next = self._issue_batch(step + 1)
        return out


    # ---- validation (pl.Trainer's val loop over val_dataloader(): framework/pipelines.py:120-129,316-318) ---------
    @torch.no_grad()
    def validate(self, rays_per_image: int = None, max_images: int = None):
        """Every image of the `rgb_test` bank through pipeline.validation_step; returns the split's means (loss, PSNR,
        semantic accuracy, per-image mIoU), the split-wide row-normalised confusion matrix and its mIoU
        (eval/eval_semantic.py:63-77,122-140) -- one host read at the very end.  Ranks share each image's rays."""
        pl = self.pipeline
        bank = pl.datasets["rgb_test"]
        # NOTE: the package's only bank factory is the synthetic one (GpuRayBank.synthetic, no image sizes), so from the product's
        # own data path the means below are per-SLICE means; a loader of real data passes `image_sizes` to GpuRayBank.
        # images: the bank's own per-image ray counts where it carries them (real data: the reference's per-image means,
        # framework/pipelines.py:120-129, eval_semantic.py:63-77); else equal slices of `rays_per_image` rows -- a SYNTHETIC
        # definition of "image" (the means are then per-slice means)
        hw = rays_per_image or (None if getattr(bank, "image_sizes", None) else min(len(bank), 64 * 64))
        n = bank.n_images(hw)
        if max_images is not None:
            n = min(n, max_images)
        acc, cm = {}, None
        for i in range(n):
            b = dict(bank.image(i, hw, self.rank, self.world), split="test")
            out = pl.validation_step(b, i)
            sse_cnt = torch.stack([out["sse"], out["count"]])
            parallel.allreduce_sum_(sse_cnt)
            vals = {"loss": out["loss"], "psnr": -10.0 * torch.log10(sse_cnt[0] / sse_cnt[1])}
            if "confusion_counts" in out:
                from ..semantic.components import metrics as M
                c = parallel.allreduce_sum_(out["confusion_counts"].clone())
                cm = c if cm is None else cm + c
                vals["mIoU"] = M.semantic_mIoU(c).to(torch.float32)
                vals["semantic_accuracy"] = torch.diagonal(c).sum() / c.sum()
            for k, v in vals.items():
                acc[k] = v if k not in acc else acc[k] + v
        res = {f"test/{k}": float(v) / max(n, 1) for k, v in acc.items()}
        if cm is not None:
            from ..semantic.components import metrics as M
            rows = cm.sum(dim=1, keepdim=True)
            res["test/confusion_matrix"] = torch.where(rows > 0, cm / rows.clamp_min(1.0), torch.zeros_like(cm)).cpu()
            res["test/mIoU_split"] = float(M.semantic_mIoU(cm))
        for k, v in res.items():
            if not torch.is_tensor(v):
                pl.log(k, v)
        return res

    # ---- checkpoints (Lightning layout; framework/util/load_ckpoint.py) ----------------------------------------
    def save_ckpoint(self, checkpoint_fp):
        from .util.load_ckpoint import save_ckpoint
        return save_ckpoint(self.pipeline, checkpoint_fp, self.optimizer, self.scheduler,
                            epoch=self.pipeline.train_steps // self.steps_per_epoch, global_step=self.pipeline.train_steps)

    def load_ckpoint(self, checkpoint_fp):
        """Resume: weights, Adam moments + step count, StepLR epoch, train_steps (trainer.fit(ckpt_path=...),
        framework/pipelines.py:321-331)."""
        from .util import load_ckpoint as lc
        ck = lc._safe_load(checkpoint_fp, self.device)
        for key, m in self.pipeline.models.items():
            lc.load_ckpoint(m, checkpoint_fp, f"model_{key}", self.device)
        if ck.get("optimizer_states"):
            self.optimizer.load_state_dict(ck["optimizer_states"][0])
        if ck.get("lr_schedulers"):
            self.scheduler.load_state_dict(ck["lr_schedulers"][0])
        self.pipeline.train_steps = int(ck.get("global_step", 0))
        return self.pipeline.train_steps


def run_pipeline(pipeline, cfgs, device=None, max_steps=None, on_step=None, on_validation=None):
    """Training loop (replaces pl.Trainer.fit of framework/pipelines.py:238-331): optimiser steps, and after every
    `check_val_every_n_epoch`-th epoch the validation loop over the test images (:316-318)."""
    loop = TrainLoop(pipeline, cfgs, device)
    steps = max_steps if max_steps is not None else cfgs.run.max_train_steps
    every = int(cfgs.run.check_val_every_n_epoch)
    t0 = time.time()
    for step in range(pipeline.train_steps, steps):
        out = loop.step(step)
        if on_step is not None:
            on_step(step, out)
        if every > 0 and (step + 1) % loop.steps_per_epoch == 0 and ((step + 1) // loop.steps_per_epoch) % every == 0:
            val = loop.validate()
            if on_validation is not None:
                on_validation(step, val)
    return time.time() - t0
