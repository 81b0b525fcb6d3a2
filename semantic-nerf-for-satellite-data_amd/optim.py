"""Optimiser of the ray pipelines: Adam(lr, weight_decay=0) + StepLR(step_size=1, gamma=0.9) per epoch
(baseline/pipelines/base_ray_pipeline.py:246-269), re-laid for the GPU:

* every parameter is a view into ONE flat fp32 buffer, every .grad a view into ONE flat gradient buffer -- the
  gradient buffer IS the RCCL all-reduce bucket (no gather / scatter copies) and zero_grad is one memset;
* the update is ONE HIP launch (snerf_adam_step) instead of torch's multi-tensor sequence over ~60 tensors;
* state_dict() / load_state_dict() speak torch.optim.Adam's format, so optimizer_states of a Lightning-style
  checkpoint written by the reference resume here and vice versa (framework/util/load_ckpoint.py).
"""
import ctypes as C
from typing import Iterable, List

import torch

from . import _lib


def _round4(n: int) -> int:
    return (n + 3) // 4 * 4


class FlatAdam:
    """torch.optim.Adam surface (param_groups, step, zero_grad, state_dict) over flat buffers."""

    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0, grad_sinks: bool = True):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdam: no trainable parameters")
        if weight_decay != 0:
            raise ValueError("FlatAdam: the reference runs Adam with weight_decay=0 (base_ray_pipeline.py:262)")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise RuntimeError("FlatAdam: parameters must live on the GPU -- the HIP path has no CPU fallback")
        for p in self.params:
            if p.device != dev or p.dtype != torch.float32:
                raise ValueError("FlatAdam: all parameters must be fp32 on one device")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += _round4(p.numel())       # every view 16-byte aligned
        self.numel = off
        self.flat_p = torch.zeros(off, device=dev)
        self.flat_g = torch.zeros(off, device=dev)
        self.exp_avg = torch.zeros(off, device=dev)
        self.exp_avg_sq = torch.zeros(off, device=dev)
        self._gviews = []
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                v = self.flat_p[o:o + p.numel()].view_as(p)
                v.copy_(p.data)
                p.data = v
                g = self.flat_g[o:o + p.numel()].view_as(p)
                p.grad = g
                self._gviews.append(g)
        # torch.optim.Adam counts steps PER PARAMETER and skips a parameter whose .grad is None; FlatAdam keeps the same
        # bookkeeping so that optimizer_states written by either side load into the other.  (In the pipelines every
        # parameter has a gradient every step -- the reference's concatenated model output makes autograd hand unused
        # heads exact zeros, not None -- so all counts are equal there and the update is one launch.)
        # the .grad views double as gradient sinks of the fused passes (ops.register_grad_sinks): while a parameter's
        # .grad IS its view, the passes accumulate into it directly and autograd launches no add kernel for it
        from . import ops
        self.sinks = bool(grad_sinks) and ops._SINKS_ON
        if self.sinks:
            ops.register_grad_sinks(self.params, self._gviews)
            # a gradient that plain autograd accumulates into a view in place (a loss outside the fused passes) marks the
            # parameter as written too, so that "still the zeroed view" can be told from "has a gradient"
            for p in self.params:
                p.register_post_accumulate_grad_hook(lambda q: ops._SINK_TOUCHED.add(id(q)))
        self.steps = [0] * len(self.params)
        self._active = None
        self.param_groups = [{"lr": float(lr), "betas": tuple(betas), "eps": float(eps), "weight_decay": 0.0,
                              "amsgrad": False, "params": list(range(len(self.params)))}]

    # ---- torch.optim surface ------------------------------------------------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        """With gradient sinks (the default): one memset, every .grad stays the view of the flat buffer that the fused
        passes accumulate into.  Without (SNERF_GRAD_SINKS=0): set_to_none (default, as torch) drops the .grad
        references -- autograd then hands over each pass's gradient tensors and step() gathers them into the flat buffer
        with one multi-tensor copy; set_to_none=False: one memset, autograd accumulates into the views in place."""
        self._active = None
        self._none_mode = bool(set_to_none)   # torch: zero_grad(set_to_none=False) leaves zero TENSORS, which Adam steps
        if self.sinks:
            from . import ops
            ops._SINK_TOUCHED.difference_update(id(p) for p in self.params)
        if set_to_none and not self.sinks:
            for p in self.params:
                p.grad = None
            return
        self.flat_g.zero_()
        for p, g in zip(self.params, self._gviews):
            if p.grad is not g:
                p.grad = g

    def _collect_foreign_grads(self):
        """Make flat_g hold every gradient: .grad tensors that are not views of it are gathered by one multi-tensor
        copy (and re-pointed at their view).  A parameter WITHOUT a gradient (autograd produced none since the last
        zero_grad) is remembered as inactive for this step -- its slice of the bucket is zeroed so that the flat
        all-reduce stays one call, and step() skips it as torch.optim.Adam does."""
        if self._active is not None:      # already gathered for this step (TrainLoop calls this before the all-reduce)
            return
        if self.sinks:
            from . import ops
            ops.wait_grad_sinks(self.flat_g.device)   # the sc pass accumulates on its own stream
        src, dst, active = [], [], []
        touched = None
        if self.sinks and getattr(self, "_none_mode", True):
            from . import ops
            touched = ops._SINK_TOUCHED
        for p, g in zip(self.params, self._gviews):
            # with sinks, a .grad that is still the (zeroed) view and that no pass has written counts as "no gradient"
            has = p.grad is not None and not (touched is not None and p.grad is g and id(p) not in touched)
            active.append(has)
            if not has:
                if p.grad is None:
                    g.zero_()     # (an untouched view is still zero from zero_grad's memset)
            elif p.grad.data_ptr() != g.data_ptr():
                src.append(p.grad.detach())
                dst.append(g)
            p.grad = g
        if src:
            torch._foreach_copy_(dst, src)
        self._active = active

    @property
    def step_count(self):
        """largest per-parameter step count (the shared count when every parameter has had a gradient every step)"""
        return max(self.steps)

    def _runs(self):
        """maximal runs of consecutive ACTIVE parameters with equal step count -> (first, last+1); one launch each
        (normally one run; two while a head is still waiting for its first gradient)"""
        runs, i, n = [], 0, len(self.params)
        while i < n:
            if not self._active[i]:
                i += 1
                continue
            j = i
            while j + 1 < n and self._active[j + 1] and self.steps[j + 1] == self.steps[i]:
                j += 1
            runs.append((i, j + 1))
            i = j + 1
        return runs

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        self._collect_foreign_grads()
        grp = self.param_groups[0]
        L = _lib.lib()
        st = torch.cuda.current_stream(self.flat_p.device).cuda_stream
        esz = 4
        for a, b in self._runs():
            lo = self.offsets[a]
            hi = self.offsets[b] if b < len(self.params) else self.numel
            for i in range(a, b):
                self.steps[i] += 1
            _lib.check(L.snerf_adam_step(self.flat_p.data_ptr() + lo * esz, self.flat_g.data_ptr() + lo * esz,
                                         self.exp_avg.data_ptr() + lo * esz, self.exp_avg_sq.data_ptr() + lo * esz,
                                         C.c_ulonglong(hi - lo), grp["lr"], grp["betas"][0], grp["betas"][1], grp["eps"],
                                         self.steps[a], float(grad_scale), C.c_void_p(st)), "snerf_adam_step")
        self._active = None

    def state_dict(self):
        """torch.optim.Adam's layout: state[i] exists only for parameters that have been stepped, each with its own
        `step`"""
        state = {}
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            if self.steps[i] == 0:
                continue
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.steps[i])),
                        "exp_avg": self.exp_avg[o:o + n].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[o:o + n].view_as(p).clone()}
        groups = [dict(g, betas=tuple(g["betas"])) for g in self.param_groups]
        return {"state": state, "param_groups": groups}

    @torch.no_grad()
    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        n_saved = sum(len(g["params"]) for g in groups)
        if n_saved != len(self.params):
            raise ValueError(f"FlatAdam: checkpoint holds {n_saved} parameters, the pipeline has {len(self.params)}")
        g0 = groups[0]
        self.param_groups[0].update(lr=float(g0["lr"]), betas=tuple(g0["betas"]), eps=float(g0["eps"]))
        ids = [i for g in groups for i in g["params"]]     # torch numbers the parameters group by group
        self.exp_avg.zero_()
        self.exp_avg_sq.zero_()
        for k, (p, o) in enumerate(zip(self.params, self.offsets)):
            s = sd["state"].get(ids[k])
            if s is None:                  # never stepped when the checkpoint was written (no gradient yet)
                self.steps[k] = 0
                continue
            n = p.numel()
            if tuple(s["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"FlatAdam: state {ids[k]} has shape {tuple(s['exp_avg'].shape)}, parameter {tuple(p.shape)}")
            self.exp_avg[o:o + n].view_as(p).copy_(s["exp_avg"])
            self.exp_avg_sq[o:o + n].view_as(p).copy_(s["exp_avg_sq"])
            self.steps[k] = int(float(s["step"]))
        self._active = None


class StepLR:
    """torch.optim.lr_scheduler.StepLR(optimizer, step_size, gamma): lr = base_lr * gamma ** (epoch // step_size);
    the pipelines step it once per epoch (base_ray_pipeline.py:264-268)."""

    def __init__(self, optimizer, step_size: int = 1, gamma: float = 0.9):
        self.optimizer, self.step_size, self.gamma = optimizer, int(step_size), float(gamma)
        self.base_lrs = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = 0

    def _apply(self):
        for g, b in zip(self.optimizer.param_groups, self.base_lrs):
            g["lr"] = b * self.gamma ** (self.last_epoch // self.step_size)

    def step(self):
        self.last_epoch += 1
        self._apply()

    def get_last_lr(self):
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self):
        return {"step_size": self.step_size, "gamma": self.gamma, "base_lrs": list(self.base_lrs),
                "last_epoch": self.last_epoch, "_last_lr": self.get_last_lr()}

    def load_state_dict(self, sd):
        self.step_size, self.gamma = int(sd["step_size"]), float(sd["gamma"])
        self.base_lrs, self.last_epoch = list(sd["base_lrs"]), int(sd["last_epoch"])
        self._apply()
