"""Fused HIP losses (snerf_loss_partial / snerf_loss_finish) as one autograd.Function.

The Function returns (total, terms): `total` is the differentiable sum of the active terms, `terms`
the eight loss_dict values (include/snerf_hip.h SNERF_TERM_*) for logging.  The gradients w.r.t. the
rendered tensors are produced by the same kernels in the forward call and scaled in backward.
Under data parallelism the per-ray sums/counts are all-reduced between the two phases so that means
with data-dependent denominators (CE over non-ignored rays, L_t over car rays) equal the single-GPU
result.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import torch

from . import _lib
from .ops import _ptr, _stream, _check_dev


@dataclass(frozen=True)
class LossSpec:
    color_mode: int = 0        # 0 none, 1 SNerfLoss, 2 SatNerfLoss
    has_sc: bool = False
    sem_mode: int = 0          # 0 none, 1 SemanticLoss, 2 SemanticUncertaintyLoss
    ignore_index: int = -100
    use_sbeta: bool = False
    detach_beta_for_s: bool = False
    car_reg: bool = False
    car_label: int = 4
    has_depth: bool = False
    sc_lambda: float = 0.0
    lambda_s: float = 0.0
    lambda_c: float = 0.0
    ds_lambda: float = 0.0
    n_classes: int = 0


_DIFF = ("rgb", "weights", "beta", "beta_semantic", "semantic_logits", "sun_sc", "depth")


def _dist_world():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_world_size()
    return 1


class _FusedLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, spec: LossSpec, aux: dict, sync: bool, *diff):
        L = _lib.lib()
        t = dict(zip(_DIFF, diff))
        ref = next(v for v in diff if v is not None)
        dev = ref.device
        N = ref.shape[0]
        S = 1
        for k in ("weights", "sun_sc"):
            if t[k] is not None:
                S = t[k].shape[1]
        cfg = _lib.SnerfLossCfg(
            n_rays=N, n_samples=S, n_classes=spec.n_classes, color_mode=spec.color_mode, has_sc=int(spec.has_sc),
            sem_mode=spec.sem_mode, ignore_index=spec.ignore_index, use_sbeta=int(spec.use_sbeta),
            detach_beta_for_s=int(spec.detach_beta_for_s), car_reg=int(spec.car_reg), car_label=spec.car_label,
            has_depth=int(spec.has_depth), sc_lambda=spec.sc_lambda, lambda_s=spec.lambda_s, lambda_c=spec.lambda_c,
            ds_lambda=spec.ds_lambda)
        li = _lib.SnerfLossIn()
        keep = []
        for k in _DIFF:
            if t[k] is not None:
                _check_dev(t[k], k)
                v = t[k].detach().contiguous()
                keep.append(v)
                setattr(li, k, v.data_ptr())
        for k in ("transparency_sc", "weights_sc", "gt_rgb", "gt_depth", "depth_weights"):
            v = aux.get(k)
            if v is not None:
                v = v.detach().to(torch.float32).contiguous()
                if not v.is_cuda:
                    raise RuntimeError(f"snerf_amd: loss input '{k}' must live on the GPU")
                keep.append(v)
                setattr(li, k, v.data_ptr())
        if aux.get("labels") is not None:
            v = aux["labels"].reshape(-1).to(torch.int64).contiguous()
            keep.append(v)
            li.labels = v.data_ptr()
        if aux.get("mask") is not None:
            v = aux["mask"].reshape(-1)
            v = (v.view(torch.uint8) if v.dtype == torch.bool else v.to(torch.uint8)).contiguous()   # bool -> uint8: a view, no launch
            keep.append(v)
            li.mask = v.data_ptr()
        nws = L.snerf_loss_workspace_bytes(C.byref(cfg))
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        totals = torch.empty(_lib.LOSS_NTOT, dtype=torch.float32, device=dev)
        terms = torch.empty(8, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _lib.check(L.snerf_loss_partial(C.byref(cfg), C.byref(li), _ptr(totals), _ptr(ws), nws, _stream()),
                       "snerf_loss_partial")
        n_global = float(N)
        if sync and _dist_world() > 1:
            from .parallel import allreduce_sum_
            allreduce_sum_(totals)           # sums and counts over all ranks (tiny: 16 floats)
            n_global = 0.0                   # = take the global ray count from totals (it is one of the summed counts):
            #                                  right for unequal shards too (e.g. the depth-ray bank)
        grads = {k: (torch.empty_like(t[k], memory_format=torch.contiguous_format) if t[k] is not None else None)
                 for k in _DIFF}
        lg = _lib.SnerfLossGrads()
        for k in _DIFF:
            if grads[k] is not None:
                setattr(lg, k, grads[k].data_ptr())
        with torch.cuda.device(dev):
            _lib.check(L.snerf_loss_finish(C.byref(cfg), C.byref(li), _ptr(totals), n_global, 1.0, _ptr(terms),
                                           C.byref(lg), _stream()), "snerf_loss_finish")
        ctx.grads = [grads[k] for k in _DIFF]
        ctx._keep = keep
        total = terms.sum()
        ctx.mark_non_differentiable(terms)
        return total, terms

    @staticmethod
    def backward(ctx, g_total, _g_terms):
        grads, ctx.grads = ctx.grads, None
        if grads is None:
            raise RuntimeError("snerf_amd: second backward through one fused loss (its gradient buffers were scaled in place by the first)")
        live = [g for g in grads if g is not None]
        if live:
            torch._foreach_mul_(live, g_total)       # one launch for all of them (the buffers are this node's own)
        return (None, None, None) + tuple(grads)


def fused_loss(spec: LossSpec, results: dict, aux: dict, typ: str = "coarse", sync: bool = True):
    """results: the renderer's dict (keys with `_coarse` suffix); aux: targets. Returns (total, terms)."""
    def get(k):
        return results.get(f"{k}_{typ}")
    need_wb = spec.color_mode == 2 or spec.sem_mode == 2 or spec.car_reg
    diff = [get("rgb") if spec.color_mode else None,
            get("weights") if need_wb else None,
            get("beta") if need_wb else None,
            get("beta_semantic") if (spec.use_sbeta and spec.sem_mode == 2) else None,
            get("semantic_logits") if spec.sem_mode else None,
            get("sun_sc") if spec.has_sc else None,
            get("depth") if spec.has_depth else None]
    aux = dict(aux)
    if spec.has_sc:
        aux["transparency_sc"] = get("transparency_sc")
        aux["weights_sc"] = get("weights_sc")
    return _FusedLoss.apply(spec, aux, sync, *diff)


# ---- several loss modules, one fused call ----------------------------------------------------------------------------
# The reference evaluates its losses module by module (SatNerfLoss, SemanticLoss, SemanticCarRegLoss: three passes over the same
# rendered tensors).  The fused kernels take ONE configuration with every term switched on or off, so modules whose terms do not
# collide run as one call: one partial / finish pair, one all-reduce of the totals under data parallelism, one backward scale --
# and the gradients of tensors several terms share (weights, beta) come out summed.  A module describes its call as a PLAN
# (LossSpec, aux dict, loss_dict keys); fused_loss(spec, ...) of a single plan is what its forward() does.
_MERGE = os.environ.get("SNERF_MERGE_LOSSES", "1") != "0"      # 0: module by module, as the reference evaluates them (A/B)
_GROUPS = (("color_mode", ("color_mode", "has_sc", "sc_lambda")),
           ("sem_mode", ("sem_mode", "ignore_index", "use_sbeta", "detach_beta_for_s", "lambda_s", "n_classes")),
           ("car_reg", ("car_reg", "car_label", "lambda_c")),
           ("has_depth", ("has_depth", "ds_lambda")))


def merge_plans(plans):
    """-> one (spec, aux, keys) equal to the SUM of the plans, or None when two of them own the same group of terms (two colour
    losses, say) or disagree about a target tensor."""
    plans = list(plans)
    if len(plans) == 1:
        return plans[0]
    fields, aux, keys = {}, {}, []
    for switch, names in _GROUPS:
        owners = [pl for pl in plans if getattr(pl[0], switch)]
        if len(owners) > 1:
            return None
        if owners:
            fields.update({n: getattr(owners[0][0], n) for n in names})
    for _spec, a, k in plans:
        for name, v in a.items():
            if name in aux and aux[name] is not v:      # (also: one module masked, the other not -- one call has ONE mask)
                return None
            aux[name] = v
        keys += [x for x in k if x not in keys]
    return LossSpec(**fields), aux, keys


def run_plans(plans, results, typ: str = "coarse", sync: bool = True):
    """Evaluate loss plans on one set of rendered tensors: merged into one fused call where they allow it, else one call each.
    Returns (total, {loss_dict key: term})."""
    merged = merge_plans(plans) if _MERGE else None
    todo = [merged] if merged is not None else list(plans)
    total, out = None, {}
    idx = {k: i for i, k in enumerate(_lib.LOSS_TERMS)}
    for spec, aux, keys in todo:
        t, terms = fused_loss(spec, results, aux, typ, sync)
        total = t if total is None else total + t
        out.update({k: terms[idx[k]] for k in keys})
    return total, out
