"""torch <-> C-ABI glue for one rendering pass: descriptor, parameter packing, autograd.Function.

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); all arithmetic of the hot
path runs in libsnerf_hip.so.  There is no fallback: a CPU tensor or a missing library raises.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import torch

import os
import weakref

_WS_POISON = os.environ.get("SNERF_WS_POISON", "0") == "1"

from . import _lib

# SNERF_MFMA=f16x2|f16x1 overrides every ModelSpec.mfma (diagnostics); unset, ModelSpec.mfma decides and its default is f16x2 --
# the arithmetic a C caller gets with no arithmetic flag
BASE_FLAGS = _lib.MFMA_FLAGS.get(os.environ.get("SNERF_MFMA", "").lower())


DEFAULT_MFMA = "f16x2"


def mfma_mode(pipeline_cfg=None, run_cfg=None) -> str:
    """Matrix-unit arithmetic of the dense layers from the reference's two precision knobs:
    `precision` (baseline/pipelines/nerf.py:65, handed to the trainer at framework/pipelines.py:319; 16 = half precision) and the
    run config's `float32_matmul_precision` (framework/configs.py:26, applied at framework/pipelines.py:254-256).  The build adds
    the pipeline field `mfma_precision`:
      "f16x2" (default)  two fp16 planes of block-scaled operands (22 significant bits), three products: fp32-class results
      "f16x1" / "bf16"   REDUCED precision: ONE fp16 plane of the same block-scaled tensors (11 significant bits -- the operand
                         precision of TF32, three bits more than bf16), one product, half the activation bytes.  `precision = 16`
                         selects it; "bf16" is the name BASELINE.json's configs[2] / [4] use and maps to the same mode
      "auto"             follow float32_matmul_precision: "highest" -> f16x2; "high" (TF32-class on the reference's hardware) and
                         "medium" -> f16x1"""
    mode = getattr(pipeline_cfg, "mfma_precision", DEFAULT_MFMA)
    if getattr(pipeline_cfg, "precision", 32) == 16:
        return "f16x1"
    if mode == "auto":
        return {"highest": "f16x2", "high": "f16x1", "medium": "f16x1"}[
            getattr(run_cfg, "float32_matmul_precision", "highest")]
    return "f16x1" if mode == "bf16" else mode

# parameter names in the reference's state_dict order -> SnerfParams field
_HEAD_FIELDS = {
    "sigma_from_xyz.0.weight": "sigma_w", "sigma_from_xyz.0.bias": "sigma_b",
    "feats_from_xyz.weight": "feats_w", "feats_from_xyz.bias": "feats_b",
    "rgb_from_xyzdir.0.weight": "rgb_w0", "rgb_from_xyzdir.0.bias": "rgb_b0",
    "rgb_from_xyzdir.2.weight": "rgb_w2", "rgb_from_xyzdir.2.bias": "rgb_b2",
    "semantic_prediction.0.weight": "sem_w0", "semantic_prediction.0.bias": "sem_b0",
    "semantic_prediction.2.weight": "sem_w2", "semantic_prediction.2.bias": "sem_b2",
    "sky_color.0.weight": "sky_w0", "sky_color.0.bias": "sky_b0",
    "sky_color.2.weight": "sky_w2", "sky_color.2.bias": "sky_b2",
    "beta_from_xyz.0.weight": "beta_w0", "beta_from_xyz.0.bias": "beta_b0",
    "beta_from_xyz.2.weight": "beta_w2", "beta_from_xyz.2.bias": "beta_b2",
    "semantic_beta_from_xyz.0.weight": "sbeta_w0", "semantic_beta_from_xyz.0.bias": "sbeta_b0",
    "semantic_beta_from_xyz.2.weight": "sbeta_w2", "semantic_beta_from_xyz.2.bias": "sbeta_b2",
}


@dataclass(frozen=True)
class ModelSpec:
    """Architecture part of SnerfDesc (field names: configs/pipelines/rs_semantic.toml:13-67)."""
    fc_units: int = 512
    fc_layers: int = 8
    feat_last: int = 256
    fc_skips: tuple = (4,)
    n_freq: int = 10            # mapping_pos_n_freq; 0 = identity (baseline SatNeRF)
    siren: bool = True
    t_dim: int = 4
    n_classes: int = 5          # 0 = baseline SatNeRF
    sem_sigmoid: bool = True
    use_tj_instead_of_beta: bool = False
    use_tj_for_s: bool = False
    use_separate_beta_for_s: bool = False
    use_separate_tj_for_semantic: bool = False
    mfma: str = "f16x2"         # see mfma_mode()

    @staticmethod
    def from_pipeline_cfg(pc, n_classes: int, model: str = "semantic", run_cfg=None) -> "ModelSpec":
        sem = model == "semantic"
        W = pc.fc_units
        return ModelSpec(
            fc_units=W, fc_layers=pc.fc_layers, feat_last=W if pc.fc_use_full_features else W // 2,
            fc_skips=tuple(pc.fc_skips), n_freq=pc.mapping_pos_n_freq if sem else 0,
            siren=pc.activation_function == "siren", t_dim=pc.t_embedding_tau,
            n_classes=n_classes if sem else 0,
            sem_sigmoid=sem and getattr(pc, "semantic_activation_function", "sigmoid") == "sigmoid",
            use_tj_instead_of_beta=sem and bool(getattr(pc, "use_tj_instead_of_beta", False)),
            use_tj_for_s=sem and bool(getattr(pc, "use_tj_for_s", False)),
            use_separate_beta_for_s=sem and bool(getattr(pc, "use_separate_beta_for_s", False)),
            use_separate_tj_for_semantic=sem and bool(getattr(pc, "use_separate_tj_for_semantic", False)),
            mfma=mfma_mode(pc, run_cfg))

    def desc(self, n_rays: int, n_samples: int, flags: int = 0) -> _lib.SnerfDesc:
        mask = 0
        for s in self.fc_skips:
            mask |= 1 << int(s)
        return _lib.SnerfDesc(
            n_rays=n_rays, n_samples=n_samples, fc_units=self.fc_units, fc_layers=self.fc_layers,
            feat_last=self.feat_last, skip_mask=mask, n_freq=self.n_freq, siren=int(self.siren), t_dim=self.t_dim,
            n_classes=self.n_classes, sem_sigmoid=int(self.sem_sigmoid),
            use_tj_instead_of_beta=int(self.use_tj_instead_of_beta), use_tj_for_s=int(self.use_tj_for_s),
            use_separate_beta_for_s=int(self.use_separate_beta_for_s),
            use_separate_tj_for_semantic=int(self.use_separate_tj_for_semantic), flags=flags | (_lib.MFMA_FLAGS[self.mfma] if BASE_FLAGS is None else BASE_FLAGS))

    def param_names(self) -> list:
        names = []
        for i in range(self.fc_layers):
            names += [f"fc_net.{2 * i}.weight", f"fc_net.{2 * i}.bias"]
        names += ["sigma_from_xyz.0.weight", "sigma_from_xyz.0.bias", "feats_from_xyz.weight", "feats_from_xyz.bias",
                  "rgb_from_xyzdir.0.weight", "rgb_from_xyzdir.0.bias", "rgb_from_xyzdir.2.weight",
                  "rgb_from_xyzdir.2.bias"]
        if self.n_classes > 0:
            names += ["semantic_prediction.0.weight", "semantic_prediction.0.bias", "semantic_prediction.2.weight",
                      "semantic_prediction.2.bias"]
        for j in (0, 2, 4, 6):
            names += [f"sun_v_net.{j}.weight", f"sun_v_net.{j}.bias"]
        names += ["sky_color.0.weight", "sky_color.0.bias", "sky_color.2.weight", "sky_color.2.bias",
                  "beta_from_xyz.0.weight", "beta_from_xyz.0.bias", "beta_from_xyz.2.weight", "beta_from_xyz.2.bias"]
        if self.n_classes > 0 and self.use_separate_beta_for_s:
            names += ["semantic_beta_from_xyz.0.weight", "semantic_beta_from_xyz.0.bias",
                      "semantic_beta_from_xyz.2.weight", "semantic_beta_from_xyz.2.bias"]
        return names


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _empty(*a, **k):
    """torch.empty; with SNERF_WS_POISON=1 (diagnostic) filled with 0xFF bytes, so that a kernel which reads bytes nobody
    wrote shows up as NaN / a wild value instead of depending on what the allocator happens to hand out"""
    t = torch.empty(*a, **k)
    if _WS_POISON:
        t.view(torch.uint8).fill_(0xFF) if t.numel() else None
    return t


# ---- pass workspaces -------------------------------------------------------------------------------------------------
# A training pass needs 8-20 GB of workspace between its forward and its backward.  Taking it from torch's caching allocator
# every step works until something else is allocated while the block lies free: the allocator carves a few MB for a result
# tensor out of the free 10 GB block, the next forward finds no block of its size and goes to hipMalloc (10 GB: milliseconds
# on a quiet box, a quarter of a second on a busy one) in the middle of training.  Workspaces are therefore LEASED from a pool
# of whole tensors keyed by (device, stream, size): a lease ends in the pass's backward (or when its autograd node dies) and the
# same tensor serves the same pass of the next step on the same stream, so stream order alone makes the reuse safe.
_WS_FREE: dict = {}                 # (device index, stream id, nbytes) -> [tensor, ...]
_WS_FREE_BYTES = 0
_WS_POOL_CAP = int(float(os.environ.get("SNERF_WS_POOL_GB", "96")) * (1 << 30))   # idle bytes kept; beyond it the oldest idle workspaces go back to torch


class _WsLease:
    __slots__ = ("key", "t")

    def __init__(self, key, t):
        self.key, self.t = key, t

    def release(self):
        global _WS_FREE_BYTES
        t, self.t = self.t, None
        if t is None:
            return
        _WS_FREE.setdefault(self.key, []).append(t)
        _WS_FREE_BYTES += t.numel()
        while _WS_FREE_BYTES > _WS_POOL_CAP and _WS_FREE:
            k = next(iter(_WS_FREE))          # dicts keep insertion order: the key idle for longest first
            lst = _WS_FREE[k]
            _WS_FREE_BYTES -= lst.pop(0).numel()
            if not lst:
                del _WS_FREE[k]

    def __del__(self):
        try:
            self.release()
        except Exception:      # interpreter shutdown: the module's globals may be gone
            pass


def lease_workspace(dev, nbytes: int) -> _WsLease:
    global _WS_FREE_BYTES
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), torch.cuda.current_stream(dev).cuda_stream, int(nbytes))
    lst = _WS_FREE.get(key)
    if lst:
        t = lst.pop()
        _WS_FREE_BYTES -= t.numel()
        if not lst:
            del _WS_FREE[key]
        if _WS_POISON:
            t.fill_(0xFF)
    else:
        try:
            t = _empty(nbytes, dtype=torch.uint8, device=dev)
        except torch.cuda.OutOfMemoryError:
            # the idle workspaces of other shapes / streams live outside torch's allocator: give them back and try once more
            release_workspaces()
            torch.cuda.empty_cache()
            t = _empty(nbytes, dtype=torch.uint8, device=dev)
    return _WsLease(key, t)


def release_workspaces():
    """hand every idle workspace back to torch's allocator (before a phase with another memory profile, e.g. full-frame inference)"""
    global _WS_FREE_BYTES
    _WS_FREE.clear()
    _WS_FREE_BYTES = 0


def _empty_like(x):
    t = torch.empty_like(x)
    if _WS_POISON and t.numel():
        t.view(torch.uint8).fill_(0xFF)
    return t


# ---- gradient sinks -------------------------------------------------------------------------------------------------
# A parameter may have a registered SINK: a preallocated gradient tensor (FlatAdam's view into its flat bucket, which is
# also p.grad).  A pass whose parameters all have one accumulates its packed gradients straight into the sinks
# (snerf_unpack_grads, accumulate = 1) and returns no parameter gradients to autograd -- the 2 x 23 AccumulateGrad add
# kernels per step (main + solar-correction pass, each parameter) disappear.  The two passes run their backward on
# different HIP streams, so the unpack launches are chained by events (each waits for the previous one; with two
# contributions the sum is the same in either order).  Parameters without sinks (tests, foreign optimisers) get their
# gradients from autograd as before.  SNERF_GRAD_SINKS=0 switches the mechanism off (A/B).
_GRAD_SINKS: dict = {}      # id(parameter) -> (weakref, sink tensor)
_SINK_EVENT: dict = {}      # device index -> event of the last accumulating unpack
_SINK_TOUCHED: set = set()  # ids of parameters whose sink a pass has accumulated into (cleared by the optimiser's zero_grad)
_SINKS_ON = os.environ.get("SNERF_GRAD_SINKS", "1") != "0"


def register_grad_sinks(params, sinks):
    for p, g in zip(params, sinks):
        _GRAD_SINKS[id(p)] = (weakref.ref(p), g)


def unregister_grad_sinks(params):
    for p in params:
        _GRAD_SINKS.pop(id(p), None)


def wait_grad_sinks(device=None):
    """Make the current stream wait for the last accumulating unpack.  A pass that writes sinks hands autograd no parameter
    gradients, so the engine's end-of-backward stream synchronisation (which follows AccumulateGrad) does not cover the
    side stream of the solar-correction pass: every consumer of the sink tensors calls this first (FlatAdam does)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    ev = _SINK_EVENT.get(dev.index if dev.index is not None else torch.cuda.current_device())
    if ev is not None:
        torch.cuda.current_stream(dev).wait_event(ev)


_SINK_SYNC_QUEUED: set = set()


def _queue_sink_sync(dev):
    """Once per backward(): when the whole graph has run, the stream that called backward() waits for the last sink write
    (autograd's own end-of-backward synchronisation only follows AccumulateGrad nodes, which the sink path bypasses)."""
    key = dev.index
    if key in _SINK_SYNC_QUEUED:
        return

    def _cb():
        _SINK_SYNC_QUEUED.discard(key)
        wait_grad_sinks(dev)

    try:
        torch.autograd.Variable._execution_engine.queue_callback(_cb)
        _SINK_SYNC_QUEUED.add(key)
    except RuntimeError:      # not inside a backward pass of the engine (direct call): the consumer synchronises
        pass


_SINKS_ACTIVE = False


class accumulate_into_sinks:
    """`with ops.accumulate_into_sinks(): loss.backward()` -- only inside this context do the passes add their parameter
    gradients straight into the registered sink tensors (and hand autograd None for them).  It is the training loop's
    statement that this backward is a plain accumulate-into-.grad one.  Anywhere else -- torch.autograd.grad,
    backward(inputs=...), gradient checks -- the passes return ordinary gradients and touch no .grad.  Entering also
    discards a stale end-of-backward marker that a backward which raised may have left behind."""

    def __enter__(self):
        global _SINKS_ACTIVE
        self._prev = _SINKS_ACTIVE
        _SINKS_ACTIVE = True
        _SINK_SYNC_QUEUED.clear()
        return self

    def __exit__(self, *exc):
        global _SINKS_ACTIVE
        _SINKS_ACTIVE = self._prev
        _SINK_SYNC_QUEUED.clear()
        return False


def _sinks_for(params):
    if not _SINKS_ON or not _SINKS_ACTIVE or not _GRAD_SINKS:
        return None
    out = []
    for p in params:
        e = _GRAD_SINKS.get(id(p))
        if e is None or e[0]() is not p or p.grad is None or p.grad.data_ptr() != e[1].data_ptr():
            return None      # not (or no longer) wired to its sink: autograd accumulates
        out.append(e[1])
    return out


def _check_dev(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"snerf_amd: '{name}' must live on the GPU (the HIP path has no CPU fallback)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"snerf_amd: '{name}' must be float32, got {t.dtype}")


def params_struct(spec: ModelSpec, tensors: dict) -> _lib.SnerfParams:
    ps = _lib.SnerfParams()
    for name in spec.param_names():
        t = tensors[name]
        _check_dev(t, name)
        if not t.is_contiguous():
            raise RuntimeError(f"snerf_amd: parameter '{name}' must be contiguous")
        if name.startswith("fc_net."):
            i = int(name.split(".")[1]) // 2
            (ps.fc_w if name.endswith("weight") else ps.fc_b)[i] = t.data_ptr()
        elif name.startswith("sun_v_net."):
            j = int(name.split(".")[1]) // 2
            (ps.sun_w if name.endswith("weight") else ps.sun_b)[j] = t.data_ptr()
        else:
            setattr(ps, _HEAD_FIELDS[name], t.data_ptr())
    return ps


def pack_params(spec: ModelSpec, tensors: dict) -> torch.Tensor:
    """state_dict tensors -> packed MFMA-friendly buffer (snerf_pack_params)."""
    L = _lib.lib()
    d = spec.desc(1, 1)
    n = L.snerf_packed_floats(C.byref(d))
    if n == 0:
        _lib.check(1, "snerf_packed_floats")
    dev = tensors[spec.param_names()[0]].device
    packed = _empty(n, dtype=torch.float32, device=dev)
    ps = params_struct(spec, tensors)
    with torch.cuda.device(dev):
        _lib.check(L.snerf_pack_params(C.byref(d), C.byref(ps), _ptr(packed), _stream()), "snerf_pack_params")
    return packed


def unpack_grads(spec: ModelSpec, packed_grads: torch.Tensor, like: dict, accumulate_into: dict | None = None) -> dict:
    """packed gradient buffer -> dict name -> gradient tensor shaped like the parameters."""
    L = _lib.lib()
    d = spec.desc(1, 1)
    if accumulate_into is None:
        grads = {n: _empty_like(like[n]) for n in spec.param_names()}
    else:
        grads = accumulate_into
    ps = params_struct(spec, grads)
    with torch.cuda.device(packed_grads.device):
        _lib.check(L.snerf_unpack_grads(C.byref(d), _ptr(packed_grads), C.byref(ps), int(accumulate_into is not None),
                                        _stream()), "snerf_unpack_grads")
    return grads


@dataclass
class PassInputs:
    """One ray batch in either of the two seams (renderer: rays [+u]; inference(): xyz + z_vals)."""
    sun_d: torch.Tensor                   # (N,3) (may be a strided view into extras (N,4))
    rays: torch.Tensor | None = None      # (N,8)
    xyz: torch.Tensor | None = None       # (N,S,3)
    z_vals: torch.Tensor | None = None    # (N,S)
    z_steps: torch.Tensor | None = None   # (S)
    u: torch.Tensor | None = None         # (N,S)
    _keep: list = field(default_factory=list)

    def struct(self, t, t_s):
        si = _lib.SnerfInputs()
        for name in ("rays", "xyz", "z_vals", "z_steps", "u"):
            v = getattr(self, name)
            if v is not None:
                _check_dev(v, name)
                v = v.contiguous()
                self._keep.append(v)
                setattr(si, name, v.data_ptr())
        sd = self.sun_d
        _check_dev(sd, "sun_d")
        if sd.dim() != 2 or sd.shape[1] != 3 or sd.stride(1) != 1:
            sd = sd.contiguous()
        self._keep.append(sd)
        si.sun_d = sd.data_ptr()
        si.sun_stride = sd.stride(0) if sd.shape[0] > 1 else 3
        si.t = t.data_ptr()
        si.t_s = t_s.data_ptr() if t_s is not None else None
        return si


_OUT_SHAPES = {
    "rgb": lambda N, S, C: (N, 3), "depth": lambda N, S, C: (N,), "weights": lambda N, S, C: (N, S),
    "transparency": lambda N, S, C: (N, S), "albedo": lambda N, S, C: (N, S, 3), "sun": lambda N, S, C: (N, S, 1),
    "sky": lambda N, S, C: (N, S, 3), "beta": lambda N, S, C: (N, S, 1), "sigmas": lambda N, S, C: (N, S),
    "beta_semantic": lambda N, S, C: (N, S, 1), "semantic_logits": lambda N, S, C: (N, C),
}


def output_keys(spec: ModelSpec, sc_pass: bool) -> list:
    if sc_pass:
        return ["weights", "transparency", "sun"]
    keys = ["rgb", "depth", "weights", "transparency", "albedo", "sun", "sky", "beta", "sigmas"]
    if spec.n_classes > 0:
        if spec.use_separate_beta_for_s:
            keys.append("beta_semantic")
        keys.append("semantic_logits")
    return keys


class _RenderPass(torch.autograd.Function):
    """forward = snerf_forward, backward = snerf_backward (+ snerf_unpack_grads)."""

    @staticmethod
    def forward(ctx, spec, pin, sc_pass, need_grad, packed, names, t, t_s, *params):
        L = _lib.lib()
        N = t.shape[0]
        S = (pin.z_vals.shape[1] if pin.z_vals is not None else
             (pin.u.shape[1] if pin.u is not None else pin.z_steps.shape[0]))
        dev = t.device
        flags = (_lib.FLAG_TRAIN if need_grad else 0) | (_lib.FLAG_SC_PASS if sc_pass else 0)
        d = spec.desc(N, S, flags)
        nbytes = L.snerf_workspace_bytes(C.byref(d))
        if nbytes == 0:
            _lib.check(1, "snerf_workspace_bytes")
        with torch.cuda.device(dev):
            lease = lease_workspace(dev, nbytes)
        ws = lease.t
        keys = output_keys(spec, sc_pass)
        outs = {k: _empty(_OUT_SHAPES[k](N, S, spec.n_classes), dtype=torch.float32, device=dev) for k in keys}
        label = _empty((N,), dtype=torch.int64, device=dev) if (spec.n_classes > 0 and not sc_pass) else None
        so = _lib.SnerfOutputs()
        for k, v in outs.items():
            setattr(so, k, v.data_ptr())
        so.semantic_label = label.data_ptr() if label is not None else None
        if pin.z_vals is not None and pin.z_vals.is_contiguous() and pin.z_vals.dtype == torch.float32:
            # depths given: they ARE the result (no copy-out launch).  ALIASING: results["z_vals"] shares storage with the caller's
            # tensor -- in fused_model_rendering the main pass's result, the sc pass's input and its result are one buffer, which
            # the backward passes read again.  Never edit it in place (tests/test_gpu_kernels.py pins the contract).
            z_out = pin.z_vals.detach()
        else:
            z_out = _empty((N, S), dtype=torch.float32, device=dev)
            so.z_vals = z_out.data_ptr()
        _check_dev(t, "t")
        tc = t.contiguous()
        tsc = t_s.contiguous() if t_s is not None else None
        si = pin.struct(tc, tsc)
        with torch.cuda.device(dev):
            _lib.check(L.snerf_forward(C.byref(d), _ptr(packed), C.byref(si), C.byref(so), _ptr(ws), nbytes,
                                       _stream()), "snerf_forward")
        if not need_grad:
            lease.release()    # nothing is kept for a backward: the next pass of this size on this stream may have it
        ctx.spec, ctx.pin, ctx.desc, ctx.lease, ctx.nbytes = spec, pin, d, lease, nbytes
        ctx.packed, ctx.names, ctx.keys, ctx.tc, ctx.tsc = packed, names, keys, tc, tsc
        ctx.param_like = params
        ctx.train = need_grad
        ctx.set_materialize_grads(False)   # results the loss does not use arrive as None, not as zero tensors
        ret = tuple(outs[k] for k in keys)
        nd = [z_out] + ([label] if label is not None else [])
        ctx.mark_non_differentiable(*nd)
        ctx.n_diff = len(keys)
        return ret + tuple(nd)

    @staticmethod
    def backward(ctx, *gouts):
        L = _lib.lib()
        if not ctx.train:
            raise RuntimeError("snerf_amd: backward through a pass that was run without SNERF_FLAG_TRAIN")
        spec, d = ctx.spec, ctx.desc
        if ctx.lease.t is None:
            raise RuntimeError("snerf_amd: second backward through one pass (its workspace went back to the pool after the first)")
        go = _lib.SnerfOutGrads()
        keep, live = [], []
        for k, g in zip(ctx.keys, gouts[:ctx.n_diff]):
            if g is not None:
                g = g.contiguous()
                keep.append(g)
                live.append(k)
                setattr(go, k, g.data_ptr())
        if not live:
            return (None,) * (8 + len(ctx.names))
        dev = ctx.tc.device
        pg = torch.zeros(int(L.snerf_grad_floats(C.byref(d))), dtype=torch.float32, device=dev)   # the fp32 region only: the backward touches nothing else
        d_t = _empty_like(ctx.tc)
        d_ts = _empty_like(ctx.tsc) if ctx.tsc is not None else None
        si = ctx.pin.struct(ctx.tc, ctx.tsc)
        with torch.cuda.device(dev):
            _lib.check(L.snerf_backward(C.byref(d), _ptr(ctx.packed), C.byref(si), C.byref(go), _ptr(pg), _ptr(d_t),
                                        _ptr(d_ts), _ptr(ctx.lease.t), ctx.nbytes, _stream()), "snerf_backward")
        like = dict(zip(ctx.names, ctx.param_like))
        ctx.lease.release()
        sinks = _sinks_for(ctx.param_like)
        if sinks is not None:
            with torch.cuda.device(dev):
                st = torch.cuda.current_stream()
                prev = _SINK_EVENT.get(dev.index)
                if prev is not None:
                    st.wait_event(prev)
                unpack_grads(spec, pg, like, accumulate_into=dict(zip(ctx.names, sinks)))
                _SINK_TOUCHED.update(id(p) for p in ctx.param_like)
                ev = torch.cuda.Event()
                ev.record(st)
                _SINK_EVENT[dev.index] = ev
            _queue_sink_sync(dev)
            return (None, None, None, None, None, None, d_t, d_ts) + (None,) * len(ctx.names)
        grads = unpack_grads(spec, pg, like)
        # EVERY parameter gets a gradient tensor, zero where no result gradient reaches it: the reference's model returns
        # one concatenated (P, 9 + C) tensor that inference() slices (semantic/models/rs_semantic.py:71-96), so autograd
        # hands e.g. the beta head exact ZEROS (not None) while epoch < first_beta_epoch, and torch.optim.Adam counts
        # those steps (probed on the reference: 45 of 45 parameters have state after one step at epoch 0)
        return (None, None, None, None, None, None, d_t, d_ts) + tuple(grads[n] for n in ctx.names)


def render_pass(spec: ModelSpec, params: dict, pin: PassInputs, t: torch.Tensor, t_s: torch.Tensor | None = None,
                sc_pass: bool = False, packed: torch.Tensor | None = None) -> dict:
    """Run one pass (main, or the solar-correction variant) and return the reference's result dict
    (semantic/models/rs_semantic.py:111-128) plus 'z_vals'.  When `pin.z_vals` is given (contiguous fp32), the returned
    'z_vals' IS that tensor's storage (no copy): do not edit either in place while the pass's backward is pending."""
    names = spec.param_names()
    plist = [params[n] for n in names]
    if packed is None:
        packed = pack_params(spec, params)
    # grad mode is always off inside Function.forward, so decide here whether activations are kept
    need_grad = torch.is_grad_enabled() and (any(p.requires_grad for p in plist) or t.requires_grad
                                             or (t_s is not None and t_s.requires_grad))
    res = _RenderPass.apply(spec, pin, sc_pass, need_grad, packed, names, t, t_s, *plist)
    keys = output_keys(spec, sc_pass)
    out = dict(zip(keys, res[:len(keys)]))
    out["z_vals"] = res[len(keys)]
    if len(res) > len(keys) + 1:
        out["semantic_label"] = res[len(keys) + 1]
    return out


@torch.no_grad()
def render_pass_into(spec: ModelSpec, params: dict, pin: PassInputs, t: torch.Tensor, t_s: torch.Tensor | None,
                     out: dict, sc_pass: bool = False, packed: torch.Tensor | None = None,
                     workspace: torch.Tensor | None = None) -> torch.Tensor:
    """Inference-only pass that produces ONLY the result tensors named in `out` and writes them in place:
    `out` maps result keys (output_keys(), 'semantic_label', 'z_vals') to preallocated contiguous tensors of the
    pass's shapes -- typically row slices of full-frame buffers, so a chunked render never concatenates
    (eval/utils/util.py:13-42 re-concatenates every key per chunk).  Every other SnerfOutputs pointer stays NULL and
    the composite kernel skips it.  Returns the workspace (pass it back in for the next chunk of the same size)."""
    L = _lib.lib()
    N = t.shape[0]
    S = (pin.z_vals.shape[1] if pin.z_vals is not None else
         (pin.u.shape[1] if pin.u is not None else pin.z_steps.shape[0]))
    dev = t.device
    d = spec.desc(N, S, _lib.FLAG_SC_PASS if sc_pass else 0)
    nbytes = L.snerf_workspace_bytes(C.byref(d))
    if nbytes == 0:
        _lib.check(1, "snerf_workspace_bytes")
    if workspace is None or workspace.numel() < nbytes or workspace.device != dev:
        workspace = _empty(nbytes, dtype=torch.uint8, device=dev)
    allowed = set(output_keys(spec, sc_pass)) | {"z_vals"} | ({"semantic_label"} if spec.n_classes > 0 and not sc_pass else set())
    so = _lib.SnerfOutputs()
    for k, v in out.items():
        if k not in allowed:
            raise KeyError(f"render_pass_into: '{k}' is not a result of this pass (have {sorted(allowed)})")
        if not v.is_cuda:
            raise RuntimeError(f"snerf_amd: out['{k}'] must live on the GPU (the HIP path has no CPU fallback)")
        want = (N,) if k == "semantic_label" else ((N, S) if k == "z_vals" else _OUT_SHAPES[k](N, S, spec.n_classes))
        if tuple(v.shape) != tuple(want) or not v.is_contiguous():
            raise ValueError(f"render_pass_into: out['{k}'] must be a contiguous {tuple(want)} tensor, got {tuple(v.shape)}")
        if v.dtype != (torch.int64 if k == "semantic_label" else torch.float32):
            raise ValueError(f"render_pass_into: out['{k}'] has dtype {v.dtype}")
        setattr(so, k, v.data_ptr())
    if packed is None:
        packed = pack_params(spec, params)
    _check_dev(t, "t")
    tc = t.contiguous()
    tsc = t_s.contiguous() if t_s is not None else None
    si = pin.struct(tc, tsc)
    with torch.cuda.device(dev):
        _lib.check(L.snerf_forward(C.byref(d), _ptr(packed), C.byref(si), C.byref(so), _ptr(workspace), workspace.numel(),
                                   _stream()), "snerf_forward")
    return workspace


class _EmbedRows(torch.autograd.Function):
    """rows = table[idx] with the library's deterministic backward (snerf_embedding_rows / snerf_embedding_backward)."""

    @staticmethod
    def forward(ctx, table, idx):
        L = _lib.lib()
        _check_dev(table, "embedding table")
        tb = table.detach().contiguous()
        ix = idx.contiguous()
        rows = torch.empty((ix.shape[0], tb.shape[1]), dtype=torch.float32, device=tb.device)
        with torch.cuda.device(tb.device):
            _lib.check(L.snerf_embedding_rows(_ptr(tb), tb.shape[0], tb.shape[1], _ptr(ix), ix.shape[0], _ptr(rows), _stream()),
                       "snerf_embedding_rows")
        ctx.save_for_backward(ix)
        ctx.shape = tuple(tb.shape)
        return rows

    @staticmethod
    def backward(ctx, g):
        (ix,) = ctx.saved_tensors
        L = _lib.lib()
        grad = torch.zeros(ctx.shape, dtype=torch.float32, device=g.device)
        gc = g.contiguous()
        with torch.cuda.device(g.device):
            _lib.check(L.snerf_embedding_backward(_ptr(ix), _ptr(gc), ix.shape[0], ctx.shape[1], ctx.shape[0], _ptr(grad), _stream()),
                       "snerf_embedding_backward")
        return grad, None


def embed_rows(embedding: torch.nn.Module, idx: torch.Tensor) -> torch.Tensor:
    """models["t"](ts) of the reference renderers (semantic/components/rendering.py:35-46): the rays' rows of an
    nn.Embedding.  torch's embedding backward is a sort + segmented-scatter chain of ~12 launches; here forward and
    backward are one launch each, the backward summed in a fixed order."""
    if idx.dtype != torch.int64:
        idx = idx.long()
    return _EmbedRows.apply(embedding.weight, idx)


def sample_z(rays: torch.Tensor, z_steps: torch.Tensor, u: torch.Tensor | None) -> torch.Tensor:
    """stratified depths (N,S) -- snerf_sample_z"""
    L = _lib.lib()
    _check_dev(rays, "rays")
    rays = rays.contiguous()
    N, S = rays.shape[0], z_steps.shape[0]
    z = torch.empty((N, S), dtype=torch.float32, device=rays.device)
    uc = u.contiguous() if u is not None else None
    with torch.cuda.device(rays.device):
        _lib.check(L.snerf_sample_z(_ptr(rays), _ptr(z_steps.contiguous()), _ptr(uc), _ptr(z), N, S, _stream()), "snerf_sample_z")
    return z
