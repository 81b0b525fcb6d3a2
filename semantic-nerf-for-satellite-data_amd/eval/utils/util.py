"""batched_inference -- mirror of eval/utils/util.py:13-42: no-grad, render_chunk_size rays at a time."""
from collections import defaultdict

import torch


_PER_RAY_OPTIONS = ("perturb_rand", "given_z_vals")   # (N, S) tensors a caller may pin for the whole frame


def _chunk_options(render_options, i, chunk, n):
    """render options of the chunk starting at ray i: per-ray tensors given for the whole frame are sliced"""
    opts = dict(render_options) if render_options else {}
    for k in _PER_RAY_OPTIONS:
        v = opts.get(k)
        if torch.is_tensor(v) and v.shape[0] == n and n > chunk:
            opts[k] = v[i:i + chunk]
    return opts


@torch.no_grad()
def batched_inference(cfgs, renderer, models, rays, extras, render_options={}, epoch=None, show_tqdm=False):
    chunk = cfgs.pipeline.render_chunk_size
    parts = defaultdict(list)
    steps = range(0, rays.shape[0], chunk)
    if show_tqdm:
        from tqdm import tqdm
        steps = tqdm(steps)
    for i in steps:
        r = renderer.render_rays(models, rays[i:i + chunk], extras[i:i + chunk] if extras is not None else None,
                                 epoch=epoch, render_options=_chunk_options(render_options, i, chunk, rays.shape[0]))
        for k, v in r.items():
            parts[k].append(v)
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in parts.items()}


_KEY_SHAPES = {  # per-ray trailing shape of the render_rays results (S = n_samples, C = classes)
    "rgb": lambda S, C: (3,), "depth": lambda S, C: (), "weights": lambda S, C: (S,), "transparency": lambda S, C: (S,),
    "albedo": lambda S, C: (S, 3), "sun": lambda S, C: (S, 1), "sky": lambda S, C: (S, 3), "beta": lambda S, C: (S, 1),
    "sigmas": lambda S, C: (S,), "beta_semantic": lambda S, C: (S, 1), "semantic_logits": lambda S, C: (C,),
    "semantic_label": lambda S, C: (), "weights_sc": lambda S, C: (S,), "transparency_sc": lambda S, C: (S,),
    "sun_sc": lambda S, C: (S, 1),
}


@torch.no_grad()
def lean_inference(cfgs, renderer, models, rays, extras, keys=("rgb_coarse", "depth_coarse", "semantic_label_coarse"),
                   render_options={}, show_tqdm=False):
    """Full-frame inference for image / point-cloud extraction (eval/extract_pointcloud.py:66-79 calls
    batched_inference and then reads only rgb and depth): the full-frame result tensors are allocated once, every
    chunk of render_chunk_size rays writes its rows in place, only the requested results are produced (no per-sample
    tensors unless asked for), the solar-correction pass is skipped unless one of its results is requested, and the
    weights are packed once for all chunks."""
    from ... import ops
    # full-frame inference has its own memory profile: the idle TRAINING workspaces (8-20 GB each, held outside torch's allocator
    # by the lease pool) go back to torch first, so a frame's result tensors and inference workspace can use that memory
    ops.release_workspaces()
    chunk = cfgs.pipeline.render_chunk_size
    n = rays.shape[0]
    S = cfgs.pipeline.n_samples
    model = models["coarse"]
    Cn = model.spec.n_classes
    out = {}
    for k in keys:
        bare = k[:-len("_coarse")] if k.endswith("_coarse") else k
        if bare not in _KEY_SHAPES:
            raise KeyError(f"lean_inference: unknown result '{k}'")
        dt = torch.int64 if bare == "semantic_label" else torch.float32
        out[bare + "_coarse"] = torch.empty((n,) + _KEY_SHAPES[bare](S, Cn), dtype=dt, device=rays.device)
    packed = ops.pack_params(model.spec, dict(model.named_parameters()))
    ws = None
    steps = range(0, n, chunk)
    if show_tqdm:
        from tqdm import tqdm
        steps = tqdm(steps)
    for i in steps:
        sl = {k: v[i:i + chunk] for k, v in out.items()}
        opts = _chunk_options(render_options, i, chunk, n)
        opts["packed_params"], opts["workspace"] = packed, ws
        ws = renderer.render_rays_into(models, rays[i:i + chunk], extras[i:i + chunk] if extras is not None else None,
                                       sl, opts)
    return out


PER_RAY_RESULTS = ("rgb", "depth", "semantic_label", "semantic_logits")   # (N, ...) results: what a frame / point cloud is made of


def shard_and_gather(render_rows, n: int, rank: int = None, world: int = None) -> dict:
    """`render_rows(lo, hi)` -> dict of results for the rays [lo, hi) of an n-ray frame; this rank renders its contiguous
    slice (parallel.frame_shard) and the per-ray results (PER_RAY_RESULTS, with or without the `_coarse` postfix) are
    all-gathered into full-frame tensors on every rank -- ragged tails and empty shards included.  Per-sample results
    ((N, S, ...): weights, sigmas, ...) stay local under their key; `out["_rows"]` = (lo, hi) names the rows they cover."""
    from ... import parallel
    lo, hi = parallel.frame_shard(n, rank, world)
    local = render_rows(lo, hi)
    out = {"_rows": (lo, hi)}
    for k, v in local.items():
        bare = k[:-len("_coarse")] if k.endswith("_coarse") else k
        out[k] = parallel.allgather_rows(v, n) if bare in PER_RAY_RESULTS else v
    return out


@torch.no_grad()
def sharded_lean_inference(cfgs, renderer, models, rays, extras, keys=("rgb_coarse", "depth_coarse", "semantic_label_coarse"),
                           render_options={}, show_tqdm=False):
    """lean_inference with the frame's rays sharded over the ranks of the process group (SURVEY 8(e), config 5's full-frame
    half; reference callers eval/utils/util.py:13-42, eval/extract_pointcloud.py:66-114 are single-device): every rank holds
    the frame's rays, renders rows frame_shard(n) of them and receives the per-ray results of all ranks; the exchange is one
    all_gather per requested per-ray result (rgb: 12 B, depth: 4 B, label: 8 B per ray).  Equal to lean_inference on one rank
    bit for bit (rays are independent; per-ray jitter given for the whole frame is sliced with the rays)."""
    n = rays.shape[0]

    def rows(lo, hi):
        if hi == lo:      # more ranks than rays: contribute nothing (a zero-ray launch has no defined result)
            S, Cn = cfgs.pipeline.n_samples, models["coarse"].spec.n_classes
            out = {}
            for k in keys:
                bare = k[:-len("_coarse")] if k.endswith("_coarse") else k
                dt = torch.int64 if bare == "semantic_label" else torch.float32
                out[bare + "_coarse"] = torch.empty((0,) + _KEY_SHAPES[bare](S, Cn), dtype=dt, device=rays.device)
            return out
        opts = _chunk_options(render_options, lo, hi - lo, n) if n > hi - lo else render_options
        return lean_inference(cfgs, renderer, models, rays[lo:hi], extras[lo:hi] if extras is not None else None, keys=keys,
                              render_options=opts, show_tqdm=show_tqdm)
    return shard_and_gather(rows, n)
