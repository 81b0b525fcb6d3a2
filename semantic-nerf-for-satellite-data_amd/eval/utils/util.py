"""batched_inference -- mirror of eval/utils/util.py:13-42: no-grad, render_chunk_size rays at a time."""
from collections import defaultdict

import torch


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
                                 epoch=epoch, render_options=render_options)
        for k, v in r.items():
            parts[k].append(v)
    return {k: (v[0] if len(v) == 1 else torch.cat(v, 0)) for k, v in parts.items()}
