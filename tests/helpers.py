"""Shared helpers for the parity tests (fixtures -> oracle config / tensors)."""
import json
import os

import numpy as np
import torch

from oracle import snerf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_fixture(name):
    z = np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False)
    meta = json.loads(str(z["meta_json"]))
    c = dict(meta["cfg"])
    c["fc_skips"] = tuple(c["fc_skips"])
    cfg = O.OracleCfg(**c)
    return z, meta, cfg


def fixture_params(z, meta, cfg):
    """Parameters regenerate from (cfg, seed); small fixtures also store them, which pins the generator."""
    params = O.init_params_numpy(cfg, meta["seed"])
    stored = [k for k in z.files if k.startswith("param_")]
    for k in stored:
        assert np.array_equal(z[k], params[k[len("param_"):]]), k
    return params


def fixture_batch(z, prefix="in_"):
    """Main batch (prefix 'in_') or the depth-ray batch ('in_depth_': rays / extras / u only)."""
    depth_keys = {"in_depth_rays", "in_depth_extras", "in_depth_u"}
    b = {}
    for k in z.files:
        if prefix == "in_" and k.startswith("in_") and k not in depth_keys:
            b[k[3:]] = z[k]
        elif prefix == "in_depth_" and k in depth_keys:
            b[k[len(prefix):]] = z[k]
    return O.batch_to_torch(b)


def max_abs(a, b):
    a = torch.as_tensor(np.asarray(a)).double()
    b = torch.as_tensor(np.asarray(b)).double()
    return float((a - b).abs().max()) if a.numel() else 0.0


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a)).double().reshape(-1)
    b = torch.as_tensor(np.asarray(b)).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def check_validation_metrics(device):
    """Device-side validation metrics (snerf_amd/semantic/components/metrics.py, snerf_amd/eval/utils/metrics.py) against
    plain numpy restatements of the reference definitions (semantic/components/metrics.py:11-87,
    eval/utils/metrics.py:8-18), with every tensor on `device`.  parity unpinned: torchmetrics / kornia are absent."""
    from snerf_amd.semantic.components import metrics as M
    from snerf_amd.eval.utils.metrics import mse, psnr, sum_squared_error
    dev = torch.device(device)
    rng = np.random.default_rng(5)
    N, S, Cn = 500, 8, 6                       # class 5 never occurs: NaN IoU -> skipped
    gt = rng.integers(0, 5, size=(N, 1)).astype(np.uint8)
    pred = np.where(rng.random(N) < 0.7, gt[:, 0], rng.integers(0, 5, size=N)).astype(np.int64)
    wn, bn = rng.random((N, S)).astype(np.float32), rng.random((N, S, 1)).astype(np.float32)
    res = {"semantic_label_coarse": torch.from_numpy(pred).to(dev), "rgb_coarse": torch.zeros(N, 3, device=dev),
           "weights_coarse": torch.from_numpy(wn).to(dev), "beta_coarse": torch.from_numpy(bn).to(dev)}
    tg = torch.from_numpy(gt).to(dev)
    err = (gt[:, 0] != pred).astype(np.float32)
    acc = M.semantic_accuracy(res, tg)
    assert acc.device.type == dev.type and abs(float(acc) - (1 - err.sum() / N)) < 1e-6
    err4 = np.where(gt[:, 0] == 4, 0.0, err)
    assert abs(float(M.semantic_accuracy(res, tg, filter_idx=4)) - (1 - err4.sum() / N)) < 1e-6
    assert M.semantic_error(res["semantic_label_coarse"], tg).shape == tg.shape
    counts = np.zeros((Cn, Cn))
    for g, p in zip(gt[:, 0], pred):
        counts[g, p] += 1
    cm_counts = M.confusion_matrix_values(res, tg, Cn, normalize=None)
    assert cm_counts.device.type == dev.type and np.array_equal(cm_counts.cpu().numpy(), counts)
    cm = M.confusion_matrix_values(res, tg, Cn).cpu().numpy()
    rows = counts.sum(1, keepdims=True)
    assert np.allclose(cm, np.divide(counts, rows, out=np.zeros_like(counts), where=rows > 0), atol=1e-6)
    ious = np.array([counts[c, c] / (counts[c].sum() + counts[:, c].sum() - counts[c, c]) if
                     (counts[c].sum() + counts[:, c].sum()) > 0 else np.nan for c in range(Cn)])
    assert abs(float(M.semantic_mIoU(cm_counts)) - np.nanmean(ious)) < 1e-9
    assert abs(float(M.semantic_mIoU(cm_counts.cpu().numpy())) - np.nanmean(ious)) < 1e-9   # the reference passes numpy
    comp = (wn[..., None] * bn).sum(-2)[:, 0]
    car = gt[:, 0] == 3
    assert abs(float(M.uncertainty_at_transient(res, tg, 3)) - comp[car].sum() / car.sum()) < 1e-5
    g = torch.Generator().manual_seed(1)
    a, c = torch.rand(40, 3, generator=g).to(dev), torch.rand(40, 3, generator=g).to(dev)
    mask = (torch.rand(40, generator=g) > 0.5).to(dev)
    want = float(-10 * torch.log10(((a - c) ** 2)[mask].mean()))
    got = psnr(a, c, mask)
    assert got.device.type == dev.type and abs(float(got) - want) < 1e-5
    assert abs(float(psnr(a, c)) - float(-10 * torch.log10(((a - c) ** 2).mean()))) < 1e-5
    sse, cnt = sum_squared_error(a, c, mask)
    assert float(cnt) == 3 * int(mask.sum()) and abs(float(sse / cnt) - float(mse(a, c, mask))) < 1e-7
    assert mse(a, c, reduction="none").shape == (40, 3) and mse(a, c, mask, reduction="none").shape == (int(mask.sum()), 3)
    img_mask = (torch.rand(5, 8, generator=g) > 0.3).to(dev)           # an (H, W) mask over an (H, W, 3) image
    ia, ic = torch.rand(5, 8, 3, generator=g).to(dev), torch.rand(5, 8, 3, generator=g).to(dev)
    assert abs(float(mse(ia, ic, img_mask)) - float(((ia - ic) ** 2)[img_mask].mean())) < 1e-6
