"""Image-error metrics of the validation step as device-side reductions.

Same definitions as the reference's `mse` / `psnr` (eval/utils/metrics.py:8-18): squared error averaged over
the selected elements, PSNR = -10 log10(MSE) for images in [0, 1].  Written as ONE masked sum-of-squares
reduction: a `valid_mask` is applied as a 0/1 weight inside the reduction (no boolean-index gather, hence no
data-dependent shape and no host synchronisation), and the result stays a 0-d device tensor until it is logged.
SSIM (kornia) is CPU tooling and stays out of scope."""
import torch


def _weights(mask, like):
    """`mask` selects leading-dimension entries of `like` (the reference indexes value[mask]); returns it as a
    float weight broadcast over the trailing dimensions, plus the number of selected ELEMENTS."""
    w = mask.to(device=like.device, dtype=like.dtype)
    per_entry = 1
    for d in like.shape[w.dim():]:
        per_entry *= int(d)
    count = w.sum() * per_entry
    return w.reshape(w.shape + (1,) * (like.dim() - w.dim())), count


def sum_squared_error(image_pred, image_gt, valid_mask=None):
    """(sum of squared differences, element count) over the valid elements -- the two numbers a sharded
    validation step all-reduces before forming the global PSNR."""
    diff = image_pred - image_gt
    if valid_mask is None:
        return torch.sum(diff * diff), torch.tensor(float(diff.numel()), device=diff.device)
    w, count = _weights(valid_mask, diff)
    # excluded elements contribute an exact zero even when they hold NaN / Inf (0 * NaN would poison the sum; the reference's
    # boolean gather never reads them)
    return torch.sum(torch.where(w > 0, w * diff * diff, torch.zeros_like(diff))), count


def mse(image_pred, image_gt, valid_mask=None, reduction="mean"):
    if reduction != "mean":
        # per-element form (only the visualisers ask for it): the reference's boolean gather
        err = torch.square(image_pred - image_gt)
        return err if valid_mask is None else err[valid_mask]
    sse, count = sum_squared_error(image_pred, image_gt, valid_mask)
    return sse / count


def psnr(image_pred, image_gt, valid_mask=None, reduction="mean"):
    return -10.0 * torch.log10(mse(image_pred, image_gt, valid_mask, reduction))
