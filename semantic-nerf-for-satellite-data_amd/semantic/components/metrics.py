"""Semantic validation metrics on the device -- mirror of semantic/components/metrics.py:11-87.

The reference moves predictions to the CPU for torchmetrics' MulticlassConfusionMatrix and loops over classes in
numpy for the mIoU; here the confusion matrix is one bincount over (gt * C + pred) on the GPU and accuracy / mIoU /
beta-at-transient are a handful of device reductions -- no host round trip until a scalar is logged.  Plotting
(plot_confusion_matrix, :63-76) stays with the visualisers (out of scope)."""
import torch


@torch.no_grad()
def semantic_error(semantic_pred, semantic_gt, filter_idx=None):
    """0 where the class is right, 1 where it is wrong; rows whose GROUND TRUTH is filter_idx count as right (:11-22)."""
    gt = semantic_gt.flatten().to(torch.int64)
    pred = semantic_pred.flatten().to(torch.int64)
    error = (gt != pred).to(torch.float32)
    if filter_idx is not None:
        error = torch.where(gt == filter_idx, torch.zeros_like(error), error)
    return error.reshape(semantic_gt.shape)


@torch.no_grad()
def semantic_accuracy(results, targets, filter_idx=None):
    """1 - errors / len(targets) (:25-29; filtered rows stay in the denominator, as in the reference)"""
    typ = "fine" if "rgb_fine" in results else "coarse"
    error = semantic_error(results[f"semantic_label_{typ}"], targets, filter_idx=filter_idx).flatten()
    return 1 - (torch.sum(error, dim=0) / len(targets))


@torch.no_grad()
def confusion_matrix_values(results, targets, n_classes: int, normalize="true"):
    """(C, C) matrix, rows = ground truth, columns = prediction, row-normalised like
    MulticlassConfusionMatrix(normalize="true") (:55-60); normalize=None returns counts."""
    typ = "fine" if "rgb_fine" in results else "coarse"
    pred = results[f"semantic_label_{typ}"].flatten().to(torch.int64)
    gt = targets.flatten().to(torch.int64).to(pred.device)
    cm = torch.bincount(gt * n_classes + pred, minlength=n_classes * n_classes).reshape(n_classes, n_classes).to(torch.float32)
    if normalize == "true":
        rows = cm.sum(dim=1, keepdim=True)
        cm = torch.where(rows > 0, cm / rows.clamp_min(1.0), torch.zeros_like(cm))  # torchmetrics: empty rows -> 0
    return cm


@torch.no_grad()
def semantic_mIoU(confusion_matrix_values_):
    """mean over classes of tp / (row + column - tp), NaN classes (absent in both) skipped (:32-42)"""
    cm = torch.as_tensor(confusion_matrix_values_, dtype=torch.float64)
    tp = torch.diagonal(cm)
    ious = tp / (cm.sum(dim=1) + cm.sum(dim=0) - tp)
    return torch.nanmean(ious)


@torch.no_grad()
def uncertainty_at_transient(results, semantic_gt, car_idx):
    """mean composited beta over the rays labelled car_idx (:79-87)"""
    typ = "fine" if "rgb_fine" in results else "coarse"
    beta = torch.sum(results[f"weights_{typ}"].unsqueeze(-1) * results[f"beta_{typ}"], -2)  # (N, 1)
    mask = (semantic_gt == car_idx).flatten().to(beta.device)
    return beta[mask].sum() / mask.sum()
