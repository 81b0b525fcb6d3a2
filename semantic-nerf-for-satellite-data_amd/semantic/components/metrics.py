"""semantic_accuracy -- mirror of semantic/components/metrics.py:25-29 (logging metric, not on the hot path)."""
import torch


@torch.no_grad()
def semantic_accuracy(results, gt, typ="coarse"):
    pred = results[f"semantic_label_{typ}"].reshape(-1)
    return (pred == gt.reshape(-1).to(pred.dtype)).float().mean()
