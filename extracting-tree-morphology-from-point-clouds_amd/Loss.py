"""point_wise_loss -- same contract as the reference's Modules/Loss.py:6-36 (part of the timed fwd+bwd unit).

semantic: cross entropy summed then divided by the number of rows; offset: mean Euclidean norm of the
residual with the squared norm clamped at 1e-8 before the square root.  ``n_points`` subsamples both terms
with independent random permutations drawn from the global CPU generator, like the reference.
"""
import torch
import torch.nn.functional as F

from .Utils import cuda_cast


@cuda_cast
def point_wise_loss(semantic_prediction_logits, offset_predictions, semantic_labels, offset_labels, n_points=None):
    n_sem, n_off = len(semantic_prediction_logits), len(offset_predictions)
    if n_points is not None and n_off >= n_points:
        pick_sem = torch.randperm(n_sem)[:n_points]
        pick_off = torch.randperm(n_off)[:n_points]
    else:
        pick_sem, pick_off = torch.arange(n_sem), torch.arange(n_off)

    if n_sem == 0:
        semantic_loss = 0 * semantic_labels.sum()
    else:
        logits = semantic_prediction_logits[pick_sem]
        semantic_loss = F.cross_entropy(logits, semantic_labels[pick_sem], reduction="sum") / len(logits)

    if n_off == 0:
        offset_loss = 0 * offset_predictions.sum()
    else:
        sq = (offset_predictions[pick_off] - offset_labels[pick_off]).pow(2).sum(1)
        offset_loss = torch.sqrt(torch.clamp(sq, min=1e-8)).mean()
    return semantic_loss, offset_loss
