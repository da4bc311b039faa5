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
    sem_logits, sem_labels = semantic_prediction_logits, semantic_labels
    off_preds, off_labels = offset_predictions, offset_labels
    if n_points is not None and n_off >= n_points:
        # same two draws from the global CPU generator as the reference, then a device-side row selection
        pick_sem = torch.randperm(n_sem)[:n_points].to(sem_logits.device)
        pick_off = torch.randperm(n_off)[:n_points].to(off_preds.device)
        sem_logits, sem_labels = sem_logits.index_select(0, pick_sem), sem_labels.index_select(0, pick_sem)
        off_preds, off_labels = off_preds.index_select(0, pick_off), off_labels.index_select(0, pick_off)
    # else: the reference indexes with torch.arange(n), i.e. the identity -- skipped here: indexing a device
    # tensor with a CPU index costs ~46 ms of host time per call in backward (accumulating index_put_) and
    # changes nothing.

    if n_sem == 0:
        semantic_loss = 0 * semantic_labels.sum()
    else:
        semantic_loss = F.cross_entropy(sem_logits, sem_labels, reduction="sum") / len(sem_logits)

    if n_off == 0:
        offset_loss = 0 * offset_predictions.sum()
    else:
        sq = (off_preds - off_labels).pow(2).sum(1)
        offset_loss = torch.sqrt(torch.clamp(sq, min=1e-8)).mean()
    return semantic_loss, offset_loss
