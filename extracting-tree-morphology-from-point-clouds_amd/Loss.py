"""point_wise_loss -- same contract as the reference's Modules/Loss.py:6-36 (part of the timed fwd+bwd unit).

semantic: cross entropy summed then divided by the number of rows; offset: mean Euclidean norm of the
residual with the squared norm clamped at 1e-8 before the square root.  ``n_points`` subsamples both terms
with independent random permutations drawn from the global CPU generator, like the reference.
"""
import torch
import torch.nn.functional as F

from . import _hip
from .Utils import cuda_cast


def mask_ranks(pad, masks_off):
    """pad [R] bool (real rows of the padded batch), masks_off [n] bool (per REAL row: its offset counts) ->
    (cum_pad int64 [R], off_mask bool [R], cum_off int64 [R]) -- what
        cum_pad = cumsum(pad); off_mask = pad & masks_off[(cum_pad - 1).clamp(0, n - 1)]; cum_off = cumsum(off_mask)
    gives (get_loss, PointNet2.py:188-196), in one launch."""
    _hip.require_device(pad, masks_off)
    if pad.dtype != torch.bool or masks_off.dtype != torch.bool:
        raise RuntimeError("mask_ranks: boolean masks expected")
    pad, masks_off = pad.contiguous(), masks_off.contiguous()
    R = pad.numel()
    if masks_off.numel() == 0:
        raise RuntimeError("mask_ranks: masks_off is empty")
    cum_pad = torch.empty(R, dtype=torch.int64, device=pad.device)
    cum_off = torch.empty(R, dtype=torch.int64, device=pad.device)
    off_mask = torch.empty(R, dtype=torch.bool, device=pad.device)
    if R:
        _hip.call("mask_ranks", _hip.lib().pn2_mask_ranks, pad.data_ptr(), masks_off.data_ptr(), R, masks_off.numel(),
                  cum_pad.data_ptr(), off_mask.data_ptr(), cum_off.data_ptr(), _hip.stream_ptr())
    return cum_pad, off_mask, cum_off


class MaskedPointLoss(torch.autograd.Function):
    """(sem [R,2], off [R,3], pad, off_mask, cum_pad, cum_off, sem_labels, off_labels[, weights]) -> tensor [2] =
    (semantic loss, offset loss) of the masked rows; one forward and one backward kernel (csrc/loss.hip).
    With weights [2] (get_loss's multipliers): -> (total [], parts [2]) = (w0 sem + w1 off, (w0 sem, w1 off)); only the
    total carries a gradient, the parts are for the log."""

    @staticmethod
    def forward(ctx, sem, off, pad, off_mask, cum_pad, cum_off, sem_labels, off_labels, weights=None):
        _hip.require_device(sem, off)
        lib = _hip.lib()
        sem, off = sem.contiguous(), off.contiguous()
        sem_labels, off_labels = sem_labels.contiguous(), off_labels.contiguous().float()
        R = sem.shape[0]
        out = torch.empty(2, dtype=torch.float32, device=sem.device)
        ws = torch.empty(max(lib.pn2_point_loss_workspace_bytes(R), 16), dtype=torch.uint8, device=sem.device)
        ctx.weighted = weights is not None
        if ctx.weighted:
            weights = weights.contiguous().float()
            total = torch.empty((), dtype=torch.float32, device=sem.device)
            _hip.call("point_loss_fwd", lib.pn2_point_loss_weighted_fwd_f32, sem.data_ptr(), off.data_ptr(), pad.data_ptr(),
                      off_mask.data_ptr(), cum_pad.data_ptr(), cum_off.data_ptr(), sem_labels.data_ptr(), sem_labels.numel(),
                      off_labels.data_ptr(), off_labels.shape[0], R, weights.data_ptr(), out.data_ptr(), total.data_ptr(),
                      ws.data_ptr(), ws.numel(), _hip.stream_ptr())
            ctx.save_for_backward(sem, off, pad, off_mask, cum_pad, cum_off, sem_labels, off_labels, weights)
            ctx.mark_non_differentiable(out)
            return total, out
        _hip.call("point_loss_fwd", lib.pn2_point_loss_fwd_f32, sem.data_ptr(), off.data_ptr(), pad.data_ptr(),
                  off_mask.data_ptr(), cum_pad.data_ptr(), cum_off.data_ptr(), sem_labels.data_ptr(), sem_labels.numel(),
                  off_labels.data_ptr(), off_labels.shape[0], R, out.data_ptr(), ws.data_ptr(), ws.numel(), _hip.stream_ptr())
        ctx.save_for_backward(sem, off, pad, off_mask, cum_pad, cum_off, sem_labels, off_labels)
        return out

    @staticmethod
    def backward(ctx, g, *_):
        sem, off, pad, off_mask, cum_pad, cum_off, sem_labels, off_labels = ctx.saved_tensors[:8]
        g = g.contiguous().float()
        dsem, doff = torch.empty_like(sem), torch.empty_like(off)
        if ctx.weighted:
            _hip.call("point_loss_bwd", _hip.lib().pn2_point_loss_weighted_bwd_f32, sem.data_ptr(), off.data_ptr(), pad.data_ptr(),
                      off_mask.data_ptr(), cum_pad.data_ptr(), cum_off.data_ptr(), sem_labels.data_ptr(), sem_labels.numel(),
                      off_labels.data_ptr(), off_labels.shape[0], sem.shape[0], g.data_ptr(), ctx.saved_tensors[8].data_ptr(),
                      dsem.data_ptr(), doff.data_ptr(), _hip.stream_ptr())
        else:
            _hip.call("point_loss_bwd", _hip.lib().pn2_point_loss_bwd_f32, sem.data_ptr(), off.data_ptr(), pad.data_ptr(),
                      off_mask.data_ptr(), cum_pad.data_ptr(), cum_off.data_ptr(), sem_labels.data_ptr(), sem_labels.numel(),
                      off_labels.data_ptr(), off_labels.shape[0], sem.shape[0], g.data_ptr(), dsem.data_ptr(), doff.data_ptr(),
                      _hip.stream_ptr())
        return dsem, doff, None, None, None, None, None, None, None


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
