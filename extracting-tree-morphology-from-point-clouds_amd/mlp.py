"""Pointwise (1x1-conv) MLP chains on channels-last rows.

Every MLP of the hot path -- the per-group Conv2d/BatchNorm2d/ReLU stacks of set abstraction (reference
blocks.py:93-98), the Conv1d/BatchNorm1d/ReLU stacks of feature propagation (:213-215) and the ConvHead
(:7-35) -- is a chain of ``rows x C_in -> rows x C_out`` contractions with per-channel batch statistics over
all rows.  The modules keep the reference's parameter containers (so state-dict keys and shapes are
unchanged) and hand their layers to ``chain_rows``.
"""
import torch
import torch.nn.functional as F


def _batch_norm_rows(bn, y):
    """BatchNorm over the row dimension with nn.BatchNorm*d's bookkeeping (momentum, num_batches_tracked)."""
    use_batch = bn.training or bn.running_mean is None
    factor = 0.0 if bn.momentum is None else bn.momentum
    if bn.training and bn.track_running_stats and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
        if bn.momentum is None:
            factor = 1.0 / float(bn.num_batches_tracked)
    return F.batch_norm(y, bn.running_mean if (not bn.training or bn.track_running_stats) else None,
                        bn.running_var if (not bn.training or bn.track_running_stats) else None,
                        bn.weight, bn.bias, use_batch, factor, bn.eps)


def chain_rows(x, layers):
    """x [R, C_in] fp32 contiguous rows; layers: iterable of (conv, bn_or_None, relu: bool).  -> [R, C_out]."""
    for conv, bn, relu in layers:
        w = conv.weight.reshape(conv.out_channels, -1)
        x = F.linear(x, w, conv.bias)
        if bn is not None:
            x = _batch_norm_rows(bn, x)
        if relu:
            x = F.relu(x)
    return x
