"""Epoch loop with the reference's signatures (Modules/train_utils.py: train / validate / run_training) plus the
data-parallel wiring the reference lacks: one gradient all-reduce per optimizer step (parallel.FlatGradAllReduce),
placed between the last backward of the step and the gradient clipping (reference lines 57-61).

Kept quirks: the loss is scaled by 50 before backward, ``clip_grad_norm_`` is called with
``max_norm=True`` (= 1.0) on the scaled gradients, ``scheduler.step(epoch)`` runs every batch.  Progress bars
(fastprogress in the reference) are optional and never required.
"""
import logging
import time
from collections import defaultdict

import numpy as np
import torch
from torch.utils.data import DataLoader

from . import ops, parallel


def _bar(iterable, parent=None, comment=""):
    try:  # display only
        from fastprogress.fastprogress import progress_bar
        pb = progress_bar(iterable, parent=parent)
        pb.comment = comment
        return pb
    except Exception:
        return iterable


def _scalar(v):
    return float(v.detach().cpu().item()) if isinstance(v, torch.Tensor) else float(v)


def train(model, train_loader, optimizer, scheduler, scaler, epoch, mb, raster_hierarchical, minibatch_streaming,
          grad_sync=None):
    """One training epoch; returns (total, offset, semantic) mean losses.  ``grad_sync`` is an optional
    parallel.FlatGradAllReduce: when given, gradients are summed over ranks once per optimizer step."""
    model.train()
    losses = defaultdict(list)
    since_flush = 0
    for batch in _bar(train_loader, mb, "Training"):
        scheduler.step(epoch)
        if grad_sync is not None:
            grad_sync.zero()
        else:
            optimizer.zero_grad()
        with torch.amp.autocast("cuda", enabled=True):
            if raster_hierarchical and minibatch_streaming:
                loss, loss_dict = model.forward_hierarchical_streaming(batch, return_loss=True, scaler=scaler)
            elif raster_hierarchical:
                loss, loss_dict = model.forward_hierarchical(batch, return_loss=True)
            else:
                loss, loss_dict = model(batch, return_loss=True)
            for key, value in loss_dict.items():
                losses[key].append(_scalar(value))
        if not raster_hierarchical and not minibatch_streaming:
            scaler.scale(loss * 50).backward()
        if grad_sync is not None:
            grad_sync.allreduce()          # ONE collective per step, after every mini-batch backward of the step
        torch.nn.utils.clip_grad_norm_(model.parameters(), True, norm_type=2)
        scaler.step(optimizer)
        scaler.update()
        if torch.cuda.is_available():
            ops.check_status()       # the loss read-back above already synchronised: a kernel that gave up raises here
        since_flush += 1
        if since_flush % 5 == 0 and torch.cuda.is_available():
            torch.cuda.empty_cache()
            since_flush = 0
    off, sem = np.mean(losses["offset_loss"]), np.mean(losses["semantic_loss"])
    return off + sem, off, sem


def validate(model, val_loader, epoch, mb, raster_hierarchical, minibatch_streaming):
    model.eval()
    losses = defaultdict(list)
    with torch.no_grad():
        for batch in _bar(val_loader, mb, "Validation"):
            with torch.amp.autocast("cuda", enabled=False):
                if raster_hierarchical and minibatch_streaming:
                    loss, loss_dict = model.forward_hierarchical_streaming(batch, return_loss=True)
                elif raster_hierarchical:
                    loss, loss_dict = model.forward_hierarchical(batch, return_loss=True)
                else:
                    loss, loss_dict = model(batch, return_loss=True)
                for key, value in loss_dict.items():
                    losses[key].append(_scalar(value))
    off, sem = np.mean(losses["offset_loss"]), np.mean(losses["semantic_loss"])
    return off + sem, off, sem


def run_training(model, train_loader: DataLoader, val_loader: DataLoader, optimizer: torch.optim.Optimizer, epochs: int,
                 scheduler=None, early_stopper=None, verbose=False, raster_hierarchical=False,
                 minibatch_streaming=False, distributed=None):
    """Reference run_training (train_utils.py:130-197).  ``distributed``: None = use data parallelism iff a process
    group is initialised; the model must already live on this rank's device."""
    device_type = "cuda" if torch.cuda.is_available() else "cpu"
    scaler = torch.amp.GradScaler(device_type, enabled=device_type == "cuda")
    if distributed is None:
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized() \
            and torch.distributed.get_world_size() > 1
    grad_sync = parallel.FlatGradAllReduce(model) if distributed else None
    rank0 = (not distributed) or torch.distributed.get_rank() == 0
    for epoch in range(epochs):
        t0 = time.time()
        tr = train(model, train_loader, optimizer, scheduler, scaler, epoch, None, raster_hierarchical,
                   minibatch_streaming, grad_sync=grad_sync)
        va = validate(model, val_loader, epoch, None, raster_hierarchical, minibatch_streaming)
        if distributed:              # every rank validated its own shard: decide on the mean, identically everywhere
            dev = next(model.parameters()).device
            tr, va = tuple(parallel.allreduce_mean(tr, dev)), tuple(parallel.allreduce_mean(va, dev))
        msg = (f"Epoch {epoch + 1}/{epochs} | Train Total Loss: {tr[0]:.4f}, Val Total Loss: {va[0]:.4f}, "
               f"Train Offset Loss: {tr[1]:.4f}, Val Offset Loss: {va[1]:.4f}, "
               f"Train Semantic Loss: {tr[2]:.4f}, Val Semantic Loss: {va[2]:.4f} | {time.time() - t0:.1f} s")
        if rank0:
            logging.info(msg)
            if verbose:
                print(msg)
        if early_stopper and rank0:
            early_stopper(model, tr[0], va[0])
        stop = bool(early_stopper and rank0 and early_stopper.early_stop)
        if distributed:
            flag = torch.tensor([int(stop)], device=next(model.parameters()).device)
            torch.distributed.broadcast(flag, src=0)
            stop = bool(flag.item())
        if stop:
            if rank0:
                logging.info(f"Early stopping triggered at epoch {epoch + 1}")
            break
