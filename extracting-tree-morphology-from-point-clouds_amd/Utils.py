"""The one helper of the reference's Modules/Utils.py that sits on the hot path: ``cuda_cast`` (Utils.py:162-177).

Everything else in that file (early stopping, cloud IO, power-law fits) is outside SURVEY.md section 8 and is
not rebuilt here; when this package is dropped into the reference tree the reference's own Utils keeps serving
those."""
import functools

import torch


def cuda_cast(func):
    """Move every tensor argument to the current HIP device before the call ("cuda" is HIP on PyTorch-ROCm)."""

    @functools.wraps(func)
    def wrapper(*args, **kwargs):
        args = [a.cuda() if isinstance(a, torch.Tensor) else a for a in args]
        kwargs = {k: (v.cuda() if isinstance(v, torch.Tensor) else v) for k, v in kwargs.items()}
        return func(*args, **kwargs)

    return wrapper
