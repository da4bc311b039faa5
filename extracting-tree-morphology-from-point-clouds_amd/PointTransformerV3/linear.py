"""The dense layers of the PointTransformerV3 mirror: nn.Linear with the product on this library's GEMM kernels."""
import os

import torch
import torch.nn as nn


class Linear(nn.Linear):
    """nn.Linear (same parameters, same state-dict keys) whose product runs on this library's GEMM kernels when it is applied to
    fp32 rows on the device (mlp.linear_rows: forward, input gradient, split-K weight gradient); anything else -- CPU tensors, other
    dtypes, inputs that are not [rows, channels], PN2_PTV3_TORCH_LINEAR=1 -- is torch's own linear."""

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and self.weight.dtype == torch.float32 and self.bias is not None
                and self.in_features % 4 == 0 and x.shape[0] >= 1024 and not os.environ.get("PN2_PTV3_TORCH_LINEAR")):
            from ..mlp import linear_rows
            return linear_rows(x if x.is_contiguous() else x.contiguous(), self)
        return super().forward(x)
