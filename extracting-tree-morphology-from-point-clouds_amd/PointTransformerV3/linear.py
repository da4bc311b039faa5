"""The dense layers and the LayerNorm of the PointTransformerV3 mirror: nn.Linear with the product on this library's GEMM
kernels, nn.LayerNorm on csrc/ptv3_norm.hip (same parameters and state-dict keys as torch's modules)."""
import ctypes
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


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, eps):
        from .. import _hip
        lib = _hip.lib()
        rows, C = x.shape
        y = torch.empty_like(x)
        keep = any(ctx.needs_input_grad[:3])          # (grad mode is off inside forward: ask what the graph needs)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device) if keep else None
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if keep else None
        _hip.call("layer_norm_fwd", lib.pn2_layer_norm_fwd_f32, x.data_ptr(), x.stride(0), _hip.ptr(weight), _hip.ptr(bias),
                  ctypes.c_float(float(eps)), rows, C, y.data_ptr(), y.stride(0), _hip.ptr(mean), _hip.ptr(rstd), _hip.stream_ptr(),
                  nbytes=8 * rows * C)
        ctx.save_for_backward(x, weight, mean, rstd)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        from .. import _hip
        x, weight, mean, rstd = ctx.saved_tensors
        lib = _hip.lib()
        rows, C = x.shape
        dy = dy.float()
        if dy.stride(1) != 1 or dy.stride(0) % 4 or dy.data_ptr() % 16:
            dy = dy.contiguous()
        dx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=x.device) if weight is not None and ctx.needs_input_grad[1] else None
        db = torch.empty(C, dtype=torch.float32, device=x.device) if ctx.has_bias and ctx.needs_input_grad[2] else None
        ws = torch.empty(lib.pn2_layer_norm_bwd_workspace_bytes(rows, C), dtype=torch.uint8, device=x.device)
        _hip.call("layer_norm_bwd", lib.pn2_layer_norm_bwd_f32, dy.data_ptr(), dy.stride(0), x.data_ptr(), x.stride(0), mean.data_ptr(),
                  rstd.data_ptr(), _hip.ptr(weight), rows, C, dx.data_ptr(), dx.stride(0), _hip.ptr(dg), _hip.ptr(db), ws.data_ptr(),
                  ws.numel(), _hip.stream_ptr(), nbytes=12 * rows * C)
        return dx, dg, db, None


class LayerNorm(nn.LayerNorm):
    """nn.LayerNorm over the last dimension of [rows, C] fp32 rows on the device, C in {32, 64, 128, 256, 512}: csrc/ptv3_norm.hip
    (forward and backward); anything else -- and PN2_PTV3_TORCH_LAYERNORM=1 -- is torch's own."""

    def forward(self, x):
        if (x.is_cuda and x.dim() == 2 and x.dtype == torch.float32 and len(self.normalized_shape) == 1 and x.shape[0] > 0
                and x.shape[1] in (32, 64, 128, 256, 512) and (self.weight is None or self.weight.dtype == torch.float32)
                and not os.environ.get("PN2_PTV3_TORCH_LAYERNORM")):
            if x.stride(1) != 1 or x.stride(0) % 4 or x.data_ptr() % 16:
                x = x.contiguous()
            return _LayerNormFn.apply(x, self.weight, self.bias, self.eps)
        return super().forward(x)
