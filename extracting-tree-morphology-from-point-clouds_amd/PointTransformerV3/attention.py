"""Serialized patch attention of PointTransformerV3 with the reference's names (Modules/PointTransformerV3/blocks.py:336-507):
`offset2bincount`, `get_padding_and_inverse` (as a function of the offsets), `patch_attention` (the non-flash attention
branch, :466-488, as ONE launch of csrc/ptv3_attention.hip -- the K x K score matrix is never written) and the module
`SerializedAttention` with the reference's constructor, parameter names (`qkv`, `proj`) and `forward(point)`.

The repository's configuration is the non-flash path (PointTransformerV3.py:283-286: enable_flash = enable_rpe = False,
head width 16 in every stage); `enable_flash=True` (flash_attn's fp16 kernels) and `enable_rpe=True` raise.  Training: the
forward then also keeps the rows' log-sum-exp and the backward is two more launches of the same file (dq; dk + dv) that rebuild
the probabilities block by block.  The two linear layers run on the chain GEMM kernels (linear.py).  Parity: the reference module cannot be imported here (spconv / torch_scatter / addict / timm at
module level), so the oracle (oracle/ptv3_attention_port.py) restates the source text: PARITY UNPINNED."""
import ctypes

import torch
import torch.nn as nn

from .. import _hip
from ..mlp import _PRECISION_CODE
from .linear import Linear

ATTENTION_PRECISION = "f32"          # "f32" (exact fp32 MFMA: the parity mode) or "bf16" (bfloat16 operands, fp32 softmax)


def offset2bincount(offset):
    """blocks.py:22-25"""
    return torch.diff(offset, prepend=torch.tensor([0], device=offset.device, dtype=torch.long))


def get_padding_and_inverse(offset, patch_size):
    """blocks.py:384-437 as a function: offset [B] int64 (cumulative cloud sizes) -> (pad [n_pad], unpad [n], cu_seqlens
    [patches + 1] int32).  One host read of the B offsets (the reference indexes them on the host cloud by cloud)."""
    _hip.require_device(offset)
    K = int(patch_size)
    off = [0] + [int(v) for v in offset.tolist()]
    B = len(off) - 1
    offpad, cu_off = [0], [0]
    for i in range(B):
        n = off[i + 1] - off[i]
        npad = (n + K - 1) // K * K if n > K else n                # only clouds longer than a patch are padded (:404-406)
        offpad.append(offpad[-1] + npad)
        cu_off.append(cu_off[-1] + len(range(0, npad, K)))
    dev = offset.device
    tab = torch.tensor([off, offpad, cu_off], dtype=torch.int64).to(dev, non_blocking=True)
    pad = torch.empty(offpad[-1], dtype=torch.int64, device=dev)
    unpad = torch.empty(off[-1], dtype=torch.int64, device=dev)
    cu = torch.empty(cu_off[-1] + 1, dtype=torch.int32, device=dev)
    _hip.call("ptv3_pad", _hip.lib().pn2_ptv3_pad_unpad_i64, tab[0].data_ptr(), tab[1].data_ptr(), tab[2].data_ptr(), B, K, offpad[-1],
              pad.data_ptr(), unpad.data_ptr(), cu.data_ptr(), _hip.stream_ptr(), nbytes=24 * offpad[-1])
    return pad, unpad, cu


def _attention_forward(qkv, order, K, H, scale, want_lse):
    C = qkv.shape[1] // 3
    n_rows = qkv.shape[0] if order is None else order.numel()
    out = torch.empty(n_rows, C, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(n_rows, H, dtype=torch.float32, device=qkv.device) if want_lse else None
    _hip.call("ptv3_attention", _hip.lib().pn2_ptv3_patch_attention_lse_f32, qkv.data_ptr(), qkv.stride(0), _hip.ptr(order), n_rows, K, H,
              C // H, ctypes.c_float(float(scale)), out.data_ptr(), _hip.ptr(lse), _PRECISION_CODE[ATTENTION_PRECISION],
              _hip.stream_ptr(), nbytes=16 * n_rows * C, flops=4 * n_rows * K * C)
    return out, lse


class _PatchAttentionFn(torch.autograd.Function):
    """Training path of `patch_attention`: the forward keeps (qkv, order, out, lse); the backward writes the gradient of the
    GATHERED rows and adds the rows several padded positions read (the repeated tail of a cloud's last patch) with index_add."""

    @staticmethod
    def forward(ctx, qkv, order, K, H, scale):
        out, lse = _attention_forward(qkv, order, K, H, scale, True)
        ctx.save_for_backward(qkv, order, out, lse)
        ctx.cfg = (K, H, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, order, out, lse = ctx.saved_tensors
        K, H, scale = ctx.cfg
        dout = _hip.f32(dout).contiguous()
        n_rows, C = out.shape
        drows = torch.empty(n_rows, 3 * C, dtype=torch.float32, device=qkv.device)
        _hip.call("ptv3_attention_bwd", _hip.lib().pn2_ptv3_patch_attention_bwd_f32, qkv.data_ptr(), qkv.stride(0), _hip.ptr(order), n_rows,
                  K, H, C // H, ctypes.c_float(float(scale)), out.data_ptr(), lse.data_ptr(), dout.data_ptr(), drows.data_ptr(),
                  _hip.stream_ptr(), nbytes=28 * n_rows * C, flops=14 * n_rows * K * C)
        if order is None:
            return drows, None, None, None, None
        return torch.zeros_like(qkv).index_add_(0, order, drows), None, None, None, None


def patch_attention(qkv, order, patch_size, num_heads, scale):
    """qkv [N, 3 C] fp32 rows ([3][H][C / H]), order [N'] int64 or None (rows of qkv behind the padded positions: the
    reference's `qkv[order]`), N' a multiple of patch_size -> feat [N', C] = softmax((q scale) k^T) v per patch and head.
    Differentiable w.r.t. qkv."""
    _hip.require_device(qkv, order)
    qkv = _hip.f32(qkv)
    if qkv.stride(1) != 1 or qkv.stride(0) % 4:
        qkv = qkv.contiguous()
    if order is not None:
        order = order.to(torch.int64).contiguous()
    K, H = int(patch_size), int(num_heads)
    if qkv.requires_grad and torch.is_grad_enabled():
        return _PatchAttentionFn.apply(qkv, order, K, H, float(scale))
    return _attention_forward(qkv, order, K, H, scale, False)[0]


class SerializedAttention(nn.Module):
    """blocks.py:336-507.  `point`: any mapping / attribute holder with feat [N, C], offset [B], serialized_order and
    serialized_inverse [orders, N] (the reference's Point after serialization, blocks.py:98-150); caches pad / unpad /
    cu_seqlens in it under the reference's keys."""

    def __init__(self, channels, num_heads, patch_size, qkv_bias=True, qk_scale=None, attn_drop=0.0, proj_drop=0.0, order_index=0,
                 enable_rpe=False, enable_flash=False, upcast_attention=False, upcast_softmax=False):
        super().__init__()
        assert channels % num_heads == 0
        if enable_flash:
            raise NotImplementedError("enable_flash=True is flash_attn's fp16 path; this build serves the non-flash branch")
        if enable_rpe:
            raise NotImplementedError("relative position encoding (RPE) is not built")
        if attn_drop or proj_drop:
            raise NotImplementedError("attention / projection dropout is not built (the repository trains with 0.0)")
        self.channels, self.num_heads = channels, num_heads
        self.scale = qk_scale or (channels // num_heads) ** -0.5
        self.order_index = order_index
        self.patch_size_max, self.patch_size = patch_size, 0
        self.qkv = Linear(channels, channels * 3, bias=qkv_bias)
        self.proj = Linear(channels, channels)

    @staticmethod
    def _get(point, key):
        return point[key] if isinstance(point, dict) else getattr(point, key)

    @staticmethod
    def _has(point, key):
        return key in point if isinstance(point, dict) else hasattr(point, key)

    @staticmethod
    def _set(point, key, value):
        if isinstance(point, dict):
            point[key] = value
        else:
            setattr(point, key, value)

    def get_padding_and_inverse(self, point):
        keys = ("pad", "unpad", "cu_seqlens_key")
        if not all(self._has(point, k) for k in keys):
            for k, v in zip(keys, get_padding_and_inverse(self._get(point, "offset"), self.patch_size)):
                self._set(point, k, v)
        return tuple(self._get(point, k) for k in keys)

    def forward(self, point):
        offset = self._get(point, "offset")
        self.patch_size = min(int(offset2bincount(offset).min()), self.patch_size_max)   # :451-454 (no masking: shortest cloud)
        pad, unpad, _ = self.get_padding_and_inverse(point)
        order = self._get(point, "serialized_order")[self.order_index][pad]
        inverse = unpad[self._get(point, "serialized_inverse")[self.order_index]]
        qkv = self.qkv(self._get(point, "feat"))
        feat = patch_attention(qkv, order, self.patch_size, self.num_heads, self.scale)
        feat = self.proj(feat[inverse])
        self._set(point, "feat", feat)
        return point
