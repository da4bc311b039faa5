"""Conditional positional encoding of PointTransformerV3: the submanifold 3 x 3 x 3 sparse convolution in front of every Block
(Modules/PointTransformerV3/blocks.py:561-568, `spconv.SubMConv3d(channels, channels, kernel_size=3, bias=True,
indice_key=...)` on the voxels of `Point.sparsify`, :153-190) as two launches of csrc/ptv3_cpe.hip:

    nbr = subm_neighbors(batch, grid_coord)      # [N, 27] int32, once per stage (spconv's indice_key)
    out = SubMConv3d(C, C)(feat, nbr)            # [N, C]

`SubMConv3d` keeps spconv 2.x's parameter names and shapes (`weight` [C_out, 3, 3, 3, C_in], `bias` [C_out]), so a state dict
of the reference's CPE loads.  Semantics = the operator's definition: the dense cross-correlation conv3d(padding = 1) of the
voxel grid evaluated at the active voxels, the three grid axes in the order of `grid_coord`'s columns (spconv's D, H, W).
spconv itself is a CUDA-only dependency that is absent here: PARITY UNPINNED against it, pinned against
torch.nn.functional.conv3d (tests/test_ptv3_cpe.py).  Training: the input gradient is the same kernel on the mirrored stencil
(distinct voxels: j(i, d) = j' <=> j(j', -d) = i, which is also what spconv's indice pairs assume), the weight gradient a gathered
split-K contraction (`pn2_ptv3_subm_wgrad_f32`), both fp32."""
import math

import torch
import torch.nn as nn

from .. import _hip

CONV_PRECISION = "f32"          # "f32" (exact fp32 MFMA: the parity mode) or "bf16" (bfloat16 operands for the layers >= 64 wide)


def subm_neighbors(batch, grid_coord, kernel_size=3):
    """batch [N] int64 cloud ids (or None: one cloud), grid_coord [N, 3] integer voxel coordinates (0 .. 65533) -> nbr
    [N, kernel_size^3] int32: row i, column ((dx + r) * k + (dy + r)) * k + (dz + r) (r = k // 2) = index of the voxel of the same
    cloud at grid_coord[i] + (dx, dy, dz), or -1."""
    from .. import ops
    _hip.require_device(grid_coord)
    lib = _hip.lib()
    grid = grid_coord.to(torch.int32).contiguous()
    N, noff = grid.shape[0], int(kernel_size) ** 3
    dev = grid.device
    b = None if batch is None else batch.to(device=dev, dtype=torch.int64).contiguous()
    nbr = torch.empty(N, noff, dtype=torch.int32, device=dev)
    ws = torch.empty(lib.pn2_ptv3_subm_workspace_bytes(N), dtype=torch.uint8, device=dev)
    _hip.call("ptv3_subm_neighbors", lib.pn2_ptv3_subm_neighbors_i32, _hip.ptr(b), grid.data_ptr(), N, int(kernel_size), nbr.data_ptr(),
              ws.data_ptr(), ws.numel(), ops.status_word(dev).data_ptr(), _hip.stream_ptr(), nbytes=N * (20 + noff * 16))
    return nbr


class SubMConv3d(nn.Module):
    """spconv.SubMConv3d(in_channels, out_channels, kernel_size, bias=...) on a neighbour table (see the module docstring).
    kernel_size 3 (the CPE) or 5 (the stem of Embedding, blocks.py:783-791; `padding` is meaningless for a submanifold conv and
    ignored, as spconv does).  An input width that is not a multiple of 16 (the stem's 4-6 raw features) is zero-padded."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=None, bias=True, indice_key=None):
        super().__init__()
        if kernel_size not in (3, 5):
            raise NotImplementedError("SubMConv3d: kernel_size 3 (CPE, blocks.py:565) or 5 (stem, :787)")
        if out_channels % 32:
            raise NotImplementedError("SubMConv3d: C_out a multiple of 32 (every stage of the repository's model is)")
        self.in_channels, self.out_channels, self.kernel_size, self.indice_key = in_channels, out_channels, kernel_size, indice_key
        k = kernel_size
        self.weight = nn.Parameter(torch.empty(out_channels, k, k, k, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()
        self._packed = None
        self._packed16 = None

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.kernel_size ** 3 * self.in_channels)
            nn.init.uniform_(self.bias, -bound, bound)

    @property
    def _cin_padded(self):
        return (self.in_channels + 15) // 16 * 16

    def _offset_major(self):
        """weight [C_out, kx, ky, kz, C_in] -> [k^3][C_in padded][C_out] (the kernel's slab per offset), cached per parameter
        version."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if self._packed is None or self._packed[0] != key:
            noff = self.kernel_size ** 3
            w = self.weight.detach().float().permute(1, 2, 3, 4, 0).reshape(noff, self.in_channels, self.out_channels)
            if self._cin_padded != self.in_channels:
                w = torch.nn.functional.pad(w, (0, 0, 0, self._cin_padded - self.in_channels))
            self._packed = (key, w.contiguous())
        return self._packed[1]

    def _offset_major_bf16(self):
        """weight -> bfloat16 [27][C_out][C_in] (eight consecutive input channels of a column contiguous), cached per version."""
        key = (self.weight.data_ptr(), self.weight._version, self.weight.device)
        if self._packed16 is None or self._packed16[0] != key:
            w = self.weight.detach().float().reshape(self.out_channels, 27, self.in_channels).permute(1, 0, 2)
            self._packed16 = (key, w.to(torch.bfloat16).contiguous())
        return self._packed16[1]

    def _prepare(self, feat, nbr):
        _hip.require_device(feat, nbr)
        feat = _hip.f32(feat)
        cin = self._cin_padded
        if cin != self.in_channels:
            feat = torch.nn.functional.pad(feat, (0, cin - self.in_channels))
        if feat.stride(1) != 1 or feat.stride(0) % 4:
            feat = feat.contiguous()
        N = feat.shape[0]
        if nbr.shape != (N, self.kernel_size ** 3):
            raise RuntimeError(f"SubMConv3d: neighbour table {tuple(nbr.shape)} does not fit {N} voxels, kernel_size {self.kernel_size}")
        return feat

    def forward(self, feat, nbr):
        feat = self._prepare(feat, nbr)
        if torch.is_grad_enabled() and (feat.requires_grad or self.weight.requires_grad):
            return _SubMConvFn.apply(feat, nbr, self.weight, self.bias, self)
        w = self._offset_major()
        w16 = self._offset_major_bf16() if (CONV_PRECISION == "bf16" and self.kernel_size == 3 and self._cin_padded >= 64) else None
        b = None if self.bias is None else self.bias.detach().float().contiguous()
        return _conv(feat, nbr, self.kernel_size, w, w16, b, self.out_channels)


def _conv(feat, nbr, kernel_size, w, w16, bias, cout):
    """One launch of pn2_ptv3_subm_conv_f32: feat [N, C_in] (C_in = w.shape[1]), w [k^3, C_in, cout] -> [N, cout]."""
    N, cin = feat.shape[0], w.shape[1]
    out = torch.empty(N, cout, dtype=torch.float32, device=feat.device)
    _hip.call("ptv3_subm_conv", _hip.lib().pn2_ptv3_subm_conv_f32, feat.data_ptr(), feat.stride(0), nbr.data_ptr(), kernel_size,
              w.data_ptr(), _hip.ptr(w16), _hip.ptr(bias), N, cin, cout, out.data_ptr(), out.stride(0), _hip.stream_ptr(),
              nbytes=4 * N * (nbr.shape[1] + cin + cout), flops=2 * nbr.shape[1] * N * cin * cout)
    return out


class _SubMConvFn(torch.autograd.Function):
    """out = SubMConv3d(feat) under autograd.  feat arrives padded to the kernel's input width; `weight` / `bias` are the module's
    parameters in spconv's layout."""

    @staticmethod
    def forward(ctx, feat, nbr, weight, bias, mod):
        w = mod._offset_major()
        w16 = mod._offset_major_bf16() if (CONV_PRECISION == "bf16" and mod.kernel_size == 3 and mod._cin_padded >= 64) else None
        b = None if bias is None else bias.detach().float().contiguous()
        ctx.save_for_backward(feat, nbr, w)
        ctx.mod = mod
        ctx.has_bias = bias is not None
        return _conv(feat, nbr, mod.kernel_size, w, w16, b, mod.out_channels)

    @staticmethod
    def backward(ctx, dout):
        feat, nbr, w = ctx.saved_tensors
        mod = ctx.mod
        k, cin, cout = mod.kernel_size, mod._cin_padded, mod.out_channels
        dout = _hip.f32(dout)
        if dout.stride(1) != 1 or dout.stride(0) % 4:
            dout = dout.contiguous()
        N = feat.shape[0]
        dfeat = dweight = dbias = None
        if ctx.needs_input_grad[0]:
            if cin % 32 or cout % 16:
                raise NotImplementedError("SubMConv3d backward: the input gradient needs C_in a multiple of 32 (the CPE layers; the "
                                          "stem's raw features take no gradient)")
            dfeat = _conv(dout, nbr, k, w.flip(0).transpose(1, 2).contiguous(), None, None, cin)   # W'[d][co][ci] = W[-d][ci][co]
        if ctx.needs_input_grad[2]:
            lib = _hip.lib()
            dw = torch.empty(k ** 3, cin, cout, dtype=torch.float32, device=feat.device)
            ws = torch.empty(lib.pn2_ptv3_subm_wgrad_workspace_bytes(N, k, cin, cout), dtype=torch.uint8, device=feat.device)
            _hip.call("ptv3_subm_wgrad", lib.pn2_ptv3_subm_wgrad_f32, feat.data_ptr(), feat.stride(0), nbr.data_ptr(), k, dout.data_ptr(),
                      dout.stride(0), N, cin, cout, dw.data_ptr(), ws.data_ptr(), ws.numel(), _hip.stream_ptr(),
                      nbytes=4 * N * k ** 3 * (1 + cout), flops=2 * k ** 3 * N * cin * cout)
            # [k^3][C_in padded][C_out] -> spconv's [C_out, k, k, k, C_in]
            dweight = dw[:, :mod.in_channels, :].reshape(k, k, k, mod.in_channels, cout).permute(4, 0, 1, 2, 3).contiguous()
        if ctx.has_bias and ctx.needs_input_grad[3]:
            dbias = dout.sum(0)
        return dfeat, None, dweight, dbias, None
