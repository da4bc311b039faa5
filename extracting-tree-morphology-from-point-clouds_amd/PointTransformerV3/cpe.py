"""Conditional positional encoding of PointTransformerV3: the submanifold 3 x 3 x 3 sparse convolution in front of every Block
(Modules/PointTransformerV3/blocks.py:561-568, `spconv.SubMConv3d(channels, channels, kernel_size=3, bias=True,
indice_key=...)` on the voxels of `Point.sparsify`, :153-190) as two launches of csrc/ptv3_cpe.hip:

    nbr = subm_neighbors(batch, grid_coord)      # [N, 27] int32, once per stage (spconv's indice_key)
    out = SubMConv3d(C, C)(feat, nbr)            # [N, C]

`SubMConv3d` keeps spconv 2.x's parameter names and shapes (`weight` [C_out, 3, 3, 3, C_in], `bias` [C_out]), so a state dict
of the reference's CPE loads.  Semantics = the operator's definition: the dense cross-correlation conv3d(padding = 1) of the
voxel grid evaluated at the active voxels, the three grid axes in the order of `grid_coord`'s columns (spconv's D, H, W).
spconv itself is a CUDA-only dependency that is absent here: PARITY UNPINNED against it, pinned against
torch.nn.functional.conv3d (tests/test_ptv3_cpe.py).  Inference only: no backward is built, a tensor that requires grad
raises."""
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

    def forward(self, feat, nbr):
        _hip.require_device(feat, nbr)
        if torch.is_grad_enabled() and feat.requires_grad:
            raise NotImplementedError("SubMConv3d: the backward pass is not built (inference only)")
        feat = _hip.f32(feat)
        cin = self._cin_padded
        if cin != self.in_channels:
            feat = torch.nn.functional.pad(feat, (0, cin - self.in_channels))
        if feat.stride(1) != 1 or feat.stride(0) % 4:
            feat = feat.contiguous()
        N = feat.shape[0]
        if nbr.shape != (N, self.kernel_size ** 3):
            raise RuntimeError(f"SubMConv3d: neighbour table {tuple(nbr.shape)} does not fit {N} voxels, kernel_size {self.kernel_size}")
        out = torch.empty(N, self.out_channels, dtype=torch.float32, device=feat.device)
        w = self._offset_major()
        w16 = self._offset_major_bf16() if (CONV_PRECISION == "bf16" and self.kernel_size == 3 and cin >= 64) else None
        b = None if self.bias is None else self.bias.detach().float().contiguous()
        _hip.call("ptv3_subm_conv", _hip.lib().pn2_ptv3_subm_conv_f32, feat.data_ptr(), feat.stride(0), nbr.data_ptr(), self.kernel_size,
                  w.data_ptr(), _hip.ptr(w16), _hip.ptr(b), N, cin, self.out_channels, out.data_ptr(), out.stride(0), _hip.stream_ptr(),
                  nbytes=4 * N * (nbr.shape[1] + cin + self.out_channels), flops=2 * nbr.shape[1] * N * cin * self.out_channels)
        return out
