"""L1 modules with the reference's constructors, attribute names and state-dict layout
(Modules/PointNet2/blocks.py): ConvHead, MLP, PointNetSetAbstraction, PointNetSetAbstractionMsg,
PointNetFeaturePropagation.

I/O is channel-first like the reference (xyz [B,3,N], points [B,D,N]); internally every feature map is a
channels-last row buffer and the channel-first tensors handed back are permuted VIEWS of those buffers, so
chaining SA -> SA -> FP never transposes or copies a feature map.
"""
import torch
import torch.nn as nn

from .. import ops
from ..mlp import chain_rows, group_bn_rows, group_hoist_ok, hoist_ok, hoisted_conv, interp_bn_rows
from .pointnet2_utils import *  # noqa: F401,F403  (the reference re-exports the L0 ops from here)
from .pointnet2_utils import _sample_and_group_i32, _draw_start


def _rows(x_cf):
    """[B,C,N] channel-first (possibly a permuted view of channels-last storage) -> ([B*N, C] rows, B, N)."""
    B, C, N = x_cf.shape
    return x_cf.permute(0, 2, 1).reshape(B * N, C), B, N


class ConvHead(nn.Module):
    """Per-point head: (Conv1d -> norm -> ReLU) x (num_layers-1) -> Conv1d (reference blocks.py:7-35).
    ``self.net`` holds the same Sequential as the reference so keys stay ``net.{0,1,3}.*``."""

    def __init__(self, in_channels, out_channels, norm_fn=None, num_layers=2):
        super().__init__()
        mods, c = [], in_channels
        for _ in range(num_layers - 1):
            mods.append(nn.Conv1d(c, c, kernel_size=1))
            if norm_fn is not None:
                mods.append(norm_fn(c))
            mods.append(nn.ReLU(inplace=True))
        mods.append(nn.Conv1d(c, out_channels, kernel_size=1))
        self.net = nn.Sequential(*mods)

    def _layers(self):
        mods, out, i = list(self.net), [], 0
        while i < len(mods):
            conv, bn, relu = mods[i], None, False
            i += 1
            if i < len(mods) and isinstance(mods[i], nn.modules.batchnorm._BatchNorm):
                bn = mods[i]
                i += 1
            if i < len(mods) and isinstance(mods[i], nn.ReLU):
                relu = True
                i += 1
            out.append((conv, bn, relu))
        return out

    def forward(self, x):
        """x [B,C_in,N] -> [B,C_out,N]."""
        rows, B, N = _rows(x)
        y = chain_rows(rows, self._layers())
        return y.view(B, N, -1).permute(0, 2, 1)


class MLP(nn.Sequential):
    """Linear/norm/ReLU stack of the reference (blocks.py:37-55); not used by PointNet2 itself."""

    def __init__(self, in_channels, out_channels, norm_fn=None, num_layers=2):
        mods = []
        for _ in range(num_layers - 1):
            mods.append(nn.Linear(in_channels, in_channels))
            if norm_fn:
                mods.append(norm_fn(in_channels))
            mods.append(nn.ReLU())
        mods.append(nn.Linear(in_channels, out_channels))
        super().__init__(*mods)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)
        nn.init.normal_(self[-1].weight, 0, 0.01)
        nn.init.constant_(self[-1].bias, 0)


def _group_mlp_max(grouped, convs, bns, seg_off=None, coords_first=0):
    """grouped [B,S,K,C] -> per-group MLP and max over K -> [B,S,C_out] (reference blocks.py:93-98).
    seg_off: row offsets of the mini-batches the B clouds belong to (whole-tree execution, streaming.py).
    coords_first: number of leading channels that are centred coordinates (no gradient flows into them: the input
    gradient of the chain is then computed for the feature channels only)."""
    B, S, K, C = grouped.shape
    y = chain_rows(grouped.reshape(B * S * K, C), [(c, b, True) for c, b in zip(convs, bns)], pool_k=K, seg_off=seg_off,
                   dx_first_col=coords_first)
    return y.view(B, S, -1)


def _hoisted_group_mlp_max(xyz_t, new_xyz, pts_t, idx, convs, bns, xyz_last, seg_off=None):
    """_group_mlp_max of the grouped rows [xyz[idx] - centre, pts[idx]] without forming them -> [B,S,C_out]."""
    B, S, K = idx.shape
    first = group_bn_rows(xyz_t, new_xyz, pts_t, idx, convs[0], bns[0], xyz_last=xyz_last, seg_off=seg_off)
    y = chain_rows(first, [(c, b, True) for c, b in zip(convs[1:], bns[1:])], pool_k=K, seg_off=seg_off)
    return y.view(B, S, -1)


class PointNetSetAbstraction(nn.Module):
    def __init__(self, npoint, radius, nsample, in_channel, mlp, group_all):
        super().__init__()
        self.npoint, self.radius, self.nsample, self.group_all = npoint, radius, nsample, group_all
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        c = in_channel
        for out_c in mlp:
            self.mlp_convs.append(nn.Conv2d(c, out_c, 1))
            self.mlp_bns.append(nn.BatchNorm2d(out_c))
            c = out_c

    def forward(self, xyz, points):
        """xyz [B,3,N], points [B,D,N] or None -> new_xyz [B,3,S], new_points [B,C_out,S] (blocks.py:74-100)."""
        xyz_t = xyz.permute(0, 2, 1)
        pts_t = None if points is None else points.permute(0, 2, 1)
        if self.group_all:
            new_xyz, grouped = sample_and_group_all(xyz_t, pts_t)
        else:
            B, N, _ = xyz_t.shape
            D = 0 if pts_t is None else pts_t.shape[2]
            if group_hoist_ok(self.mlp_convs[0], self.mlp_bns[0], len(self.mlp_convs), B, N, self.npoint, min(self.nsample, N), D,
                              xyz_t.device):
                # the feature share of the first conv runs on the N source points, not on the S*K grouped rows (mlp.group_bn_rows)
                _, new_xyz = ops.furthest_point_sample(xyz_t, self.npoint, _draw_start(B, N, xyz_t.device))
                idx = ops.ball_query(self.radius, self.nsample, xyz_t, new_xyz)
                pooled = _hoisted_group_mlp_max(xyz_t, new_xyz, pts_t, idx, self.mlp_convs, self.mlp_bns, False)
                return new_xyz.permute(0, 2, 1), pooled.permute(0, 2, 1)
            new_xyz, grouped, _, _ = _sample_and_group_i32(self.npoint, self.radius, self.nsample, xyz_t, pts_t)
        # [xyz - centroid, feats]: the 3 leading channels need no gradient (GroupPoints differentiates the features only)
        pooled = _group_mlp_max(grouped, self.mlp_convs, self.mlp_bns, coords_first=0 if self.group_all or pts_t is None else 3)
        return new_xyz.permute(0, 2, 1), pooled.permute(0, 2, 1)


class PointNetSetAbstractionMsg(nn.Module):
    """Multi-scale grouping: one FPS, then ball query / group / MLP / max per radius, channels concatenated
    (reference blocks.py:103-160).  Grouped channel order is [feats, xyz - centroid]."""

    def __init__(self, npoint, radius_list, nsample_list, in_channel, mlp_list):
        super().__init__()
        self.npoint, self.radius_list, self.nsample_list = npoint, radius_list, nsample_list
        self.conv_blocks, self.bn_blocks = nn.ModuleList(), nn.ModuleList()
        for widths in mlp_list:
            convs, bns, c = nn.ModuleList(), nn.ModuleList(), in_channel
            for out_c in widths:
                convs.append(nn.Conv2d(c, out_c, 1))
                bns.append(nn.BatchNorm2d(out_c))
                c = out_c
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward(self, xyz, points):
        xyz_t = xyz.permute(0, 2, 1)
        pts_t = None if points is None else points.permute(0, 2, 1)
        B, N, _ = xyz_t.shape
        _, new_xyz = ops.furthest_point_sample(xyz_t, self.npoint, _draw_start(B, N, xyz_t.device))
        scales = []
        for radius, K, convs, bns in zip(self.radius_list, self.nsample_list, self.conv_blocks, self.bn_blocks):
            idx = ops.ball_query(radius, K, xyz_t, new_xyz)
            if pts_t is not None and group_hoist_ok(convs[0], bns[0], len(convs), B, N, idx.shape[1], idx.shape[2], pts_t.shape[2],
                                                    xyz_t.device):
                scales.append(_hoisted_group_mlp_max(xyz_t, new_xyz, pts_t, idx, convs, bns, True))
                continue
            grouped = ops.GroupPoints.apply(xyz_t, new_xyz, pts_t, idx, True)
            scales.append(_group_mlp_max(grouped, convs, bns))
        return new_xyz.permute(0, 2, 1), torch.cat(scales, dim=-1).permute(0, 2, 1)


class PointNetFeaturePropagation(nn.Module):
    def __init__(self, in_channel, mlp):
        super().__init__()
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        c = in_channel
        for out_c in mlp:
            self.mlp_convs.append(nn.Conv1d(c, out_c, 1))
            self.mlp_bns.append(nn.BatchNorm1d(out_c))
            c = out_c

    def forward(self, xyz1, xyz2, points1, points2, lazy_rows=False):
        """xyz1 [B,3,N] dense, xyz2 [B,3,S] sampled, points1 [B,D1,N] or None, points2 [B,D2,S] -> [B,D',N]
        (reference blocks.py:174-216): 3-NN inverse-distance interpolation, skip concat, MLP.
        lazy_rows: return (rows, B, N) with rows a mlp.LazyRows when the last BatchNorm + ReLU can be left to the consumer
        (the prediction heads, linked chains) -- otherwise ([B*N, D'] rows, B, N)."""
        x1, x2 = xyz1.permute(0, 2, 1), xyz2.permute(0, 2, 1)
        p2 = points2.permute(0, 2, 1)
        p1 = None if points1 is None else points1.permute(0, 2, 1)
        B, N, _ = x1.shape
        S = x2.shape[1]
        layers = [(c, b, True) for c, b in zip(self.mlp_convs, self.mlp_bns)]
        if S == 1:
            feats = p2.repeat(1, N, 1)
            if p1 is not None:
                feats = torch.cat([p1, feats], dim=-1)
        else:
            idx, w = ops.three_nn(x1, x2)
            if p1 is None and hoist_ok(self.mlp_convs[0], self.mlp_bns[0], len(self.mlp_convs), x1.device):
                # no skip connection (fp1): conv(interp(P)) = interp(conv(P)) -- the first contraction runs over the B*S sampled
                # rows, the interpolation carries the layer's BatchNorm statistics (mlp.interp_bn_rows)
                q = hoisted_conv(p2.reshape(B * S, -1), self.mlp_convs[0])
                feats = interp_bn_rows(q.view(B, S, -1), idx, w, self.mlp_bns[0], bias=self.mlp_convs[0].bias)
                layers = layers[1:]
            else:
                feats = ops.ThreeInterpolateConcat.apply(p1, p2, idx, w).reshape(B * N, -1)
        y = chain_rows(feats.reshape(B * N, -1) if torch.is_tensor(feats) else feats, layers, lazy_out=lazy_rows)
        if lazy_rows:
            return y, B, N
        return y.view(B, N, -1).permute(0, 2, 1)
