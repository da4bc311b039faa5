"""PointTransformerV3's building blocks with the reference's names, constructors and parameter layout
(Modules/PointTransformerV3/blocks.py): `Point` (:65-191), `PointModule` / `PointSequential` (:194-269), `MLP` (:510-533),
`Block` (:536-623), `SerializedPooling` (:626-729), `SerializedUnpooling` (:732-767), `Embedding` (:770-800).

What runs where: the three operators that make up the backbone's time -- the space-filling-curve codes (csrc/serialize.hip),
the serialized patch attention (csrc/ptv3_attention.hip) and the submanifold convolutions of the stem and of every Block's
conditional positional encoding (csrc/ptv3_cpe.hip) -- are this library's kernels, and so are the dense layers' GEMMs (linear.py); the LayerNorm / BatchNorm / GELU
layers are plain library calls (torch), and so is the index bookkeeping of the pooling (unique / sort / segment reductions: the
reference's torch + torch_scatter calls, torch_scatter.segment_csr spelled torch.segment_reduce).  The reference's dependencies
(spconv, torch_scatter, addict, timm) are absent here, so `Point` is a plain dict with attribute access, a voxel set carries its
neighbour tables instead of a spconv.SparseConvTensor.

Inference and training (the attention and the sparse convolutions have backward kernels, attention.py / cpe.py; the rest is
torch autograd); PARITY UNPINNED against the reference, whose module cannot be imported here -- the restatement the tests use is oracle/ptv3_model_port.py."""
import math
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from .attention import SerializedAttention
from .linear import LayerNorm, Linear
from .cpe import SubMConv3d, subm_neighbors
from .serialization.default import serialize


def offset2bincount(offset):
    return torch.diff(offset, prepend=torch.tensor([0], device=offset.device, dtype=torch.long))


def offset2batch(offset):
    bincount = offset2bincount(offset)
    return torch.arange(len(bincount), device=offset.device, dtype=torch.long).repeat_interleave(bincount)


def batch2offset(batch):
    return torch.cumsum(batch.bincount(), dim=0).long()


class Point(dict):
    """blocks.py:65-191: a dict of per-point properties with attribute access ("coord", "grid_coord", "feat", "batch" /
    "offset", ...).  `sparsify` keeps the reference's "sparse_shape"; the spconv tensor is replaced by per-kernel-size neighbour
    tables built on first use and shared by every convolution on this voxel set (spconv's indice_key)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if "batch" not in self.keys() and "offset" in self.keys():
            self["batch"] = offset2batch(self["offset"])
        elif "offset" not in self.keys() and "batch" in self.keys():
            self["offset"] = batch2offset(self["batch"])

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def _grid(self):
        if "grid_coord" not in self.keys():
            assert {"grid_size", "coord"}.issubset(self.keys())
            self["grid_coord"] = torch.div(self.coord - self.coord.min(0)[0], self.grid_size, rounding_mode="trunc").int()
        return self["grid_coord"]

    def serialization(self, order="z", depth=None, shuffle_orders=False):
        """:98-148; all orders' codes come from one launch (serialization.default.serialize)."""
        assert "batch" in self.keys()
        grid = self._grid()
        if depth is None:
            depth = int(grid.max()).bit_length()
        self["serialized_depth"] = depth
        assert depth * 3 + len(self.offset).bit_length() <= 63 and depth <= 16
        order = [order] if isinstance(order, str) else list(order)
        code, perm, inverse = serialize(grid, self.batch, depth, order)
        if shuffle_orders:
            p = torch.randperm(code.shape[0])
            code, perm, inverse = code[p], perm[p], inverse[p]
        self["serialized_code"], self["serialized_order"], self["serialized_inverse"] = code, perm, inverse

    def sparsify(self, pad=96):
        """:150-190 without spconv: the voxel set is (batch, grid_coord); convolutions ask `neighbors(kernel_size)`."""
        assert {"feat", "batch"}.issubset(self.keys())
        grid = self._grid()
        if "sparse_shape" not in self.keys():
            self["sparse_shape"] = torch.add(torch.max(grid, dim=0).values, pad).tolist()
        self.setdefault("_neighbors", {})

    def neighbors(self, kernel_size):
        tabs = self.setdefault("_neighbors", {})
        if kernel_size not in tabs:
            tabs[kernel_size] = subm_neighbors(self.batch, self._grid(), kernel_size)
        return tabs[kernel_size]


class PointModule(nn.Module):
    """:194-201 placeholder: a module that takes and returns a Point."""


class PointSequential(PointModule):
    """:203-269: a sequential container that feeds Point modules the Point, sparse convolutions the voxel set and everything
    else the feature rows."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        return list(self._modules.values())[idx]

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        for module in self._modules.values():
            if isinstance(module, PointModule):
                input = module(input)
            elif isinstance(module, SubMConv3d):
                input.feat = module(input.feat, input.neighbors(module.kernel_size))
            elif isinstance(input, Point):
                input.feat = module(input.feat)
            else:
                input = module(input)
        return input


class MLP(nn.Module):
    """:510-533"""

    def __init__(self, in_channels, hidden_channels=None, out_channels=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        out_channels = out_channels or in_channels
        hidden_channels = hidden_channels or in_channels
        self.fc1 = Linear(in_channels, hidden_channels)
        self.act = act_layer()
        self.fc2 = Linear(hidden_channels, out_channels)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class DropPath(nn.Module):
    """timm.layers.DropPath (stochastic depth, scale_by_keep=True): in training every ROW of the residual branch (the first
    dimension is the sample dimension -- here a point, as in the reference, where the module sees `point.feat` [N, C]) is kept with
    probability 1 - drop_prob and scaled by 1 / (1 - drop_prob); the identity at inference or with rate 0."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = drop_prob, scale_by_keep

    def forward(self, x):
        if not self.training or self.drop_prob == 0.0:
            return x
        keep = 1.0 - self.drop_prob
        mask = x.new_empty((x.shape[0],) + (1,) * (x.ndim - 1)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class Block(PointModule):
    """:536-623: conditional positional encoding (sparse conv + Linear + norm), serialized attention, MLP, pre- or post-norm."""

    def __init__(self, channels, num_heads, patch_size=48, mlp_ratio=4.0, qkv_bias=True, qk_scale=None, attn_drop=0.0,
                 proj_drop=0.0, drop_path=0.0, norm_layer=LayerNorm, act_layer=nn.GELU, pre_norm=True, order_index=0,
                 cpe_indice_key=None, enable_rpe=False, enable_flash=True, upcast_attention=True, upcast_softmax=True):
        super().__init__()
        self.channels, self.pre_norm = channels, pre_norm
        self.cpe = PointSequential(SubMConv3d(channels, channels, kernel_size=3, bias=True, indice_key=cpe_indice_key),
                                   Linear(channels, channels), norm_layer(channels))
        self.norm1 = PointSequential(norm_layer(channels))
        self.attn = SerializedAttention(channels=channels, patch_size=patch_size, num_heads=num_heads, qkv_bias=qkv_bias,
                                        qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop, order_index=order_index,
                                        enable_rpe=enable_rpe, enable_flash=enable_flash, upcast_attention=upcast_attention,
                                        upcast_softmax=upcast_softmax)
        self.norm2 = PointSequential(norm_layer(channels))
        self.mlp = PointSequential(MLP(in_channels=channels, hidden_channels=int(channels * mlp_ratio), out_channels=channels,
                                       act_layer=act_layer, drop=proj_drop))
        self.drop_path = PointSequential(DropPath(drop_path) if drop_path > 0.0 else nn.Identity())

    def forward(self, point):
        shortcut = point.feat
        point = self.cpe(point)
        point.feat = shortcut + point.feat
        shortcut = point.feat
        if self.pre_norm:
            point = self.norm1(point)
        point = self.drop_path(self.attn(point))
        point.feat = shortcut + point.feat
        if not self.pre_norm:
            point = self.norm1(point)
        shortcut = point.feat
        if self.pre_norm:
            point = self.norm2(point)
        point = self.drop_path(self.mlp(point))
        point.feat = shortcut + point.feat
        if not self.pre_norm:
            point = self.norm2(point)
        return point


class SerializedPooling(PointModule):
    """:626-729: voxels that share the leading bits of their (order-0) code become one voxel of the next stage."""

    def __init__(self, in_channels, out_channels, stride=2, norm_layer=None, act_layer=None, reduce="max", shuffle_orders=True,
                 traceable=True):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        assert stride == 2 ** (math.ceil(stride) - 1).bit_length()
        self.stride = stride
        assert reduce in ["sum", "mean", "min", "max"]
        self.reduce, self.shuffle_orders, self.traceable = reduce, shuffle_orders, traceable
        self.proj = Linear(in_channels, out_channels)
        self.norm = PointSequential(norm_layer(out_channels)) if norm_layer is not None else None
        self.act = PointSequential(act_layer()) if act_layer is not None else None

    def forward(self, point):
        pooling_depth = (math.ceil(self.stride) - 1).bit_length()
        if pooling_depth > point.serialized_depth:
            pooling_depth = 0
        assert {"serialized_code", "serialized_order", "serialized_inverse", "serialized_depth"}.issubset(point.keys()), \
            "Run point.serialization() point cloud before SerializedPooling"
        code = point.serialized_code >> pooling_depth * 3
        code_, cluster, counts = torch.unique(code[0], sorted=True, return_inverse=True, return_counts=True)
        _, indices = torch.sort(cluster, stable=True)        # (the reference's sort leaves the order inside a cluster open)
        idx_ptr = torch.cat([counts.new_zeros(1), torch.cumsum(counts, dim=0)])
        head_indices = indices[idx_ptr[:-1]]
        code = code[:, head_indices]
        order = torch.argsort(code)
        inverse = torch.zeros_like(order).scatter_(
            dim=1, index=order, src=torch.arange(0, code.shape[1], device=order.device).repeat(code.shape[0], 1))
        if self.shuffle_orders:
            perm = torch.randperm(code.shape[0])
            code, order, inverse = code[perm], order[perm], inverse[perm]
        point_dict = dict(
            feat=torch.segment_reduce(self.proj(point.feat)[indices], self.reduce, lengths=counts, axis=0),   # segment_csr
            coord=torch.segment_reduce(point.coord[indices], "mean", lengths=counts, axis=0),
            grid_coord=point.grid_coord[head_indices] >> pooling_depth,
            serialized_code=code, serialized_order=order, serialized_inverse=inverse,
            serialized_depth=point.serialized_depth - pooling_depth,
            batch=point.batch[head_indices],
        )
        for k in ("condition", "context"):
            if k in point.keys():
                point_dict[k] = point[k]
        if self.traceable:
            point_dict["pooling_inverse"] = cluster
            point_dict["pooling_parent"] = point
        point = Point(point_dict)
        if self.norm is not None:
            point = self.norm(point)
        if self.act is not None:
            point = self.act(point)
        point.sparsify()
        return point


class SerializedUnpooling(PointModule):
    """:732-767"""

    def __init__(self, in_channels, skip_channels, out_channels, norm_layer=None, act_layer=None, traceable=False):
        super().__init__()
        self.proj = PointSequential(Linear(in_channels, out_channels))
        self.proj_skip = PointSequential(Linear(skip_channels, out_channels))
        if norm_layer is not None:
            self.proj.add(norm_layer(out_channels))
            self.proj_skip.add(norm_layer(out_channels))
        if act_layer is not None:
            self.proj.add(act_layer())
            self.proj_skip.add(act_layer())
        self.traceable = traceable

    def forward(self, point):
        assert "pooling_parent" in point.keys() and "pooling_inverse" in point.keys()
        parent = point.pop("pooling_parent")
        inverse = point.pop("pooling_inverse")
        point = self.proj(point)
        parent = self.proj_skip(parent)
        parent.feat = parent.feat + point.feat[inverse]
        if self.traceable:
            parent["unpooling_parent"] = point
        return parent


class Embedding(PointModule):
    """:770-800: the stem, a 5 x 5 x 5 submanifold conv on the raw features."""

    def __init__(self, in_channels, embed_channels, norm_layer=None, act_layer=None):
        super().__init__()
        self.in_channels, self.embed_channels = in_channels, embed_channels
        self.stem = PointSequential(conv=SubMConv3d(in_channels, embed_channels, kernel_size=5, padding=1, bias=False,
                                                    indice_key="stem"))
        if norm_layer is not None:
            self.stem.add(norm_layer(embed_channels), name="norm")
        if act_layer is not None:
            self.stem.add(act_layer(), name="act")

    def forward(self, point):
        return self.stem(point)
