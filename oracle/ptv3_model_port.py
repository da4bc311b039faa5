"""TEST INFRASTRUCTURE (oracle): CPU float64 restatement of the PointTransformerV3 backbone forward at inference
(/root/reference/Modules/PointTransformerV3/PointTransformerV3.py:261-460 with blocks.py: Embedding :770-800, Block :536-623,
SerializedPooling :626-729, SerializedUnpooling :732-767, MLP :510-533), driven by a state dict with the reference's parameter
names.  It is built from the other restatements -- codes (serialization_port), attention (ptv3_attention_port), submanifold
convolutions (ptv3_cpe_port) -- and plain torch for the dense layers.
PARITY UNPINNED: the reference module cannot be imported here (spconv, torch_scatter, addict, timm at module level) and it holds
no fixture of this model.  Only tests/ and tools' cpu-baseline legs may import this module."""
import math

import numpy as np
import torch

from . import ptv3_attention_port as A
from . import ptv3_cpe_port as C
from . import serialization_port as S

F64 = torch.float64


def _t(sd, k):
    """Every parameter / buffer read goes through here (tests/test_ptv3_train.py swaps it for float64 leaves to differentiate)."""
    return sd[k].detach().cpu().to(F64)


def _linear(sd, p, x):
    return x @ _t(sd, p + ".weight").t() + _t(sd, p + ".bias")


def _layernorm(sd, p, x, eps=1e-5):
    mu = x.mean(-1, keepdim=True)
    var = ((x - mu) ** 2).mean(-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * _t(sd, p + ".weight") + _t(sd, p + ".bias")


def _batchnorm_eval(sd, p, x, eps=1e-3):
    return (x - _t(sd, p + ".running_mean")) / torch.sqrt(_t(sd, p + ".running_var") + eps) * _t(sd, p + ".weight") + _t(sd, p + ".bias")


def _gelu(x):
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def _serialize(grid, batch, depth, orders):
    code = np.stack([S.encode(grid, batch, depth, order=o) for o in orders])
    order = np.argsort(code, axis=1, kind="stable")
    inverse = np.empty_like(order)
    for k in range(len(orders)):
        inverse[k, order[k]] = np.arange(code.shape[1])
    return code, order, inverse


def _block(sd, p, pt, num_heads, patch_size, order_index, parts=None):
    """Block.forward :598-623 (pre-norm)."""
    feat = pt["feat"]
    nbr = pt["nbr"].setdefault(3, C.subm_neighbors(pt["batch"], pt["grid"], 3))
    x = C.subm_conv(feat, nbr, _t(sd, p + ".cpe.0.weight"), _t(sd, p + ".cpe.0.bias"))
    x = _layernorm(sd, p + ".cpe.2", _linear(sd, p + ".cpe.1", x))
    if parts is not None:
        parts["cpe"] = x.clone()
    feat = feat + x
    y = _layernorm(sd, p + ".norm1.0", feat)
    y = A.serialized_attention(y, pt["offset"], pt["order"][order_index], pt["inverse"][order_index],
                               _t(sd, p + ".attn.qkv.weight"), _t(sd, p + ".attn.qkv.bias"),
                               _t(sd, p + ".attn.proj.weight"), _t(sd, p + ".attn.proj.bias"), num_heads, patch_size)
    if parts is not None:
        parts["attn"] = y.clone()
    feat = feat + y
    y = _layernorm(sd, p + ".norm2.0", feat)
    y = _linear(sd, p + ".mlp.0.fc2", _gelu(_linear(sd, p + ".mlp.0.fc1", y)))
    pt["feat"] = feat + y
    return pt


def _pool(sd, p, pt, stride, reduce="max"):
    """SerializedPooling.forward :658-729 (reduce = "max" is the constructor's default, :640; shuffle_orders off)."""
    depth = (math.ceil(stride) - 1).bit_length()
    if depth > pt["depth"]:
        depth = 0
    code = pt["code"] >> (depth * 3)
    _, cluster, counts = np.unique(code[0], return_inverse=True, return_counts=True)
    indices = np.argsort(cluster, kind="stable")
    idx_ptr = np.concatenate([[0], np.cumsum(counts)])
    head = indices[idx_ptr[:-1]]
    code = code[:, head]
    order = np.argsort(code, axis=1, kind="stable")
    inverse = np.empty_like(order)
    for k in range(code.shape[0]):
        inverse[k, order[k]] = np.arange(code.shape[1])
    proj = _linear(sd, p + ".proj", pt["feat"])[torch.from_numpy(indices)]
    seg = [proj[idx_ptr[i]:idx_ptr[i + 1]] for i in range(len(counts))]
    feat = torch.stack([r.max(0).values if reduce == "max" else r.mean(0) for r in seg])
    coord = np.stack([pt["coord"][indices[idx_ptr[i]:idx_ptr[i + 1]]].mean(0) for i in range(len(counts))])
    batch = pt["batch"][head]
    new = {"feat": _gelu(_batchnorm_eval(sd, p + ".norm.0", feat)), "coord": coord, "grid": pt["grid"][head] >> depth, "batch": batch,
           "offset": np.cumsum(np.bincount(batch)), "code": code, "order": order, "inverse": inverse, "depth": pt["depth"] - depth,
           "nbr": {}, "pooling_inverse": cluster, "pooling_parent": pt}
    return new


def _unpool(sd, p, pt):
    """SerializedUnpooling.forward :757-767."""
    parent, inverse = pt.pop("pooling_parent"), pt.pop("pooling_inverse")
    x = _gelu(_batchnorm_eval(sd, p + ".proj.1", _linear(sd, p + ".proj.0", pt["feat"])))
    skip = _gelu(_batchnorm_eval(sd, p + ".proj_skip.1", _linear(sd, p + ".proj_skip.0", parent["feat"])))
    parent["feat"] = skip + x[torch.from_numpy(inverse)]
    return parent


def backbone_forward(sd, cfg, feat, coord, grid_coord, batch, trace=None):
    """sd: state dict of PointTransformerV3 (reference names); cfg: dict with order, stride, enc_depths, enc_num_head,
    enc_patch_size, dec_depths, dec_num_head, dec_patch_size; -> the finest stage's features [N, dec_channels[0]] (float64)."""
    grid = np.asarray(grid_coord, dtype=np.int64)
    batch = np.asarray(batch, dtype=np.int64)
    depth = int(grid.max()).bit_length()
    code, order, inverse = _serialize(grid, batch, depth, cfg["order"])
    pt = {"feat": torch.as_tensor(np.asarray(feat), dtype=F64), "coord": np.asarray(coord, dtype=np.float64), "grid": grid, "batch": batch,
          "offset": np.cumsum(np.bincount(batch)), "code": code, "order": order, "inverse": inverse, "depth": depth, "nbr": {}}
    # Embedding :770-800
    nbr5 = C.subm_neighbors(batch, grid, 5)
    x = C.subm_conv(pt["feat"], nbr5, _t(sd, "embedding.stem.conv.weight"), None)
    pt["feat"] = _gelu(_batchnorm_eval(sd, "embedding.stem.norm", x))
    n_orders = len(cfg["order"])
    note = (lambda name: trace.append((name, pt["feat"].clone()))) if trace is not None else (lambda name: None)
    note("embedding")
    for s in range(len(cfg["enc_depths"])):
        if s > 0:
            pt = _pool(sd, f"enc.enc{s}.down", pt, cfg["stride"][s - 1], cfg.get("pool_reduce", "max"))
            note(f"enc.enc{s}.down")
        for i in range(cfg["enc_depths"][s]):
            parts = {} if trace is not None else None
            pt = _block(sd, f"enc.enc{s}.block{i}", pt, cfg["enc_num_head"][s], cfg["enc_patch_size"][s], i % n_orders, parts)
            if trace is not None:
                trace.append((f"enc.enc{s}.block{i}.cpe", parts["cpe"]))
                trace.append((f"enc.enc{s}.block{i}.attn", parts["attn"]))
            note(f"enc.enc{s}.block{i}")
    for s in reversed(range(len(cfg["enc_depths"]) - 1)):
        pt = _unpool(sd, f"dec.dec{s}.up", pt)
        note(f"dec.dec{s}.up")
        for i in range(cfg["dec_depths"][s]):
            pt = _block(sd, f"dec.dec{s}.block{i}", pt, cfg["dec_num_head"][s], cfg["dec_patch_size"][s], i % n_orders)
            note(f"dec.dec{s}.block{i}")
    return pt["feat"]
