"""Space-filling-curve codes of a voxelised cloud -- the interface of Modules/PointTransformerV3/serialization/default.py
(encode :8-25, decode :28-40, the four *_encode / *_decode helpers :43-58) on csrc/serialize.hip.  `serialize` is the code /
order / inverse triple of Point.serialization (PointTransformerV3/blocks.py:131-142) with all orders encoded by ONE launch.
No CPU fallback: a tensor that is not on the GPU raises."""
import ctypes

import torch

from ... import _hip

ORDERS = {"z": 0, "z-trans": 1, "hilbert": 2, "hilbert-trans": 3}   # PN2_ORDER_*


def _encode_many(grid_coord, batch, depth, orders):
    _hip.require_device(grid_coord)
    if grid_coord.dim() != 2 or grid_coord.shape[1] != 3:
        raise RuntimeError(f"grid_coord must be [N, 3], got {tuple(grid_coord.shape)}")
    if not 1 <= int(depth) <= 16:
        raise RuntimeError(f"depth {depth} outside 1..16 (blocks.py:126)")
    for o in orders:
        if o not in ORDERS:
            raise NotImplementedError(o)
    g = grid_coord if grid_coord.dtype == torch.int32 else grid_coord.to(torch.int32)   # only the low `depth` <= 16 bits count
    n = g.shape[0]
    out = torch.empty(len(orders), n, dtype=torch.int64, device=g.device)
    if n == 0:
        return out
    b = None
    if batch is not None:
        b = batch.to(device=g.device, dtype=torch.int64).contiguous()
        if b.numel() != n:
            raise RuntimeError("batch and grid_coord disagree on the number of points")
    codes = (ctypes.c_int32 * len(orders))(*[ORDERS[o] for o in orders])
    _hip.call("serialize_encode", _hip.lib().pn2_serialize_encode_i64, g.data_ptr(), g.stride(0), g.stride(1), _hip.ptr(b), n,
              int(depth), codes, len(orders), out.data_ptr(), _hip.stream_ptr(), nbytes=n * (12 + 8 * len(orders)))
    return out


@torch.inference_mode()
def encode(grid_coord, batch=None, depth=16, order="z"):
    assert order in {"z", "z-trans", "hilbert", "hilbert-trans"}
    return _encode_many(grid_coord, batch, depth, [order])[0]


@torch.inference_mode()
def decode(code, depth=16, order="z"):
    assert order in {"z", "hilbert"}
    _hip.require_device(code)
    code = code.to(torch.int64).contiguous()
    n = code.numel()
    grid = torch.empty(n, 3, dtype=torch.int64, device=code.device)
    batch = torch.empty(n, dtype=torch.int64, device=code.device)
    if n:
        _hip.call("serialize_decode", _hip.lib().pn2_serialize_decode_i64, code.data_ptr(), n, int(depth), ORDERS[order],
                  grid.data_ptr(), batch.data_ptr(), _hip.stream_ptr(), nbytes=40 * n)
    return grid, batch


def z_order_encode(grid_coord, depth=16):
    return encode(grid_coord, None, depth, "z")


def z_order_decode(code, depth):
    return decode(code, depth, "z")[0]


def hilbert_encode(grid_coord, depth=16):
    return encode(grid_coord, None, depth, "hilbert")


def hilbert_decode(code, depth=16):
    return decode(code, depth, "hilbert")[0]


@torch.inference_mode()
def serialize(grid_coord, batch, depth, order=("z",)):
    """-> (code [k, n], order [k, n], inverse [k, n]) like Point.serialization (blocks.py:131-142): all k codes from one
    launch; the sort is stable (the reference's argsort leaves the order of equal codes -- duplicate voxels -- open)."""
    if isinstance(order, str):
        order = [order]
    code = _encode_many(grid_coord, batch, depth, list(order))
    perm = torch.argsort(code, dim=1, stable=True)
    inverse = torch.empty_like(perm)
    inverse.scatter_(1, perm, torch.arange(code.shape[1], device=code.device).expand_as(perm))
    return code, perm, inverse
