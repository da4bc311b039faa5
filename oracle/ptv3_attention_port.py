"""TEST INFRASTRUCTURE (oracle): CPU restatement of PointTransformerV3's serialized patch attention, following the source text
of /root/reference/Modules/PointTransformerV3/blocks.py -- get_padding_and_inverse :384-437 (numpy, the reference's loops) and
the non-flash branch of SerializedAttention.forward :457-488 (torch CPU).
PARITY UNPINNED: blocks.py imports spconv, torch_scatter, addict and timm at module level, none of which exists here, so no
fixture can be generated from the reference itself; these functions are checked by hand-derived cases (tests/test_ptv3_attention.py).
Only tests/ and tools' cpu-baseline legs may import this module."""
import numpy as np
import torch


def get_padding_and_inverse(offset, patch_size):
    """offset: int64 array of cumulative cloud sizes -> (pad, unpad, cu_seqlens), blocks.py:392-436 line by line."""
    offset = np.asarray(offset, dtype=np.int64)
    bincount = np.diff(offset, prepend=0)                                               # offset2bincount :22-25
    bincount_pad = (bincount + patch_size - 1) // patch_size * patch_size               # :394-401
    mask_pad = bincount > patch_size                                                    # :403
    bincount_pad = (~mask_pad) * bincount + mask_pad * bincount_pad                     # :404
    _offset = np.concatenate([[0], offset])                                             # :405
    _offset_pad = np.concatenate([[0], np.cumsum(bincount_pad)])                        # :406
    pad = np.arange(_offset_pad[-1])                                                    # :407
    unpad = np.arange(_offset[-1])                                                      # :408
    cu_seqlens = []
    for i in range(len(offset)):                                                        # :410
        unpad[_offset[i]:_offset[i + 1]] += _offset_pad[i] - _offset[i]                 # :411
        if bincount[i] != bincount_pad[i]:                                              # :412
            a = _offset_pad[i + 1] - patch_size + (bincount[i] % patch_size)
            b = _offset_pad[i + 1] - 2 * patch_size + (bincount[i] % patch_size)
            pad[a:_offset_pad[i + 1]] = pad[b:_offset_pad[i + 1] - patch_size]          # :413-423
        pad[_offset_pad[i]:_offset_pad[i + 1]] -= _offset_pad[i] - _offset[i]           # :424
        cu_seqlens.append(np.arange(_offset_pad[i], _offset_pad[i + 1], patch_size, dtype=np.int32))   # :425-433
    cu = np.concatenate(cu_seqlens + [np.array([_offset_pad[-1]], dtype=np.int32)])     # :435-437
    return pad.astype(np.int64), unpad.astype(np.int64), cu.astype(np.int32)


def patch_attention(qkv, order, K, H, scale, dtype=torch.float32):
    """qkv [N, 3C] -> feat [N', C], blocks.py:463-484 (non-flash, no RPE, no upcast, dropout 0)."""
    qkv = torch.as_tensor(qkv, dtype=dtype)
    C = qkv.shape[1] // 3
    if order is not None:
        qkv = qkv[torch.as_tensor(order, dtype=torch.long)]                             # :463
    q, k, v = qkv.reshape(-1, K, 3, H, C // H).permute(2, 0, 3, 1, 4).unbind(dim=0)     # :467-469
    attn = (q * scale) @ k.transpose(-2, -1)                                            # :474
    attn = torch.softmax(attn, dim=-1)                                                  # :479
    return (attn @ v).transpose(1, 2).reshape(-1, C)                                    # :481


def serialized_attention(feat, offset, order, inverse, wqkv, bqkv, wproj, bproj, H, patch_size_max, dtype=torch.float64):
    """SerializedAttention.forward :448-505 for one serialization order; returns the new feat."""
    feat = torch.as_tensor(feat, dtype=dtype)
    counts = np.diff(np.asarray(offset, dtype=np.int64), prepend=0)
    K = int(min(counts.min(), patch_size_max))                                          # :451-454
    pad, unpad, _ = get_padding_and_inverse(offset, K)
    C = feat.shape[1]
    o = torch.as_tensor(order, dtype=torch.long)[torch.from_numpy(pad)]                 # :460
    inv = torch.from_numpy(unpad)[torch.as_tensor(inverse, dtype=torch.long)]           # :461
    qkv = feat @ torch.as_tensor(wqkv, dtype=dtype).t() + torch.as_tensor(bqkv, dtype=dtype)
    out = patch_attention(qkv, o, K, H, (C // H) ** -0.5, dtype=dtype)[inv]             # :463-498
    return out @ torch.as_tensor(wproj, dtype=dtype).t() + torch.as_tensor(bproj, dtype=dtype)
