"""PointTransformerV3 serialized patch attention (second stage; reference Modules/PointTransformerV3/blocks.py:336-507).

PARITY UNPINNED: the reference module imports spconv / torch_scatter / addict / timm at module level (none exists here), so no
fixture can come from the reference itself.  CPU: the oracle's index maps against hand-derived cases and their defining
properties.  GPU: the index-map kernel bit for bit against the oracle (integer work); the attention kernel against the torch
restatement of the non-flash branch -- fp32 mode within 2e-5 of the largest output (exact fp32 products; exp and the order of
the sums differ from torch's), bf16 mode within its own measured tolerance -- and the whole module at BASELINE configs[3]'s
shape through the pinned serialization orders."""
import numpy as np
import pytest
import torch

import helpers
from oracle import ptv3_attention_port as P


# ------------------------------------------------------------------------------------------------------------- CPU
def test_oracle_index_maps_hand_case():
    # two clouds of 5 and 3 points, patch 2: cloud 0 is padded to 6 (its last patch repeats point 3 behind point 4), cloud 1 to 4
    pad, unpad, cu = P.get_padding_and_inverse([5, 8], 2)
    assert pad.tolist() == [0, 1, 2, 3, 4, 3, 5, 6, 7, 6]
    assert unpad.tolist() == [0, 1, 2, 3, 4, 6, 7, 8]
    assert cu.tolist() == [0, 2, 4, 6, 8, 10]
    # a cloud not longer than the patch is left alone (:403-404)
    pad, unpad, cu = P.get_padding_and_inverse([3, 10], 4)
    assert pad.tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 6] and unpad.tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 9]
    assert cu.tolist() == [0, 3, 7, 11]


@pytest.mark.parametrize("seed", range(5))
def test_oracle_index_map_properties(seed):
    rng = np.random.default_rng(seed)
    K = int(rng.integers(2, 40))
    counts = rng.integers(1, 200, size=int(rng.integers(1, 6)))
    off = np.cumsum(counts)
    pad, unpad, cu = P.get_padding_and_inverse(off, K)
    assert np.array_equal(pad[unpad], np.arange(off[-1]))                 # every point sits at its unpadded place
    starts = np.concatenate([[0], off[:-1]])
    for s, e, c in zip(starts, off, counts):                               # a cloud's padding repeats that cloud's own points
        seg = pad[unpad[s]:unpad[e - 1] + 1]
        assert seg.min() >= s and seg.max() < e
    assert cu[-1] == len(pad) and np.all(np.diff(cu) > 0)


def test_oracle_attention_rows_are_convex_combinations():
    torch.manual_seed(0)
    qkv = torch.randn(64, 3 * 32)
    out = P.patch_attention(qkv, None, 16, 2, 0.25)
    v = qkv[:, 64:].reshape(4, 16, 2, 16)
    o = out.reshape(4, 16, 2, 16)
    assert torch.all(o <= v.max(dim=1, keepdim=True)[0] + 1e-6) and torch.all(o >= v.min(dim=1, keepdim=True)[0] - 1e-6)


# ------------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("counts,K", [([5, 3], 2), ([3, 7], 4), ([1000, 2500, 1024], 1024), ([4096], 1024), ([17, 900, 33], 16),
                                      ([70000, 123457], 1000)])
def test_hip_index_maps_equal_the_oracle(counts, K):
    helpers.load_pkg()
    from pn2_amd.PointTransformerV3 import attention as A
    off = np.cumsum(counts)
    pad, unpad, cu = A.get_padding_and_inverse(torch.from_numpy(off).cuda(), K)
    opad, ounpad, ocu = P.get_padding_and_inverse(off, K)
    assert pad.dtype == unpad.dtype == torch.int64 and cu.dtype == torch.int32
    assert np.array_equal(pad.cpu().numpy(), opad) and np.array_equal(unpad.cpu().numpy(), ounpad)
    assert np.array_equal(cu.cpu().numpy(), ocu)


@pytest.mark.gpu
@pytest.mark.parametrize("n_patch,K,H", [(3, 16, 2), (2, 100, 4), (2, 1000, 2), (4, 1024, 2), (1, 1024, 32), (5, 257, 8)])
@pytest.mark.parametrize("ordered", [False, True])
def test_hip_patch_attention_vs_torch_restatement(n_patch, K, H, ordered):
    helpers.load_pkg()
    from pn2_amd.PointTransformerV3 import attention as A
    torch.manual_seed(n_patch * K + H)
    C = 16 * H
    n_rows = n_patch * K
    n = n_rows if not ordered else n_rows - K // 3
    qkv = torch.randn(n, 3 * C) * 1.3
    order = torch.randint(0, n, (n_rows,)) if ordered else None
    scale = 16 ** -0.5
    want = P.patch_attention(qkv.double(), order, K, H, scale, dtype=torch.float64)
    got = A.patch_attention(qkv.cuda(), None if order is None else order.cuda(), K, H, scale)
    err = float((got.cpu().double() - want).abs().max()) / float(want.abs().max())
    assert err <= 2e-5, f"fp32 mode: {err:.2e} of the largest output"
    ref32 = P.patch_attention(qkv, order, K, H, scale)                    # torch's own fp32 evaluation for scale
    print(f"K={K} H={H}: hip {err:.2e}, torch fp32 {float((ref32.double() - want).abs().max()) / float(want.abs().max()):.2e}")
    A.ATTENTION_PRECISION = "bf16"
    try:
        got16 = A.patch_attention(qkv.cuda(), None if order is None else order.cuda(), K, H, scale)
    finally:
        A.ATTENTION_PRECISION = "f32"
    err16 = float((got16.cpu().double() - want).abs().max()) / float(want.abs().max())
    assert err16 <= 2e-2, f"bf16 mode: {err16:.2e}"                       # measured 4e-3 .. 9e-3 (bfloat16 q, k, p, v)


@pytest.mark.gpu
def test_serialized_attention_module_at_config3_shape():
    """The module on a 1 048 576-voxel plot (BASELINE configs[3]) in two clouds, through the z-order serialization of the first
    stage (pinned, tests/test_serialization.py): spot-checked rows against the float64 restatement of the whole forward."""
    helpers.load_pkg()
    from pn2_amd.PointTransformerV3 import attention as A
    from pn2_amd.PointTransformerV3.serialization import serialize
    torch.manual_seed(3)
    counts = [600_000, 448_576]
    n, C, H = sum(counts), 32, 2
    grid = torch.randint(0, 1 << 10, (n, 3), dtype=torch.int64, device="cuda")
    batch = torch.repeat_interleave(torch.arange(2, device="cuda"), torch.tensor(counts, device="cuda"))
    code, order, inverse = serialize(grid, batch, depth=10, order=["z"])
    feat = torch.randn(n, C, device="cuda")
    mod = A.SerializedAttention(C, H, 1024).cuda().eval()
    point = {"feat": feat, "offset": torch.tensor(np.cumsum(counts), device="cuda"), "serialized_order": order, "serialized_inverse": inverse}
    with torch.no_grad():
        out = mod(point)["feat"]
    assert out.shape == (n, C) and torch.isfinite(out).all()
    assert point["pad"].numel() == sum((c + 1023) // 1024 * 1024 for c in counts)
    # float64 restatement on the patches that hold 64 sampled points
    pad, unpad = point["pad"].cpu(), point["unpad"].cpu()
    ord0, inv0 = order[0].cpu(), inverse[0].cpu()
    sample = torch.randint(0, n, (64,))
    w = {k: v.detach().cpu().double() for k, v in mod.state_dict().items()}
    f64 = feat.cpu().double()
    for i in sample.tolist():
        p = int(unpad[inv0[i]])                       # padded position of point i
        rows = ord0[pad[p // 1024 * 1024:(p // 1024 + 1) * 1024]]
        qkv = f64[rows] @ w["qkv.weight"].t() + w["qkv.bias"]
        o = P.patch_attention(qkv, None, 1024, H, (C // H) ** -0.5, dtype=torch.float64)[p % 1024]
        want = o @ w["proj.weight"].t() + w["proj.bias"]
        assert float((out[i].cpu().double() - want).abs().max()) <= 1e-4 * float(want.abs().max()) + 1e-6


@pytest.mark.gpu
def test_attention_refuses_what_is_not_built():
    helpers.load_pkg()
    from pn2_amd.PointTransformerV3 import attention as A
    with pytest.raises(NotImplementedError):
        A.SerializedAttention(32, 2, 1024, enable_flash=True)
    with pytest.raises(NotImplementedError):
        A.SerializedAttention(32, 2, 1024, enable_rpe=True)
    with pytest.raises(RuntimeError):
        A.patch_attention(torch.randn(32, 3 * 64, device="cuda"), None, 16, 2, 0.25)      # head width 32: not the built one
