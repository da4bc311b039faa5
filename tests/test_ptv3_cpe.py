"""PointTransformerV3 conditional positional encoding (third stage): the submanifold 3 x 3 x 3 sparse convolution
(Modules/PointTransformerV3/blocks.py:561-568).  PARITY UNPINNED against spconv (absent); the oracle restatement
(oracle/ptv3_cpe_port.py) is pinned to torch.nn.functional.conv3d on the densified grid (CPU test), the device path to the
oracle: neighbour tables bit-exact, outputs to fp32 rounding."""
import numpy as np
import pytest
import torch

import helpers


def _voxels(n, clouds, extent, seed, dup=0):
    rng = np.random.default_rng(seed)
    pts = []
    for b in range(clouds):
        # a thin curved sheet + a blob: surface-like occupancy with dense and isolated parts
        u, v = rng.uniform(0, extent, n // 2), rng.uniform(0, extent, n // 2)
        sheet = np.stack([u, v, 0.5 * extent + 0.2 * extent * np.sin(u / extent * 6.0) * np.cos(v / extent * 4.0)], 1)
        blob = rng.normal(0.5 * extent, 0.08 * extent, (n - n // 2, 3))
        g = np.clip(np.concatenate([sheet, blob]), 0, extent - 1e-3).astype(np.int64)
        g = np.unique(g, axis=0)
        pts.append(np.concatenate([np.full((len(g), 1), b), g], 1))
    vox = np.concatenate(pts)
    vox = vox[rng.permutation(len(vox))]
    if dup:
        vox = np.concatenate([vox, vox[:dup]])
    return vox[:, 0].copy(), vox[:, 1:].astype(np.int32).copy()


def test_oracle_is_the_dense_conv3d_at_the_active_voxels():
    """The restatement against its definition: conv3d(padding = 1) of the densified grid, read at the active voxels."""
    from oracle import ptv3_cpe_port as P
    batch, grid = _voxels(600, 2, 12, seed=1)
    rng = np.random.default_rng(2)
    feat = rng.standard_normal((len(grid), 16))
    w = rng.standard_normal((32, 3, 3, 3, 16)) * 0.1
    bias = rng.standard_normal(32)
    nbr = P.subm_neighbors(batch, grid)
    assert (nbr[:, 13] == np.arange(len(grid))).all()                 # the centre offset is the voxel itself
    assert (nbr >= 0).sum(1).min() >= 1 and (nbr >= 0).sum(1).max() > 9
    got = P.subm_conv(feat, nbr, w, bias)
    want = P.dense_reference(batch, grid, feat, w, bias)
    assert float((got - want).abs().max()) <= 1e-12 * float(want.abs().max())
    # the stem's 5 x 5 x 5 kernel on a few raw features (Embedding, blocks.py:783-791)
    w5 = rng.standard_normal((32, 5, 5, 5, 6)) * 0.1
    f5 = rng.standard_normal((len(grid), 6))
    nbr5 = P.subm_neighbors(batch, grid, 5)
    assert nbr5.shape[1] == 125 and (nbr5[:, 62] == np.arange(len(grid))).all()
    got5, want5 = P.subm_conv(f5, nbr5, w5, None), P.dense_reference(batch, grid, f5, w5, None)
    assert float((got5 - want5).abs().max()) <= 1e-12 * float(want5.abs().max())
    # a neighbour of another cloud at the same coordinates is not a neighbour
    nb2 = P.subm_neighbors(np.array([0, 1]), np.array([[3, 3, 3], [3, 3, 4]]))
    assert (nb2[0] >= 0).sum() == 1 and (nb2[1] >= 0).sum() == 1


@pytest.mark.gpu
@pytest.mark.parametrize("n,clouds,extent,cin,cout,dup", [
    (3000, 2, 40, 32, 32, 0),        # the first stage's width; rows not a multiple of the 128-row tile
    (5000, 1, 64, 64, 64, 0),
    (2000, 3, 30, 48, 96, 0),        # C_in != C_out, 32-column tiles
    (4000, 2, 50, 128, 128, 7),      # 128-column tiles; seven duplicate voxels (lowest index represents the cell)
    (130, 1, 400, 16, 256, 0),       # isolated voxels: almost every offset slab skipped; two column tiles
])
@pytest.mark.parametrize("kernel", ["default", "compact", "dense"])
def test_subm_conv_matches_oracle(n, clouds, extent, cin, cout, dup, kernel, monkeypatch):
    """Both kernels at every shape: whole 128-voxel tiles per offset (default below 128 input channels) and the rows compacted per
    offset (default from 128 on)."""
    if kernel == "compact":
        monkeypatch.setenv("PN2_CPE_COMPACT", "1")
    elif kernel == "dense":
        monkeypatch.setenv("PN2_CPE_DENSE_TILES", "1")
    helpers.load_pkg()
    from oracle import ptv3_cpe_port as P
    from pn2_amd.PointTransformerV3 import cpe
    batch, grid = _voxels(n, clouds, extent, seed=n + cin, dup=dup)
    N = len(grid)
    rng = np.random.default_rng(5)
    feat = rng.standard_normal((N, cin)).astype(np.float32)
    want_nbr = P.subm_neighbors(batch, grid)
    nbr = cpe.subm_neighbors(torch.from_numpy(batch).cuda(), torch.from_numpy(grid).cuda())
    assert np.array_equal(nbr.cpu().numpy(), want_nbr)
    torch.manual_seed(3)
    conv = cpe.SubMConv3d(cin, cout).cuda()
    with torch.no_grad():
        conv.bias.uniform_(-0.5, 0.5)
        out = conv(torch.from_numpy(feat).cuda(), nbr)
    want = P.subm_conv(feat, want_nbr, conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy())
    err = float((out.cpu().double() - want).abs().max()) / float(want.abs().max())
    assert err <= 2e-6, err
    # the cached offset-major weight follows an in-place update of the parameter
    with torch.no_grad():
        conv.weight.mul_(0.5)
        out2 = conv(torch.from_numpy(feat).cuda(), nbr)
    want2 = P.subm_conv(feat, want_nbr, conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy())
    assert float((out2.cpu().double() - want2).abs().max()) <= 2e-6 * float(want2.abs().max())


@pytest.mark.gpu
def test_subm_neighbors_rejects_out_of_range_voxels_loudly():
    helpers.load_pkg()
    from pn2_amd import ops
    from pn2_amd.PointTransformerV3 import cpe
    grid = torch.tensor([[1, 1, 1], [1, 1, 2], [70000, 0, 0]], dtype=torch.int32).cuda()
    nbr = cpe.subm_neighbors(None, grid)
    assert (nbr[2] == -1).all() and int(nbr[0, 14]) == 1 and int(nbr[1, 12]) == 0
    with pytest.raises(RuntimeError):
        ops.check_status()


@pytest.mark.gpu
def test_stem_conv_5x5x5_on_raw_features():
    """Embedding's stem (blocks.py:783-791): SubMConv3d(in_channels = 6, 32, kernel_size = 5, bias = False) -- 125 offsets, the
    input width zero-padded to 16."""
    helpers.load_pkg()
    from oracle import ptv3_cpe_port as P
    from pn2_amd.PointTransformerV3 import cpe
    batch, grid = _voxels(2500, 2, 30, seed=77)
    N = len(grid)
    feat = np.random.default_rng(6).standard_normal((N, 6)).astype(np.float32)
    want_nbr = P.subm_neighbors(batch, grid, 5)
    nbr = cpe.subm_neighbors(torch.from_numpy(batch).cuda(), torch.from_numpy(grid).cuda(), kernel_size=5)
    assert np.array_equal(nbr.cpu().numpy(), want_nbr)
    torch.manual_seed(4)
    conv = cpe.SubMConv3d(6, 32, kernel_size=5, padding=1, bias=False).cuda()
    with torch.no_grad():
        out = conv(torch.from_numpy(feat).cuda(), nbr)
    want = P.subm_conv(feat, want_nbr, conv.weight.detach().cpu().numpy(), None)
    assert float((out.cpu().double() - want).abs().max()) <= 2e-6 * float(want.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("n,extent,c", [(4000, 30, 64), (3000, 24, 128), (2500, 20, 256)])
def test_subm_conv_bf16_mode(n, extent, c):
    """bf16 mode (operands rounded to bfloat16, fp32 accumulation): within 1.5e-2 of the largest value of the float64 restatement
    (measured 4-6e-3: ~250 products of bfloat16-rounded factors per output), and exactly the fp32 kernel's result on inputs that
    ARE bfloat16 numbers."""
    helpers.load_pkg()
    from oracle import ptv3_cpe_port as P
    from pn2_amd.PointTransformerV3 import cpe
    batch, grid = _voxels(n, 2, extent, seed=c)
    N = len(grid)
    feat = np.random.default_rng(9).standard_normal((N, c)).astype(np.float32)
    nbr = cpe.subm_neighbors(torch.from_numpy(batch).cuda(), torch.from_numpy(grid).cuda())
    torch.manual_seed(8)
    conv = cpe.SubMConv3d(c, c).cuda()
    want = P.subm_conv(feat, nbr.cpu().numpy(), conv.weight.detach().cpu().numpy(), conv.bias.detach().cpu().numpy())
    old = cpe.CONV_PRECISION
    try:
        cpe.CONV_PRECISION = "bf16"
        with torch.no_grad():
            out = conv(torch.from_numpy(feat).cuda(), nbr)
            err = float((out.cpu().double() - want).abs().max()) / float(want.abs().max())
            assert err <= 1.5e-2, err
            # bfloat16-representable inputs and weights: both modes multiply the same numbers
            conv.weight.copy_(conv.weight.to(torch.bfloat16).float())
            fb = torch.from_numpy(feat).cuda().to(torch.bfloat16).float()
            out16 = conv(fb, nbr)
            cpe.CONV_PRECISION = "f32"
            out32 = conv(fb, nbr)
        assert float((out16 - out32).abs().max()) <= 2e-5 * float(out32.abs().max())
    finally:
        cpe.CONV_PRECISION = old
