"""Training side of the PointTransformerV3 mirror: backward kernels of the serialized patch attention
(csrc/ptv3_attention.hip: dq; dk + dv) and of the submanifold convolutions (csrc/ptv3_cpe.hip: gathered split-K weight gradient;
the input gradient is the forward kernel on the mirrored stencil), checked against torch autograd in float64 on the oracle's
restatements (oracle/ptv3_attention_port.py, ptv3_cpe_port.py, ptv3_model_port.py).  PARITY UNPINNED against the reference
(spconv / torch_scatter / flash_attn are absent here), like the forward tests."""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max()) / max(float(b.double().abs().max()), 1e-30)


@pytest.mark.parametrize("patches,K,H,gather", [(2, 1024, 2, False), (5, 48, 4, True), (3, 100, 1, True), (1, 16, 3, False),
                                                  (2, 1000, 2, True)])
def test_attention_backward_vs_float64_autograd(patches, K, H, gather):
    helpers.load_pkg()
    from oracle import ptv3_attention_port as A
    from pn2_amd.PointTransformerV3.attention import patch_attention
    C, n_rows = 16 * H, patches * K
    g = torch.Generator().manual_seed(patches * 1000 + K)
    n_src = n_rows - 7 if gather else n_rows
    qkv = torch.randn(n_src, 3 * C, generator=g) * 1.5
    order = None
    if gather:   # a permutation whose tail repeats rows (the padded tail of a cloud's last patch reads rows twice)
        order = torch.cat([torch.randperm(n_src, generator=g), torch.randint(0, n_src, (7,), generator=g)])
    gout = torch.randn(n_rows, C, generator=g)
    scale = 16 ** -0.5
    q64 = qkv.double().requires_grad_(True)
    want = A.patch_attention(q64, order, K, H, scale, dtype=torch.float64)
    want.backward(gout.double())
    x = qkv.cuda().requires_grad_(True)
    got = patch_attention(x, None if order is None else order.cuda(), K, H, scale)
    assert _rel(got.detach(), want.detach()) <= 2e-5
    got.backward(gout.cuda())
    assert _rel(x.grad, q64.grad) <= 5e-5, _rel(x.grad, q64.grad)


def _voxels(n_pts, seed, cell=0.05):
    from pn2_amd.synthetic import gaussian_branch_tree
    xyz = gaussian_branch_tree(n_pts, seed=seed)[0]
    return np.unique(np.floor((xyz - xyz.min(0)) / cell).astype(np.int64), axis=0)


@pytest.mark.parametrize("cin,cout,k,n_pts", [(32, 32, 3, 9000), (64, 64, 3, 5000), (128, 128, 3, 3000), (4, 32, 5, 3000),
                                               (32, 64, 3, 150)])
def test_subm_conv_backward_vs_float64_autograd(cin, cout, k, n_pts):
    helpers.load_pkg()
    from oracle import ptv3_cpe_port as Cp
    from pn2_amd.PointTransformerV3.cpe import SubMConv3d, subm_neighbors
    grid = _voxels(n_pts, seed=cin + k)
    N = len(grid)
    torch.manual_seed(cin * 7 + cout)
    conv = SubMConv3d(cin, cout, kernel_size=k, bias=(k == 3)).cuda()
    feat = torch.randn(N, cin)
    gout = torch.randn(N, cout)
    nbr = subm_neighbors(None, torch.from_numpy(grid).cuda(), k)
    want_nbr = Cp.subm_neighbors(np.zeros(N, np.int64), grid, k)
    assert np.array_equal(nbr.cpu().numpy(), want_nbr)
    f64 = feat.double().requires_grad_(True)
    w64 = conv.weight.detach().cpu().double().requires_grad_(True)
    b64 = None if conv.bias is None else conv.bias.detach().cpu().double().requires_grad_(True)
    want = Cp.subm_conv(f64, want_nbr, w64, b64)
    want.backward(gout.double())
    needs_dx = cin % 32 == 0                      # (the stem's raw features take no gradient)
    x = feat.cuda().requires_grad_(needs_dx)
    got = conv(x, nbr)
    assert _rel(got.detach(), want.detach()) <= 2e-5
    got.backward(gout.cuda())
    assert _rel(conv.weight.grad, w64.grad) <= 2e-5
    if b64 is not None:
        assert _rel(conv.bias.grad, b64.grad) <= 2e-5
    if needs_dx:
        assert _rel(x.grad, f64.grad) <= 2e-5


def _small_cfg():
    return dict(in_channels=4, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2, 2, 2), enc_depths=(1,) * 5,
                enc_channels=(32, 64, 128, 256, 512), enc_num_head=(2, 4, 8, 16, 32), enc_patch_size=(64,) * 5, dec_depths=(1,) * 4,
                dec_channels=(64, 64, 128, 256), dec_num_head=(4, 4, 8, 16), dec_patch_size=(64,) * 4, shuffle_orders=False)


def test_backbone_parameter_gradients_vs_float64_restatement(monkeypatch):
    """d(sum(out * R)) / d(every parameter) of the backbone (running-statistics BatchNorm, no stochastic depth: the restatement's
    setting) against torch autograd through oracle/ptv3_model_port.py in float64.
    The poolings reduce with "mean" here instead of the constructor's default "max": a max over a cluster is not differentiable
    where two rows tie, and with ~360 000 (cluster, channel) maxima per stage one of them is closer than fp32 rounding in about
    every second pass -- the forward's last-bit noise (LDS float atomics in the wide convolutions) then moves that gradient
    element to another row, 7e-4 of the weight gradients' norm (tools/debug_ptv3_grad.py: the forward agrees to 7e-7, the
    gradients are bimodal from pass to pass).  The max reduction is torch's segment_reduce, not a kernel of this library."""
    helpers.load_pkg()
    from oracle import ptv3_model_port as P
    from pn2_amd.PointTransformerV3.PointTransformerV3 import PointTransformerV3
    cfg = _small_cfg()
    torch.manual_seed(0)
    model = PointTransformerV3(**cfg).cuda().eval()
    from pn2_amd.PointTransformerV3.blocks import SerializedPooling
    for m in model.modules():
        if hasattr(m, "shuffle_orders"):
            m.shuffle_orders = False
        if isinstance(m, SerializedPooling):
            m.reduce = "mean"
    cfg = dict(cfg, pool_reduce="mean")
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():
        for m in model.modules():
            if isinstance(m, torch.nn.BatchNorm1d):
                m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
                m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.LayerNorm)):
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
    clouds = []
    for b in range(2):
        grid = _voxels(5000, seed=20 + b)
        clouds.append(np.concatenate([np.full((len(grid), 1), b), grid], 1))
    vox = np.concatenate(clouds)
    batch, grid = vox[:, 0].copy(), vox[:, 1:].copy()
    N = len(grid)
    rng = np.random.default_rng(2)
    feat = rng.standard_normal((N, 4)).astype(np.float32)
    coord = (grid * 0.05).astype(np.float32)
    R = torch.from_numpy(rng.standard_normal((N, 64)))
    point = model({"feat": torch.from_numpy(feat).cuda(), "coord": torch.from_numpy(coord).cuda(),
                   "grid_coord": torch.from_numpy(grid).cuda().int(), "batch": torch.from_numpy(batch).cuda()})
    (point.feat * R.float().cuda()).sum().backward()
    sd64 = {k: (v.detach().cpu().double().requires_grad_(True) if v.is_floating_point() else v.detach().cpu())
            for k, v in model.state_dict().items()}
    monkeypatch.setattr(P, "_t", lambda sd, k: sd[k])
    want = P.backbone_forward(sd64, cfg, feat, coord, grid, batch)
    assert _rel(point.feat.detach(), want.detach()) <= 2e-4
    (want * R).sum().backward()
    errs = {}
    for name, p in model.named_parameters():
        ref = sd64[name].grad
        assert ref is not None and p.grad is not None, name
        errs[name] = float((p.grad.double().cpu() - ref).norm()) / max(float(ref.norm()), 1e-30)
    bad = sorted(((e, k) for k, e in errs.items() if not e <= 1e-4), reverse=True)      # measured: 2.4e-6 at worst
    print(f"backbone gradients: {len(errs)} parameters, worst relative L2 error {max(errs.values()):.2e}")
    assert not bad, bad[:12]
    assert len(errs) > 150


def test_training_steps_with_stochastic_depth_reduce_the_loss():
    """PointTransformerWithHeads in train mode (batch-statistics BatchNorm, DropPath 0.3 spread over the blocks, the reference's
    loss): a few AdamW steps on one synthetic tree lower the loss; eval mode afterwards still runs."""
    helpers.load_pkg()
    from pn2_amd.PointTransformerV3.PointTransformerV3 import PointTransformerWithHeads
    from pn2_amd.synthetic import gaussian_branch_tree
    torch.manual_seed(0)
    model = PointTransformerWithHeads(dim_feat=4, use_feats=True, voxel_size=0.05).cuda().train()
    xyz, off, _ = gaussian_branch_tree(20000, seed=5)
    n = len(xyz)
    batch = {"coords": torch.from_numpy(xyz), "feats": torch.randn(n, 4), "batch_ids": torch.zeros(n, dtype=torch.long),
             "semantic_labels": (torch.arange(n) % 2).cuda(), "offset_labels": torch.from_numpy(np.asarray(off, np.float32)).cuda(),
             "masks_off": torch.ones(n, dtype=torch.bool).cuda()}
    opt = torch.optim.AdamW(model.parameters(), lr=2e-3)
    losses = []
    for _ in range(8):
        opt.zero_grad(set_to_none=True)
        loss, ld = model(batch, return_loss=True)
        assert bool(torch.isfinite(loss))
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    missing = [k for k, p in model.named_parameters() if p.grad is None]
    assert not missing, missing[:5]
    assert min(losses[-3:]) < losses[0], losses
    model.eval()
    with torch.no_grad():
        out = model(batch, return_loss=False)
    assert bool(torch.isfinite(out["offset_predictions"]).all())


@pytest.mark.parametrize("rows,cin,cout", [(70000, 32, 96), (70000, 128, 32), (5000, 512, 2048), (3000, 64, 2), (1024, 36, 100)])
def test_linear_on_library_gemms_equals_torch_linear(rows, cin, cout):
    """PointTransformerV3/linear.py: the dense layers of the mirror run on this library's GEMM kernels (forward, input gradient,
    split-K weight gradient, bias column sums) -- against torch.nn.functional.linear in float64."""
    helpers.load_pkg()
    from pn2_amd.PointTransformerV3.linear import Linear
    torch.manual_seed(rows + cin)
    lin = Linear(cin, cout).cuda()
    x0 = torch.randn(rows, cin)
    gout = torch.randn(rows, cout)
    x64 = x0.double().requires_grad_(True)
    w64, b64 = lin.weight.detach().cpu().double().requires_grad_(True), lin.bias.detach().cpu().double().requires_grad_(True)
    want = torch.nn.functional.linear(x64, w64, b64)
    want.backward(gout.double())
    x = x0.cuda().requires_grad_(True)
    got = lin(x)
    assert _rel(got.detach(), want.detach()) <= 2e-6
    got.backward(gout.cuda())
    assert _rel(x.grad, x64.grad) <= 2e-6
    assert _rel(lin.weight.grad, w64.grad) <= 2e-5 and _rel(lin.bias.grad, b64.grad) <= 2e-5
    with torch.no_grad():                                     # inference path, and torch's own for what the kernels do not take
        assert _rel(lin(x0.cuda()), want.detach()) <= 2e-6
        assert _rel(lin(x0.cuda()[:100]), want.detach()[:100]) <= 2e-6


@pytest.mark.parametrize("rows,C", [(70001, 32), (33333, 64), (9000, 128), (4097, 256), (1500, 512), (3, 32)])
def test_layer_norm_kernels_equal_torch_layer_norm(rows, C):
    """PointTransformerV3/linear.py LayerNorm on csrc/ptv3_norm.hip, forward and backward, against torch's in float64; an
    unsupported width falls through to torch."""
    helpers.load_pkg()
    from pn2_amd.PointTransformerV3.linear import LayerNorm
    torch.manual_seed(rows + C)
    ln = LayerNorm(C).cuda()
    with torch.no_grad():
        ln.weight.copy_(torch.rand(C) + 0.5)
        ln.bias.copy_(torch.randn(C) * 0.2)
    x0 = torch.randn(rows, C) * 2.0 + 0.7
    gout = torch.randn(rows, C)
    x64 = x0.double().requires_grad_(True)
    w64, b64 = ln.weight.detach().cpu().double().requires_grad_(True), ln.bias.detach().cpu().double().requires_grad_(True)
    want = torch.nn.functional.layer_norm(x64, (C,), w64, b64, ln.eps)
    want.backward(gout.double())
    x = x0.cuda().requires_grad_(True)
    got = ln(x)
    assert _rel(got.detach(), want.detach()) <= 2e-6
    got.backward(gout.cuda())
    assert _rel(x.grad, x64.grad) <= 5e-6
    assert _rel(ln.weight.grad, w64.grad) <= 2e-5 and _rel(ln.bias.grad, b64.grad) <= 2e-5
    with torch.no_grad():
        assert _rel(ln(x0.cuda()), want.detach()) <= 2e-6
    odd = LayerNorm(48).cuda()
    assert tuple(odd(torch.randn(10, 48, device="cuda")).shape) == (10, 48)
