"""GPU: a feature-propagation level without a skip connection runs its first convolution on the SAMPLED rows and interpolates
the result (conv(interp(P)) = interp(conv(P)), include/pn2_hip.h "first convolution HOISTED": pn2_interp_bn_{fwd,bwd}_f32)
-- against the launch-per-layer path on the interpolated rows (PN2_NO_HOIST=1) and against a float64 torch evaluation of
the reference's expression order (blocks.py:194-215): outputs, running statistics and every gradient."""
import os

import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


class env:
    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kw}
        for k, v in self.kw.items():
            os.environ[k] = str(v)

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _fp(d2, widths):
    helpers.load_pkg()
    from pn2_amd.PointNet2.blocks import PointNetFeaturePropagation
    torch.manual_seed(11)
    fp = PointNetFeaturePropagation(d2, widths).cuda().train()
    with torch.no_grad():
        for bn in fp.mlp_bns:
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
    return fp


def _inputs(B, N, S, d2):
    g = torch.Generator().manual_seed(3)
    xyz1 = torch.rand(B, 3, N, generator=g).cuda()
    xyz2 = xyz1[:, :, torch.randperm(N, generator=g)[:S]].contiguous()
    p2 = torch.randn(B, d2, S, generator=g).cuda()
    return xyz1, xyz2, p2


def _run(fp, xyz1, xyz2, p2, hoist, no_link_sums=False):
    for p in fp.parameters():
        p.grad = None
    for bn in fp.mlp_bns:
        bn.running_mean.zero_()
        bn.running_var.fill_(1.0)
        bn.num_batches_tracked.zero_()
    pin = p2.clone().requires_grad_(True)
    kw = {} if hoist else {"PN2_NO_HOIST": 1}
    if no_link_sums:
        kw["PN2_NO_LINK_SUMS"] = 1
    with env(**kw):
        y = fp(xyz1, xyz2, None, pin)
        wgt = torch.cos(torch.arange(y.numel(), device="cuda", dtype=torch.float32) * 0.37).view_as(y)
        (y * wgt).sum().backward()
    torch.cuda.synchronize()
    out = {"y": y.detach().clone(), "dp2": pin.grad.clone()}
    for i, (c, b) in enumerate(zip(fp.mlp_convs, fp.mlp_bns)):
        out[f"w{i}"], out[f"b{i}"] = c.weight.grad.clone(), c.bias.grad.clone()
        out[f"g{i}"], out[f"be{i}"] = b.weight.grad.clone(), b.bias.grad.clone()
        out[f"rm{i}"], out[f"rv{i}"] = b.running_mean.clone(), b.running_var.clone()
        out[f"nbt{i}"] = b.num_batches_tracked.clone().float()
    return out


def _reference64(fp, xyz1, xyz2, p2):
    """The reference's forward (blocks.py:204-215) in float64 with autograd, on the product's 3-NN indices and weights."""
    x1, x2 = xyz1.double().permute(0, 2, 1), xyz2.double().permute(0, 2, 1)
    pin = p2.double().clone().requires_grad_(True)
    from pn2_amd import ops
    idx, w = ops.three_nn(x1.float(), x2.float())          # the neighbours and fp32 weights both paths use (tested elsewhere)
    idx, w = idx.long(), w.double()
    pts = pin.permute(0, 2, 1)
    B, N, _ = x1.shape
    gathered = torch.stack([pts[b][idx[b]] for b in range(B)])          # [B,N,3,D]
    x = (gathered * w.unsqueeze(-1)).sum(dim=2).permute(0, 2, 1)        # [B,D,N]
    params = []
    for conv, bn in zip(fp.mlp_convs, fp.mlp_bns):
        W = conv.weight.detach().double().squeeze(-1).requires_grad_(True)
        bias = conv.bias.detach().double().requires_grad_(True)
        g = bn.weight.detach().double().requires_grad_(True)
        be = bn.bias.detach().double().requires_grad_(True)
        params.append((W, bias, g, be))
        z = torch.einsum("oc,bcn->bon", W, x) + bias[None, :, None]
        mean = z.mean(dim=(0, 2), keepdim=True)
        var = z.var(dim=(0, 2), unbiased=False, keepdim=True)
        x = torch.relu((z - mean) / torch.sqrt(var + bn.eps) * g[None, :, None] + be[None, :, None])
    wgt = torch.cos(torch.arange(x.numel(), device="cuda", dtype=torch.float32) * 0.37).view_as(x).double()
    (x * wgt).sum().backward()
    out = {"y": x.detach(), "dp2": pin.grad}
    for i, (W, bias, g, be) in enumerate(params):
        out[f"w{i}"], out[f"g{i}"], out[f"be{i}"] = W.grad.unsqueeze(-1), g.grad, be.grad
    return out


def _close(a, b, tol, what):
    a, b = a.double(), b.double().reshape(a.shape)
    scale = float(b.abs().max()) + 1e-30
    err = float((a - b).abs().max()) / scale
    assert err <= tol, f"{what}: max error {err:.3g} of the largest magnitude (limit {tol:g})"


@pytest.mark.parametrize("B,N,S,d2,widths", [
    (2, 5000, 64, 128, [128, 128]),          # fp1's shape class: ragged last block, two clouds
    (1, 20000, 256, 96, [64, 64, 32]),       # C = 64 (256-row blocks), input width != output width
    (3, 1111, 16, 64, [256, 128]),           # C = 256 (64-row blocks)
])
def test_hoisted_level_matches_plain_path_and_float64(B, N, S, d2, widths):
    fp = _fp(d2, widths)
    xyz1, xyz2, p2 = _inputs(B, N, S, d2)
    plain = _run(fp, xyz1, xyz2, p2, hoist=False)
    hoisted = _run(fp, xyz1, xyz2, p2, hoist=True)
    ref = _reference64(fp, xyz1, xyz2, p2)
    for k in plain:
        # conv biases in front of a train-mode BatchNorm have a zero gradient: both paths leave rounding noise there
        if k.startswith("b") and not k.startswith("be"):
            continue
        _close(hoisted[k], plain[k], 2e-5, f"hoisted vs plain: {k}")
    for k in ref:
        _close(hoisted[k], ref[k], 5e-5, f"hoisted vs float64: {k}")
        _close(plain[k], ref[k], 5e-5, f"plain vs float64: {k}")


def test_hoisted_level_really_runs_and_reduces_its_own_sums():
    """The hoisted kernels are what ran (launch names), and the backward without the consumer's sums (PN2_NO_LINK_SUMS) gives the
    same gradients."""
    from pn2_amd import _hip
    fp = _fp(128, [128, 128, 128])
    xyz1, xyz2, p2 = _inputs(2, 4096, 128, 128)
    names = []
    orig = _hip.call

    def spy(name, fn, *a, **kw):
        names.append(name)
        return orig(name, fn, *a, **kw)

    _hip.call = spy
    try:
        linked = _run(fp, xyz1, xyz2, p2, hoist=True)
    finally:
        _hip.call = orig
    assert "interp_bn_fwd" in names and "interp_bn_bwd" in names and "three_interpolate" not in names
    own = _run(fp, xyz1, xyz2, p2, hoist=True, no_link_sums=True)
    for k in linked:
        if k.startswith("b") and not k.startswith("be"):
            continue
        _close(own[k], linked[k], 1e-5, f"own sums vs linked sums: {k}")


def test_hoisted_level_on_ragged_clouds_with_segments():
    """Whole-tree execution (streaming.run_tree: ragged level-0 clouds, one BatchNorm segment per mini-batch) with fp1's first
    conv hoisted against the same pass with PN2_NO_HOIST=1: losses, accumulated gradients, BatchNorm buffers."""
    import numpy as np
    from test_streaming import _Scaler, _grads_close, _tree_minibatches
    helpers.load_pkg()
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    mbs, labels = _tree_minibatches(30000, seed=5, mbs=8)
    assert len(mbs) >= 3
    res = {}
    for hoist in (False, True):
        with env(PN2_STREAMING="fused", **({} if hoist else {"PN2_NO_HOIST": 1})):
            torch.manual_seed(11)
            model = PointNet2(depth=5).cuda().train()
            torch.manual_seed(12)
            loss, ld = model.forward_hierarchical_streaming(dict(labels, mini_batches=iter(mbs)), return_loss=True, scaler=_Scaler())
            res[hoist] = (float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters()},
                          {n: b.detach().clone() for n, b in model.named_buffers()})
    (l0, g0, b0), (l1, g1, b1) = res[False], res[True]
    assert abs(l0 - l1) <= 1e-5 * abs(l0)
    _grads_close(g0, g1)
    for n in b0:
        if "num_batches" in n:
            assert int(b0[n]) == int(b1[n])
        else:
            np.testing.assert_allclose(b1[n].cpu().numpy(), b0[n].cpu().numpy(), rtol=1e-4, atol=1e-5)


def _sa_run(sa, xyz, pts, hoist):
    for p in sa.parameters():
        p.grad = None
    for m in sa.modules():
        if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
            m.running_mean.zero_()
            m.running_var.fill_(1.0)
            m.num_batches_tracked.zero_()
    pin = pts.clone().requires_grad_(True)
    kw = {"PN2_HOIST_GROUP_MIN_ROWS": 1} if hoist else {"PN2_NO_HOIST_GROUP": 1}
    names = []
    from pn2_amd import _hip
    orig = _hip.call

    def spy(name, fn, *a, **k):
        names.append(name)
        return orig(name, fn, *a, **k)

    _hip.call = spy
    try:
        with env(**kw):
            torch.manual_seed(4)                          # the FPS start draws
            new_xyz, y = sa(xyz, pin)
            wgt = torch.cos(torch.arange(y.numel(), device="cuda", dtype=torch.float32) * 0.37).view_as(y)
            (y * wgt).sum().backward()
    finally:
        _hip.call = orig
    torch.cuda.synchronize()
    out = {"new_xyz": new_xyz.detach().clone(), "y": y.detach().clone(), "dpts": pin.grad.clone()}
    for n, p in sa.named_parameters():
        out["grad:" + n] = p.grad.clone()
    for n, b in sa.named_buffers():
        out["buf:" + n] = b.detach().clone().float()
    return out, names


@pytest.mark.parametrize("msg,B,N,S,K,D,widths", [
    (False, 3, 400, 64, 32, 64, [64, 64, 128]),        # second level of a raster pass
    (False, 2, 100, 50, 32, 128, [128, 128, 256]),     # more groups than a source point can fill: padded balls
    (False, 5, 300, 40, 16, 32, [32, 64]),             # C = 32 (512-row blocks, half a wavefront of channels), K = 16
    (True, 2, 500, 48, 32, 64, [64, 128]),             # multi-scale channel order [feats, xyz]
])
def test_hoisted_set_abstraction_matches_plain_path(msg, B, N, S, K, D, widths):
    helpers.load_pkg()
    from pn2_amd.PointNet2.blocks import PointNetSetAbstraction, PointNetSetAbstractionMsg
    torch.manual_seed(21)
    if msg:
        sa = PointNetSetAbstractionMsg(S, [0.25, 0.4], [K, K], D + 3, [widths, widths]).cuda().train()
    else:
        sa = PointNetSetAbstraction(S, 0.3, K, D + 3, widths, False).cuda().train()
    with torch.no_grad():
        for m in sa.modules():
            if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                m.weight.uniform_(0.5, 1.5)
                m.bias.uniform_(-0.3, 0.3)
    g = torch.Generator().manual_seed(8)
    xyz = (torch.rand(B, 3, N, generator=g) + 3.0).cuda()           # off-centre clouds: the coordinates are centred per group
    pts = torch.randn(B, D, N, generator=g).cuda()
    plain, n0 = _sa_run(sa, xyz, pts, hoist=False)
    hoisted, n1 = _sa_run(sa, xyz, pts, hoist=True)
    assert "group_bn_fwd" in n1 and "group_bn_bwd" in n1 and "group_points" not in n1
    assert "group_points" in n0 and "group_bn_fwd" not in n0
    for k in plain:
        if k.startswith("grad:") and k.endswith("0.bias") and "convs" in k:
            continue                                                  # (zero gradient: rounding noise on the plain path)
        if "num_batches" in k or k == "new_xyz":
            assert torch.equal(hoisted[k], plain[k]), k
        else:
            _close(hoisted[k], plain[k], 3e-5, f"hoisted vs plain: {k}")


def test_hoisted_set_abstraction_with_row_segments():
    """Whole-tree execution hands a level the clouds of ALL mini-batches with one BatchNorm segment per mini-batch: the hoisted
    level against the grouped-tensor path on the same segments (per-segment statistics, coefficient blocks, running buffers)."""
    helpers.load_pkg()
    from pn2_amd import ops
    from pn2_amd.PointNet2.blocks import PointNetSetAbstraction, _group_mlp_max, _hoisted_group_mlp_max
    B, N, S, K, D = 7, 300, 40, 32, 64
    torch.manual_seed(31)
    sa = PointNetSetAbstraction(S, 0.3, K, D + 3, [64, 64, 128], False).cuda().train()
    g = torch.Generator().manual_seed(9)
    xyz = (torch.rand(B, N, 3, generator=g) - 2.0).cuda()
    pts = torch.randn(B, N, D, generator=g).cuda()
    start = torch.zeros(B, dtype=torch.int64, device="cuda")
    _, new_xyz = ops.furthest_point_sample(xyz, S, start)
    idx = ops.ball_query(0.3, K, xyz, new_xyz)
    seg = [0, 2 * S * K, 3 * S * K, 7 * S * K]
    res = {}
    for hoist in (False, True):
        for p in sa.parameters():
            p.grad = None
        for m in sa.mlp_bns:
            m.running_mean.zero_()
            m.running_var.fill_(1.0)
            m.num_batches_tracked.zero_()
        pin = pts.clone().requires_grad_(True)
        if hoist:
            y = _hoisted_group_mlp_max(xyz, new_xyz, pin, idx, sa.mlp_convs, sa.mlp_bns, False, seg_off=seg)
        else:
            grouped = ops.GroupPoints.apply(xyz, new_xyz, pin, idx, False)
            y = _group_mlp_max(grouped, sa.mlp_convs, sa.mlp_bns, seg_off=seg, coords_first=3)
        wgt = torch.cos(torch.arange(y.numel(), device="cuda", dtype=torch.float32) * 0.37).view_as(y)
        (y * wgt).sum().backward()
        torch.cuda.synchronize()
        out = {"y": y.detach().clone(), "dpts": pin.grad.clone()}
        for n, p in sa.named_parameters():
            out["grad:" + n] = p.grad.clone()
        for n, b in sa.named_buffers():
            out["buf:" + n] = b.detach().clone().float()
        res[hoist] = out
    for k in res[False]:
        if k == "grad:mlp_convs.0.bias":
            continue
        _close(res[True][k], res[False][k], 3e-5, f"hoisted vs plain with segments: {k}")


@pytest.mark.parametrize("linked,segments,precision", [(True, False, "f32"), (True, True, "f32"), (False, False, "f32"),
                                                       (True, False, "bf16")])
def test_pair_dgrad_as_one_contraction(linked, segments, precision):
    """The two heads' first-layer dgrad as ONE contraction over K = 2 * 128 (pn2_mlp_pair_dgrad_f32) against the two launches
    of which the second accumulates (PN2_NO_PAIR_DGRAD=1): input gradient, the producer's gradients (its BatchNorm-backward sums
    come out of the fused launch's epilogue), the heads' own gradients."""
    import torch.nn as nn
    helpers.load_pkg()
    from pn2_amd import _hip, mlp

    def build():
        torch.manual_seed(0)
        mk = lambda ci, co, bn=True: (nn.Conv1d(ci, co, 1).cuda(), nn.BatchNorm1d(co).cuda().train() if bn else None, bn)
        return [mk(128, 128), mk(128, 128)], [mk(128, 128), mk(128, 2, False)], [mk(128, 128), mk(128, 3, False)]

    rows = 20000
    seg = [0, 7000, 7000 + 5120, rows] if segments else None
    x0 = torch.randn(rows, 128, device="cuda")
    res = {}
    old = mlp.GEMM_PRECISION
    mlp.GEMM_PRECISION = precision
    try:
        for mode, e in (("fused", {}), ("two_launches", {"PN2_NO_PAIR_DGRAD": 1})):
            trunk, ha, hb = build()
            x = x0.clone().requires_grad_(True)
            names = []
            orig = _hip.call

            def spy(name, fn, *a, **k):
                names.append(name)
                return orig(name, fn, *a, **k)

            _hip.call = spy
            try:
                with env(**e):
                    feats = mlp.chain_rows(x, trunk, seg_off=seg, lazy_out=True) if linked else x
                    a, b = mlp.chain_pair_rows(feats, ha, hb, seg_off=seg)
                    ((a * a).sum() + (b * torch.arange(3, device="cuda")).sum()).backward()
            finally:
                _hip.call = orig
            assert ("mlp_pair_dgrad" in names) == (mode == "fused")
            chains = (trunk, ha, hb) if linked else (ha, hb)
            params = [p for chain in chains for conv, bn, _ in chain for p in ([conv.weight] + ([bn.weight, bn.bias] if bn else [conv.bias]))]
            res[mode] = [x.grad.clone()] + [p.grad.clone() for p in params]
    finally:
        mlp.GEMM_PRECISION = old
    tol = 2e-5 if precision == "f32" else 2e-2
    for g, h in zip(res["fused"], res["two_launches"]):
        assert float((g.float() - h.float()).norm()) <= tol * float(h.float().norm()) + 1e-7, (float((g - h).norm()), float(h.norm()))


def test_hoisted_layers_in_eval_mode():
    """Inference (running statistics): the hoisted feature-propagation level and the hoisted set-abstraction level against their
    plain paths, and the whole depth-5 model's eval forward hoisted vs not."""
    helpers.load_pkg()
    from pn2_amd import _hip
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    from pn2_amd.PointNet2.blocks import PointNetSetAbstraction
    g = torch.Generator().manual_seed(12)

    def randomise(mod):
        with torch.no_grad():
            for m in mod.modules():
                if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                    m.running_mean.copy_(torch.randn(m.num_features, generator=g) * 0.2)
                    m.running_var.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                    m.weight.copy_(torch.rand(m.num_features, generator=g) + 0.5)
                    m.bias.copy_(torch.randn(m.num_features, generator=g) * 0.1)

    def run(fn, hoist, **extra):
        names, orig = [], _hip.call

        def spy(name, f, *a, **k):
            names.append(name)
            return orig(name, f, *a, **k)

        _hip.call = spy
        try:
            with env(**(dict(extra) if hoist else {"PN2_NO_HOIST": 1})), torch.no_grad():
                out = fn()
        finally:
            _hip.call = orig
        torch.cuda.synchronize()
        return out, names

    fp = _fp(128, [128, 128, 128]).eval()
    randomise(fp)
    xyz1, xyz2, p2 = _inputs(2, 5000, 64, 128)
    a, na = run(lambda: fp(xyz1, xyz2, None, p2), True)
    b, nb = run(lambda: fp(xyz1, xyz2, None, p2), False)
    assert "interp_bn_fwd" in na and "interp_bn_fwd" not in nb
    _close(a, b, 2e-5, "eval feature propagation, hoisted vs plain")
    with pytest.raises(NotImplementedError):      # no backward through eval-mode BatchNorm, hoisted or not
        with env():
            fp(xyz1, xyz2, None, p2.clone().requires_grad_(True)).sum().backward()

    torch.manual_seed(21)
    sa = PointNetSetAbstraction(64, 0.3, 32, 64 + 3, [64, 64, 128], False).cuda().eval()
    randomise(sa)
    xyz = (torch.rand(3, 3, 400, generator=g) + 3.0).cuda()
    pts = torch.randn(3, 64, 400, generator=g).cuda()

    def sa_run():
        torch.manual_seed(4)
        return sa(xyz, pts)[1]

    a, na = run(sa_run, True, PN2_HOIST_GROUP_MIN_ROWS=1)
    b, nb = run(sa_run, False)
    assert "group_bn_fwd" in na and "group_bn_fwd" not in nb
    _close(a, b, 3e-5, "eval set abstraction, hoisted vs plain")

    torch.manual_seed(0)
    model = PointNet2(depth=5).cuda().eval()
    randomise(model)
    from pn2_amd.synthetic import gaussian_branch_tree
    xyz = torch.from_numpy(gaussian_branch_tree(16384, seed=1)[0].T[None].copy()).cuda()
    batch = {"coords": xyz, "feats": torch.ones(1, 4, 16384).cuda(), "masks_pad": torch.ones(1, 16384, dtype=torch.bool).cuda()}

    def model_run():
        torch.manual_seed(5)
        return model(batch, return_loss=False)["offset_predictions"]

    a, na = run(model_run, True)
    b, nb = run(model_run, False)
    assert "interp_bn_fwd" in na
    _close(a, b, 2e-5, "eval model offsets, hoisted vs plain")
