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
