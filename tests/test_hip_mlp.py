"""GPU: the MFMA MLP-chain kernels (pn2_mlp_chain_{fwd,bwd}_f32) against a float64 torch-CPU evaluation of the
same conv -> BatchNorm -> ReLU [-> max] stack (a floating-point kernel, so the checker is a torch reference in
higher precision; tolerance 2e-5 of the tensor's largest magnitude forward, 1e-4 for gradients)."""
import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import helpers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def chain_rows():
    helpers.load_pkg()
    from pn2_amd.mlp import chain_rows
    return chain_rows


def build(widths, cin, conv_cls, bn_cls, last_bn=True, seed=0):
    torch.manual_seed(seed)
    layers, c = [], cin
    for i, w in enumerate(widths):
        conv = conv_cls(c, w, 1)
        bn = bn_cls(w) if (last_bn or i < len(widths) - 1) else None
        if bn is not None:
            with torch.no_grad():
                bn.weight.uniform_(0.5, 1.5)
                bn.bias.uniform_(-0.5, 0.5)
        layers.append((conv, bn, bn is not None))
        c = w
    return layers


def ref64(x, layers, pool_k, train=True):
    """float64 CPU reference; returns out and leaves grads on the float64 parameter copies."""
    x = x.double()
    params = []
    for conv, bn, relu in layers:
        w = conv.weight.detach().double().reshape(conv.out_channels, -1).requires_grad_(True)
        b = conv.bias.detach().double().requires_grad_(True)
        x = F.linear(x, w, b)
        g = be = None
        if bn is not None:
            g = bn.weight.detach().double().requires_grad_(True)
            be = bn.bias.detach().double().requires_grad_(True)
            x = F.batch_norm(x, None, None, g, be, True, 0.1, bn.eps)
        if relu:
            x = F.relu(x)
        params.append((w, b, g, be))
    if pool_k > 1:
        x = x.view(-1, pool_k, x.shape[-1]).max(dim=1)[0]
    return x, params


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


@pytest.mark.parametrize("rows,cin,widths,pool_k,last_bn,conv,bn", [
    (32 * 100, 7, [32, 32, 64], 32, True, nn.Conv2d, nn.BatchNorm2d),        # SA1-like, unaligned cin
    (32 * 37, 67, [64, 64, 128], 32, True, nn.Conv2d, nn.BatchNorm2d),       # ragged rows, cin = 64 + 3
    (16 * 8, 259, [256, 256, 512], 16, True, nn.Conv2d, nn.BatchNorm2d),     # wide, multi-tile N and K
    (32 * 1024, 7, [32, 32, 64], 32, True, nn.Conv2d, nn.BatchNorm2d),       # SA1 of the depth-4 table: 128 scatter blocks
    (32 * 512, 67, [128, 128, 256], 32, True, nn.Conv2d, nn.BatchNorm2d),     # pooled layer on the 128-tile kernels (TR_DYP)
    (20 * 53, 67, [64, 64, 128], 20, True, nn.Conv2d, nn.BatchNorm2d),       # K = min(nsample, N) when a cloud is small
    (5000, 128, [128, 128, 128], 1, True, nn.Conv1d, nn.BatchNorm1d),        # FP1-like
    (3001, 128, [128, 3], 1, False, nn.Conv1d, nn.BatchNorm1d),              # ConvHead: last conv bare, cout = 3
    (777, 320, [256, 128], 1, True, nn.Conv1d, nn.BatchNorm1d),              # FP2-like
    (64, 12, [8], 1, True, nn.Conv1d, nn.BatchNorm1d),                       # tiny
])
def test_chain_forward_backward(chain_rows, rows, cin, widths, pool_k, last_bn, conv, bn):
    layers = build(widths, cin, conv, bn, last_bn=last_bn, seed=rows)
    torch.manual_seed(rows + 1)
    x_cpu = torch.randn(rows, cin) * 2 + 0.5
    gout = None
    want, p64 = ref64(x_cpu.clone().requires_grad_(False), layers, pool_k)
    x64 = x_cpu.double().requires_grad_(True)
    want, p64 = ref64(x64, layers, pool_k)
    gout = torch.randn_like(want)
    want.backward(gout)

    for c_, b_, _ in layers:
        c_.cuda()
        if b_ is not None:
            b_.cuda().train()
    x = x_cpu.cuda().requires_grad_(True)
    got = chain_rows(x, layers, pool_k=pool_k)
    assert got.shape == want.shape
    assert rel(got.detach().cpu().double(), want.detach()) < 2e-5
    got.backward(gout.float().cuda())
    assert rel(x.grad.cpu().double(), x64.grad) < 1e-4, "dx"
    for (c_, b_, _), (w, b, g, be) in zip(layers, p64):
        assert rel(c_.weight.grad.cpu().double().reshape(w.shape), w.grad) < 1e-4, "dW"
        if b_ is not None:
            assert rel(b_.weight.grad.cpu().double(), g.grad) < 1e-4, "dgamma"
            assert rel(b_.bias.grad.cpu().double(), be.grad) < 1e-4, "dbeta"
            assert float(c_.bias.grad.abs().max()) == 0.0           # analytically zero, kept exactly zero
        else:
            assert rel(c_.bias.grad.cpu().double(), b.grad) < 1e-4, "dbias"


def test_pooled_chain_two_pass_sums(chain_rows, monkeypatch):
    """The pooled layer's BatchNorm-backward sums come out of the max-pool scatter by default; PN2_POOL_NO_SUMS=1 takes the
    scatter + bn_bwd_reduce pair instead: both against the float64 reference."""
    monkeypatch.setenv("PN2_POOL_NO_SUMS", "1")
    test_chain_forward_backward(chain_rows, 32 * 100, 7, [32, 32, 64], 32, True, nn.Conv2d, nn.BatchNorm2d)
    test_chain_forward_backward(chain_rows, 16 * 8, 259, [256, 256, 512], 16, True, nn.Conv2d, nn.BatchNorm2d)


@pytest.mark.parametrize("rows,cin,widths,pool_k,mode", [
    (32 * 1024, 7, [32, 32, 64], 32, "f32"),       # 64-tile kernels
    (32 * 512, 67, [128, 128, 256], 32, "f32"),    # 128-tile kernels
    (32 * 512, 67, [128, 128, 256], 32, "bf16"),   # ... with bfloat16 MFMA operands (fp32 rows)
    (16 * 700, 36, [64], 16, "f32"),               # the pooled layer is the chain's first: dx straight from the rebuilt gradient
    (32 * 1024, 8, [32, 32, 64], 32, "f32 segments"),        # three mini-batch segments (per-segment BatchNorm), 64-tile kernels
    (32 * 512, 68, [128, 128, 256], 32, "f32 segments"),     # ... 128-tile kernels
])
def test_pooled_gradient_rebuilt_while_staging_equals_dense_scatter(chain_rows, monkeypatch, rows, cin, widths, pool_k, mode):
    """TR_DYP (mlp_tile.h): the pooled layer's wgrad / dgrad rebuild the max-pool's gradient from (dout, arg-max bytes) instead
    of reading the dense scattered tensor.  Same values enter the same contractions in the same order: bit-identical
    gradients to the scatter path (PN2_NO_POOL_DYP=1)."""
    import pn2_amd.mlp as M
    monkeypatch.setenv("PN2_BF16_STORAGE", "0")
    monkeypatch.setenv("PN2_POOL_DYP_MIN_ROWS", "1")   # (the default leaves the small, latency-bound levels on the dense path)
    layers = build(widths, cin, nn.Conv2d, nn.BatchNorm2d, seed=5)
    for c_, b_, _ in layers:
        c_.cuda()
        b_.cuda().train()
    x0 = torch.randn(rows, cin, device="cuda", generator=torch.Generator("cuda").manual_seed(3))
    res = []
    seg_off = None
    if mode.endswith("segments"):
        mode = mode.split()[0]
        g = rows // pool_k
        seg_off = [0, pool_k * (g // 5), pool_k * (g // 2 + 3), rows]
    old = M.GEMM_PRECISION
    M.GEMM_PRECISION = mode
    try:
        for dense in (False, True):
            if dense:
                monkeypatch.setenv("PN2_NO_POOL_DYP", "1")
            for c_, b_, _ in layers:
                for p_ in list(c_.parameters()) + list(b_.parameters()):
                    p_.grad = None
            x = x0.clone().requires_grad_(True)
            out = chain_rows(x, layers, pool_k=pool_k, seg_off=seg_off)
            out.backward(torch.randn(out.shape, device="cuda", generator=torch.Generator("cuda").manual_seed(4)))
            res.append([x.grad.clone()] + [p_.grad.clone() for c_, b_, _ in layers for p_ in list(c_.parameters()) + list(b_.parameters())])
    finally:
        M.GEMM_PRECISION = old
    for a, b in zip(*res):
        assert torch.equal(a, b)
    assert float(res[0][0].abs().max()) > 0.0


@pytest.mark.parametrize("pool_k", [32, 1])
def test_leading_columns_of_dx_come_back_zero(chain_rows, pool_k):
    """dx_first_col: the input gradient is computed from that column on and the columns in front of it come back as exact
    zeros (zeroed inside the max-pool scatter launch of a pooled chain, by a launch of its own otherwise)."""
    rows, cin = 32 * 64, 67
    layers = build([64, 64, 128], cin, nn.Conv2d, nn.BatchNorm2d, seed=3)
    for c_, b_, _ in layers:
        c_.cuda()
        b_.cuda().train()
    torch.manual_seed(9)
    x0 = torch.randn(rows, cin, device="cuda")
    grads = []
    for skip in (0, 3):
        x = x0.clone().requires_grad_(True)
        out = chain_rows(x, layers, pool_k=pool_k, dx_first_col=skip)
        torch.manual_seed(10)
        out.backward(torch.randn_like(out))
        grads.append(x.grad.clone())
    assert float(grads[1][:, :3].abs().max()) == 0.0
    assert torch.equal(grads[1][:, 3:], grads[0][:, 3:]) or rel(grads[1][:, 3:].double().cpu(), grads[0][:, 3:].double().cpu()) < 1e-6
    assert float(grads[0][:, :3].abs().max()) > 0.0


def test_running_stats_and_eval_mode(chain_rows):
    layers = build([16, 8], 5, nn.Conv1d, nn.BatchNorm1d, seed=3)
    ref = [(nn.Conv1d(5, 16, 1), nn.BatchNorm1d(16)), (nn.Conv1d(16, 8, 1), nn.BatchNorm1d(8))]
    for (c, b, _), (rc, rb) in zip(layers, ref):
        rc.load_state_dict(c.state_dict())
        rb.load_state_dict(b.state_dict())
    x = torch.randn(500, 5)
    y = x.t()[None]
    for rc, rb in ref:
        rc.train()
        rb.train()
        y = F.relu(rb(rc(y)))
    for c, b, _ in layers:
        c.cuda()
        b.cuda().train()
    got = chain_rows(x.cuda(), layers)
    np.testing.assert_allclose(got.detach().cpu().numpy(), y[0].t().detach().numpy(), rtol=1e-4, atol=1e-5)
    for (c, b, _), (rc, rb) in zip(layers, ref):
        np.testing.assert_allclose(b.running_mean.cpu().numpy(), rb.running_mean.numpy(), rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(b.running_var.cpu().numpy(), rb.running_var.numpy(), rtol=1e-5, atol=1e-6)
        assert int(b.num_batches_tracked) == int(rb.num_batches_tracked) == 1
    # eval: running statistics
    y = x.t()[None]
    for rc, rb in ref:
        rb.eval()
        y = F.relu(rb(rc(y)))
    for c, b, _ in layers:
        b.eval()
    with torch.no_grad():
        got = chain_rows(x.cuda(), layers)
    np.testing.assert_allclose(got.detach().cpu().numpy(), y[0].t().detach().numpy(), rtol=1e-4, atol=1e-5)
