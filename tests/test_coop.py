"""GPU: cooperative chain launches (csrc/chain_coop.hip: a deep level's whole MLP chain as ONE persistent launch per
direction, layers separated by a grid-wide arrival counter) against the launch-per-layer path (PN2_NO_COOP=1) on the same
inputs -- same tile arithmetic, so outputs, arg-max rows, running statistics and every gradient agree to fp32 rounding of the
BatchNorm merges -- and against a float64 torch evaluation of the same stack."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

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


def _stack(widths, cin, conv2d):
    torch.manual_seed(5)
    layers, c = [], cin
    for w in widths:
        conv = (nn.Conv2d(c, w, 1) if conv2d else nn.Conv1d(c, w, 1)).cuda()
        bn = (nn.BatchNorm2d(w) if conv2d else nn.BatchNorm1d(w)).cuda().train()
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.uniform_(-0.3, 0.3)
        layers.append((conv, bn, True))
        c = w
    return layers


def _run(x, layers, pool_k, first_col, coop):
    from pn2_amd import mlp
    for conv, bn, _ in layers:
        conv.weight.grad = conv.bias.grad = bn.weight.grad = bn.bias.grad = None
        bn.running_mean.zero_()
        bn.running_var.fill_(1.0)
    xin = x.clone().requires_grad_(True)
    with env(**({"PN2_COOP_MAX_ROWS": 100000} if coop else {"PN2_NO_COOP": 1})):   # (every shape here, whatever the default limit)
        y = mlp.chain_rows(xin, layers, pool_k=pool_k, dx_first_col=first_col)
        w = torch.cos(torch.arange(y.numel(), device="cuda", dtype=torch.float32) * 0.37).view_as(y)
        (y * w).sum().backward()
    torch.cuda.synchronize()
    out = {"y": y.detach().clone(), "dx": xin.grad.clone()}
    for i, (conv, bn, _) in enumerate(layers):
        out[f"dw{i}"], out[f"dg{i}"], out[f"db{i}"] = conv.weight.grad.clone(), bn.weight.grad.clone(), bn.bias.grad.clone()
        out[f"rm{i}"], out[f"rv{i}"] = bn.running_mean.clone(), bn.running_var.clone()
    return out


def _f64(x, layers, pool_k, first_col):
    xin = x.double().clone().requires_grad_(True)
    h = xin
    ps = []
    for conv, bn, _ in layers:
        w = conv.weight.detach().double().reshape(conv.out_channels, -1).requires_grad_(True)
        g, b = bn.weight.detach().double().requires_grad_(True), bn.bias.detach().double().requires_grad_(True)
        ps.append((w, g, b))
        z = h @ w.t() + conv.bias.detach().double()
        mu, var = z.mean(0), z.var(0, unbiased=False)
        h = torch.relu((z - mu) / torch.sqrt(var + bn.eps) * g + b)
    y = h.view(-1, pool_k, h.shape[1]).max(1)[0] if pool_k > 1 else h
    w = torch.cos(torch.arange(y.numel(), device="cuda", dtype=torch.float32) * 0.37).view_as(y).double()
    (y * w).sum().backward()
    dx = xin.grad.clone()
    dx[:, :first_col] = 0
    out = {"y": y.detach(), "dx": dx}
    for i, (w_, g, b) in enumerate(ps):
        out[f"dw{i}"], out[f"dg{i}"], out[f"db{i}"] = w_.grad, g.grad, b.grad
    return out


SHAPES = [  # rows, cin, widths, pool_k, first_col, conv2d
    (2048, 131, [128, 128, 256], 32, 3, True),      # SA3 of the depth-4 table
    (512, 259, [256, 256, 512], 32, 3, True),       # SA4
    (8192, 67, [64, 64, 128], 32, 3, True),         # SA2
    (1024, 320, [256, 128], 1, 0, False),           # FP2
    (256, 384, [256, 256], 1, 0, False),            # FP3
    (64, 768, [256, 256], 1, 0, False),             # FP4
    (1000, 40, [64, 32], 1, 0, False),              # ragged rows
    (96 * 16, 35, [32, 64, 64], 16, 0, True),       # K = 16 groups
    (3200, 7, [32, 32, 64], 32, 0, True),           # SA1 of the depth-5 table (no input gradient columns skipped)
    (72, 12, [16], 8, 0, True),                     # one layer, tiny
]


@pytest.mark.parametrize("rows,cin,widths,pool_k,first_col,conv2d", SHAPES)
def test_cooperative_chain_equals_launch_per_layer(rows, cin, widths, pool_k, first_col, conv2d):
    helpers.load_pkg()
    from pn2_amd import _hip, ops
    torch.manual_seed(rows + cin)
    x = torch.randn(rows, cin, device="cuda") * 1.5 + 0.2
    layers = _stack(widths, cin, conv2d)
    ref = _run(x, layers, pool_k, first_col, coop=False)
    names = []
    for rep in range(3):                                   # run to run: the barriers, the counters' reset
        got = _run(x, layers, pool_k, first_col, coop=True)
        ops.check_status()
        groups = _hip.kernel_profile(lambda: _run(x, layers, pool_k, first_col, coop=True)) if rep == 0 else []
        names += [g["name"] for g in groups]
        for k, v in ref.items():
            scale = float(v.abs().max()) + 1e-30
            err = float((got[k] - v).abs().max())
            assert err <= 2e-5 * scale, f"{k} (run {rep}): {err:.3e} vs scale {scale:.3e}"
    assert "chain_coop_fwd" in names and "chain_coop_bwd" in names, names     # the cooperative path really ran
    assert not any(n.startswith("gemm_") or n.startswith("bn_") for n in names), names
    want = _f64(x, layers, pool_k, first_col)
    for k, v in want.items():
        scale = float(v.abs().max()) + 1e-30
        err = float((got[k].double().reshape(v.shape) - v).abs().max())
        assert err <= 2e-4 * scale, f"{k} vs float64: {err:.3e} vs scale {scale:.3e}"


def test_whole_model_with_and_without_cooperative_chains():
    """depth-4 model, one 20 000-point tree: loss and every parameter gradient, cooperative vs launch-per-layer."""
    helpers.load_pkg()
    from pn2_amd import ops
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    from pn2_amd.synthetic import gaussian_branch_tree
    xyz, off, _ = gaussian_branch_tree(20000, seed=3)
    n = len(xyz)
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    batch = {"coords": dev(xyz.T[None].copy()), "feats": dev(np.sin(np.arange(4 * n, dtype=np.float32)).reshape(1, 4, n)),
             "semantic_labels": torch.zeros(n, dtype=torch.long, device="cuda"), "offset_labels": dev(off),
             "masks_off": torch.ones(n, dtype=torch.bool, device="cuda"), "masks_pad": torch.ones(1, n, dtype=torch.bool, device="cuda")}
    res = {}
    for mode, e in (("coop", {"PN2_COOP_MAX_ROWS": 10000}), ("plain", {"PN2_NO_COOP": 1})):
        torch.manual_seed(0)
        model = PointNet2(depth=4).cuda().train()
        torch.manual_seed(1)
        with env(**e):
            loss, _ = model(batch, return_loss=True)
            loss.backward()
        ops.check_status()
        res[mode] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None},
                     {k: b.detach().clone() for k, b in model.named_buffers() if b.dtype.is_floating_point})
    assert abs(res["coop"][0] - res["plain"][0]) <= 1e-5 * abs(res["plain"][0])
    gmax = max(float(g.abs().max()) for g in res["plain"][1].values())
    for k, g in res["plain"][1].items():
        assert float((res["coop"][1][k] - g).abs().max()) <= 2e-3 * gmax, k      # (atomics in the grouping backward: last bits differ anyway)
    for k, b in res["plain"][2].items():
        assert float((res["coop"][2][k] - b).abs().max()) <= 1e-5 * (float(b.abs().max()) + 1e-6), k
