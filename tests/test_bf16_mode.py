"""GPU: the bf16-operand MFMA throughput mode (mlp.GEMM_PRECISION = "bf16"; BASELINE configs[1] names bf16) -- a separate
mode with its own tolerance; the parity mode stays fp32.

What is rounded: the two operands of every large (128-row-tile) contraction, to bfloat16 (8 significant bits), in front of
v_mfma_f32_32x32x16_bf16, and [round 3] the large chains' pre-BatchNorm rows and gradient rows IN MEMORY (bfloat16 storage,
include/pn2_hip.h PN2_CHAIN_STORE_BF16; PN2_BF16_STORAGE=0 keeps them fp32); accumulation, statistics, weights, weight
gradients and all small layers stay fp32.  Expected error per contraction ~ 2^-9 * sqrt(K) relative to the operand magnitudes;
stated bars = measured value + 30 % (measured values are printed):
  * one 128 -> 128 -> 128 chain on Gaussian inputs against the fp32 mode: forward 9.5e-3 of the largest magnitude (measured
    7.3e-3; 5.7e-3 with fp32 rows); gradients in relative L2 norm: 1.2e-1 (measured 9.2e-2 for dx and for dW; 8.6e-2 with
    fp32 rows).  A bf16-sized change of a pre-activation
    flips the ReLU of every element that sits within ~0.5 % of zero -- with Gaussian pre-activations that is ~0.5 % of all
    elements per layer -- and each flip switches one entry of dY on or off: relative L2 ~ sqrt(fraction flipped) ~ 7-9 %,
    for dx and (the flips have random signs) for the row sums dW alike.  This is the mode's honest gradient noise on a
    random chain -- comparable to mini-batch noise, and the reason it is a separate mode; the maximum error says nothing;
  * whole PointNet2(depth 5) forward on a 16 384-point tree against the float64-layer-arithmetic yardstick of the torch-CPU
    restatement (the same one the fp32 mode meets at 1e-4): 2.3e-2 of the largest offset (measured 1.7e-2);
  * 20 optimizer steps (AdamW, lr 0.01) of the depth-4 model on one 16 384-point tree, bf16 mode against fp32 mode from the same
    initial weights and FPS draws: the two loss curves (test_bf16_loss_curve_tracks_fp32).
"""
import contextlib

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


@contextlib.contextmanager
def precision(mode):
    from pn2_amd import mlp
    old = mlp.GEMM_PRECISION
    mlp.GEMM_PRECISION = mode
    try:
        yield
    finally:
        mlp.GEMM_PRECISION = old


def test_bf16_chain_close_to_fp32_chain():
    helpers.load_pkg()
    from pn2_amd.mlp import chain_rows
    from test_streaming import _mlp
    rows, cin = 65536, 128
    convs, bns = _mlp([128, 128, 128], cin, False, seed=3)
    layers = [(c, b, True) for c, b in zip(convs, bns)]
    x = torch.randn(rows, cin, device="cuda", generator=torch.Generator("cuda").manual_seed(1))
    gout = torch.randn(rows, 128, device="cuda", generator=torch.Generator("cuda").manual_seed(2))
    res = {}
    for mode in ("f32", "bf16"):
        for p in list(convs.parameters()) + list(bns.parameters()):
            p.grad = None
        with precision(mode):
            xa = x.clone().requires_grad_(True)
            y = chain_rows(xa, layers)
            y.backward(gout)
        res[mode] = (y.detach(), xa.grad, [p.grad.clone() for p in convs.parameters() if p.grad is not None])
    (y0, d0, w0), (y1, d1, w1) = res["f32"], res["bf16"]
    ey = float((y0 - y1).abs().max() / y0.abs().max())
    ed = float((d0 - d1).norm() / d0.norm())
    ew = max(float((a - b).norm() / a.norm()) for a, b in zip(w0, w1) if float(a.abs().max()) > 1e-6)
    print(f"bf16 vs fp32 chain: out {ey:.2e} of the largest magnitude; relative L2: dx {ed:.2e}, dW {ew:.2e}")
    assert 1e-5 < ey <= 9.5e-3 and ed <= 1.2e-1 and ew <= 1.2e-1        # > 1e-5: the mode really ran


def test_bf16_model_forward_against_f64_yardstick():
    helpers.load_pkg()
    from oracle import torch_port as P
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    from test_round2 import _f64_layers, _tree_batch
    _, batch = _tree_batch(16384, seed=0)
    torch.manual_seed(77)
    ref = P.PortPointNet2(depth=5).train()
    torch.manual_seed(77)
    model = PointNet2(depth=5).train().cuda()
    P.STABLE_SORT = True
    try:
        with _f64_layers(), torch.no_grad():
            torch.manual_seed(5)
            _, off64 = ref(batch["coords"], batch["feats"])
    finally:
        P.STABLE_SORT = False
    gb = {k: v.cuda() for k, v in batch.items()}
    out = {}
    for mode in ("f32", "bf16"):
        state = {k: v.clone() for k, v in model.state_dict().items()}
        with precision(mode), torch.no_grad():
            torch.manual_seed(5)
            out[mode] = model(gb, return_loss=False)["offset_predictions"].cpu().numpy()
        model.load_state_dict(state)
    scale = float(np.abs(off64.numpy()).max())
    e32 = float(np.abs(out["f32"] - off64.numpy()).max()) / scale
    e16 = float(np.abs(out["bf16"] - off64.numpy()).max()) / scale
    print(f"depth-5 offsets vs the float64 yardstick: fp32 mode {e32:.2e}, bf16 mode {e16:.2e} (of the largest offset)")
    assert e32 <= 1e-4 and 1e-5 < e16 <= 2.3e-2


def test_bf16_loss_curve_tracks_fp32():
    """20 training steps of the depth-4 model on a 16 384-point tree in both modes, same initial weights, same FPS start draws:
    the bf16 mode's loss curve stays with the fp32 one (it is a mode to train in, not only to time)."""
    helpers.load_pkg()
    from pn2_amd import parallel
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    from test_round2 import _tree_batch
    _, batch = _tree_batch(16384, seed=1)
    gb = {k: v.cuda() for k, v in batch.items()}
    curves = {}
    for mode in ("f32", "f32 again", "bf16"):
        torch.manual_seed(11)
        model = PointNet2(depth=4, loss_multiplier_semantic=0).cuda().train()
        grads = parallel.FlatGradAllReduce(model, flatten_params=True)
        opt = torch.optim.AdamW(grads.optimizer_params(), lr=0.01, weight_decay=1e-3, fused=True)
        losses = []
        with precision(mode.split()[0]):
            for step in range(20):
                torch.manual_seed(1000 + step)                 # the step's FPS start draws
                grads.zero()
                loss, _ = model(gb, return_loss=True)
                (loss * 50).backward()
                opt.step()
                losses.append(float(loss.detach()))
        curves[mode] = np.array(losses)
    a, b = curves["f32"], curves["bf16"]
    rel = np.abs(a - b) / a
    print("fp32 loss curve:", np.round(a, 4).tolist())
    print("bf16 loss curve:", np.round(b, 4).tolist())
    print(f"relative difference: first step {rel[0]:.2e}, mean {rel.mean():.2e}, max {rel.max():.2e}")
    assert a[-1] < 0.8 * a[0] and b[-1] < 0.8 * b[0]                  # both train
    # The trajectories are chaotic: the float atomics of the scatter kernels differ run to run, and over 20 AdamW steps at
    # lr = 0.01 two fp32 runs of the SAME code already differ by 1.6-4.1e-2 on average and 6-13e-2 at worst (ten runs, round 3).
    # Measured bf16 vs fp32: first step 8e-5, mean 3.4-5.7e-2, max 1.2-1.9e-1 (twenty runs), worst seen 7.2e-2 / 2.2e-1.  The
    # yardstick is therefore this run's own fp32-vs-fp32 distance: the bf16 curve may be three times as far, with the absolute
    # floor the measurement gives (worst seen + 30 %: 9.5e-2 mean, 2.8e-1 max).
    own = np.abs(curves["f32 again"] - a) / a
    print(f"fp32 vs fp32 (same code, run to run): mean {own.mean():.2e}, max {own.max():.2e}")
    assert rel[0] <= 1e-3
    assert rel.mean() <= max(9.5e-2, 3.0 * own.mean()) and rel.max() <= max(2.8e-1, 3.0 * own.max())
