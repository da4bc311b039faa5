"""GPU: the HIP path (through the C ABI) against the golden vectors of the imported reference and against the
C oracle on seeded inputs.

Bars (north_star): FPS / ball-query indices bit-exact; 3-NN distances, weights and interpolation bit-exact
(indices bit-exact wherever torch.sort specifies them, see test_oracle_golden.py); MLP outputs and per-point
offsets within 1e-4 relative (RTOL below, with an absolute floor of 1e-5 x the tensor's largest magnitude);
gradients within 2e-4 relative to the gradient's largest magnitude (sums over up to 65k rows in a different
order than the reference's CPU kernels).
"""
import os

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu

RTOL = 1e-4
ATOL_REL = 1e-5
GRAD_REL = 2e-4


@pytest.fixture(scope="module")
def pn2():
    return helpers.load_pkg()


@pytest.fixture(scope="module")
def U(pn2):
    from pn2_amd.PointNet2 import pointnet2_utils
    return pointnet2_utils


@pytest.fixture(scope="module")
def O():
    from oracle import pn2_oracle
    pn2_oracle.build()
    return pn2_oracle


def gold(name):
    return np.load(os.path.join(helpers.GOLDEN, name))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(got, want, rtol=RTOL, atol_rel=ATOL_REL, what=""):
    got = got.detach().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    atol = atol_rel * max(float(np.abs(want).max()), 1e-30)
    bad = np.abs(got - want) > atol + rtol * np.abs(want)
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.size} outside tolerance, max abs err " \
                          f"{float(np.abs(got - want).max()):.3e} (scale {float(np.abs(want).max()):.3e})"


def grad_close(got, want, what="", rel=GRAD_REL):
    got = got.detach().cpu().numpy()
    scale = max(float(np.abs(want).max()), 1e-30)
    err = float(np.abs(got - want).max())
    assert err <= rel * scale, f"{what}: max abs err {err:.3e} vs scale {scale:.3e}"


def bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


# ------------------------------------------------------------------------------------------------ golden: ops
def test_loaded_library_is_the_hip_extension(pn2):
    from pn2_amd import _hip
    assert _hip.lib().pn2_arch() == b"gfx950"
    assert os.path.samefile(_hip.LIB_PATH, os.path.join(helpers.PKG_DIR, "libpn2hip.so"))


def test_square_distance_golden(U):
    g = gold("ops.npz")
    xyz = dev(g["coords"]).permute(0, 2, 1)
    d = U.square_distance(dev(g["new_xyz"][:, :8]), xyz)
    assert np.array_equal(bits(d), g["sqdist_rows"].view(np.uint32))


def test_fps_golden(pn2, U):
    from pn2_amd import ops
    g = gold("ops.npz")
    xyz = dev(g["coords"]).permute(0, 2, 1)                      # permuted view, as the SA module passes it
    idx, new_xyz = ops.furthest_point_sample(xyz, 128, dev(g["fps_start"]))
    assert np.array_equal(idx.cpu().numpy(), g["fps_idx"])
    assert np.array_equal(bits(new_xyz), g["new_xyz"].view(np.uint32))
    idx2, _ = ops.furthest_point_sample(xyz.contiguous(), 128, dev(g["fps_start"]))   # AoS layout
    assert torch.equal(idx, idx2)
    # public API: same RNG stream as the reference's torch.randint draw
    torch.manual_seed(1234)
    api = U.farthest_point_sample(xyz, 128)
    assert api.dtype == torch.long and np.array_equal(api.cpu().numpy(), g["fps_idx"])


@pytest.mark.parametrize("tag,r,K", [("r01", 0.1, 32), ("r02", 0.2, 32), ("r005", 0.05, 16)])
def test_ball_query_golden(U, tag, r, K):
    g = gold("ops.npz")
    xyz = dev(g["coords"]).permute(0, 2, 1)
    idx = U.query_ball_point(r, K, xyz, dev(g["new_xyz"]))
    assert idx.dtype == torch.long
    assert np.array_equal(idx.cpu().numpy(), g[f"bq_{tag}"])


def test_ball_query_empty_balls_and_short_clouds(U):
    g = gold("ops.npz")
    xyz = dev(g["coords"]).permute(0, 2, 1)
    idx = U.query_ball_point(0.1, 32, xyz, dev(g["q_shift"]))
    assert np.array_equal(idx.cpu().numpy(), g["bq_shift"])
    small = U.query_ball_point(0.4, 32, xyz[:, :20], dev(g["new_xyz"][:, :5]))
    assert tuple(small.shape) == (2, 5, 20)
    assert np.array_equal(small.cpu().numpy(), g["bq_small"])


def test_sample_and_group_golden(U):
    g = gold("ops.npz")
    xyz = dev(g["coords"]).permute(0, 2, 1)
    feats = dev(g["feats"]).permute(0, 2, 1)
    torch.manual_seed(77)
    new_xyz, new_points, grouped_xyz, fps = U.sample_and_group(64, 0.2, 32, xyz, feats, returnfps=True)
    assert np.array_equal(fps.cpu().numpy(), g["sg_fps"])
    assert np.array_equal(bits(new_xyz), g["sg_new_xyz"].view(np.uint32))
    assert np.array_equal(bits(new_points[:, :16]), g["sg_new_points_head"].view(np.uint32))
    s = new_points.double()
    np.testing.assert_allclose([float(s.sum()), float(s.abs().sum())], g["sg_new_points_sum"], rtol=1e-12)
    assert tuple(grouped_xyz.shape) == (2, 64, 32, 3)


def test_three_nn_and_interpolate_golden(pn2, U, O):
    from pn2_amd import ops
    g = gold("ops.npz")
    xyz = dev(g["coords"]).permute(0, 2, 1)
    idx, w, dist = ops.three_nn(xyz, dev(g["new_xyz"]), want_dist=True)
    assert np.array_equal(bits(dist), g["nn_dist"].view(np.uint32))
    assert np.array_equal(bits(w), g["nn_weight"].view(np.uint32))
    d = O.square_distance(np.ascontiguousarray(g["coords"].transpose(0, 2, 1)), g["new_xyz"])
    s4 = np.sort(d, axis=-1)[:, :, :4]
    tie = (s4[:, :, 1:] == s4[:, :, :-1]).any(-1)
    got = idx.cpu().numpy()
    assert np.array_equal(got[~tie], g["nn_idx"][~tie])
    assert np.array_equal(got, np.argsort(d, axis=-1, kind="stable")[:, :, :3])      # our tie rule: lower index first
    # interpolation with the reference's own neighbour choice -> bit-exact
    out = U.three_interpolate(dev(g["interp_points2"]), dev(g["nn_idx"].astype(np.int64)), dev(g["nn_weight"]))
    assert np.array_equal(bits(out.permute(0, 2, 1).contiguous()), np.ascontiguousarray(g["interp_out"]).view(np.uint32))


# ------------------------------------------------------------------------------------- oracle: shapes and edges
def _cloud(B, N, seed, scale=1.0, shift=(0, 0, 0)):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(B, N, 3)) * scale + np.asarray(shift)).astype(np.float32)


@pytest.mark.parametrize("B,N,npoint", [(3, 5000, 37), (1, 64, 64), (2, 1, 3), (5, 333, 400), (1, 20000, 64),
                                         (2, 70001, 50), (300, 257, 9),
                                         # multi-pick rounds (>= 200 samples, >= 8 members): 4 / 8 / 16 points per lane
                                         (3, 20000, 512), (2, 70001, 300), (1, 150000, 256), (9, 16390, 200)])
def test_fps_vs_oracle(pn2, O, B, N, npoint):
    """single- and multi-workgroup clouds, ragged sizes, npoint > N (repeats), more clouds than workgroups, and the
    multi-pick kernel with zero-padded tails (thousands of identical points: every key tie is broken by index)."""
    from pn2_amd import ops
    xyz = _cloud(B, N, seed=N + B, scale=0.5, shift=(10.0, -20.0, 15.0))
    if N > 100:
        xyz[:, N - N // 5:] = 0.0                                # zero padding: exact ties
    start = np.random.default_rng(1).integers(0, N, size=B)
    want = O.farthest_point_sample(xyz, npoint, start)
    got, new_xyz = ops.furthest_point_sample(dev(xyz), npoint, dev(start))
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(new_xyz.cpu().numpy(), O.index_points(xyz, want))


@pytest.mark.parametrize("B,N,S,r,K", [(3, 5000, 70, 0.3, 32), (1, 100000, 16, 0.05, 32), (2, 999, 1000, 0.5, 8),
                                        (1, 4096, 4096, 0.1, 100), (70, 300, 40, 0.4, 32), (1, 65, 3, 10.0, 32)])
def test_ball_query_vs_oracle(U, O, B, N, S, r, K):
    """one segment and many segments, Q = 1..8, nsample > 64, every ball full, ragged tails."""
    xyz = _cloud(B, N, seed=N, scale=0.6, shift=(5.0, 5.0, 12.0))
    q = _cloud(B, S, seed=S + 1, scale=0.6, shift=(5.0, 5.0, 12.0))
    q[:, : min(S, N) // 2] = xyz[:, : min(S, N) // 2]            # half the queries are cloud members
    want = O.query_ball_point(r, K, xyz, q)
    got = U.query_ball_point(r, K, dev(xyz), dev(q))
    assert np.array_equal(got.cpu().numpy(), want)


def test_three_nn_vs_oracle_large_s(pn2, O):
    from pn2_amd import ops
    xyz1 = _cloud(2, 3001, 3, shift=(3, 3, 3))
    xyz2 = _cloud(2, 2500, 4, shift=(3, 3, 3))                   # more than one LDS tile
    dist, idx = O.three_nn(xyz1, xyz2)
    gi, gw, gd = ops.three_nn(dev(xyz1), dev(xyz2), want_dist=True)
    assert np.array_equal(gi.cpu().numpy(), idx)
    assert np.array_equal(bits(gd), dist.view(np.uint32))
    assert np.array_equal(bits(gw), O.three_weights(dist).view(np.uint32))
    with pytest.raises(RuntimeError):
        ops.three_nn(dev(xyz1), dev(xyz2[:, :2]))


def test_gather_and_group_grads_vs_oracle(U, O):
    rng = np.random.default_rng(0)
    B, N, S, K, D = 2, 500, 40, 16, 6
    pts = rng.normal(size=(B, N, D)).astype(np.float32)
    xyz = _cloud(B, N, 5)
    idx = rng.integers(0, N, size=(B, S, K))
    p = dev(pts).requires_grad_(True)
    out = U.index_points(p, dev(idx))
    assert np.array_equal(out.detach().cpu().numpy(), O.index_points(pts, idx))
    gout = rng.normal(size=out.shape).astype(np.float32)
    out.backward(dev(gout))
    np.testing.assert_allclose(p.grad.cpu().numpy(), O.index_points_grad(gout, idx, N), rtol=1e-5, atol=1e-5)

    from pn2_amd import ops
    new_xyz = xyz[:, :S].copy()
    for xyz_last in (False, True):
        f = dev(pts).requires_grad_(True)
        g = ops.GroupPoints.apply(dev(xyz), dev(new_xyz), f, dev(idx), xyz_last)
        assert np.array_equal(g.detach().cpu().numpy(), O.group(xyz, new_xyz, pts, idx, xyz_last=xyz_last))
        gg = rng.normal(size=g.shape).astype(np.float32)
        g.backward(dev(gg))
        sl = slice(0, D) if xyz_last else slice(3, 3 + D)
        np.testing.assert_allclose(f.grad.cpu().numpy(), O.index_points_grad(gg[..., sl], idx, N), rtol=1e-5, atol=1e-5)


def test_interpolate_concat_and_grad_vs_oracle(pn2, O):
    from pn2_amd import ops
    rng = np.random.default_rng(2)
    B, N, S, D1, D2 = 2, 700, 50, 5, 12
    idx = rng.integers(0, S, size=(B, N, 3))
    w = rng.uniform(0.1, 1.0, size=(B, N, 3)).astype(np.float32)
    p1 = rng.normal(size=(B, N, D1)).astype(np.float32)
    p2 = rng.normal(size=(B, S, D2)).astype(np.float32)
    t1, t2 = dev(p1).requires_grad_(True), dev(p2).requires_grad_(True)
    out = ops.ThreeInterpolateConcat.apply(t1, t2, dev(idx), dev(w))
    want = np.concatenate([p1, O.three_interpolate(p2, idx, w)], axis=-1)
    assert np.array_equal(out.detach().cpu().numpy(), want)
    g = rng.normal(size=want.shape).astype(np.float32)
    out.backward(dev(g))
    assert np.array_equal(t1.grad.cpu().numpy(), g[..., :D1])
    np.testing.assert_allclose(t2.grad.cpu().numpy(), O.three_interpolate_grad(np.ascontiguousarray(g[..., D1:]), idx, w, S),
                               rtol=1e-5, atol=1e-5)


# --------------------------------------------------------------------------------------------- golden: blocks
def _load_params(module, g):
    helpers.closed_form_init(module)
    return module.cuda().train()


def _check_param_grads(module, g, what):
    params = dict(module.named_parameters())
    for n, p in params.items():
        if helpers.is_pre_bn_bias(n):
            wn = float(params[n[:-4] + "weight"].grad.abs().max())
            assert float(p.grad.abs().max()) <= 1e-2 * wn, f"{what} {n}: pre-BN bias gradient should vanish"
        else:
            grad_close(p.grad, g["g__" + n], f"{what} grad {n}")


def test_set_abstraction_golden(pn2):
    from pn2_amd.PointNet2.blocks import PointNetSetAbstraction
    g = gold("sa.npz")
    sa = _load_params(PointNetSetAbstraction(64, 0.2, 32, 7, [16, 16, 32], False), g)
    feats = dev(g["feats"]).requires_grad_(True)
    torch.manual_seed(5)
    nx, npts = sa(dev(g["coords"]), feats)
    assert np.array_equal(nx.cpu().numpy(), g["new_xyz"])
    close(npts, g["new_points"], what="SA new_points")
    (npts * dev(g["G"])).sum().backward()
    grad_close(feats.grad, g["d_feats"], "SA d_feats")
    _check_param_grads(sa, g, "SA")
    for n, b in sa.named_buffers():
        if "num_batches" in n:
            assert int(b) == int(g["buf__" + n])
        else:
            close(b, g["buf__" + n], what=f"SA buffer {n}")


def test_feature_propagation_golden(pn2):
    from pn2_amd.PointNet2.blocks import PointNetFeaturePropagation
    g = gold("fp.npz")
    fp = _load_params(PointNetFeaturePropagation(32, [32, 16]), g)
    p1, p2 = dev(g["points1"]).requires_grad_(True), dev(g["points2"]).requires_grad_(True)
    y = fp(dev(g["coords"]), dev(g["xyz2"]), p1, p2)
    close(y, g["out"], what="FP out")
    (y * dev(g["G"])).sum().backward()
    grad_close(p1.grad, g["d_points1"], "FP d_points1")
    grad_close(p2.grad, g["d_points2"], "FP d_points2")
    _check_param_grads(fp, g, "FP")
    for n, b in fp.named_buffers():
        if "num_batches" not in n:
            close(b, g["buf__" + n], what=f"FP buffer {n}")
    fp1 = _load_params(PointNetFeaturePropagation(24, [8]), g)                       # S == 1 branch
    y1 = fp1(dev(g["coords"]), dev(g["xyz2"][:, :, :1]), None, dev(g["points2"][:, :, :1]))
    close(y1, g["out_s1"], what="FP S==1")


def test_set_abstraction_msg_golden(pn2):
    from pn2_amd.PointNet2.blocks import PointNetSetAbstractionMsg
    g = gold("msg.npz")
    msg = _load_params(PointNetSetAbstractionMsg(48, [0.05, 0.1, 0.2], [8, 16, 16], 7, [[8, 16], [8, 16], [16, 16]]), g)
    feats = dev(g["feats"]).requires_grad_(True)
    torch.manual_seed(6)
    nx, npts = msg(dev(g["coords"]), feats)
    assert np.array_equal(nx.cpu().numpy(), g["new_xyz"])
    close(npts, g["new_points"], what="MSG new_points")
    (npts * dev(g["G"])).sum().backward()
    grad_close(feats.grad, g["d_feats"], "MSG d_feats")
    _check_param_grads(msg, g, "MSG")


def test_conv_head_golden(pn2):
    import functools
    from pn2_amd.PointNet2.blocks import ConvHead
    g = gold("head.npz")
    head = _load_params(ConvHead(16, 3, norm_fn=functools.partial(torch.nn.BatchNorm1d, eps=1e-4, momentum=0.1),
                                 num_layers=2), g)
    x = dev(g["x"]).requires_grad_(True)
    y = head(x)
    close(y, g["out"], what="head out")
    (y * dev(g["G"])).sum().backward()
    grad_close(x.grad, g["d_x"], "head d_x")
    _check_param_grads(head, g, "head")


# --------------------------------------------------------------------------------------------- golden: models
def seeded_model(cls, g, depth):
    """Default (random) init reproduced by seed -- same nn modules created in the same order as the reference --
    and verified against the per-parameter checksums stored with the fixture."""
    torch.manual_seed(int(g["weight_seed"]))
    model = cls(depth=depth)
    ps = sorted(model.named_parameters(), key=lambda kv: kv[0])
    assert [n for n, _ in ps] == [str(n) for n in g["grad_names"]]
    got_sum = np.array([float(p.detach().double().sum()) for _, p in ps])
    got_abs = np.array([float(p.detach().double().abs().sum()) for _, p in ps])
    np.testing.assert_allclose(got_sum, g["param_sum"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(got_abs, g["param_abs"], rtol=1e-12)
    return model


def close_to_reference(got, ref32, ref64, what, tol=RTOL):
    """The bar for float outputs of a whole network: within `tol` x (largest magnitude) of the reference evaluated
    with float64 layer arithmetic, and no further from the reference's fp32 output than tol + the reference's own
    distance from that float64 evaluation (networks of this depth amplify fp32 rounding to 1e-5..1e-4)."""
    got = got.detach().cpu().numpy()
    scale = float(np.abs(ref64).max())
    e_ref = float(np.abs(ref32 - ref64).max())
    e_got = float(np.abs(got - ref64).max())
    e_dir = float(np.abs(got - ref32).max())
    msg = f"{what}: |hip-f64| {e_got / scale:.2e}, |ref32-f64| {e_ref / scale:.2e}, |hip-ref32| {e_dir / scale:.2e} (of max)"
    print(msg)
    assert e_got <= tol * scale, msg
    assert e_dir <= tol * scale + e_ref, msg


@pytest.mark.parametrize("depth", [5, 4, 6, 3, 2])
def test_model_golden(pn2, depth):
    """Whole forward + loss + backward of PointNet2 against the imported reference: same FPS starts (same RNG
    stream), per-point offsets within 1e-4 relative, losses to 1e-4.

    Parameter-gradient NORMS are compared with the reference's float64-layer-arithmetic values.  The bar per parameter is
    max(5e-4, 2 x fixture noise, 3 x this parameter's own |ref32 - ref64| / ref64) x norm + 1e-6 x (largest norm), where
    "fixture noise" is the largest relative distance between the reference's OWN fp32 norms and their float64 values
    (~5e-3 at depth 4): gradients of these chains are that ill-conditioned in fp32 for the reference as much as for us,
    so the effective bar is ~1 % of a norm, not 5e-4.  A norm is also a blunt instrument -- element-wise noise of 6e-3
    (measured for sa2.mlp_bns.2.bias at depth 3, tests/golden/model_d3_bn_grads.npz) moves a 128-element norm by
    anything up to a few percent depending on how it correlates with the gradient -- which is why the contested
    parameters are additionally compared element by element (tests/test_round2.py::test_depth3_bn_gradients_elementwise)
    and why the selected `g__*` tensors below are."""
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    g = gold(f"model_d{depth}.npz")
    model = seeded_model(PointNet2, g, depth).cuda().train()
    batch = {k: dev(g[k]) for k in ["coords", "feats", "masks_pad", "masks_off", "semantic_labels", "offset_labels"]}
    torch.manual_seed(int(g["torch_seed"]))
    loss, ld = model(batch, return_loss=True)
    (loss * 50).backward()
    assert abs(float(loss.detach()) - float(g["loss_f64"])) <= 1e-4 * abs(float(g["loss_f64"]))
    assert abs(float(loss.detach()) - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    assert abs(float(ld["offset_loss"].detach()) - float(g["offset_loss"])) <= 1e-4 * abs(float(g["offset_loss"]))
    assert abs(float(ld["semantic_loss"].detach()) - float(g["semantic_loss"])) <= 1e-4 * abs(float(g["semantic_loss"]))

    params = dict(model.named_parameters())
    gmax = float(g["grad_l2_f64"].max())       # absolute floor: gradients that all but vanish are rounding noise
    # noise level of this fixture: the largest relative distance between the reference's own fp32 gradient norms
    # and their float64-arithmetic values (non-vanishing, non-noise parameters).  It is ~5e-3 at depth 4: the
    # gradients of the deep chains are that ill-conditioned in fp32, for the reference as much as for us.
    noise = max(abs(l2 - l64) / l64 for n_, l2, l64 in zip(g["grad_names"], g["grad_l2"], g["grad_l2_f64"])
                if not helpers.is_pre_bn_bias(str(n_)) and l64 > 1e-3 * gmax)
    print(f"depth {depth}: reference fp32 gradient-norm noise level {noise:.2e}")
    for name, l2, l2_64 in zip(g["grad_names"], g["grad_l2"], g["grad_l2_f64"]):
        name = str(name)
        got = float(params[name].grad.double().norm())
        if helpers.is_pre_bn_bias(name):
            wn = float(params[name[:-4] + "weight"].grad.double().norm())
            assert got <= 1e-2 * wn, f"{name}: pre-BN bias gradient should vanish, got {got} (weight grad {wn})"
        else:
            bar = max(5e-4 * l2_64, 2 * noise * l2_64, 3 * abs(l2 - l2_64)) + 1e-6 * gmax
            assert abs(got - l2_64) <= bar, f"grad norm of {name}: hip {got}, ref32 {l2}, ref f64 {l2_64}"
    for key in g.files:
        if key.startswith("g__") and not helpers.is_pre_bn_bias(key[3:]):
            ref32, ref64 = g[key], g["g64__" + key[3:]]
            got = params[key[3:]].grad.detach().cpu().numpy()
            scale = float(np.abs(ref64).max())
            bar = max(GRAD_REL * scale, 3 * float(np.abs(ref32 - ref64).max()), 2 * noise * scale)
            assert float(np.abs(got - ref64).max()) <= bar, f"depth {depth} grad {key[3:]}"
    bufs = dict(model.named_buffers())
    for key in g.files:
        if key.startswith("buf__"):
            close(bufs[key[5:]], g[key], what=f"depth {depth} buffer {key[5:]}")

    torch.manual_seed(int(g["torch_seed"]))
    with torch.no_grad():
        out = model(batch, return_loss=False)
    close_to_reference(out["offset_predictions"], g["offset_predictions"], g["offset_predictions_f64"], f"depth {depth} offsets")
    close_to_reference(out["semantic_prediction_logits"], g["semantic_logits"], g["semantic_logits_f64"], f"depth {depth} logits")
    close_to_reference(out["backbone_feats"][:, :, :64], g["backbone_head"], g["backbone_head_f64"], f"depth {depth} backbone")


# ----------------------------------------------------------------------------------- full size (BASELINE config 2)
def test_full_size_tree_indices_vs_oracle(pn2, U, O):
    """262 144-point synthetic tree, depth-4 first level (S = 1024, r = 0.1, K = 32): FPS and ball query
    bit-exact against the C oracle, plus the size-independent properties (distinct samples, ascending rows,
    padding = first hit, every hit inside the ball by the reference's own expression)."""
    from pn2_amd import ops
    from pn2_amd.synthetic import gaussian_branch_tree
    xyz, _, _ = gaussian_branch_tree(262144, seed=0)
    coords = dev(xyz.T.copy()[None])                                  # [1,3,N] channel-first
    x = coords.permute(0, 2, 1)
    start = np.array([4242])
    idx, new_xyz = ops.furthest_point_sample(x, 1024, dev(start))
    want = O.farthest_point_sample(xyz[None], 1024, start)
    assert np.array_equal(idx.cpu().numpy(), want)
    assert len(np.unique(want)) == 1024
    bq = U.query_ball_point(0.1, 32, x, new_xyz).cpu().numpy()
    assert np.array_equal(bq, O.query_ball_point(0.1, 32, xyz[None], new_xyz.cpu().numpy()))
    row = bq[0]
    first_pad = (np.diff(row, axis=1) <= 0)
    # rows are ascending up to the padding, and the padding repeats the first entry
    for r_ in row[:: 37]:
        k = 1
        while k < 32 and r_[k] > r_[k - 1]:
            k += 1
        assert (r_[k:] == r_[0]).all()
    assert first_pad.shape == (1024, 31)
    # 3-NN at FP1 size against the oracle on a slice of the tree
    gi, gw, gd = ops.three_nn(x[:, :20000], new_xyz, want_dist=True)
    dist, oi = O.three_nn(xyz[None, :20000], new_xyz.cpu().numpy())
    assert np.array_equal(gi.cpu().numpy(), oi)
    assert np.array_equal(bits(gd), dist.view(np.uint32))


# ------------------------------------------------------------------------------ streaming mode (the whole-tree path)
class _FakeScaler:
    def scale(self, x):
        return x


@pytest.mark.parametrize("mode", ["fused", "sequential"])
def test_forward_hierarchical_streaming_golden(pn2, mode, monkeypatch):
    """PointNet2.forward_hierarchical_streaming (reference PointNet2.py:210-327) on a 6-raster tree in mini-batches of
    two: per-mini-batch backward with gradient accumulation, scatter-averaged predictions per original point id.
    The fixture was produced by the reference's own loop (tests/golden/make_golden.py:make_streaming).  Both execution
    modes: the whole-tree pass with per-mini-batch BatchNorm segments (default) and the mini-batch-by-mini-batch loop."""
    monkeypatch.setenv("PN2_STREAMING", mode)
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    from pn2_amd.synthetic import gaussian_branch_tree, rasterize
    g = gold("streaming_d5.npz")
    xyz, off, _ = gaussian_branch_tree(20000, seed=5)
    n = len(xyz)
    feats_all = (np.sin(0.61 * np.arange(n * 4, dtype=np.float64) + 7)).astype(np.float32).reshape(n, 4)
    rasters = [g[f"raster{i}"].astype(np.int64) for i in range(int(g["n_rasters"]))]
    assert all(np.array_equal(a, b) for a, b in zip(rasters, [r for r in rasterize(xyz, 2.0, 2.0) if len(r) >= 40][:6]))
    sem_lab = (np.arange(n) % 3 == 0).astype(np.int64)

    def mini_batches():
        for k in range(0, len(rasters), 2):
            group = rasters[k:k + 2]
            nmax = max(len(r) for r in group)
            coords = np.zeros((len(group), 3, nmax), np.float32)
            fts = np.zeros((len(group), 4, nmax), np.float32)
            mpad = np.zeros((len(group), nmax), bool)
            for i, r in enumerate(group):
                coords[i, :, :len(r)] = (xyz[r] - np.floor(xyz[r].min(axis=0))).T
                fts[i, :, :len(r)] = feats_all[r].T
                mpad[i, :len(r)] = True
            ids = np.concatenate(group)
            moff = (np.arange(len(ids)) % 5) != 2
            yield {"coords": dev(coords), "feats": dev(fts), "masks_pad": dev(mpad), "masks_off": dev(moff),
                   "point_ids": dev(ids)}

    torch.manual_seed(20250718)
    model = PointNet2(depth=5).cuda().train()
    batch = {"cloud_length": n, "mini_batches": mini_batches(), "semantic_labels": torch.from_numpy(sem_lab)[:, None],
             "offset_labels": torch.from_numpy(off)}
    torch.manual_seed(31)
    avg_loss, ld = model.forward_hierarchical_streaming(batch, return_loss=True, scaler=_FakeScaler())
    assert abs(avg_loss - float(g["avg_loss_f64"])) <= 2e-4 * abs(float(g["avg_loss_f64"]))
    assert abs(float(ld["offset_loss"].detach()) - float(g["offset_loss_f64"])) <= 2e-4 * abs(float(g["offset_loss_f64"]))
    assert abs(float(ld["semantic_loss"].detach()) - float(g["semantic_loss_f64"])) <= 2e-4 * abs(float(g["semantic_loss_f64"]))
    params = dict(model.named_parameters())
    gmax = float(g["grad_l2_f64"].max())
    # small rasters (40..200 points, 100 centroids) make several BatchNorm layers nearly degenerate: the reference's own
    # fp32 gradient norms sit up to ~4e-2 from their float64-arithmetic values on this fixture
    noise = max(abs(l2 - l64) / l64 for n_, l2, l64 in zip(g["grad_names"], g["grad_l2"], g["grad_l2_f64"])
                if not helpers.is_pre_bn_bias(str(n_)) and l64 > 1e-3 * gmax)
    print(f"streaming: reference fp32 gradient-norm noise level {noise:.2e}")
    for name, l2, l64 in zip(g["grad_names"], g["grad_l2"], g["grad_l2_f64"]):
        name = str(name)
        if helpers.is_pre_bn_bias(name):
            continue
        got = float(params[name].grad.double().norm())
        bar = max(5e-4 * l64, 2 * noise * l64, 3 * abs(l2 - l64)) + 1e-6 * gmax
        assert abs(got - l64) <= bar, f"accumulated grad norm of {name}: {got} vs {l64}"

    batch["mini_batches"] = mini_batches()
    torch.manual_seed(32)
    with torch.no_grad():
        pred = model.forward_hierarchical_streaming(batch, return_loss=False)
    close_to_reference(pred["offset_predictions"], g["pred_offsets"], g["pred_offsets_f64"], "streaming offsets", tol=2e-4)
    close_to_reference(pred["semantic_prediction_logits"], g["pred_logits"], g["pred_logits_f64"], "streaming logits", tol=2e-4)
    visited = np.zeros(n, bool)
    visited[np.concatenate(rasters)] = True
    assert float(pred["offset_predictions"][torch.from_numpy(~visited).cuda()].abs().max()) == 0.0


@pytest.mark.parametrize("mode", ["fused", "sequential"])
def test_forward_hierarchical_golden(pn2, mode, monkeypatch):
    """PointNet2.forward_hierarchical (the NON-streaming raster mode, reference PointNet2.py:329-394): predictions are
    accumulated WITH autograd history, averaged per point id, ONE loss over the whole cloud (unvisited points predict
    zero), one backward.  Fixture: the reference's own method on the streaming fixture's tree
    (tests/golden/make_golden_hier.py); both execution modes."""
    monkeypatch.setenv("PN2_STREAMING", mode)
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    from pn2_amd.synthetic import gaussian_branch_tree
    g, gs = gold("hierarchical_d5.npz"), gold("streaming_d5.npz")
    xyz, off, _ = gaussian_branch_tree(20000, seed=5)
    n = len(xyz)
    feats_all = (np.sin(0.61 * np.arange(n * 4, dtype=np.float64) + 7)).astype(np.float32).reshape(n, 4)
    rasters = [gs[f"raster{i}"].astype(np.int64) for i in range(int(gs["n_rasters"]))]
    mbs = []
    for k in range(0, len(rasters), 2):
        group = rasters[k:k + 2]
        nmax = max(len(r) for r in group)
        coords = np.zeros((len(group), 3, nmax), np.float32)
        fts = np.zeros((len(group), 4, nmax), np.float32)
        mpad = np.zeros((len(group), nmax), bool)
        for i, r in enumerate(group):
            coords[i, :, :len(r)] = (xyz[r] - np.floor(xyz[r].min(axis=0))).T
            fts[i, :, :len(r)] = feats_all[r].T
            mpad[i, :len(r)] = True
        ids = np.concatenate(group)
        mbs.append({"coords": dev(coords), "feats": dev(fts), "masks_pad": dev(mpad), "masks_off": dev((np.arange(len(ids)) % 5) != 2),
                    "point_ids": dev(ids)})
    torch.manual_seed(20250718)
    model = PointNet2(depth=5).cuda().train()
    batch = {"cloud_length": n, "mini_batches": iter(mbs), "semantic_labels": torch.from_numpy((np.arange(n) % 3 == 0).astype(np.int64))[:, None],
             "offset_labels": torch.from_numpy(off)}
    torch.manual_seed(41)
    loss, ld = model.forward_hierarchical(batch, return_loss=True)
    loss.backward()
    for key, val in (("loss", loss), ("offset_loss", ld["offset_loss"]), ("semantic_loss", ld["semantic_loss"])):
        assert abs(float(val.detach()) - float(g[key + "_f64"])) <= 2e-4 * abs(float(g[key + "_f64"])), key
    params = dict(model.named_parameters())
    gmax = float(g["grad_l2_f64"].max())
    noise = max(abs(l2 - l64) / l64 for n_, l2, l64 in zip(g["grad_names"], g["grad_l2"], g["grad_l2_f64"])
                if not helpers.is_pre_bn_bias(str(n_)) and l64 > 1e-3 * gmax)
    print(f"hierarchical ({mode}): reference fp32 gradient-norm noise level {noise:.2e}")
    for name, l2, l64 in zip(g["grad_names"], g["grad_l2"], g["grad_l2_f64"]):
        name = str(name)
        if helpers.is_pre_bn_bias(name):
            continue
        got = float(params[name].grad.double().norm())
        bar = max(5e-4 * l64, 2 * noise * l64, 3 * abs(l2 - l64)) + 1e-6 * gmax
        assert abs(got - l64) <= bar, f"grad norm of {name}: {got} vs {l64}"


def test_get_loss_matches_compacted_form(pn2):
    """The sync-free masked loss of PointNet2.get_loss equals point_wise_loss on the compacted rows (the reference's
    formulation, PointNet2.py:180-207) for values and for the gradients w.r.t. the predictions."""
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    from pn2_amd.Loss import point_wise_loss
    torch.manual_seed(3)
    B, N = 3, 500
    model = PointNet2(depth=5, loss_multiplier_semantic=0.7, loss_multiplier_offset=1.3)
    pad = torch.rand(B, N) > 0.3
    pad[2, :] = False                                           # a fully padded sample
    n_valid = int(pad.sum())
    moff = torch.rand(n_valid) > 0.4
    sem_lab = torch.randint(0, 2, (n_valid,))
    off_lab = torch.randn(int(moff.sum()), 3)
    outs = []
    for masked in (True, False):
        sem = torch.randn(B, 2, N, generator=torch.Generator().manual_seed(1)).cuda().requires_grad_(True)
        off = torch.randn(B, 3, N, generator=torch.Generator().manual_seed(2)).cuda().requires_grad_(True)
        mo = {"semantic_prediction_logits": sem, "offset_predictions": off}
        if masked:
            loss, ld = model.get_loss(mo, sem_lab.cuda(), off_lab.cuda(), moff.cuda(), pad.cuda())
        else:
            s_v, o_v = model._valid_rows(sem, off, pad.cuda(), moff.cuda())
            ls, lo = point_wise_loss(s_v.float(), o_v.float(), sem_lab.cuda(), off_lab.cuda())
            loss = ls * 0.7 + lo * 1.3
        loss.backward()
        outs.append((float(loss), sem.grad.cpu(), off.grad.cpu()))
    assert abs(outs[0][0] - outs[1][0]) <= 1e-6 * abs(outs[1][0])
    np.testing.assert_allclose(outs[0][1].numpy(), outs[1][1].numpy(), rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(outs[0][2].numpy(), outs[1][2].numpy(), rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("B,N,S,scale,shift", [(1, 70000, 1024, 1.0, (0, 0, 0)), (2, 20000, 300, 10.0, (5, -3, 20)),
                                                (1, 66000, 64, 1.0, (100, 100, 100)), (1, 40000, 128, 0.0, (1, 2, 3))])
def test_three_nn_large_clouds_vs_oracle(pn2, O, B, N, S, scale, shift):
    """Three-NN at full-resolution sizes, bit for bit: sampled points drawn FROM the cloud (zero distances),
    far-from-origin coordinates (large rounding error of the expanded distance), a degenerate cloud of identical
    points (every distance ties -> the three lowest indices)."""
    from pn2_amd import ops
    xyz1 = _cloud(B, N, 11, scale=scale, shift=shift)
    rng = np.random.default_rng(5)
    xyz2 = np.stack([xyz1[b][rng.choice(N, S, replace=False)] for b in range(B)])
    dist, idx = O.three_nn(xyz1, xyz2)
    gi, gw, gd = ops.three_nn(dev(xyz1), dev(xyz2), want_dist=True)
    assert np.array_equal(bits(gd), dist.view(np.uint32))
    assert np.array_equal(gi.cpu().numpy(), idx)
    assert np.array_equal(bits(gw), O.three_weights(dist).view(np.uint32))


def test_graphed_step_matches_eager(pn2):
    """graphs.GraphedTrainStep: forward + backward captured in one HIP graph replays with fresh FPS start indices drawn
    in the reference's RNG order -- the third pass after seeding equals the third eager pass bit for bit (forward) and
    to atomics' rounding (gradients)."""
    from pn2_amd.graphs import GraphedTrainStep
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    from pn2_amd.synthetic import gaussian_branch_tree
    n = 20000
    xyz, off, _ = gaussian_branch_tree(n, seed=7)
    batch = {"coords": torch.from_numpy(xyz.T.copy()[None]).cuda(), "feats": torch.ones(1, 4, n).cuda(),
             "masks_pad": torch.ones(1, n, dtype=torch.bool).cuda(), "masks_off": torch.ones(n, dtype=torch.bool).cuda(),
             "semantic_labels": torch.zeros(n, dtype=torch.long).cuda(), "offset_labels": torch.from_numpy(off).cuda()}
    torch.manual_seed(21)
    model = PointNet2(depth=4, loss_multiplier_semantic=0).cuda().train()
    state = {k: v.clone() for k, v in model.state_dict().items()}

    def step():
        for p in model.parameters():
            if p.grad is not None:
                p.grad.zero_()
        loss, _ = model(batch, return_loss=True)
        loss.backward()
        return loss

    torch.manual_seed(22)
    for _ in range(3):
        eager = float(step())
    eager_grads = [p.grad.clone() for p in model.parameters()]
    model.load_state_dict(state)                                   # BatchNorm running statistics back to the start
    torch.manual_seed(22)
    graphed = GraphedTrainStep(step, warmup=1)                     # pass 1 = warm-up, pass 2 = capture (draws, no run)
    loss = graphed()                                               # pass 3
    torch.cuda.synchronize()
    assert float(loss) == eager
    for a, b in zip(eager_grads, [p.grad for p in model.parameters()]):
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=1e-3, atol=2e-4 * float(a.abs().max()) + 1e-12)
