"""Round-2 parity cases.

CPU part: the C oracle against the round-2 fixtures of the imported reference (tests/golden/make_golden_r2.py).
GPU part (-m gpu, through the C ABI):
  * every FPS kernel variant and fallback against the C oracle (PN2_FPS_NO_XCD, PN2_FPS_NO_MULTI,
    PN2_FPS_FORCE_FALLBACK), and a hand-off timeout that must be LOUD (status word, -1 / NaN rows, no fault downstream);
  * BASELINE configs[0] (1 x 16 384 points, depth 5, forward) and configs[2] (per-GPU shape 8 x 65 536, depth 4): FPS, ball
    query and three-NN bit-exact against the oracle at those shapes, offsets against the torch-CPU restatement, one full
    step, and the two-shard data-parallel gradient against its sequential single-process emulation;
  * sample_and_group_all, the grid kNN against the reference's own add_features output, run-to-run bounds of
    the atomic backward kernels.
"""
import contextlib
import os

import numpy as np
import pytest
import torch

import helpers
from oracle import pn2_oracle as O

GOLD = helpers.GOLDEN


def gold(name):
    return np.load(os.path.join(GOLD, name))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def bits(t):
    return t.detach().cpu().numpy().view(np.uint32)


@pytest.fixture(scope="module")
def pn2():
    return helpers.load_pkg()


def _features_cloud(g):
    from pn2_amd.synthetic import gaussian_branch_tree
    n = int(g["n"])
    xyz, off, _ = gaussian_branch_tree(n, seed=int(g["seed"]))
    cloud = np.concatenate([xyz.astype(np.float64), off.astype(np.float64), np.zeros((n, 1))], axis=1)
    np.testing.assert_allclose([cloud.sum(), np.abs(cloud).sum()], g["cloud_checksum"], rtol=1e-13)
    return cloud


def _check_features(got, g, evals15):
    ref = g["appended"]
    np.testing.assert_array_equal(got[:, 11], ref[:, 4])                                    # density
    np.testing.assert_allclose(got[:, 10], ref[:, 3], rtol=1e-8, atol=1e-12)                # curvature
    np.testing.assert_allclose(got[:, 12], ref[:, 5], rtol=1e-12, atol=0)                   # height
    np.testing.assert_allclose(got[:, 14], ref[:, 7], rtol=1e-12, atol=1e-14)               # distance to the centre
    gap = np.minimum(evals15[:, 1] - evals15[:, 0], evals15[:, 2] - evals15[:, 1]) / evals15[:, 2]
    ok = gap > 1e-3
    assert ok.mean() > 0.9
    err = np.abs(np.abs(got[ok, 7:10]) - np.abs(ref[ok, 0:3])).max(1)
    assert (err <= 1e-9 / gap[ok]).all(), float((err * gap[ok]).max())


# ------------------------------------------------------------------------------------------------------ CPU: oracle
def test_oracle_features_at_grid_size(pn2):
    """The oracle on the 6000-point fixture (the size class where the product takes the cell-grid path)."""
    O.build()
    g = gold("features_grid.npz")
    cloud = _features_cloud(g)
    idx, _, _ = O.knn_radius(cloud[:, :3], 15, 0.1)
    np.testing.assert_array_equal(idx, g["nn15"].astype(np.int64))
    evals15, _ = O.cov_eig(cloud[:, :3], idx, 15)
    _check_features(O.add_features(cloud), g, evals15)


def test_a6_fixture_is_the_plain_concat():
    """sample_and_group_all is pure data movement: the fixture pins channel order [xyz, feats] and the zero centroid."""
    g = gold("a6.npz")
    xyz = g["coords"].transpose(0, 2, 1)
    want = np.concatenate([xyz, g["feats"].transpose(0, 2, 1)], axis=-1)[:, None]
    assert np.array_equal(g["group_new_points"], want)
    assert np.array_equal(g["group_new_points_nofeat"], xyz[:, None])
    assert not g["group_new_xyz"].any() and g["group_new_xyz"].shape == (2, 1, 3)


# ------------------------------------------------------------------------------------------- GPU: FPS variants, loudness
@contextlib.contextmanager
def env(**kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _cloud(B, N, seed, scale=1.0, shift=(0, 0, 0)):
    rng = np.random.default_rng(seed)
    return (rng.normal(size=(B, N, 3)) * scale + np.asarray(shift)).astype(np.float32)


FPS_MODES = {"no_xcd": {"PN2_FPS_NO_XCD": 1}, "no_multi": {"PN2_FPS_NO_MULTI": 1}, "fallback": {"PN2_FPS_FORCE_FALLBACK": 1},
             "fallback_no_multi": {"PN2_FPS_FORCE_FALLBACK": 1, "PN2_FPS_NO_MULTI": 1},
             # multi-pick rounds in index order (round 1's kernel), and the ordered kernel with one listed candidate per member
             "no_sort": {"PN2_FPS_NO_SORT": 1}, "one_listed": {"PN2_FPS_PER": 1}}


@pytest.mark.gpu
@pytest.mark.parametrize("mode", sorted(FPS_MODES))
@pytest.mark.parametrize("B,N,npoint", [(1, 20000, 64), (2, 70001, 50), (3, 20000, 512), (2, 70001, 300), (1, 150000, 256),
                                         (9, 16390, 200), (1, 262144, 300)])
def test_fps_variants_vs_oracle(pn2, mode, B, N, npoint):
    """Every multi-workgroup FPS path -- consecutive-block groups with write-through hand-off (NO_XCD), one sample per
    exchange (NO_MULTI), and the XCD kernels forced onto their placement-independent grouping (FORCE_FALLBACK: what a
    busy GPU gives them) -- returns the oracle's indices, zero-padded tails (exact ties) included."""
    from pn2_amd import ops
    O.build()
    xyz = _cloud(B, N, seed=N + B, scale=0.5, shift=(10.0, -20.0, 15.0))
    xyz[:, N - N // 5:] = 0.0
    start = np.random.default_rng(1).integers(0, N, size=B)
    want = O.farthest_point_sample(xyz, npoint, start)
    with env(**FPS_MODES[mode]):
        got, new_xyz = ops.furthest_point_sample(dev(xyz), npoint, dev(start))
        ops.check_status()
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(new_xyz.cpu().numpy(), O.index_points(xyz, want))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["identical", "flat", "two_places", "many_clouds", "lattice", "few_distinct", "huge_offset"])
def test_fps_ordered_kernel_edges(pn2, case):
    """The spatially ordered multi-pick kernel (cell counting sort, box-test skips, two listed candidates per member,
    sequential simulation on the list) on inputs that stress its shortcuts: degenerate boxes, exact distance ties inside and
    across wavefronts and members, distances that reach zero, more clouds than groups, coordinates far from the origin."""
    from pn2_amd import ops
    O.build()
    rng = np.random.default_rng(11)
    B, N, npoint = 2, 20000, 256
    if case == "identical":
        xyz = np.tile(np.array([[1.5, -2.0, 3.25]], np.float32), (B, N, 1))
    elif case == "flat":
        xyz = _cloud(B, N, seed=5)
        xyz[:, :, 2] = 7.0
    elif case == "two_places":
        xyz = np.where(rng.integers(0, 2, size=(B, N, 1)) == 0, np.float32([0.0, 0.0, 0.0]), np.float32([1.0, 2.0, 3.0])).astype(np.float32)
    elif case == "many_clouds":
        B, N, npoint = 11, 65536, 200                      # 32 members per cloud, 8 groups: some groups own two clouds
        xyz = _cloud(B, N, seed=6, scale=2.0)
    elif case == "lattice":
        xyz = (np.round(_cloud(B, N, seed=7, scale=1.5) * 2.0) / 2.0).astype(np.float32)   # half-metre lattice: ties everywhere
    elif case == "few_distinct":
        B, N, npoint = 1, 17000, 300
        places = _cloud(1, 100, seed=8)[0]
        xyz = places[rng.integers(0, 100, size=(B, N))]
    else:
        xyz = _cloud(B, N, seed=9, scale=0.3, shift=(4.0e5, -3.0e5, 250.0))   # UTM-like coordinates: coarse fp32 grid
    start = rng.integers(0, N, size=B)
    want = O.farthest_point_sample(xyz, npoint, start)
    got, new_xyz = ops.furthest_point_sample(dev(xyz), npoint, dev(start))
    ops.check_status()
    assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(new_xyz.cpu().numpy(), O.index_points(xyz, want))


@pytest.mark.gpu
@pytest.mark.parametrize("mode,N,npoint", [("default", 262144, 1024), ("default", 30000, 64), ("no_xcd", 50000, 64),
                                            ("fallback", 100000, 256)])
def test_fps_timeout_is_loud(pn2, mode, N, npoint):
    """A hand-off that never completes (provoked with a spin limit of 0 polls: the first miss counts as a timeout) must
    not return garbage silently: the launch ends at once, the status word says why, unfinished rows hold -1 / NaN,
    check_status() raises, and a gather fed with those rows neither faults nor goes unnoticed."""
    from pn2_amd import ops
    xyz = dev(_cloud(1, N, seed=3))
    start = dev(np.array([5]))
    ops.furthest_point_sample(xyz[:, :64], 4, start)       # loads the code object outside the timed bracket
    ops.check_status()
    t0 = torch.cuda.Event(enable_timing=True)
    t1 = torch.cuda.Event(enable_timing=True)
    with env(PN2_FPS_SPIN_LIMIT=0, **(FPS_MODES[mode] if mode != "default" else {})):
        t0.record()
        idx, new_xyz = ops.furthest_point_sample(xyz, npoint, start)
        t1.record()
    torch.cuda.synchronize()
    assert t0.elapsed_time(t1) < 200.0, "a dead launch must drain at once, not re-spin every round"
    word = int(ops.status_word(xyz.device).item())
    assert word & 3, "no FPS failure bit in the status word"
    i = idx.cpu().numpy()
    assert (i == -1).any() and i[0, 0] in (5, -1)
    assert np.isnan(new_xyz.cpu().numpy()[0][i[0] == -1]).all()
    with pytest.raises(RuntimeError, match="farthest_point_sample"):
        ops.check_status()
    # downstream: the -1 rows are not followed, and flagged
    pts = torch.ones(1, N, 4, device="cuda")
    out = ops.GatherPoints.apply(pts, idx)
    torch.cuda.synchronize()
    assert tuple(out.shape) == (1, npoint, 4)
    with pytest.raises(RuntimeError, match="index outside"):
        ops.check_status()
    # and the same call without the provocation is exact again
    O.build()
    idx2, _ = ops.furthest_point_sample(xyz, npoint, start)
    ops.check_status()
    assert np.array_equal(idx2.cpu().numpy(), O.farthest_point_sample(xyz.cpu().numpy(), npoint, np.array([5])))


@pytest.mark.gpu
def test_bad_index_is_flagged_not_followed(pn2):
    from pn2_amd import ops
    pts = dev(np.arange(2 * 10 * 3, dtype=np.float32).reshape(2, 10, 3))
    idx = dev(np.array([[0, 9, 10], [3, -1, 2]]))
    ops.check_status()
    out = ops.GatherPoints.apply(pts.clone().requires_grad_(True), idx)
    out.sum().backward()
    with pytest.raises(RuntimeError, match="index outside"):
        ops.check_status()
    assert np.array_equal(out.detach().cpu().numpy()[0, :2], pts.cpu().numpy()[0, [0, 9]])
    w = torch.full((1, 4, 3), 1 / 3, device="cuda")
    bad = dev(np.array([[[0, 1, 2], [0, 1, 7], [2, 1, 0], [-5, 0, 0]]]))
    y = ops.ThreeInterpolateConcat.apply(None, torch.ones(1, 3, 8, device="cuda"), bad, w)
    assert torch.isfinite(y).all()
    with pytest.raises(RuntimeError, match="index outside"):
        ops.check_status()
    ops.check_status()          # sticky bits are cleared by the raising check


# ------------------------------------------------------------------------------------ BASELINE configs[0]: 1 x 16 384, d5
def _f64_layers():
    """Run every Conv/BatchNorm module with float64 arithmetic (inputs/outputs stay fp32): the yardstick for fp32
    rounding amplification, same construction as tests/golden/make_golden.py:_f64_layers."""
    import torch.nn as nn
    import torch.nn.functional as F

    def conv_fwd(self, x):
        f = F.conv2d if isinstance(self, nn.Conv2d) else F.conv1d
        return f(x.double(), self.weight.double(), self.bias.double()).float()

    def bn_fwd(self, x):
        if self.training:
            self.num_batches_tracked.add_(1)
        return F.batch_norm(x.double(), self.running_mean.double(), self.running_var.double(), self.weight.double(),
                            self.bias.double(), self.training, self.momentum, self.eps).float()

    @contextlib.contextmanager
    def ctx():
        saved = (nn.Conv1d.forward, nn.Conv2d.forward, nn.BatchNorm1d.forward, nn.BatchNorm2d.forward)
        nn.Conv1d.forward = nn.Conv2d.forward = conv_fwd
        nn.BatchNorm1d.forward = nn.BatchNorm2d.forward = bn_fwd
        try:
            yield
        finally:
            nn.Conv1d.forward, nn.Conv2d.forward, nn.BatchNorm1d.forward, nn.BatchNorm2d.forward = saved
    return ctx()


def _tree_batch(n, seed, trees=1, centre=False):
    from pn2_amd.synthetic import gaussian_branch_tree
    clouds = [gaussian_branch_tree(n, seed=seed + t) for t in range(trees)]
    xyz = np.stack([c[0] for c in clouds])                                            # [B,N,3]
    if centre:
        xyz = xyz - np.floor(xyz.min(axis=1, keepdims=True))
    off = np.concatenate([c[1] for c in clouds])
    feats = np.sin(0.61 * np.arange(trees * 4 * n, dtype=np.float64) + 3).astype(np.float32).reshape(trees, 4, n)
    total = trees * n
    return xyz, {"coords": torch.from_numpy(np.ascontiguousarray(xyz.transpose(0, 2, 1))), "feats": torch.from_numpy(feats),
                 "masks_pad": torch.ones(trees, n, dtype=torch.bool), "masks_off": torch.ones(total, dtype=torch.bool),
                 "semantic_labels": torch.from_numpy((np.arange(total) % 5 == 0).astype(np.int64)),
                 "offset_labels": torch.from_numpy(off)}


def _geometry_vs_oracle(xyz, npoint, radius, K, seed):
    """FPS -> ball query -> three-NN at one level, all clouds, bit for bit against the C oracle."""
    from pn2_amd import ops
    from pn2_amd.PointNet2 import pointnet2_utils as U
    O.build()
    B, N, _ = xyz.shape
    start = np.random.default_rng(seed).integers(0, N, size=B)
    x = dev(np.ascontiguousarray(xyz.transpose(0, 2, 1))).permute(0, 2, 1)            # view of channel-first storage
    idx, new_xyz = ops.furthest_point_sample(x, npoint, dev(start))
    want = O.farthest_point_sample(xyz, npoint, start)
    assert np.array_equal(idx.cpu().numpy(), want), "FPS"
    assert np.array_equal(new_xyz.cpu().numpy(), O.index_points(xyz, want))
    nx = new_xyz.cpu().numpy()
    bq = U.query_ball_point(radius, K, x, new_xyz)
    assert np.array_equal(bq.cpu().numpy(), O.query_ball_point(radius, K, xyz, nx)), "ball query"
    gi, gw, gd = ops.three_nn(x, new_xyz, want_dist=True)
    dist, oi = O.three_nn(xyz, nx)
    assert np.array_equal(bits(gd), dist.view(np.uint32)), "three-NN distances"
    assert np.array_equal(gi.cpu().numpy(), oi), "three-NN indices"
    assert np.array_equal(bits(gw), O.three_weights(dist).view(np.uint32)), "three-NN weights"
    ops.check_status()


@pytest.mark.gpu
def test_config0_geometry_vs_oracle(pn2):
    """BASELINE configs[0] shape: one 16 384-point tree, the depth-5 first level (S = 100, r = 0.1, K = 32)."""
    xyz, _ = _tree_batch(16384, seed=0)
    _geometry_vs_oracle(xyz, 100, 0.1, 32, seed=2)


@pytest.mark.gpu
def test_config0_forward_vs_cpu_restatement(pn2):
    """configs[0]: PointNet2(depth=5) forward on one 16 384-point tree against the torch-CPU restatement of the reference
    (oracle/torch_port.py, pinned by the reference's fixtures): same seeded weights, same FPS start draws; offsets within
    1e-4 of the float64-layer-arithmetic evaluation and no further from the fp32 one than 1e-4 + its own distance."""
    from oracle import torch_port as P
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    _, batch = _tree_batch(16384, seed=0)
    torch.manual_seed(77)
    ref = P.PortPointNet2(depth=5).train()
    torch.manual_seed(77)
    model = PointNet2(depth=5).train()
    for (n1, p1), (n2, p2) in zip(sorted(ref.named_parameters()), sorted(model.named_parameters())):
        assert n1 == n2 and torch.equal(p1, p2)
    outs = {}
    P.STABLE_SORT = True                                 # three-NN ties: lower index first, the product's documented rule
    try:
        for tag in ("f32", "f64"):
            state = {k: v.clone() for k, v in ref.state_dict().items()}
            with (_f64_layers() if tag == "f64" else contextlib.nullcontext()), torch.no_grad():
                torch.manual_seed(5)
                sem, off = ref(batch["coords"], batch["feats"])
            ref.load_state_dict(state)
            outs[tag] = (sem.numpy(), off.numpy())
    finally:
        P.STABLE_SORT = False
    model = model.cuda()
    gb = {k: v.cuda() for k, v in batch.items()}
    torch.manual_seed(5)
    with torch.no_grad():
        out = model(gb, return_loss=False)
    from pn2_amd import ops
    ops.check_status()
    for key, i in (("semantic_prediction_logits", 0), ("offset_predictions", 1)):
        got = out[key].cpu().numpy()
        r32, r64 = outs["f32"][i], outs["f64"][i]
        scale = float(np.abs(r64).max())
        e_ref, e_got, e_dir = (float(np.abs(a - b).max()) for a, b in ((r32, r64), (got, r64), (got, r32)))
        print(f"config0 {key}: |hip-f64| {e_got / scale:.2e} |port32-f64| {e_ref / scale:.2e} |hip-port32| {e_dir / scale:.2e}")
        assert e_got <= 1e-4 * scale and e_dir <= 1e-4 * scale + e_ref


# ----------------------------------------------------------------------------- BASELINE configs[2]: 8 x 65 536 per GPU, d4
@pytest.mark.gpu
def test_config2_geometry_vs_oracle(pn2):
    """The per-GPU shape of configs[2]: 8 trees x 65 536 points, depth-4 first level (S = 1024: multi-pick FPS at 4 points
    per lane with 32-member groups on all 256 workgroups; ball query on its (8, nseg) plan), every cloud bit-exact."""
    xyz, _ = _tree_batch(65536, seed=20, trees=8)
    _geometry_vs_oracle(xyz, 1024, 0.1, 32, seed=4)


def _step_grads(model, batch, seed):
    for p in model.parameters():
        p.grad = None
    torch.manual_seed(seed)
    loss, _ = model(batch, return_loss=True)
    (loss * 50).backward()
    return loss.detach(), [p.grad.detach().clone() for p in model.parameters()]


@pytest.mark.gpu
def test_config2_full_step(pn2):
    """One full training step at 8 x 65 536, depth 4: finite loss and gradients, no kernel gave up, and the run is
    reproducible (same seeds -> same loss bit for bit; gradients to the atomics' rounding)."""
    from pn2_amd import ops
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    _, batch = _tree_batch(65536, seed=20, trees=8)
    gb = {k: v.cuda() for k, v in batch.items()}
    torch.manual_seed(1)
    model = PointNet2(depth=4, loss_multiplier_semantic=0).cuda().train()
    state = {k: v.clone() for k, v in model.state_dict().items()}
    loss1, g1 = _step_grads(model, gb, seed=9)
    ops.check_status()
    assert torch.isfinite(loss1) and all(torch.isfinite(g).all() for g in g1)
    model.load_state_dict(state)
    loss2, g2 = _step_grads(model, gb, seed=9)
    assert float(loss1) == float(loss2)
    for a, b in zip(g1, g2):
        assert float((a - b).abs().max()) <= 1e-4 * float(a.abs().max()) + 1e-12


def _shard(batch, lo, hi, n):
    return {"coords": batch["coords"][lo:hi], "feats": batch["feats"][lo:hi], "masks_pad": batch["masks_pad"][lo:hi],
            "masks_off": batch["masks_off"][lo * n:hi * n], "semantic_labels": batch["semantic_labels"][lo * n:hi * n],
            "offset_labels": batch["offset_labels"][lo * n:hi * n]}


def _dp_worker(rank, world, port, n, trees, depth, q):
    """One data-parallel rank (both ranks share the box's single GPU; gloo carries the collective, like
    PN2_DIST_BACKEND=gloo in bench.py): deliberately DIFFERENT initial weights per rank -- FlatGradAllReduce must
    broadcast rank 0's."""
    import torch.distributed as dist
    helpers.load_pkg()
    from pn2_amd import ops, parallel
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    _, batch = _tree_batch(n, seed=20, trees=trees)
    per = trees // world
    gb = {k: v.cuda() for k, v in _shard(batch, rank * per, (rank + 1) * per, n).items()}
    torch.manual_seed(100 + rank)
    model = PointNet2(depth=depth, loss_multiplier_semantic=0).cuda().train()
    sync = parallel.FlatGradAllReduce(model)
    sync.zero()
    torch.manual_seed(1000 + rank)                       # FPS start draws: a per-rank stream
    loss, _ = model(gb, return_loss=True)
    (loss * 50).backward()
    sync.allreduce()
    ops.check_status()
    # numpy, not tensors: torch shares tensor storage through file descriptors that die with this process
    if rank == 0:
        q.put(([p.detach().cpu().numpy() for p in model.parameters()], sync.packed().cpu().numpy(), float(loss)))
    else:
        q.put((None, sync.packed().cpu().numpy(), float(loss)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("n,trees,depth", [(65536, 8, 4), (3000, 4, 5)])
def test_two_rank_gradient_equals_sequential_emulation(pn2, n, trees, depth):
    """Data parallelism on the REAL model (not a toy): two ranks, each with half of the trees (configs[2]'s per-GPU batch
    cut in two), one flat all-reduce -- the averaged gradient equals the single-process emulation that runs the two
    shards one after the other with the same weights and the same per-shard FPS start draws and averages.  Both replicas
    end with rank 0's weights although they were initialised differently."""
    import torch.multiprocessing as mp
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, n, trees, depth, q)) for r in range(2)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=240) for _ in procs]       # (well inside the GPU box's 7-minute silence limit)
    except Exception:
        for p in procs:
            p.kill()
        raise
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    params0 = [torch.from_numpy(a) for a in next(r[0] for r in res if r[0] is not None)]
    flats = [torch.from_numpy(r[1]) for r in res]
    assert torch.equal(flats[0], flats[1]), "ranks disagree after the all-reduce"
    # sequential emulation with rank 0's initial weights
    _, batch = _tree_batch(n, seed=20, trees=trees)
    torch.manual_seed(100)
    model = PointNet2(depth=depth, loss_multiplier_semantic=0).cuda().train()
    for p, q0 in zip(model.parameters(), params0):
        assert torch.equal(p.detach().cpu(), q0), "rank 0's weights are the seed-100 initialisation"
    state = {k: v.clone() for k, v in model.state_dict().items()}
    per = trees // 2
    total = None
    for r in range(2):
        model.load_state_dict(state)                     # per-replica BatchNorm buffers: each rank starts from the same
        gb = {k: v.cuda() for k, v in _shard(batch, r * per, (r + 1) * per, n).items()}
        _, g = _step_grads(model, gb, seed=1000 + r)
        flat = torch.cat([t.reshape(-1) for t in g])
        total = flat if total is None else total + flat
    want = (total / 2).cpu()
    err = float((flats[0] - want).abs().max())
    assert err <= 2e-4 * float(want.abs().max()), f"two-rank gradient differs from the sequential emulation: {err:.3e}"


@pytest.mark.gpu
def test_depth3_bn_gradients_elementwise(pn2):
    """The parameter whose gradient NORM sat at 3x the reference's own fp32 noise in round 1 (sa2.mlp_bns.2.bias, depth 3),
    and its neighbours, element by element: the HIP gradient is no further from the float64-layer-arithmetic reference
    than 3x the reference's own fp32 gradient is (vector 2-norm of the difference), i.e. the deviation is rounding noise
    of an ill-conditioned sum, not a reduction defect."""
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    g = gold("model_d3.npz")
    e = gold("model_d3_bn_grads.npz")
    torch.manual_seed(int(g["weight_seed"]))
    model = PointNet2(depth=3).cuda().train()
    batch = {k: dev(g[k]) for k in ["coords", "feats", "masks_pad", "masks_off", "semantic_labels", "offset_labels"]}
    torch.manual_seed(int(g["torch_seed"]))
    loss, _ = model(batch, return_loss=True)
    (loss * 50).backward()
    params = dict(model.named_parameters())
    worst = 0.0
    for key in sorted(e.files):
        if not key.startswith("g__"):
            continue
        name = key[3:]
        r32, r64 = e[key].astype(np.float64), e["g64__" + name].astype(np.float64)
        got = params[name].grad.cpu().numpy().astype(np.float64)
        scale = np.linalg.norm(r64)
        if scale < 1e-4:                                   # sa3's last beta: analytically ~0 (max-pool over a BN output)
            assert np.linalg.norm(got) <= 1e-4
            continue
        d_ref, d_hip = np.linalg.norm(r32 - r64), np.linalg.norm(got - r64)
        print(f"{name}: |ref32-f64| {d_ref / scale:.2e}  |hip-f64| {d_hip / scale:.2e}  |hip-ref32| "
              f"{np.linalg.norm(got - r32) / scale:.2e} (of the f64 norm)")
        worst = max(worst, d_hip / max(d_ref, 1e-30))
        assert d_hip <= max(3 * d_ref, 2e-4 * scale), name
    print(f"worst |hip-f64| / |ref32-f64| = {worst:.2f}")


# ------------------------------------------------------------------------------------------- a6, grid features, atomics
@pytest.mark.gpu
def test_sample_and_group_all_golden(pn2):
    from pn2_amd.PointNet2 import pointnet2_utils as U
    from pn2_amd.PointNet2.blocks import PointNetSetAbstraction
    g = gold("a6.npz")
    c = dev(g["coords"])
    feats = dev(g["feats"]).requires_grad_(True)
    nx, npts = U.sample_and_group_all(c.permute(0, 2, 1), feats.detach().permute(0, 2, 1))
    assert np.array_equal(nx.cpu().numpy(), g["group_new_xyz"]) and np.array_equal(npts.cpu().numpy(), g["group_new_points"])
    _, npts0 = U.sample_and_group_all(c.permute(0, 2, 1), None)
    assert np.array_equal(npts0.cpu().numpy(), g["group_new_points_nofeat"])
    sa = PointNetSetAbstraction(None, None, None, 3 + 5, [16, 32], True)
    helpers.closed_form_init(sa)
    sa = sa.cuda().train()
    new_xyz, new_points = sa(c, feats)
    assert np.array_equal(new_xyz.cpu().numpy(), g["new_xyz"])
    want = g["new_points"]
    assert float(np.abs(new_points.detach().cpu().numpy() - want).max()) <= 1e-4 * float(np.abs(want).max())
    (new_points * dev(g["G"])).sum().backward()
    assert float((feats.grad.cpu() - torch.from_numpy(g["d_feats"])).abs().max()) <= 2e-4 * float(np.abs(g["d_feats"]).max())
    for n, p in sa.named_parameters():
        if not helpers.is_pre_bn_bias(n):
            ref = g["g__" + n]
            assert float(np.abs(p.grad.cpu().numpy() - ref).max()) <= 2e-4 * float(np.abs(ref).max()), n
    for n, b in sa.named_buffers():
        if "num_batches" not in n:
            np.testing.assert_allclose(b.cpu().numpy(), g["buf__" + n], rtol=1e-4, atol=1e-6)


@pytest.mark.gpu
def test_grid_knn_features_vs_reference(pn2):
    """add_features on 6000 points -- above Features.GRID_MIN_POINTS, so the hashed cell grid (csrc/knn_grid.hip) is the
    path that runs -- against the imported reference's own output and against the oracle's neighbour lists."""
    from pn2_amd import Features as F
    O.build()
    g = gold("features_grid.npz")
    cloud = _features_cloud(g)
    assert len(cloud) >= F.GRID_MIN_POINTS
    idx, cnt, pts = F.neighbourhoods(cloud[:, :3], 15, 0.1)
    np.testing.assert_array_equal(idx.cpu().numpy(), g["nn15"].astype(np.int64))
    oi, _, oc = O.knn_radius(cloud[:, :3], 15, 0.1)
    np.testing.assert_array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_array_equal(cnt.cpu().numpy(), oc)
    evals15, _ = O.cov_eig(cloud[:, :3], oi, 15)
    got = F.add_features(cloud.copy())
    assert np.array_equal(got[:, :7], cloud)
    _check_features(got, g, evals15)


@pytest.mark.gpu
def test_atomic_backward_run_to_run_bound(pn2):
    """The gather / group / interpolation backward kernels add with float atomics, so their sums depend on arrival
    order (the reference's CPU index_put_ is deterministic).  Bound: repeated runs agree to a few ulps of the
    float64 sum of magnitudes, and each run is within the same bound of the float64 result."""
    from pn2_amd import ops
    rng = np.random.default_rng(0)
    B, N, S, K, D = 2, 4000, 256, 32, 64
    idx = dev(rng.integers(0, 40, size=(B, S, K)))            # 40 hot rows: ~400 colliding atomics per address
    g = dev(rng.normal(size=(B, S, K, D)).astype(np.float32))
    exact = np.zeros((B, N, D))
    np.add.at(exact, (np.arange(B)[:, None, None], idx.cpu().numpy()), g.cpu().numpy().astype(np.float64))
    mags = np.zeros((B, N, D))
    np.add.at(mags, (np.arange(B)[:, None, None], idx.cpu().numpy()), np.abs(g.cpu().numpy().astype(np.float64)))
    runs = []
    for _ in range(5):
        pts = torch.zeros(B, N, D, device="cuda", requires_grad=True)
        ops.GatherPoints.apply(pts, idx).backward(g)
        runs.append(pts.grad.cpu().numpy().astype(np.float64))
    bound = 64 * np.finfo(np.float32).eps * mags.max()
    for r in runs:
        assert np.abs(r - exact).max() <= bound
        assert np.abs(r - runs[0]).max() <= bound
    # interpolation backward, bucketed path (large) and atomic path (small): both bounded the same way
    for n_dense, d2 in [(70000, 64), (500, 8)]:
        s = 50
        nn_idx = dev(rng.integers(0, s, size=(1, n_dense, 3)))
        w = dev(rng.uniform(0.1, 1, size=(1, n_dense, 3)).astype(np.float32))
        go = dev(rng.normal(size=(1, n_dense, d2)).astype(np.float32))
        want = O.three_interpolate_grad(go.cpu().numpy(), nn_idx.cpu().numpy(), w.cpu().numpy(), s).astype(np.float64)
        scale = float(np.abs(want).max())
        outs = []
        for _ in range(3):
            p2 = torch.zeros(1, s, d2, device="cuda", requires_grad=True)
            ops.ThreeInterpolateConcat.apply(None, p2, nn_idx, w).backward(go)
            outs.append(p2.grad.cpu().numpy().astype(np.float64))
        for o in outs:
            assert np.abs(o - want).max() <= 2e-4 * scale and np.abs(o - outs[0]).max() <= 2e-4 * scale
    ops.check_status()


@pytest.mark.gpu
def test_three_nn_in_cell_order_matches_oracle(pn2):
    """three_nn handed the cloud an ordered FPS call just sampled walks it in that call's cell order (a permutation left in
    the FPS workspace): neighbours, distances and weights must be what the index-order scan and the oracle give, ties
    included (a fifth of the cloud is zero padding)."""
    from pn2_amd import ops
    O.build()
    xyz = _cloud(2, 20000, seed=21, scale=0.7)
    xyz[:, 16000:] = 0.0
    dense = dev(xyz)
    idx, new_xyz = ops.furthest_point_sample(dense, 256, dev(np.array([3, 4])))
    ops.check_status()
    order = ops._cloud_memo.lookup(dense)[0]
    assert order is not None and tuple(order.shape) == (2, 20000)
    assert np.array_equal(np.sort(order.cpu().numpy(), axis=1), np.tile(np.arange(20000, dtype=np.int32), (2, 1)))
    gi, gw, gd = ops.three_nn(dense, new_xyz, want_dist=True)
    with env(PN2_TNN_NO_ORDER=1):
        hi, hw, hd = ops.three_nn(dense, new_xyz, want_dist=True)
    dist, want = O.three_nn(xyz, new_xyz.cpu().numpy())
    assert np.array_equal(gi.cpu().numpy(), want) and np.array_equal(hi.cpu().numpy(), want)
    assert torch.equal(gd, hd) and torch.equal(gw, hw)
    assert np.array_equal(gd.cpu().numpy().view(np.uint32), dist.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["members", "strangers", "dense", "far_origin", "tiny_radius", "big_k", "huge_k"])
def test_ball_query_by_cells_matches_oracle(pn2, case):
    """query_ball_point on a cloud an ordered FPS call just sampled searches the cells that call left in its workspace.
    Same rows as the oracle's index-order scan: queries that are cloud members and queries that are not (empty balls take
    the arg-min), balls with far more hits than nsample (list cuts + index limit), coordinates far from the origin (the
    reach of the expanded distance's rounding covers many cells: index-order walk), a radius below the point spacing,
    nsample > 64 and nsample too large for the list."""
    from pn2_amd import ops
    O.build()
    B, N, S, r, K = 2, 30000, 300, 0.15, 32
    xyz = _cloud(B, N, seed=31, scale=1.0)
    if case == "dense":
        xyz = _cloud(B, N, seed=32, scale=0.05)              # thousands of points inside every ball
    elif case == "far_origin":
        xyz = _cloud(B, N, seed=33, scale=1.0, shift=(5000.0, -7000.0, 300.0))
    elif case == "tiny_radius":
        r = 0.004
    elif case == "big_k":
        K, r = 100, 0.3
    elif case == "huge_k":
        K, r = 300, 0.5
    dense = dev(xyz)
    idx, new_xyz = ops.furthest_point_sample(dense, S, dev(np.array([1, 2])))
    ops.check_status()
    assert ops._cloud_memo.lookup(dense)[1] is not None
    q = new_xyz if case != "strangers" else new_xyz + 0.37
    want = O.query_ball_point(r, K, xyz, q.cpu().numpy())
    got = ops.ball_query(r, K, dense, q)
    with env(PN2_BQ_NO_CELLS=1):
        scan = ops.ball_query(r, K, dense, q)
    assert np.array_equal(scan.cpu().numpy(), want)
    assert np.array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
def test_deferred_weight_gradients_and_head_pair(pn2):
    """Weight-gradient slab reductions deferred to the engine callback at the end of the backward pass (default) against
    reductions per chain call (PN2_NO_DEFER_WGRAD): not a bit differs on a stack of chains (same slabs, same summation
    order), and nothing is left pending when backward() returns.  On the real model (whose grouping backward uses float
    atomics, so runs differ in the last bits anyway) the three ways -- deferred, per chain, heads as separate autograd
    nodes (PN2_NO_CHAIN_PAIR) -- agree to 2e-3 of the largest gradient."""
    import torch.nn as nn
    from pn2_amd import mlp, _hip
    from pn2_amd.PointNet2.PointNet2 import PointNet2
    from pn2_amd.synthetic import gaussian_branch_tree

    def stack():
        torch.manual_seed(0)
        mk = lambda ci, co: (nn.Conv1d(ci, co, 1).cuda(), nn.BatchNorm1d(co).cuda().train(), True)
        return [[mk(35, 64), mk(64, 64)], [mk(64, 128), mk(128, 32)], [mk(32, 32)]]

    x = torch.randn(5000, 35, device="cuda")
    got = {}
    for mode, e in (("deferred", {}), ("per_chain", {"PN2_NO_DEFER_WGRAD": 1})):
        chains = stack()
        with env(**e):
            y = x
            for layers in chains:
                y = mlp.chain_rows(y, layers)
            (y * y).sum().backward()
            assert not mlp._DeferredWgrad._passes, "weight-gradient reductions left pending after backward()"
        got[mode] = [p.grad.clone() for layers in chains for conv, bn, _ in layers for p in (conv.weight, bn.weight, bn.bias)]
    assert all(torch.equal(a, b) for a, b in zip(got["deferred"], got["per_chain"]))
    assert all(float(g.abs().max()) > 0 for g in got["deferred"])

    xyz, off, _ = gaussian_branch_tree(20000, seed=3)
    n = len(xyz)
    batch = {"coords": dev(xyz.T[None].copy()), "feats": dev(np.ones((1, 4, n), np.float32)),
             "semantic_labels": torch.zeros(n, dtype=torch.long, device="cuda"), "offset_labels": dev(off),
             "masks_off": torch.ones(n, dtype=torch.bool, device="cuda"), "masks_pad": torch.ones(1, n, dtype=torch.bool, device="cuda")}
    grads = {}
    for mode, e in (("deferred", {}), ("per_chain", {"PN2_NO_DEFER_WGRAD": 1}), ("separate_heads", {"PN2_NO_CHAIN_PAIR": 1})):
        torch.manual_seed(0)
        model = PointNet2(depth=4).cuda().train()
        torch.manual_seed(1)
        with env(**e):
            loss, _ = model(batch, return_loss=True)
            loss.backward()
            assert not mlp._DeferredWgrad._passes
        grads[mode] = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    gmax = max(float(g.abs().max()) for g in grads["deferred"].values())
    for other in ("per_chain", "separate_heads"):
        for k, g in grads["deferred"].items():
            assert float((g - grads[other][k]).abs().max()) <= 2e-3 * gmax, (other, k)


@pytest.mark.gpu
@pytest.mark.parametrize("segments", [False, True])
def test_linked_chains_equal_materialised_rows(pn2, segments):
    """A chain that leaves its last BatchNorm + ReLU to the two head chains reading it (mlp.LazyRows: no activated rows in
    memory, the heads' second dgrad hands the BatchNorm-backward sums back) against the same stack with the rows
    materialised (PN2_NO_LAZY_ROWS): outputs bit-identical (the staging transform is the apply kernel's expression),
    gradients equal up to the summation order of the handed-over sums."""
    import torch.nn as nn
    from pn2_amd import mlp

    def build():
        torch.manual_seed(0)
        mk = lambda ci, co, bn=True: (nn.Conv1d(ci, co, 1).cuda(), nn.BatchNorm1d(co).cuda().train() if bn else None, bn)
        return [mk(35, 64), mk(64, 128)], [mk(128, 128), mk(128, 2, False)], [mk(128, 128), mk(128, 3, False)]

    rows = 6000
    seg = [0, 1500, 1500 + 2600, rows] if segments else None
    x0 = torch.randn(rows, 35, device="cuda")
    res = {}
    for mode, e in (("linked", {}), ("materialised", {"PN2_NO_LAZY_ROWS": 1})):
        trunk, ha, hb = build()
        x = x0.clone().requires_grad_(True)
        with env(**e):
            feats = mlp.chain_rows(x, trunk, seg_off=seg, lazy_out=True)
            assert isinstance(feats, mlp.LazyRows) == (mode == "linked")
            a, b = mlp.chain_pair_rows(feats, ha, hb, seg_off=seg)
            ((a * a).sum() + (b * torch.arange(3, device="cuda")).sum()).backward()
        params = [p for chain in (trunk, ha, hb) for conv, bn, _ in chain for p in ([conv.weight] + ([bn.weight, bn.bias] if bn else [conv.bias]))]
        res[mode] = (a.detach(), b.detach(), x.grad.clone(), [p.grad.clone() for p in params])
    assert torch.equal(res["linked"][0], res["materialised"][0]) and torch.equal(res["linked"][1], res["materialised"][1])
    for g, h in zip([res["linked"][2]] + res["linked"][3], [res["materialised"][2]] + res["materialised"][3]):
        assert float((g - h).norm()) <= 2e-5 * float(h.norm()) + 1e-7, (float((g - h).norm()), float(h.norm()))


@pytest.mark.gpu
def test_cell_structure_is_never_used_for_another_cloud(pn2):
    """The cell structure an ordered FPS call leaves behind serves ball_query / three_nn only for the very tensor it sampled:
    not for a cloud of the same shape elsewhere, not after an in-place write to the cloud, not for another cloud that the
    allocator would otherwise place at the same address (the memo pins the storage)."""
    from pn2_amd import ops
    O.build()
    a = dev(_cloud(1, 20000, seed=41))
    idx, new_xyz = ops.furthest_point_sample(a, 256, dev(np.array([0])))
    assert ops._cloud_memo.lookup(a)[1] is not None
    assert ops._cloud_memo.lookup(a.clone())[1] is None                      # same values, other storage
    ptr = a.data_ptr()
    a.mul_(1.5)                                                               # same storage, other values
    assert ops._cloud_memo.lookup(a)[1] is None
    want = O.query_ball_point(0.2, 16, a.cpu().numpy(), new_xyz.cpu().numpy())
    assert np.array_equal(ops.ball_query(0.2, 16, a, new_xyz).cpu().numpy(), want)
    del a
    torch.cuda.empty_cache()
    b = dev(_cloud(1, 20000, seed=42))                                        # would reuse a's address if a's memory were free
    assert b.data_ptr() != ptr or ops._cloud_memo.lookup(b)[1] is None
    want = O.query_ball_point(0.2, 16, b.cpu().numpy(), new_xyz.cpu().numpy())
    assert np.array_equal(ops.ball_query(0.2, 16, b, new_xyz).cpu().numpy(), want)


@pytest.mark.gpu
def test_mask_ranks_equal_the_torch_expressions(pn2):
    """get_loss's mask arithmetic (two prefix sums + a gather, PointNet2.py:188-196) as one launch: integer work, so equal
    -- for ragged sizes, unaligned views, all-true / all-false masks, non-0/1 true bytes, and a masks_off that is shorter
    than the number of real rows (the torch expression clamps)."""
    from pn2_amd.Loss import mask_ranks
    g = torch.Generator(device="cpu").manual_seed(5)

    def check(pad, moff):
        cum_pad = torch.cumsum(pad, 0)
        off_mask = pad & moff.index_select(0, (cum_pad - 1).clamp(0, moff.numel() - 1))
        cum_off = torch.cumsum(off_mask, 0)
        a, b, c = mask_ranks(pad, moff)
        assert a.dtype == torch.int64 and b.dtype == torch.bool and c.dtype == torch.int64
        assert torch.equal(a, cum_pad) and torch.equal(b, off_mask) and torch.equal(c, cum_off)

    for R, p_pad, p_off in [(262144, 0.8, 0.5), (1, 1.0, 1.0), (1023, 0.5, 0.5), (1025, 0.5, 0.9), (70001, 0.3, 0.1),
                            (40000, 1.0, 1.0), (40000, 0.0, 0.5), (300000, 0.97, 0.0)]:
        pad = (torch.rand(R, generator=g) < p_pad).cuda()
        n_real = max(int(pad.sum().item()), 1)
        moff = (torch.rand(n_real, generator=g) < p_off).cuda()
        check(pad, moff)
        if n_real > 10:
            check(pad, moff[: n_real // 2])          # too short: clamped to its last entry
            check(pad, moff[: n_real // 2 - 1])
    # views at odd byte offsets, and "true" stored as another non-zero byte
    base = (torch.rand(100000, generator=g) < 0.6).cuda()
    mo = (torch.rand(100000, generator=g) < 0.4).cuda()
    check(base[3:90001], mo[5:])
    raw = (base.view(torch.uint8) * 255).view(torch.bool)
    a, _, _ = mask_ranks(raw, mo)
    assert torch.equal(a, torch.cumsum(base, 0))
    with pytest.raises(RuntimeError):
        mask_ranks(base, mo[:0])


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,S", [(1, 1024, 256), (1, 256, 64), (8, 2000, 33), (3, 77, 2048), (2, 8192, 1000), (16, 1000, 32)])
def test_three_nn_small_clouds_threads_per_point(pn2, monkeypatch, B, N, S):
    """Small clouds split every dense point's scan over several threads and merge the lists in (distance, index) order:
    neighbours, distances and weights must be what the one-thread scan (PN2_TNN_NO_SMALL=1) and the oracle give -- with
    ties: duplicated samples, samples drawn from the cloud, a block of identical points."""
    from pn2_amd import ops
    O.build()
    rng = np.random.default_rng(B * 1000 + N + S)
    xyz = _cloud(B, N, seed=N + S, scale=0.8, shift=(2.0, -1.0, 5.0))
    xyz[:, N // 2:N // 2 + 9] = xyz[:, :1]                          # identical dense points
    pick = rng.integers(0, N, size=S)
    new_xyz = xyz[:, pick].copy()
    new_xyz[:, S // 3:S // 3 + 5] = new_xyz[:, :1]                  # identical samples: equal distances, lowest index first
    dist, want = O.three_nn(xyz, new_xyz)
    x, q = torch.as_tensor(xyz, device="cuda"), torch.as_tensor(new_xyz, device="cuda")
    gi, gw, gd = ops.three_nn(x, q, want_dist=True)
    monkeypatch.setenv("PN2_TNN_NO_SMALL", "1")
    hi, hw, hd = ops.three_nn(x, q, want_dist=True)
    assert torch.equal(gi, hi) and torch.equal(gd.view(torch.int32), hd.view(torch.int32))
    assert torch.equal(gw.view(torch.int32), hw.view(torch.int32))
    assert np.array_equal(gi.cpu().numpy(), want)
    assert np.array_equal(gd.cpu().numpy().view(np.uint32), dist.view(np.uint32))


@pytest.mark.gpu
def test_deferral_never_loses_a_returned_weight_gradient(pn2):
    """Round-2 advisor finding: with mlp.FUSED_GRAD_ACCUMULATION = False (the documented switch for torch.autograd.grad-style
    use) the chain backward RETURNS its weight gradients; a deferred slab reduction would then write into the returned tensor
    after AccumulateGrad / the engine's input-buffer sum had already read it (zeros).  Deferral now happens only when every
    target is the parameter's own .grad buffer.  Two accumulated backward() calls, a weight shared by two chains, and
    torch.autograd.grad, each against PN2_NO_DEFER_WGRAD=1 -- bit for bit -- and against fused accumulation to rounding."""
    import torch.nn as nn
    from pn2_amd import mlp

    def build():
        torch.manual_seed(0)
        mk = lambda ci, co: (nn.Conv1d(ci, co, 1).cuda(), nn.BatchNorm1d(co).cuda().train(), True)
        return [mk(40, 64), mk(64, 64)], [mk(64, 32)]

    xs = [torch.randn(6000, 40, device="cuda"), torch.randn(4000, 40, device="cuda")]

    def params(chains):
        return [p for layers in chains for conv, bn, _ in layers for p in (conv.weight, bn.weight, bn.bias)]

    def run(fused, e):
        a, b = build()
        old = mlp.FUSED_GRAD_ACCUMULATION
        mlp.FUSED_GRAD_ACCUMULATION = fused
        try:
            with env(**e):
                for x in xs:                                  # two backward() calls accumulate; `a` is used by two chains
                    h = mlp.chain_rows(x, a)
                    loss = (mlp.chain_rows(h, b) ** 2).sum() + (mlp.chain_rows(x[:3000], a) ** 2).sum()
                    loss.backward()
                    assert not mlp._DeferredWgrad._passes
                acc = [p.grad.clone() for p in params((a, b))]
                ag = []
                if not fused:                                 # (fused accumulation returns nothing to autograd.grad by design)
                    h = mlp.chain_rows(xs[0], a)
                    ag = torch.autograd.grad((mlp.chain_rows(h, b) ** 2).sum(), params((a, b)))
        finally:
            mlp.FUSED_GRAD_ACCUMULATION = old
        return acc, list(ag)

    ret_defer, ag_defer = run(False, {})
    ret_plain, ag_plain = run(False, {"PN2_NO_DEFER_WGRAD": 1})
    fused, _ = run(True, {})
    for a, b in zip(ret_defer + ag_defer, ret_plain + ag_plain):
        assert torch.equal(a, b)
    assert all(float(g.abs().max()) > 0 for g in ret_defer + ag_defer)
    for a, b in zip(ret_defer, fused):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.gpu
def test_nested_backward_keeps_the_outer_pass_reductions(pn2):
    """A backward pass started INSIDE another one (here: from a tensor hook) has its own graph-task id: its deferred
    reductions are its own record and its own callback, the outer pass's pending reductions survive it (round-2 advisor:
    a second task id used to be taken for a dead pass and the outer list dropped)."""
    import torch.nn as nn
    from pn2_amd import mlp
    torch.manual_seed(0)
    outer = [(nn.Conv1d(48, 64, 1).cuda(), nn.BatchNorm1d(64).cuda().train(), True), (nn.Conv1d(64, 64, 1).cuda(), nn.BatchNorm1d(64).cuda().train(), True)]
    inner = [(nn.Conv1d(16, 32, 1).cuda(), nn.BatchNorm1d(32).cuda().train(), True)]
    x, z = torch.randn(5000, 48, device="cuda"), torch.randn(3000, 16, device="cuda")

    def run(nested):
        for layers in (outer, inner):
            for conv, bn, _ in layers:
                conv.weight.grad = bn.weight.grad = bn.bias.grad = None
        h = mlp.chain_rows(x, outer[:1])
        if nested:
            def hook(g):
                with torch.enable_grad():
                    (mlp.chain_rows(z, inner) ** 2).sum().backward()
                return g
            h.register_hook(hook)
        (mlp.chain_rows(h, outer[1:]) ** 2).sum().backward()
        assert not mlp._DeferredWgrad._passes
        if not nested:
            (mlp.chain_rows(z, inner) ** 2).sum().backward()
        return [l[0].weight.grad.clone() for l in outer + inner]

    for a, b in zip(run(True), run(False)):
        assert torch.equal(a, b) and float(a.abs().max()) > 0


@pytest.mark.gpu
def test_layer_used_several_times_in_one_backward_pass(pn2):
    """forward_hierarchical runs the same layers once per mini-batch and calls backward() once: the deferred weight-gradient
    reductions of one layer then meet in the flush and must not share a launch (their blocks would read-modify-write the
    same gradient elements -- lost updates).  Many repeats of one 128 x 128 layer against per-call reductions (which the
    stream orders) -- bit for bit, several times."""
    import torch.nn as nn
    from pn2_amd import mlp, _hip
    torch.manual_seed(0)
    layers = [(nn.Conv1d(128, 128, 1).cuda(), nn.BatchNorm1d(128).cuda().train(), True),
              (nn.Conv1d(128, 128, 1).cuda(), nn.BatchNorm1d(128).cuda().train(), True)]
    xs = [torch.randn(3000 + 500 * i, 128, device="cuda") for i in range(12)]
    got = {}
    for mode, e in (("deferred", {}), ("per_call", {"PN2_NO_DEFER_WGRAD": 1})):
        for rep in range(3):
            for conv, bn, _ in layers:
                conv.weight.grad = bn.weight.grad = bn.bias.grad = None
            with env(**e):
                total = sum((mlp.chain_rows(x, layers) ** 2).sum() for x in xs)
                total.backward()
                assert not mlp._DeferredWgrad._passes
            grads = [layers[0][0].weight.grad.clone(), layers[1][0].weight.grad.clone()]
            if mode in got:
                assert all(torch.equal(a, b) for a, b in zip(grads, got[mode])), f"{mode}: run-to-run difference"
            got[mode] = grads
    for a, b in zip(got["deferred"], got["per_call"]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("B,N,S,K,D,xyz_last", [(2, 1024, 256, 32, 64, False), (3, 100, 50, 32, 64, True), (1, 64, 16, 32, 256, False),
                                                (2, 500, 40, 16, 6, False), (1, 300, 64, 8, 300, False), (2, 90, 7, 64, 130, False)])
def test_group_grad_presummed_per_group(pn2, monkeypatch, B, N, S, K, D, xyz_last):
    """The grouping backward sums the rows of a group that share row 0's destination (the padding of a sparse ball) before
    they leave as one atomic per channel: equal to the float64 scatter-add and to the row-wise atomic kernel
    (PN2_GROUP_GRAD_ROWS=1) within rounding -- padded balls, fully distinct balls, arbitrary duplicates."""
    from pn2_amd import ops
    rng = np.random.default_rng(N + D)
    dev = lambda a: torch.as_tensor(a, device="cuda")
    idx = rng.integers(0, N, size=(B, S, K)).astype(np.int32)
    cnt = rng.integers(1, K + 1, size=(B, S))
    pad = np.arange(K)[None, None, :] >= cnt[:, :, None]
    idx = np.where(pad, idx[:, :, :1], idx)                       # short balls repeat their first hit
    idx[:, 0, :] = idx[:, 0, ::-1]                                # one group with duplicates elsewhere
    xyz, new_xyz = dev(_cloud(B, N, 1)), dev(_cloud(B, S, 2))
    go = rng.normal(size=(B, S, K, 3 + D)).astype(np.float32)
    cols = slice(0, D) if xyz_last else slice(3, 3 + D)
    want = np.zeros((B, N, D))
    np.add.at(want, (np.arange(B)[:, None, None], idx.astype(np.int64)), go[..., cols].astype(np.float64))
    scale = np.abs(want).max()

    def run():
        feats = torch.zeros(B, N, D, device="cuda", requires_grad=True)
        ops.GroupPoints.apply(xyz, new_xyz, feats, dev(idx), xyz_last).backward(dev(go))
        return feats.grad.clone()

    monkeypatch.setenv("PN2_GROUP_GRAD_GROUPS", "1")              # the group kernel whatever the number of groups
    a = run()
    assert np.abs(a.cpu().numpy().astype(np.float64) - want).max() <= 2e-5 * scale
    monkeypatch.delenv("PN2_GROUP_GRAD_GROUPS")
    monkeypatch.setenv("PN2_GROUP_GRAD_ROWS", "1")
    c = run()
    assert np.abs(c.cpu().numpy().astype(np.float64) - want).max() <= 2e-5 * scale
