"""Closest-cylinder projection (Modules/Projection.py:19-144).

PARITY UNPINNED: the reference module imports `fastprogress` (absent), so no golden vectors exist; the reference has no
tests of its own.  CPU: the C oracle (a line-by-line restatement) against closed-form geometry.  GPU: the HIP kernel
against the oracle, bit for bit."""
import numpy as np
import pytest

import helpers
from oracle import pn2_oracle as O


def _forest(m, seed):
    rng = np.random.default_rng(seed)
    start = rng.uniform(-5, 5, size=(m, 3)).astype(np.float32)
    d = rng.normal(size=(m, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    end = (start + d * rng.uniform(0.2, 3.0, size=(m, 1))).astype(np.float32)
    radius = rng.uniform(0.02, 0.4, size=m).astype(np.float32)
    return start, end, radius


def test_oracle_matches_closed_form_geometry():
    """One cylinder along z, radius r, length L: the reference's distance is | rho - r | beside the cylinder, and the
    distance to the cap disk beyond its ends; offsets point from the point to the mantle / the cap rim."""
    O.build()
    r, L = 0.5, 2.0
    start, unit, length = O.cylinder_axes(np.array([[0, 0, 0]], np.float32), np.array([[0, 0, L]], np.float32))
    assert np.allclose(unit, [[0, 0, 1]]) and np.allclose(length, [L])
    pts = np.array([[2.0, 0, 1.0],       # beside, outside
                    [0.1, 0, 1.0],       # beside, inside
                    [0.0, 0.3, 3.0],     # beyond the top cap, above the disk
                    [2.0, 0.0, 3.0],     # beyond the top cap, outside the rim
                    [0.0, -0.2, -1.0]],  # below the bottom cap
                   np.float32)
    ids, dist, off = O.cylinder_project(pts, start, unit, length, np.array([r], np.float32), mantle=False)
    assert (ids == 0).all()
    np.testing.assert_allclose(dist[0], 1.5, rtol=1e-6)
    np.testing.assert_allclose(off[0], [-1.5, 0, 0], atol=1e-6)
    np.testing.assert_allclose(dist[1], 0.4, rtol=1e-6)
    np.testing.assert_allclose(off[1], [0.4, 0, 0], atol=1e-6)
    np.testing.assert_allclose(dist[2], 1.0, rtol=1e-6)                      # straight down onto the cap disk
    np.testing.assert_allclose(off[2], [0, 0, -1.0], atol=1e-6)
    np.testing.assert_allclose(dist[3], np.hypot(1.5, 1.0), rtol=1e-6)       # to the rim
    np.testing.assert_allclose(dist[4], 1.0, rtol=1e-6)
    # mantle variant: the beyond-the-cap points are moved to the nearer END of the diameter segment (the rim)
    _, dist_m, off_m = O.cylinder_project(pts, start, unit, length, np.array([r], np.float32), mantle=True)
    np.testing.assert_array_equal(dist_m, dist)                              # the distance itself is not changed
    np.testing.assert_allclose(pts[2] + off_m[2], [0, 0.5, 2.0], atol=1e-6)
    np.testing.assert_allclose(pts[4] + off_m[4], [0, -0.5, 0.0], atol=1e-6)


def test_oracle_picks_the_first_of_equal_cylinders_and_maps_ids():
    O.build()
    start, end, radius = _forest(5, 1)
    start, end, radius = np.concatenate([start, start]), np.concatenate([end, end]), np.concatenate([radius, radius])
    s, u, l = O.cylinder_axes(start, end)
    pts = np.random.default_rng(2).uniform(-6, 6, size=(200, 3)).astype(np.float32)
    ids, _, _ = O.cylinder_project(pts, s, u, l, radius)
    assert ids.max() < 5                                                     # duplicates 5..9 never win a tie
    named, _, _ = O.cylinder_project(pts, s, u, l, radius, ids=np.arange(10, dtype=np.int32) * 7 + 3)
    np.testing.assert_array_equal(named, ids * 7 + 3)


@pytest.mark.gpu
@pytest.mark.parametrize("n,m,mantle", [(5000, 1, True), (1, 700, True), (70000, 2500, True), (3333, 1025, False)])
def test_hip_projection_equals_oracle_bitwise(n, m, mantle):
    import torch
    helpers.load_pkg()
    from pn2_amd import Projection as P
    O.build()
    start, end, radius = _forest(m, seed=m)
    s, u, l = O.cylinder_axes(start, end)
    rng = np.random.default_rng(n)
    pts = rng.uniform(-6, 6, size=(n, 3)).astype(np.float32)
    pts[: n // 4] = (start[rng.integers(0, m, n // 4)] + rng.normal(size=(n // 4, 3)) * 0.05).astype(np.float32)   # near the axes' ends
    ids = (np.arange(m, dtype=np.int32) * 3 + 11)
    want = O.cylinder_project(pts, s, u, l, radius, ids=ids, mantle=mantle)
    dev = torch.device("cuda")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)         # noqa: E731
    got = P.closest_cylinder_cuda_batch(pts, t(s), t(radius), t(l)[:, None], t(u), t(ids), dev, move_points_to_mantle=mantle)
    np.testing.assert_array_equal(got[0], want[0])
    np.testing.assert_array_equal(got[1].view(np.uint32), want[1].view(np.uint32))
    np.testing.assert_array_equal(got[2].view(np.uint32), want[2].view(np.uint32))
    # the preparation block and the [N,7] label array of generate_offset_cloud_cuda_batched
    cyl = {"startX": start[:, 0], "startY": start[:, 1], "startZ": start[:, 2], "endX": end[:, 0], "endY": end[:, 1],
           "endZ": end[:, 2], "radius": radius, "ID": ids}
    s2, r2, l2, u2, i2 = P.cylinder_tensors(cyl, dev)
    # (torch.norm on the device need not round like sqrt((x*x + y*y) + z*z): last-bit agreement is not required here)
    np.testing.assert_allclose(u2.cpu().numpy(), u, rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(l2.cpu().numpy().reshape(-1), l, rtol=2e-6)
    if mantle and np.array_equal(u2.cpu().numpy(), u) and np.array_equal(l2.cpu().numpy().reshape(-1), l):
        lab = P.generate_offset_cloud_cuda_batched(np.concatenate([pts, np.zeros((n, 2), np.float32)], 1).astype(np.float64), cyl, dev)
        assert lab.shape == (n, 7) and lab.dtype == np.float64
        np.testing.assert_array_equal(lab[:, 6].astype(np.int32), want[0])
        np.testing.assert_array_equal(lab[:, 3:6].astype(np.float32), want[2])
