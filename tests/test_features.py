"""kNN feature helpers (Modules/Features.py): oracle vs the reference's own output (CPU), HIP path vs both (GPU).

Fixture: tests/golden/features.npz = add_features of the imported reference on a 3000-point synthetic tree
(tests/golden/make_golden_features.py).  What can be compared how:
  * kNN lists, density: exactly (the cloud has no distance ties at the list boundary);
  * curvature, height, distance: 1e-9 relative (float64 end to end; summation order differs);
  * normals = z-components of the three principal directions: the reference's SIGNS are whatever LAPACK's svd returned,
    and a direction is only defined when its eigenvalue is separated from the others -- compared as absolute values on
    the points whose relative eigenvalue gaps exceed 1e-3, with a tolerance that grows with 1 / gap.
"""
import os

import numpy as np
import pytest

from helpers import load_pkg

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "features.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def O():
    from oracle import pn2_oracle as o
    o.build()
    return o


def check_against_reference(got, gold, evals15):
    ref = gold["enriched"]
    assert got.shape == ref.shape
    np.testing.assert_array_equal(got[:, :7], ref[:, :7])
    np.testing.assert_array_equal(got[:, 11], ref[:, 11])                                   # density
    np.testing.assert_allclose(got[:, 10], ref[:, 10], rtol=1e-8, atol=1e-12)               # curvature
    np.testing.assert_allclose(got[:, 12], ref[:, 12], rtol=1e-12, atol=0)                  # height
    np.testing.assert_allclose(got[:, 14], ref[:, 14], rtol=1e-12, atol=1e-14)              # distance to the centre
    lam = evals15                                                                           # ascending
    gap = np.minimum(lam[:, 1] - lam[:, 0], lam[:, 2] - lam[:, 1]) / lam[:, 2]
    ok = gap > 1e-3
    assert ok.mean() > 0.9
    err = np.abs(np.abs(got[ok, 7:10]) - np.abs(ref[ok, 7:10])).max(1)
    assert (err <= 1e-9 / gap[ok]).all(), float((err * gap[ok]).max())
    np.testing.assert_allclose(got[ok, 13], ref[ok, 13], rtol=0, atol=float((1e-9 / gap[ok]).max()))   # verticality


def test_oracle_matches_reference_features(gold, O):
    cloud = gold["cloud"]
    idx, d2, cnt = O.knn_radius(cloud[:, :3], 15, 0.1)
    np.testing.assert_array_equal(idx, gold["nn15"])
    assert (np.diff(d2, axis=1) >= 0).all() and (idx[:, 0] == np.arange(len(cloud))).all()
    evals15, evecs15 = O.cov_eig(cloud[:, :3], idx, 15)
    np.testing.assert_allclose(np.linalg.norm(evecs15, axis=2), 1.0, atol=1e-12)
    check_against_reference(O.add_features(cloud), gold, evals15)
    # the sklearn-variant normals (v[-1], the direction of least variance), up to sign
    ev10, vec10 = O.cov_eig(cloud[:500, :3], O.knn_radius(cloud[:500, :3], 10)[0], 10)
    ref = gold["normals_sklearn"]
    gap = (ev10[:, 1] - ev10[:, 0]) / ev10[:, 2]
    ok = gap > 1e-3
    dots = np.abs((vec10[ok, 0, :] * ref[ok]).sum(1))
    assert (1.0 - dots <= 1e-9 / gap[ok]).all()


@pytest.mark.gpu
def test_hip_features_match_oracle_and_reference(gold, O):
    load_pkg()
    import torch
    from pn2_amd import Features as F
    cloud = gold["cloud"]
    idx, cnt, pts = F.neighbourhoods(cloud[:, :3], 15, 0.1)
    oi, _, oc = O.knn_radius(cloud[:, :3], 15, 0.1)
    np.testing.assert_array_equal(idx.cpu().numpy(), oi)
    np.testing.assert_array_equal(cnt.cpu().numpy(), oc)
    evals, evecs = F._cov_eig(pts, idx, 15)
    oe, ov = O.cov_eig(cloud[:, :3], oi, 15)
    np.testing.assert_allclose(evals.cpu().numpy(), oe, rtol=1e-9, atol=1e-18)
    gap = np.minimum(oe[:, 1] - oe[:, 0], oe[:, 2] - oe[:, 1]) / oe[:, 2]
    ok = gap > 1e-3
    assert (np.abs(evecs.cpu().numpy()[ok] - ov[ok]).reshape(ok.sum(), -1).max(1) <= 1e-10 / gap[ok]).all()
    check_against_reference(F.add_features(cloud.copy()), gold, oe)
    # single-feature entry points and the reference's fallbacks
    np.testing.assert_array_equal(F.compute_density_ckdtree(cloud[:, :3]), gold["enriched"][:, 11].astype(np.int64))
    np.testing.assert_allclose(F.compute_curvature_ckdtree(cloud[:, :3], k=10), gold["enriched"][:, 10], rtol=1e-8, atol=1e-12)
    with pytest.raises(RuntimeError):
        F.neighbourhoods(cloud[:10, :3], 15)                      # k > N


@pytest.mark.gpu
def test_hip_knn_ragged_sizes_vs_oracle(O):
    load_pkg()
    from pn2_amd import Features as F
    rng = np.random.default_rng(0)
    for n, k in [(17, 16), (300, 10), (1025, 15), (2500, 3)]:
        pts = rng.normal(size=(n, 3))
        pts[n // 2] = pts[0]                                       # an exact duplicate: tie at distance 0
        idx, cnt, _ = F.neighbourhoods(pts, k, 0.5)
        oi, _, oc = O.knn_radius(pts, k, 0.5)
        np.testing.assert_array_equal(idx.cpu().numpy(), oi)
        np.testing.assert_array_equal(cnt.cpu().numpy(), oc)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,k,r", [("tree", 60000, 15, 0.1), ("tree", 5000, 10, 0.1), ("uniform", 30000, 16, 0.05),
                                        ("lumps", 20000, 15, 0.2), ("line", 8192, 3, 0.01), ("same", 4500, 5, 0.1)])
def test_grid_knn_is_the_brute_force_result(kind, n, k, r):
    """csrc/knn_grid.hip vs the full scan of csrc/features.hip, bit for bit (indices, squared distances, radius counts):
    a tree, uniform noise, tight lumps far apart (most 27-cell searches fail their certificate -> slow path), a line
    with duplicated points (ties), and a cloud of identical points (one cell, every query on the slow path)."""
    load_pkg()
    import ctypes
    import torch
    from pn2_amd import _hip
    from pn2_amd.synthetic import gaussian_branch_tree
    rng = np.random.default_rng(7)
    if kind == "tree":
        pts = gaussian_branch_tree(n, seed=9)[0].astype(np.float64)
    elif kind == "uniform":
        pts = rng.uniform(-3, 5, size=(n, 3))
    elif kind == "lumps":
        centres = rng.uniform(-50, 50, size=(40, 3))
        pts = centres[rng.integers(0, 40, n)] + rng.normal(size=(n, 3)) * 0.01
    elif kind == "line":
        pts = np.stack([np.linspace(0, 1, n), np.zeros(n), np.zeros(n)], axis=1)
        pts[1::2] = pts[0::2]                                          # every point twice
    else:
        pts = np.tile(np.array([[1.5, -2.0, 7.0]]), (n, 1))
    p = torch.from_numpy(np.ascontiguousarray(pts)).cuda()
    lib = _hip.lib()
    out = {}
    for tag in ("brute", "grid"):
        idx = torch.empty(n, k, dtype=torch.int32, device="cuda")
        d2 = torch.empty(n, k, dtype=torch.float64, device="cuda")
        cnt = torch.empty(n, dtype=torch.int32, device="cuda")
        if tag == "brute":
            st = lib.pn2_knn_radius_f64(p.data_ptr(), n, k, r * r, idx.data_ptr(), d2.data_ptr(), cnt.data_ptr(), _hip.stream_ptr())
        else:
            ws = torch.empty(lib.pn2_knn_grid_workspace_bytes(n), dtype=torch.uint8, device="cuda")
            st = lib.pn2_knn_radius_grid_f64(p.data_ptr(), n, k, r * r, idx.data_ptr(), d2.data_ptr(), cnt.data_ptr(),
                                             ws.data_ptr(), ws.numel(), _hip.stream_ptr())
        assert st == 0
        torch.cuda.synchronize()
        out[tag] = (idx.cpu().numpy(), d2.cpu().numpy(), cnt.cpu().numpy())
    np.testing.assert_array_equal(out["grid"][1].view(np.uint64), out["brute"][1].view(np.uint64))
    np.testing.assert_array_equal(out["grid"][0], out["brute"][0])
    np.testing.assert_array_equal(out["grid"][2], out["brute"][2])
