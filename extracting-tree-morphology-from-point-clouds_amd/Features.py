"""kNN feature helpers with the reference's names and numpy-in / numpy-out signatures (Modules/Features.py),
neighbourhood work on the GPU (csrc/features.hip through the C ABI, float64 like the reference).

One brute-force pass over the cloud (`pn2_knn_radius_f64`) yields the 15 nearest neighbours of every point -- their
first 10 are the k = 10 neighbours the curvature uses -- and the radius-0.1 density; `pn2_cov_eig_f64` replaces the
per-point Python loop of np.cov + np.linalg.svd / eigvalsh.  The O(N) leftovers (height, verticality, distance to
the centre) stay numpy on the host, identical to the reference.

Reproduced quirk: compute_normals_ckdtree returns ``v[:, -1]`` of ``_, _, v = np.linalg.svd(cov)`` (reference
:129-130) -- the LAST COLUMN of V^T, i.e. the z-components of the three principal directions (largest variance first),
not the direction of least variance.  The sign of every entry is LAPACK's choice in the reference; here each principal
direction is normalised to a positive largest component.  `compute_normals` (the sklearn variant, :11-29) takes the
actual smallest-variance direction ``v[-1]``.
"""
import os

import numpy as np
import torch

from . import _hip

__all__ = ["compute_normals", "compute_height", "compute_density", "compute_verticality", "compute_distance_to_center",
           "compute_curvature", "compute_normals_ckdtree", "compute_curvature_ckdtree", "compute_density_ckdtree",
           "add_features", "neighbourhoods"]


GRID_MIN_POINTS = 4096     # below this the brute-force scan (csrc/features.hip) is as fast as building a grid


def _device_points(points):
    pts = np.ascontiguousarray(np.asarray(points)[:, :3], dtype=np.float64)
    if not torch.cuda.is_available():
        raise RuntimeError("pn2_amd.Features runs on a HIP device only")
    return torch.from_numpy(pts).cuda()


def neighbourhoods(points, k=15, radius=None):
    """-> (nn_idx int32 [N,k] on the device, radius counts int32 [N] or None, device points).  Neighbours ascending by
    (squared distance, index), the point itself first (cKDTree.query(points, k) order)."""
    pts = _device_points(points)
    n = pts.shape[0]
    idx = torch.empty(n, k, dtype=torch.int32, device=pts.device)
    cnt = torch.empty(n, dtype=torch.int32, device=pts.device) if radius is not None else None
    r2 = float(radius) ** 2 if radius is not None else -1.0
    lib = _hip.lib()
    if n >= GRID_MIN_POINTS and not os.environ.get("PN2_KNN_BRUTE"):
        ws = torch.empty(lib.pn2_knn_grid_workspace_bytes(n), dtype=torch.uint8, device=pts.device)
        _hip.call("knn_radius_grid", lib.pn2_knn_radius_grid_f64, pts.data_ptr(), n, int(k), r2, idx.data_ptr(), None,
                  _hip.ptr(cnt), ws.data_ptr(), ws.numel(), _hip.stream_ptr())
    else:
        _hip.call("knn_radius", lib.pn2_knn_radius_f64, pts.data_ptr(), n, int(k), r2, idx.data_ptr(), None,
                  _hip.ptr(cnt), _hip.stream_ptr())
    return idx, cnt, pts


def _cov_eig(pts, idx, k):
    n = pts.shape[0]
    evals = torch.empty(n, 3, dtype=torch.float64, device=pts.device)
    evecs = torch.empty(n, 3, 3, dtype=torch.float64, device=pts.device)
    _hip.call("cov_eig", _hip.lib().pn2_cov_eig_f64, pts.data_ptr(), n, idx.data_ptr(), idx.shape[1], int(k),
              evals.data_ptr(), evecs.data_ptr(), _hip.stream_ptr())
    return evals, evecs


def _normals_ckdtree_from(evecs):
    # svd orders by decreasing variance: rows 2, 1, 0 of the ascending eigenvectors; v[:, -1] = their z-components
    return torch.stack([evecs[:, 2, 2], evecs[:, 1, 2], evecs[:, 0, 2]], dim=1)


def _curvature_from(evals):
    return evals[:, 0] / (evals.sum(dim=1) + 1e-6)


def compute_normals_ckdtree(points, k=10):
    """Reference :111-133 (see the module docstring for what ``v[:, -1]`` is)."""
    idx, _, pts = neighbourhoods(points, k)
    return _normals_ckdtree_from(_cov_eig(pts, idx, k)[1]).cpu().numpy()


def compute_normals(points, k=10):
    """Reference :11-29 (sklearn kneighbors + ``v[-1]``): the unit direction of least variance."""
    idx, _, pts = neighbourhoods(points, k)
    return _cov_eig(pts, idx, k)[1][:, 0, :].cpu().numpy()


def compute_curvature_ckdtree(points, k=10):
    """Reference :136-158: smallest eigenvalue / (sum of eigenvalues + 1e-6)."""
    idx, _, pts = neighbourhoods(points, k)
    return _curvature_from(_cov_eig(pts, idx, k)[0]).cpu().numpy()


compute_curvature = compute_curvature_ckdtree      # reference :76-107, same quantity through sklearn


def compute_density_ckdtree(points, radius=0.1):
    """Reference :161-173: number of points within `radius` of every point (itself included)."""
    _, cnt, _ = neighbourhoods(points, 1, radius)
    return cnt.cpu().numpy().astype(np.int64)


compute_density = compute_density_ckdtree          # reference :42-52


def compute_height(points):
    """Reference :31-40."""
    z_min = np.min(points[:, 2])
    z_max = np.max(points[:, 2])
    return (points[:, 2] - z_min) / (z_max - z_min)


def compute_verticality(normals):
    """Reference :54-63."""
    return np.abs(np.dot(normals, np.array([0, 0, 1])))


def compute_distance_to_center(points):
    """Reference :65-74."""
    center_xy = np.mean(points[:, :2], axis=0)
    return np.linalg.norm(points[:, :2] - center_xy, axis=1)


def add_features(labeled_cloud, use_normals=True, use_heights=True, use_densities=True, use_verticalities=True,
                 use_distances=True, use_curvatures=True):
    """Reference :178-229, same column order: [cloud | normals(3) | curvature | density | height | verticality |
    distance].  One kNN pass (k = 15, radius 0.1) serves normals, curvature (its first 10 neighbours) and density."""
    points = labeled_cloud[:, :3]
    cols = [labeled_cloud]
    normals = None
    need_nn = use_normals or use_curvatures or use_densities or use_verticalities
    if need_nn:
        idx, cnt, pts = neighbourhoods(points, 15, 0.1 if use_densities else None)
    if use_normals or use_verticalities:
        evecs15 = _cov_eig(pts, idx, 15)[1]
        # with use_normals=False the reference falls back to compute_normals (the v[-1] variant) for the verticality
        normals = (_normals_ckdtree_from(evecs15) if use_normals else evecs15[:, 0, :]).cpu().numpy()
    if use_normals:
        cols.append(normals)
    if use_curvatures:
        cols.append(_curvature_from(_cov_eig(pts, idx, 10)[0]).cpu().numpy()[:, np.newaxis])
    if use_densities:
        cols.append(cnt.cpu().numpy().astype(np.int64)[:, np.newaxis])
    if use_heights:
        cols.append(compute_height(points)[:, np.newaxis])
    if use_verticalities:
        cols.append(compute_verticality(normals)[:, np.newaxis])
    if use_distances:
        cols.append(compute_distance_to_center(points)[:, np.newaxis])
    return np.concatenate(cols, axis=1)
