"""CPU restatement of Modules/Features.py:111-229 with the reference's own structure (scipy cKDTree query, then a
Python loop of np.cov + np.linalg.svd / eigvalsh per point) -- TEST INFRASTRUCTURE / cpu_baseline only, never imported by
the product.  `sample` restricts the per-point loops to the first `sample` points (the tree is still built over the
whole cloud) so that a baseline run stays bounded."""
import numpy as np
from scipy.spatial import cKDTree


def add_features_port(labeled_cloud, sample=None):
    points = labeled_cloud[:, :3]
    n = points.shape[0] if sample is None else min(int(sample), points.shape[0])
    tree = cKDTree(points)
    q = points[:n]
    _, nn15 = tree.query(q, k=15)
    normals = np.zeros((n, 3))
    for i in range(n):
        nb = points[nn15[i]] - q[i]
        _, _, v = np.linalg.svd(np.cov(nb.T))
        normals[i] = v[:, -1]
    _, nn10 = tree.query(q, k=10)
    curvature = np.zeros(n)
    for i in range(n):
        nb = points[nn10[i]] - q[i]
        ev = np.linalg.eigvalsh(np.cov(nb.T))
        curvature[i] = ev[0] / (np.sum(ev) + 1e-6)
    density = np.array([len(tree.query_ball_point(p, r=0.1)) for p in q])
    z = points[:, 2]
    height = ((z - z.min()) / (z.max() - z.min()))[:n]
    vert = np.abs(normals @ np.array([0, 0, 1]))
    dist = np.linalg.norm(points[:, :2] - points[:, :2].mean(0), axis=1)[:n]
    return np.concatenate([labeled_cloud[:n], normals, curvature[:, None], density[:, None], height[:, None], vert[:, None],
                           dist[:, None]], axis=1)
