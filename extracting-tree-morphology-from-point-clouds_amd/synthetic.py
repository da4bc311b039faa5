"""Seeded synthetic inputs for tests and bench.py: the "Gaussian-branch tree" of SURVEY.md section 8(d).

There is no dataset in the reference tree and no network, so every measurement uses these clouds.
Coordinates are raw metres and are NOT centred (the reference feeds raw coordinates to the network,
Modules/DataLoading/RasterizedTreeSet.py:427), which is what makes the fp32 cancellation error of the
expanded squared distance part of the behaviour to reproduce.
"""
import numpy as np


def gaussian_branch_tree(n_points, seed=0, n_segments=64):
    """Return (xyz [n,3] f32, offset_labels [n,3] f32, segment_id [n] i64).

    Trunk (0,0,0)->(0,0,20) radius 0.25; branch i starts on a random earlier segment p at
    a_p + t (b_p - a_p), t~U(0.2,1), direction ~N(0,I) with z <- 0.7|z| normalised, length U(1,5),
    radius max(0.02, r_p U(0.4,0.8)).  Points are assigned to segments independently with probability
    proportional to length*radius (so the array order is spatially incoherent), point = axis point +
    N(0, r^2 I); offset label = axis point - point.
    """
    rng = np.random.default_rng(seed)
    a = np.zeros((n_segments, 3))
    b = np.zeros((n_segments, 3))
    rad = np.zeros(n_segments)
    b[0] = (0.0, 0.0, 20.0)
    rad[0] = 0.25
    for i in range(1, n_segments):
        p = int(rng.integers(0, i))
        t = rng.uniform(0.2, 1.0)
        start = a[p] + t * (b[p] - a[p])
        d = rng.normal(size=3)
        d[2] = 0.7 * abs(d[2])
        d /= np.linalg.norm(d)
        length = rng.uniform(1.0, 5.0)
        a[i] = start
        b[i] = start + length * d
        rad[i] = max(0.02, rad[p] * rng.uniform(0.4, 0.8))
    weight = np.linalg.norm(b - a, axis=1) * rad
    seg = rng.choice(n_segments, size=n_points, p=weight / weight.sum())
    t = rng.uniform(0.0, 1.0, size=n_points)
    axis = a[seg] + t[:, None] * (b[seg] - a[seg])
    pts = axis + rng.normal(size=(n_points, 3)) * rad[seg][:, None]
    return pts.astype(np.float32), (axis - pts).astype(np.float32), seg.astype(np.int64)


def rasterize(xyz, size=1.0, stride=1.0):
    """Half-open axis-aligned boxes on a grid from the bbox minimum, like the reference's on-the-fly
    rasteriser (Modules/Pipeline/ModelPredicting.py:98-163, PreProcessing/RasterizeClouds.py:52-63).
    Returns a list of int64 index arrays (ascending), one per non-empty box."""
    lo = xyz.min(axis=0)
    hi = xyz.max(axis=0)
    out = []
    nx, ny, nz = (int(np.floor((hi[d] - lo[d]) / stride)) + 1 for d in range(3))
    if size == stride:
        cell = np.floor((xyz - lo) / stride).astype(np.int64)
        key = (cell[:, 0] * ny + cell[:, 1]) * nz + cell[:, 2]
        order = np.argsort(key, kind="stable")
        ks = key[order]
        cuts = np.flatnonzero(np.diff(ks)) + 1
        for chunk in np.split(order, cuts):
            out.append(np.sort(chunk))
        return out
    for ix in range(nx):
        for iy in range(ny):
            for iz in range(nz):
                c0 = lo + stride * np.array([ix, iy, iz])
                m = np.all((xyz >= c0) & (xyz < c0 + size), axis=1)
                if m.any():
                    out.append(np.flatnonzero(m))
    return out


def pad_rasters(xyz, feats, rasters):
    """Zero-pad a list of rasters to the longest one: coords [R,3,Nmax], feats [R,F,Nmax], masks_pad [R,Nmax]
    (the collate of Modules/DataLoading/RasterizedTreeSet.py:407-429)."""
    nmax = max(len(r) for r in rasters)
    R = len(rasters)
    coords = np.zeros((R, 3, nmax), np.float32)
    fts = np.zeros((R, feats.shape[1], nmax), np.float32)
    mask = np.zeros((R, nmax), bool)
    for i, r in enumerate(rasters):
        coords[i, :, : len(r)] = xyz[r].T
        fts[i, :, : len(r)] = feats[r].T
        mask[i, : len(r)] = True
    return coords, fts, mask
