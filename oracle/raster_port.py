"""numpy restatement of the reference's raster data path -- TEST INFRASTRUCTURE ONLY, PARITY UNPINNED.

The reference modules (Modules/Pipeline/ModelPredicting.py, Modules/DataLoading/RasterizedTreeSet.py) import
`fastprogress`, which is not installed, so they cannot be imported to generate golden vectors; they hold no fixtures of
their own.  This file follows their source text loop by loop (same O(#rasters x N) masks), so it is only usable on small
clouds; tests/test_rasters.py compares the device path (rasters.py, csrc/raster.hip) with it element for element.
"""
import numpy as np


def rasterize_clouds(points64, raster_size, stride):
    """ModelPredicting.py:119-152: the grid from np.arange and the non-empty boxes (float64 test), x-major order.
    -> list of (min [3], max [3]) float64 bounds."""
    mn, mx = points64.min(axis=0), points64.max(axis=0)
    xs, ys, zs = (np.arange(mn[a], mx[a], stride) for a in range(3))
    out = []
    for x in xs:
        for y in ys:
            for z in zs:
                m = ((points64[:, 0] >= x) & (points64[:, 0] < x + raster_size) & (points64[:, 1] >= y) &
                     (points64[:, 1] < y + raster_size) & (points64[:, 2] >= z) & (points64[:, 2] < z + raster_size))
                if m.any():
                    out.append(([x, y, z], [x + raster_size, y + raster_size, z + raster_size]))
    return out


def getitem_rasters(points32, features32, offset_mask, bounds):
    """RasterizedTreeSet.py:226-254: one float32 box mask per raster (torch compares a float32 tensor with the JSON's
    Python floats in float32)."""
    idx = np.arange(len(points32))
    out = []
    for lo, hi in bounds:
        lo, hi = np.asarray(lo, np.float64).astype(np.float32), np.asarray(hi, np.float64).astype(np.float32)
        m = np.ones(len(points32), bool)
        for a in range(3):
            m &= (points32[:, a] >= lo[a]) & (points32[:, a] < hi[a])
        out.append({"points": points32[m], "features": features32[m], "offset_mask": offset_mask[m], "point_ids": idx[m]})
    return out


def collate_streaming(rasters, minibatch_size):
    """RasterizedTreeSet.py:390-455: the size adjustment and the zero padding to the group maximum."""
    orig = mb = minibatch_size
    while len(rasters) % mb == 1 and mb > 1:
        mb -= 1
        if mb == 1:
            mb = orig
            while len(rasters) % mb == 1:
                mb += 1
            break
    out = []
    for i in range(0, len(rasters), mb):
        group = rasters[i:i + mb]
        nmax = max(len(r["points"]) for r in group)
        F = group[0]["features"].shape[1]
        coords = np.zeros((len(group), 3, nmax), np.float32)
        feats = np.zeros((len(group), F, nmax), np.float32)
        pad = np.zeros((len(group), nmax), bool)
        for j, r in enumerate(group):
            n = len(r["points"])
            coords[j, :, :n] = r["points"].T
            feats[j, :, :n] = r["features"].T
            pad[j, :n] = True
        out.append({"coords": coords, "feats": feats, "masks_pad": pad,
                    "masks_off": np.concatenate([r["offset_mask"] for r in group]),
                    "point_ids": np.concatenate([r["point_ids"] for r in group])})
    return out
