"""Raster data path on the device: from a tree's raw cloud to the network's level-0 input in a handful of launches.

Replaces the host loops either side of the hot path:
  * rasterize_clouds (Modules/Pipeline/ModelPredicting.py:98-163): a triple loop over the grid with one boolean mask over
    the whole cloud per box;
  * RasterizedTreeSet_Hierarchical.__getitem__ (Modules/DataLoading/RasterizedTreeSet.py:201-268): again one mask per
    raster, O(#rasters x N);
  * collate_fn_streaming (:390-459): per-mini-batch zero padding on the host and an upload per mini-batch.
Here the boxes containing a point are found by binary search on the boxes' float32 bounds (csrc/raster.hip), one sort
groups the points by box, and one kernel writes the zero-padded channel-first buffers of ALL mini-batches of the tree --
which is exactly what the whole-tree pass consumes (streaming.run_tree); the per-mini-batch tensors of the reference's
API are views of those buffers.

Semantics kept: the grid is np.arange(min, max, stride) per axis in float64 (so the cloud's maximum may lie outside every
box when max - min is a multiple of the stride -- such points get no prediction, as in the reference); membership is
lo <= p < hi on float32 coordinates against float32-rounded bounds (torch compares a float32 tensor with a Python float in
float32); rasters are ordered x-major / y / z and keep ascending point ids; mini-batches are consecutive groups with the
reference's size adjustment (a trailing mini-batch of one raster is avoided: BatchNorm over a single sample).
Difference, documented: a box is kept when it is non-empty under the float32 test (the reference decides with the float64
coordinates and may keep a box that the float32 test then leaves empty, or vice versa, for points within an ulp of a seam).

Parity: UNPINNED (the reference modules import `fastprogress`); tests/test_rasters.py checks this path against a numpy
restatement of the reference loops on small clouds, element for element.
"""
import numpy as np
import torch

from . import _hip


def grid_bounds(points_min, points_max, raster_size, stride):
    """-> per axis (lo float32 [n], hi float32 [n]) of the boxes, from the reference's np.arange grid."""
    out = []
    for a in range(3):
        vals = np.arange(points_min[a], points_max[a], stride)            # float64, like rasterize_clouds
        if len(vals) == 0:                                                  # degenerate extent: np.arange(min, min) is empty
            vals = np.array([points_min[a]], dtype=np.float64)
        out.append((vals.astype(np.float32), (vals + raster_size).astype(np.float32), vals))
    return out


def adjusted_minibatch_size(n_rasters, minibatch_size):
    """collate_fn_streaming's adjustment (RasterizedTreeSet.py:395-404): avoid a trailing mini-batch of ONE raster.
    (With a single raster the reference's loop never terminates -- 1 % k == 1 for every k > 1; here it is one mini-batch.)"""
    if n_rasters <= 1:
        return max(int(minibatch_size), 1)
    orig = mb = minibatch_size
    while n_rasters % mb == 1 and mb > 1:
        mb -= 1
        if mb == 1:
            mb = orig
            while n_rasters % mb == 1:
                mb += 1
            break
    return mb


class RasterStream(list):
    """The mini-batches of one tree: a list of the reference's mini-batch dicts (coords [B,3,N], feats [B,F,N], masks_pad,
    masks_off, point_ids) whose tensors are VIEWS of flat whole-tree buffers, kept in `flat` for the fused pass."""
    flat = None
    disjoint = False     # True: stride >= raster size, no point id repeats inside a mini-batch (streaming.py skips the overlap pass)


def rasterize_points(points, raster_size=1.0, stride=1.0, bounds=None):
    """points [N,3] float32 on the device -> (sorted point ids int64 [K], raster lengths (host list), box ids int64 [R],
    grid dims).  K = number of (point, box) memberships."""
    _hip.require_device(points)
    pts = _hip.f32(points)
    N = pts.shape[0]
    dev = pts.device
    if bounds is None:
        lo_hi = torch.stack([pts.min(dim=0)[0], pts.max(dim=0)[0]]).double().cpu().numpy()      # one small read-back
        bounds = grid_bounds(lo_hi[0], lo_hi[1], raster_size, stride)
    nx, ny, nz = (len(b[0]) for b in bounds)
    flat = np.concatenate([np.concatenate([b[0], b[1]]) for b in bounds]).astype(np.float32)
    bdev = torch.from_numpy(flat).to(dev, non_blocking=True)
    lib = _hip.lib()
    ranges = torch.empty(N, 6, dtype=torch.int32, device=dev)
    count = torch.empty(N, dtype=torch.int32, device=dev)
    _hip.call("raster_ranges", lib.pn2_raster_ranges_f32, pts.data_ptr(), pts.stride(0), N, bdev.data_ptr(), nx, ny, nz,
              ranges.data_ptr(), count.data_ptr(), _hip.stream_ptr())
    incl = torch.cumsum(count, 0, dtype=torch.int64)
    total = int(incl[-1])                                                  # host sync: sizes of what follows
    if total == 0:
        return torch.empty(0, dtype=torch.int64, device=dev), [], torch.empty(0, dtype=torch.int64, device=dev), (nx, ny, nz), bounds
    offset = incl - count
    keys = torch.empty(total, dtype=torch.int64, device=dev)
    _hip.call("raster_keys", lib.pn2_raster_keys, ranges.data_ptr(), offset.data_ptr(), N, ny, nz, keys.data_ptr(),
              _hip.stream_ptr())
    keys, _ = torch.sort(keys)
    box = torch.div(keys, N, rounding_mode="floor")
    ids = keys - box * N
    boxes, lengths = torch.unique_consecutive(box, return_counts=True)
    return ids, lengths.cpu().tolist(), boxes, (nx, ny, nz), bounds


def bounds_from_metadata(raster_entries):
    """The per-axis grid behind the boxes rasterize_clouds stored in its JSON ({"bounds": {"min": [...], "max": [...]}} per
    non-empty raster, ModelPredicting.py:133-149 / RasterizeClouds.py:52-97) -> (bounds as grid_bounds returns them, raster
    size, stride or None).  The reference's dataset takes the boxes from the JSON (RasterizedTreeSet.py:228-238), whatever
    raster_size / stride wrote them; so does the mirror when the JSON carries them."""
    lo = np.array([r["bounds"]["min"] for r in raster_entries], dtype=np.float64)
    hi = np.array([r["bounds"]["max"] for r in raster_entries], dtype=np.float64)
    size = float(np.median(hi - lo))
    out, strides = [], []
    for a in range(3):
        vals = np.unique(lo[:, a])
        if len(vals) > 1:
            strides.append(float(np.min(np.diff(vals))))
        out.append((vals.astype(np.float32), (vals + size).astype(np.float32), vals))
    return out, size, (min(strides) if strides else None)


def build_stream(points, features, offset_mask, raster_size=1.0, stride=1.0, minibatch_size=20, bounds=None):
    """The tree as forward_hierarchical_streaming expects it: -> RasterStream (see class).  points [N,3], features [N,F]
    (or None), offset_mask [N] bool, all on the device.  bounds: the grid to use instead of the one computed from the cloud
    (bounds_from_metadata: boxes stored by rasterize_clouds)."""
    ids, lengths, boxes, dims, bounds = rasterize_points(points, raster_size, stride, bounds=bounds)
    stream = RasterStream()
    if not lengths:
        return stream
    dev = points.device
    pts = _hip.f32(points)
    feats = None if features is None else _hip.f32(features)
    F = 0 if feats is None else feats.shape[1]
    mb = adjusted_minibatch_size(len(lengths), minibatch_size)
    table, groups = [], []
    row = first = 0
    for g0 in range(0, len(lengths), mb):
        group = lengths[g0:g0 + mb]
        npad = max(group)
        groups.append((len(group), npad, row, first, sum(group)))
        for n in group:
            table.append((row, npad, first, n))
            row += npad
            first += n
    rows = row
    tab = torch.tensor(table, dtype=torch.int32).to(dev, non_blocking=True)
    xyz_cf = torch.empty(3 * rows, dtype=torch.float32, device=dev)
    feats_cf = torch.empty(F * rows, dtype=torch.float32, device=dev) if F else None
    mask = torch.empty(rows, dtype=torch.bool, device=dev)
    _hip.call("raster_pack", _hip.lib().pn2_raster_pack_f32, pts.data_ptr(), pts.stride(0), _hip.ptr(feats),
              0 if feats is None else feats.stride(0), F, ids.data_ptr(), tab.data_ptr(), len(table), max(lengths), xyz_cf.data_ptr(),
              _hip.ptr(feats_cf), mask.data_ptr(), _hip.stream_ptr())
    moff = offset_mask.index_select(0, ids) if offset_mask is not None else torch.ones_like(ids, dtype=torch.bool)
    for b, npad, row0, first0, nreal in groups:
        stream.append({
            "coords": xyz_cf[3 * row0:3 * (row0 + b * npad)].view(b, 3, npad),
            "feats": (feats_cf[F * row0:F * (row0 + b * npad)].view(b, F, npad) if F
                      else torch.zeros(b, 0, npad, device=dev)),
            "masks_pad": mask[row0:row0 + b * npad].view(b, npad),
            "masks_off": moff[first0:first0 + nreal],
            "point_ids": ids[first0:first0 + nreal],
        })
    stream.flat = {"xyz_cf": xyz_cf, "feats_cf": feats_cf, "masks_pad": mask, "point_ids": ids, "masks_off": moff,
                   "lengths": [t[1] for t in table], "rasters": len(lengths), "boxes": boxes, "dims": dims,
                   "bounds": bounds}
    stream.disjoint = float(stride) >= float(raster_size)      # half-open boxes on a grid of pitch `stride`: no point in two
    return stream


def raster_bounds_metadata(boxes, dims, bounds, raster_size):
    """The per-raster metadata rasterize_clouds writes to JSON: raster_id and float64 box bounds."""
    nx, ny, nz = dims
    b = boxes.cpu().numpy()
    kx, ky, kz = b // (ny * nz), (b // nz) % ny, b % nz
    out = []
    for r, (i, j, k) in enumerate(zip(kx, ky, kz)):
        lo = [float(bounds[0][2][i]), float(bounds[1][2][j]), float(bounds[2][2][k])]
        out.append({"raster_id": r, "bounds": {"min": lo, "max": [v + raster_size for v in lo]}})
    return out
