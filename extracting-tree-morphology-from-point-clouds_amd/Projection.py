"""Closest-cylinder projection with the reference's names and signatures (Modules/Projection.py:19-144; the same code is
duplicated in PreProcessing/LabelGenerationCuda.py:20-135): label generation (offset of every cloud point to its QSM
cylinder) and the kNN-to-QSM step of the prediction pipeline (BASELINE configs[4]).

`closest_cylinder_cuda_batch` keeps the reference's arguments and numpy outputs; the device work is ONE launch of
libpn2hip's `pn2_cylinder_project_f32` for all points (the reference broadcasts 1024 points at a time against [1024, M, 3]
temporaries and comes back to the host after every batch).  `generate_offset_cloud_cuda_batched` keeps its `batch_size`
argument for compatibility and ignores it.

Parity: UNPINNED -- the reference module imports `fastprogress` (not installed), so no golden vectors can be generated;
oracle/pn2_oracle.c restates the reference's expressions line by line and the kernel matches it bit for bit
(tests/test_projection.py), plus closed-form geometric properties.
"""
import numpy as np
import torch

from . import _hip


def _dev_f32(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def cylinder_project(points, start, axis_unit, axis_length, radius, IDs=None, move_points_to_mantle=True):
    """Device-level form: tensors on a HIP device in, tensors out (ids int32 [N], distances [N], offsets [N,3])."""
    _hip.require_device(points, start)
    points = _hip.f32(points)
    if points.stride(-1) != 1:
        points = points.contiguous()
    N, M = points.shape[0], start.shape[0]
    dev = points.device
    start, axis_unit = _hip.f32(start).contiguous(), _hip.f32(axis_unit).contiguous()
    axis_length, radius = _hip.f32(axis_length).reshape(-1).contiguous(), _hip.f32(radius).reshape(-1).contiguous()
    ids = None if IDs is None else IDs.to(device=dev, dtype=torch.int32).contiguous()
    out_id = torch.empty(N, dtype=torch.int32, device=dev)
    out_d = torch.empty(N, dtype=torch.float32, device=dev)
    out_o = torch.empty(N, 3, dtype=torch.float32, device=dev)
    if N == 0:
        return out_id, out_d, out_o
    if M == 0:
        raise RuntimeError("cylinder_project: no cylinders")
    _hip.call("cylinder_project", _hip.lib().pn2_cylinder_project_f32, points.data_ptr(), points.stride(0), N, start.data_ptr(),
              axis_unit.data_ptr(), axis_length.data_ptr(), radius.data_ptr(), _hip.ptr(ids), M, int(bool(move_points_to_mantle)),
              out_id.data_ptr(), out_d.data_ptr(), out_o.data_ptr(), _hip.stream_ptr(), nbytes=32 * N + 32 * M,
              flops=90 * N * M)
    return out_id, out_d, out_o


def closest_cylinder_cuda_batch(points, start, radius, axis_length, axis_unit, IDs, device, move_points_to_mantle=True):
    """Reference signature (Projection.py:19): points numpy [N,3]; cylinder tensors on `device`;
    -> (ids, distances, offsets) as numpy arrays."""
    pts = _dev_f32(points, device)
    ids, dist, off = cylinder_project(pts, start, axis_unit, axis_length, radius, IDs, move_points_to_mantle)
    return ids.cpu().numpy(), dist.cpu().numpy(), off.cpu().numpy()


def cylinder_tensors(cylinders, device):
    """The preparation block of generate_offset_cloud_cuda_batched (Projection.py:121-132): start / end / radius / ID
    columns of the QSM table -> (start, radius, axis_length [M,1], axis_unit, IDs) on the device.  `cylinders` is a pandas
    DataFrame or a dict of arrays with the reference's column names."""
    col = (lambda names: np.stack([np.asarray(cylinders[n], dtype=np.float64) for n in names], axis=1))
    start = torch.tensor(col(["startX", "startY", "startZ"]), dtype=torch.float32, device=device)
    end = torch.tensor(col(["endX", "endY", "endZ"]), dtype=torch.float32, device=device)
    radius = torch.tensor(np.asarray(cylinders["radius"], dtype=np.float64), dtype=torch.float32, device=device)
    IDs = torch.tensor(np.asarray(cylinders["ID"]), dtype=torch.int32, device=device)
    axis = end - start
    axis_length = torch.norm(axis, dim=1, keepdim=True)
    safe = axis_length.clone()
    safe[safe < 1e-8] = 1e-8
    return start, radius, axis_length, axis / safe, IDs


def generate_offset_cloud_cuda_batched(cloud, cylinders, device, masterBar=None, batch_size=1024):
    """Reference signature (Projection.py:117): -> float64 [N,7] = xyz, offset vector, cylinder ID."""
    out = np.zeros((len(cloud), 7))
    start, radius, axis_length, axis_unit, IDs = cylinder_tensors(cylinders, device)
    ids, _, offsets = closest_cylinder_cuda_batch(cloud[:, :3], start, radius, axis_length, axis_unit, IDs, device)
    out[:, :3] = cloud[:, :3]
    out[:, 3:6] = offsets
    out[:, 6] = ids
    return out
