"""Tensor-level wrappers of the C-ABI entry points (include/pn2_hip.h) and their autograd Functions.

Layout conventions used inside the package (DESIGN.md "Data layout in HBM"):
  * coordinates are handed to the kernels with explicit strides, so the channel-first ``[B,3,N]`` tensors the
    model receives are consumed as SoA planes without a transpose;
  * feature maps are channels-last ``[B,N,C]`` (one contiguous row per point) between kernels;
  * indices are int32 on the device.
Nothing here falls back to the CPU: a CPU tensor raises.
"""
import ctypes
import os

import torch

from . import _hip

_DEBUG = bool(int(os.environ.get("PN2_DEBUG", "0")))


def _strides3(t):
    """Element strides (batch, point, channel) of a [B,N,C] view."""
    return t.stride(0), t.stride(1), t.stride(2)


def _workspace(nbytes, device):
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


# One sticky int32 status word per device (include/pn2_hip.h: PN2_STATUS_*).  Kernels OR a bit in when they gave up
# instead of producing results -- an FPS hand-off that timed out on a busy GPU, an index outside its range -- and leave
# -1 / NaN behind; nothing is read back on the hot path.  `check_status()` reads the word (a device sync) and raises:
# call it where the host synchronises anyway (the loss read-back of a step, the end of an inference pass, a test).
_status_words = {}
_STATUS_TEXT = {_hip.STATUS_FPS_HANDOFF: "farthest_point_sample: a workgroup hand-off timed out (GPU shared with another "
                                         "kernel?); the unfinished rows hold -1 / NaN",
                _hip.STATUS_FPS_ARRIVAL: "farthest_point_sample: the launch's workgroups were not co-resident",
                _hip.STATUS_BAD_INDEX: "a gather was handed an index outside [0, N)"}


def status_word(device):
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    t = _status_words.get(dev)
    if t is None:
        t = _status_words[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


def check_status(device=None, clear=True):
    """Raise RuntimeError if any kernel since the last check reported a failure on `device` (default: every device
    used so far).  Synchronises the device."""
    words = [status_word(device)] if device is not None else list(_status_words.values())
    bad = []
    for t in words:
        v = int(t.item())
        if v:
            if clear:
                t.zero_()
            bad += [f"{t.device}: {text}" for bit, text in _STATUS_TEXT.items() if v & bit]
    if bad:
        raise RuntimeError("libpn2hip reported failed kernels -- " + "; ".join(bad))


# ----------------------------------------------------------------------------------------------- forward-only ops
def square_distance(src, dst):
    """[B,N,3] x [B,M,3] -> [B,N,M], the reference's expanded form (pointnet2_utils.py:21-42)."""
    _hip.require_device(src, dst)
    src, dst = _hip.f32(src), _hip.f32(dst)
    B, N, _ = src.shape
    M = dst.shape[1]
    out = torch.empty(B, N, M, dtype=torch.float32, device=src.device)
    _hip.call("square_distance", _hip.lib().pn2_square_distance_f32, src.data_ptr(), *_strides3(src), dst.data_ptr(),
              *_strides3(dst), B, N, M, out.data_ptr(), _hip.stream_ptr(), nbytes=B * (12 * (N + M) + 4 * N * M))
    return out


def furthest_point_sample(xyz, npoint, start):
    """xyz [B,N,3] (any strides), start [B] int64 on the device -> (idx int32 [B,npoint], new_xyz [B,npoint,3])."""
    _hip.require_device(xyz, start)
    xyz = _hip.f32(xyz)
    B, N, _ = xyz.shape
    lib = _hip.lib()
    idx = torch.empty(B, npoint, dtype=torch.int32, device=xyz.device)
    new_xyz = torch.empty(B, npoint, 3, dtype=torch.float32, device=xyz.device)
    nbytes = lib.pn2_fps_workspace_bytes(B, N, npoint)
    if nbytes == 0:
        raise RuntimeError(f"farthest_point_sample: unsupported size B={B} N={N} npoint={npoint}")
    ws = _workspace(nbytes, xyz.device)
    start = start.to(device=xyz.device, dtype=torch.int64).contiguous()
    _hip.call("farthest_point_sample", lib.pn2_fps_f32, xyz.data_ptr(), *_strides3(xyz), B, N, npoint, start.data_ptr(),
              idx.data_ptr(), new_xyz.data_ptr(), ws.data_ptr(), ws.numel(), status_word(xyz.device).data_ptr(),
              _hip.stream_ptr(), nbytes=B * (12 * N + 8 * npoint))
    if _DEBUG:
        check_status(xyz.device)
    return idx, new_xyz


def ball_query(radius, nsample, xyz, new_xyz):
    """-> int32 [B,S,min(nsample,N)]; r^2 is float32(double(radius)**2) like the reference's comparison."""
    _hip.require_device(xyz, new_xyz)
    xyz, new_xyz = _hip.f32(xyz), _hip.f32(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    keff = min(int(nsample), N)
    lib = _hip.lib()
    out = torch.empty(B, S, keff, dtype=torch.int32, device=xyz.device)
    ws = _workspace(lib.pn2_ball_query_workspace_bytes(B, N, S, int(nsample)), xyz.device)
    r2 = ctypes.c_float(float(radius) ** 2).value  # float32(double(radius)**2), pointnet2_utils.py:107
    _hip.call("query_ball_point", lib.pn2_ball_query_f32, xyz.data_ptr(), *_strides3(xyz), new_xyz.data_ptr(),
              *_strides3(new_xyz), B, N, S, r2, int(nsample), out.data_ptr(), ws.data_ptr(), ws.numel(),
              _hip.stream_ptr(), nbytes=B * (12 * N + 12 * S + 8 * S * keff))
    return out


def three_nn(xyz1, xyz2, want_dist=False):
    """-> (idx int32 [B,N,3], weight f32 [B,N,3][, dist f32 [B,N,3]]); blocks.py:194-203."""
    _hip.require_device(xyz1, xyz2)
    xyz1, xyz2 = _hip.f32(xyz1), _hip.f32(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    if S < 3:
        raise RuntimeError(f"three_nn needs at least 3 sampled points, got {S}")
    idx = torch.empty(B, N, 3, dtype=torch.int32, device=xyz1.device)
    w = torch.empty(B, N, 3, dtype=torch.float32, device=xyz1.device)
    dist = torch.empty(B, N, 3, dtype=torch.float32, device=xyz1.device) if want_dist else None
    _hip.call("three_nn", _hip.lib().pn2_three_nn_f32, xyz1.data_ptr(), *_strides3(xyz1), xyz2.data_ptr(),
              *_strides3(xyz2), B, N, S, idx.data_ptr(), w.data_ptr(), _hip.ptr(dist), _hip.stream_ptr(),
              nbytes=B * (12 * N + 12 * S + N * 3 * (8 + 4)))
    return (idx, w, dist) if want_dist else (idx, w)


def _check_idx(idx, n, what):
    # The reference asserts the range on every call (pointnet2_utils.py:54), which costs a device sync.  The kernels do
    # not follow an out-of-range index and flag it in the status word (check_status()); PN2_DEBUG=1 asserts right here.
    if _DEBUG and idx.numel():
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= n:
            raise AssertionError(f"{what}: index out of range [{lo}, {hi}] for N={n}")


def _as_i32(idx):
    return idx if idx.dtype == torch.int32 else idx.to(torch.int32)


# ------------------------------------------------------------------------------------------- differentiable ops
class GatherPoints(torch.autograd.Function):
    """index_points(points [B,N,C], idx [B,...]) -> [B,...,C]   (pointnet2_utils.py:45-63)."""

    @staticmethod
    def forward(ctx, points, idx):
        _hip.require_device(points, idx)
        points = _hip.f32(points)
        B, N, C = points.shape
        idx32 = _as_i32(idx).contiguous()
        _check_idx(idx32, N, "index_points")
        S = idx32.numel() // B
        out = torch.empty(*idx.shape, C, dtype=torch.float32, device=points.device)
        _hip.call("index_points", _hip.lib().pn2_gather_f32, points.data_ptr(), *_strides3(points), idx32.data_ptr(), B, N,
                  S, C, out.data_ptr(), status_word(points.device).data_ptr(), _hip.stream_ptr(), nbytes=B * S * (8 + 8 * C))
        ctx.save_for_backward(idx32)
        ctx.dims = (B, N, S, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx32,) = ctx.saved_tensors
        B, N, S, C = ctx.dims
        dout = dout.contiguous()
        dpoints = torch.empty(B, N, C, dtype=torch.float32, device=dout.device)
        _hip.call("index_points_grad", _hip.lib().pn2_gather_grad_f32, dout.data_ptr(), idx32.data_ptr(), B, N, S, C,
                  dpoints.data_ptr(), _hip.stream_ptr(), nbytes=B * S * (8 + 8 * C) + 4 * B * N * C)
        return dpoints, None


class GroupPoints(torch.autograd.Function):
    """Gather + centre + concat of sample_and_group: -> [B,S,K,3+D] (channels [xyz_norm, feats], or
    [feats, xyz_norm] with xyz_last, the MSG order)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, feats, idx, xyz_last):
        _hip.require_device(xyz, new_xyz, feats, idx)
        xyz = _hip.f32(xyz)
        new_xyz = _hip.f32(new_xyz).contiguous()
        B, N, _ = xyz.shape
        _, S, K = idx.shape
        idx32 = _as_i32(idx).contiguous()
        _check_idx(idx32, N, "group_points")
        if feats is None:
            D, fptr, fs = 0, None, (0, 0, 0)
        else:
            feats = _hip.f32(feats)
            D, fptr, fs = feats.shape[2], feats.data_ptr(), _strides3(feats)
        out = torch.empty(B, S, K, 3 + D, dtype=torch.float32, device=xyz.device)
        _hip.call("group_points", _hip.lib().pn2_group_f32, xyz.data_ptr(), *_strides3(xyz), new_xyz.data_ptr(), fptr, *fs,
                  idx32.data_ptr(), B, N, S, K, D, int(bool(xyz_last)), out.data_ptr(), status_word(xyz.device).data_ptr(),
                  _hip.stream_ptr(), nbytes=B * S * K * (8 + 8 * (3 + D)))
        ctx.save_for_backward(idx32)
        ctx.dims = (B, N, S, K, D, int(bool(xyz_last)))
        return out

    @staticmethod
    def backward(ctx, dout):
        (idx32,) = ctx.saved_tensors
        B, N, S, K, D, xyz_last = ctx.dims
        if D == 0 or not ctx.needs_input_grad[2]:
            return None, None, None, None, None
        dout = dout.contiguous()
        dfeats = torch.empty(B, N, D, dtype=torch.float32, device=dout.device)
        _hip.call("group_points_grad", _hip.lib().pn2_group_grad_f32, dout.data_ptr(), idx32.data_ptr(), B, N, S, K, D,
                  xyz_last, dfeats.data_ptr(), _hip.stream_ptr(), nbytes=B * S * K * (8 + 8 * D) + 4 * B * N * D)
        return None, None, dfeats, None, None


class ThreeInterpolateConcat(torch.autograd.Function):
    """cat([points1, sum_k w_k * points2[idx_k]], -1) as one channels-last [B,N,D1+D2] buffer (blocks.py:204-208).
    points1 may be None (D1 = 0)."""

    @staticmethod
    def forward(ctx, points1, points2, idx, w):
        _hip.require_device(points1, points2, idx, w)
        points2 = _hip.f32(points2)
        B, S, D2 = points2.shape
        N = idx.shape[1]
        D1 = 0 if points1 is None else points1.shape[2]
        idx32 = _as_i32(idx).contiguous()
        _check_idx(idx32, S, "three_interpolate")
        w = w.contiguous()
        out = torch.empty(B, N, D1 + D2, dtype=torch.float32, device=points2.device)
        if D1:
            out[:, :, :D1].copy_(points1)
        _hip.call("three_interpolate", _hip.lib().pn2_three_interpolate_f32, points2.data_ptr(), *_strides3(points2),
                  idx32.data_ptr(), w.data_ptr(), B, N, S, D2, out.data_ptr(), D1 + D2, D1,
                  status_word(points2.device).data_ptr(), _hip.stream_ptr(), nbytes=B * N * (3 * 12 + 4 * D2) + 4 * B * S * D2)
        ctx.save_for_backward(idx32, w)
        ctx.dims = (B, N, S, D1, D2)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx32, w = ctx.saved_tensors
        B, N, S, D1, D2 = ctx.dims
        dout = dout.contiguous()
        d1 = dout[:, :, :D1] if (D1 and ctx.needs_input_grad[0]) else None
        d2 = None
        if ctx.needs_input_grad[1]:
            d2 = torch.empty(B, S, D2, dtype=torch.float32, device=dout.device)
            lib = _hip.lib()
            ws = _workspace(lib.pn2_three_interpolate_grad_workspace_bytes(B, N, S, D2), dout.device)
            _hip.call("three_interpolate_grad", lib.pn2_three_interpolate_grad_f32, dout.data_ptr(), D1 + D2, D1,
                      idx32.data_ptr(), w.data_ptr(), B, N, S, D2, d2.data_ptr(), ws.data_ptr(), ws.numel(), _hip.stream_ptr(),
                      nbytes=B * N * (3 * 12 + 4 * D2) + 4 * B * S * D2)
        return d1, d2, None, None
