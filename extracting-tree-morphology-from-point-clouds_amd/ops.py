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
                _hip.STATUS_BAD_INDEX: "a gather was handed an index outside [0, N)",
                _hip.STATUS_COOP_BARRIER: "a cooperative MLP chain launch: a workgroup never reached a layer barrier (GPU shared "
                                          "with another kernel?); the chain's results are garbage"}


def status_word(device):
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    t = _status_words.get(dev)
    if t is None:
        t = _status_words[dev] = torch.zeros(1, dtype=torch.int32, device=dev)
    return t


# Arrival counters of the cooperative chain launches (include/pn2_hip.h: pn2_coop.sync): 64 zeroed uint32 per (device, stream),
# allocated once; the kernels leave them zeroed when they finish.  One buffer per stream because launches of one stream are
# ordered, launches of two streams are not.
_coop_ctl = {}


def coop_ctl(device):
    """-> ctypes pointer to the pn2_coop of `device`'s current stream (kept alive here)."""
    dev = torch.device(device)
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    key = (dev.index, _hip._raw_stream(dev.index) if _hip._raw_stream is not None else torch.cuda.current_stream(dev).cuda_stream)
    rec = _coop_ctl.get(key)
    if rec is None:
        words = torch.zeros(64, dtype=torch.int32, device=dev)
        c = _hip.Coop(words.data_ptr(), status_word(dev).data_ptr(), 0, 0)
        rec = _coop_ctl[key] = (ctypes.pointer(c), c, words)
    return rec[0]


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
                if v & _hip.STATUS_COOP_BARRIER:      # a dead cooperative launch leaves its arrival counters behind
                    for (di, _), rec in _coop_ctl.items():
                        if di == t.device.index:
                            rec[2].zero_()
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


# The cell structure the last ordered FPS call left in its workspace (include/pn2_hip.h: pn2_fps_*_offset), remembered for the
# cloud tensor it sampled: three_nn walks its dense points in that cell order and ball_query searches the cells when they are
# handed the SAME cloud (set abstraction level 1 and feature propagation level 1 both see the level-0 cloud).
# "The same cloud" must be airtight -- a ball query over another cloud's cells is simply wrong --, so an entry holds a strong
# reference to the cloud's STORAGE (the allocator cannot give that memory to another tensor while the entry lives: equal
# data_ptr / shape / strides then means a view of the very same storage) and the storage's version counter (an in-place
# write to the cloud bumps it).  One entry: it keeps a workspace (~8 MB at 262144 points) and one cloud alive.
class _CloudMemo:
    def __init__(self):
        self.key = self.storage = self.version = self.order = self.cells = None

    @staticmethod
    def _key(xyz):
        return (xyz.data_ptr(), tuple(xyz.shape), tuple(xyz.stride()), xyz.device.index)

    def remember(self, xyz, order, cells):
        self.key, self.storage, self.version = self._key(xyz), xyz.untyped_storage(), xyz._version
        self.order, self.cells = order, cells

    def lookup(self, xyz):
        """-> (order, cells) or (None, None)"""
        if self.key is None or self._key(xyz) != self.key or xyz._version != self.version:
            return None, None
        if xyz.untyped_storage().data_ptr() != self.storage.data_ptr():
            return None, None
        return self.order, self.cells

    def clear(self):
        self.__init__()


_cloud_memo = _CloudMemo()


def last_fps_rounds():
    """Exchanges (rounds of up to 16 samples) the last ordered FPS call needed for its first cloud, or None; synchronises."""
    r = getattr(_cloud_memo, "rounds", None)
    return None if r is None else int(r.item())
_NO_ORDER = ctypes.c_size_t(-1).value


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
    off = lib.pn2_fps_order_offset(B, N, npoint)
    if off != _NO_ORDER and not os.environ.get("PN2_FPS_NO_SORT"):
        boff, coff = lib.pn2_fps_box_offset(B, N, npoint), lib.pn2_fps_cellstart_offset(B, N, npoint)
        xoff = lib.pn2_fps_sorted_xyz_offset(B, N, npoint)
        roff = lib.pn2_fps_rounds_offset(B, N, npoint)
        _cloud_memo.rounds = ws[roff:roff + 4].view(torch.int32)
        _cloud_memo.remember(xyz, ws[off:off + 4 * B * N].view(torch.int32).view(B, N),
                             (ws[boff:boff + 32 * B].view(torch.int32), ws[coff:coff + 4 * 4097 * B].view(torch.int32),
                              ws[xoff:xoff + 12 * B * N].view(torch.float32)))
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
    r2 = ctypes.c_float(float(radius) ** 2).value  # float32(double(radius)**2), pointnet2_utils.py:107
    order, cells = _cloud_memo.lookup(xyz)
    if cells is not None and not os.environ.get("PN2_BQ_NO_CELLS"):
        # the cloud was just sampled by an ordered FPS call: sparse balls are searched in the cells it left behind
        box, cellstart, sorted_xyz = cells
        _hip.call("query_ball_point", lib.pn2_ball_query_cells_f32, xyz.data_ptr(), *_strides3(xyz), new_xyz.data_ptr(),
                  *_strides3(new_xyz), B, N, S, r2, int(nsample), box.data_ptr(), cellstart.data_ptr(),
                  order.data_ptr(), sorted_xyz.data_ptr(), out.data_ptr(), _hip.stream_ptr(),
                  nbytes=B * (12 * N + 12 * S + 8 * S * keff))
        return out
    ws = _workspace(lib.pn2_ball_query_workspace_bytes(B, N, S, int(nsample)), xyz.device)
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
    order = _cloud_memo.lookup(xyz1)[0] if not os.environ.get("PN2_TNN_NO_ORDER") else None
    _hip.call("three_nn", _hip.lib().pn2_three_nn_f32, xyz1.data_ptr(), *_strides3(xyz1), xyz2.data_ptr(),
              *_strides3(xyz2), B, N, S, idx.data_ptr(), w.data_ptr(), _hip.ptr(dist), _hip.ptr(order), _hip.stream_ptr(),
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
        if D1:   # the skip rows are copied by the same launch
            points1 = _hip.f32(points1)
            _hip.call("three_interpolate", _hip.lib().pn2_three_interpolate_concat_f32, points1.data_ptr(), *_strides3(points1), D1,
                      points2.data_ptr(), *_strides3(points2), idx32.data_ptr(), w.data_ptr(), B, N, S, D2, out.data_ptr(), D1 + D2,
                      D1, status_word(points2.device).data_ptr(), _hip.stream_ptr(),
                      nbytes=B * N * (3 * 12 + 4 * D2 + 8 * D1) + 4 * B * S * D2)
        else:
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


# ------------------------------------------------------------------------------- ragged clouds (whole-tree execution)
class RaggedClouds:
    """Level-0 input of a whole tree: C clouds of different lengths in flat CHANNEL-FIRST buffers (include/pn2_hip.h,
    "Ragged clouds") -- the reference's per-mini-batch tensors [B_j, CH, N_j] laid end to end, no transposition.
    xyz_cf: float32 [3 * rows]; feats_cf: float32 [D * rows] or None; lengths: host list of C ints."""

    def __init__(self, xyz_cf, feats_cf, dim_feat, lengths):
        self.xyz_cf, self.feats_cf, self.D = xyz_cf, feats_cf, int(dim_feat) if feats_cf is not None else 0
        self.lengths = [int(n) for n in lengths]
        self.C, self.n_max, self.n_min = len(self.lengths), max(self.lengths), min(self.lengths)
        off = [0]
        for n in self.lengths:
            off.append(off[-1] + n)
        self.rows = off[-1]
        if self.rows >= 2 ** 31 // 4:
            raise RuntimeError("RaggedClouds: too many points for one pass")
        self.coff_host = off
        self.coff = torch.tensor(off, dtype=torch.int32).to(xyz_cf.device, non_blocking=True)
        self._row_cloud = None

    @property
    def row_cloud(self):
        """int32 [rows]: the cloud of every packed row (built on first use; pn2_interp_bn_fwd_f32 reads it)."""
        if self._row_cloud is None:
            dev = self.xyz_cf.device
            reps = torch.tensor(self.lengths, dtype=torch.int64).to(dev, non_blocking=True)
            self._row_cloud = torch.repeat_interleave(torch.arange(self.C, dtype=torch.int32, device=dev), reps,
                                                      output_size=self.rows)
        return self._row_cloud


def fps_ragged(rc, npoint, start):
    """-> (idx int32 [C,npoint] cloud-local, new_xyz [C,npoint,3]); same samples as per-cloud farthest_point_sample."""
    lib = _hip.lib()
    dev = rc.xyz_cf.device
    idx = torch.empty(rc.C, npoint, dtype=torch.int32, device=dev)
    new_xyz = torch.empty(rc.C, npoint, 3, dtype=torch.float32, device=dev)
    nbytes = lib.pn2_fps_ragged_workspace_bytes(rc.C, rc.n_max, npoint)
    if nbytes == 0:
        raise RuntimeError(f"fps_ragged: unsupported size (longest cloud {rc.n_max} > 16384?)")
    ws = _workspace(nbytes, dev)      # single-workgroup clouds: no hand-off, the error word is never touched
    start = start.to(device=dev, dtype=torch.int64).contiguous()
    _hip.call("farthest_point_sample", lib.pn2_fps_ragged_f32, rc.xyz_cf.data_ptr(), rc.coff.data_ptr(), rc.C, rc.n_max,
              npoint, start.data_ptr(), idx.data_ptr(), new_xyz.data_ptr(), ws.data_ptr(), ws.numel(), _hip.stream_ptr(),
              nbytes=12 * rc.rows + 8 * rc.C * npoint)
    return idx, new_xyz


def ball_query_ragged(radius, nsample, rc, new_xyz):
    """-> int32 [C,S,nsample] cloud-local indices; every cloud must hold at least nsample points."""
    if rc.n_min < nsample:
        raise RuntimeError("ball_query_ragged needs every cloud to hold at least nsample points")
    new_xyz = new_xyz.contiguous()
    S = new_xyz.shape[1]
    out = torch.empty(rc.C, S, int(nsample), dtype=torch.int32, device=new_xyz.device)
    r2 = ctypes.c_float(float(radius) ** 2).value
    _hip.call("query_ball_point", _hip.lib().pn2_ball_query_ragged_f32, rc.xyz_cf.data_ptr(), rc.coff.data_ptr(),
              new_xyz.data_ptr(), rc.C, rc.n_max, S, r2, int(nsample), out.data_ptr(), _hip.stream_ptr(),
              nbytes=12 * rc.rows + rc.C * (12 * S + 8 * S * int(nsample)))
    return out


def group_ragged(rc, new_xyz, idx, xyz_last=False):
    """sample_and_group's gather + centre + concat on ragged level-0 clouds -> [C,S,K,3+D].  The level-0 inputs are
    data (no gradient flows into them), so this is a plain function."""
    new_xyz = new_xyz.contiguous()
    _, S, K = idx.shape
    out = torch.empty(rc.C, S, K, 3 + rc.D, dtype=torch.float32, device=new_xyz.device)
    _hip.call("group_points", _hip.lib().pn2_group_ragged_f32, rc.xyz_cf.data_ptr(), _hip.ptr(rc.feats_cf), rc.D,
              rc.coff.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), rc.C, S, K, int(bool(xyz_last)), out.data_ptr(),
              status_word(new_xyz.device).data_ptr(), _hip.stream_ptr(), nbytes=rc.C * S * K * (8 + 8 * (3 + rc.D)))
    return out


def three_nn_ragged(rc, xyz2):
    """dense side = the ragged level-0 clouds, sampled side xyz2 [C,S,3] -> (idx int32 [rows,3], weight [rows,3])."""
    xyz2 = xyz2.contiguous()
    S = xyz2.shape[1]
    if S < 3:
        raise RuntimeError(f"three_nn needs at least 3 sampled points, got {S}")
    idx = torch.empty(rc.rows, 3, dtype=torch.int32, device=xyz2.device)
    w = torch.empty(rc.rows, 3, dtype=torch.float32, device=xyz2.device)
    _hip.call("three_nn", _hip.lib().pn2_three_nn_ragged_f32, rc.xyz_cf.data_ptr(), rc.coff.data_ptr(), xyz2.data_ptr(), rc.C,
              rc.n_max, S, idx.data_ptr(), w.data_ptr(), _hip.stream_ptr(), nbytes=48 * rc.rows + 12 * rc.C * S)
    return idx, w


class ThreeInterpolateRagged(torch.autograd.Function):
    """points2 [C,S,D] -> packed rows [rows, D] = sum_k w_k * points2[cloud(row), idx_k]  (blocks.py:204 on ragged
    clouds; there is no skip connection at level 0: fp1 is called with points1 = None, PointNet2.py:156)."""

    @staticmethod
    def forward(ctx, points2, idx, w, rc):
        points2 = _hip.f32(points2).contiguous()
        C, S, D = points2.shape
        out = torch.empty(rc.rows, D, dtype=torch.float32, device=points2.device)
        _hip.call("three_interpolate", _hip.lib().pn2_three_interpolate_ragged_f32, points2.data_ptr(), idx.data_ptr(),
                  w.data_ptr(), rc.coff.data_ptr(), C, rc.n_max, rc.rows, S, D, out.data_ptr(), D, 0,
                  status_word(points2.device).data_ptr(), _hip.stream_ptr(), nbytes=rc.rows * (36 + 4 * D) + 4 * C * S * D)
        ctx.save_for_backward(idx, w)
        ctx.rc, ctx.dims = rc, (C, S, D)
        return out

    @staticmethod
    def backward(ctx, dout):
        idx, w = ctx.saved_tensors
        rc, (C, S, D) = ctx.rc, ctx.dims
        dout = dout.contiguous()
        d2 = torch.empty(C, S, D, dtype=torch.float32, device=dout.device)
        lib = _hip.lib()
        ws = _workspace(lib.pn2_three_interpolate_grad_ragged_workspace_bytes(C, rc.rows, S), dout.device)
        _hip.call("three_interpolate_grad", lib.pn2_three_interpolate_grad_ragged_f32, dout.data_ptr(), D, 0, idx.data_ptr(),
                  w.data_ptr(), rc.coff.data_ptr(), C, rc.n_max, rc.rows, S, D, d2.data_ptr(), ws.data_ptr(), ws.numel(),
                  _hip.stream_ptr(), nbytes=rc.rows * (36 + 4 * D) + 4 * C * S * D)
        return d2, None, None, None
