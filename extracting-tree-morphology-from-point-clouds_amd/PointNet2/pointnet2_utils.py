"""L0 ops with the reference's Python signatures (Modules/PointNet2/pointnet2_utils.py), served by libpn2hip.

Every function takes and returns the same shapes and dtypes as its reference namesake (indices are
``torch.long`` at this API); the device work is one C-ABI call each (include/pn2_hip.h).  Tensors must live on
a HIP device -- there is no CPU path.
"""
import torch

from .. import ops

__all__ = ["square_distance", "index_points", "farthest_point_sample", "query_ball_point", "sample_and_group",
           "sample_and_group_all", "three_nn", "three_interpolate"]


def square_distance(src, dst):
    """src [B,N,3], dst [B,M,3] -> [B,N,M] expanded squared distance (reference :21-42), bit-identical
    rounding: ((-2*dot) + |src|^2) + |dst|^2 with dot = fma(z,z', fma(y,y', x*x'))."""
    return ops.square_distance(src, dst)


def index_points(points, idx):
    """points [B,N,C], idx [B,S] or [B,S,K] (any integer dtype) -> [B,S,(K,)C] (reference :45-63).
    The reference's range assert forces a device sync; it is checked only with PN2_DEBUG=1."""
    return ops.GatherPoints.apply(points, idx)


class StartIndexFeed:
    """Persistent pinned host slots (+ their device twins) for the FPS start indices, so that a forward pass can be
    captured in a HIP graph: a capture may not allocate pinned memory, and a replay must see NEW random starts.
    While a feed is active every `_draw_start` call takes the next slot -- the draw lands in the slot's pinned buffer
    and a host-to-device copy from that fixed address is enqueued (a memcpy node under capture).  `redraw()` before
    each replay draws again, slot by slot, i.e. in the reference's order (sa1, sa2, ... per forward)."""
    active = None

    def __init__(self):
        self.slots, self.cursor = [], 0

    def begin_pass(self):
        self.cursor = 0

    def __enter__(self):
        StartIndexFeed.active = self
        self.begin_pass()
        return self

    def __exit__(self, *exc):
        StartIndexFeed.active = None

    def take(self, B, N, device):
        if self.cursor == len(self.slots):          # first (uncaptured) pass: create the slot
            self.slots.append((torch.empty(B, dtype=torch.long, pin_memory=True),
                               torch.empty(B, dtype=torch.long, device=device), N))
        host, dev, n = self.slots[self.cursor]
        if host.numel() != B or n != N or dev.device != torch.device(device):
            raise RuntimeError("StartIndexFeed: the captured forward changed shape")
        self.cursor += 1
        torch.randint(0, N, (B,), dtype=torch.long, out=host)
        dev.copy_(host, non_blocking=True)
        return dev

    def redraw(self):
        for host, _, n in self.slots:
            torch.randint(0, n, (host.numel(),), dtype=torch.long, out=host)


class start_batch:
    """Context: the start indices of the FPS calls made inside -- `sizes` = the cloud size of each, in call order -- are
    drawn up front, one torch.randint per call in that order (the global CPU generator sees exactly the draws it would
    see anyway), and reach the device in ONE copy instead of one 8-byte copy per level.  A call that does not match the
    announced (B, N) sequence falls back to its own draw."""
    active = None

    def __init__(self, B, sizes, device):
        self.todo = None
        if StartIndexFeed.active is None and start_batch.active is None and torch.device(device).type == "cuda":
            host = torch.empty(len(sizes), B, dtype=torch.long, pin_memory=True)
            for row, n in zip(host, sizes):
                torch.randint(0, int(n), (B,), dtype=torch.long, out=row)
            dev = host.to(device, non_blocking=True)
            self.todo = [(B, int(n), dev[i]) for i, n in enumerate(sizes)]

    def __enter__(self):
        if self.todo is not None:
            start_batch.active = self
        return self

    def __exit__(self, *exc):
        if start_batch.active is self:
            start_batch.active = None

    def take(self, B, N):
        if self.todo and self.todo[0][0] == B and self.todo[0][1] == N:
            return self.todo.pop(0)[2]
        self.todo = []          # out of step: the remaining announced draws were made already, later calls draw for themselves
        return None


def _draw_start(B, N, device):
    # One draw per call from the global CPU generator, then moved to the device: same RNG stream
    # consumption as the reference (:79), so seeded runs pick the same first centroid.
    if StartIndexFeed.active is not None:
        return StartIndexFeed.active.take(B, N, device)
    if start_batch.active is not None:
        got = start_batch.active.take(B, N)
        if got is not None:
            return got
    return torch.randint(0, N, (B,), dtype=torch.long, pin_memory=True).to(device, non_blocking=True)


def farthest_point_sample(xyz, npoint):
    """xyz [B,N,3] -> centroid indices [B,npoint] long (reference :66-89)."""
    B, N, _ = xyz.shape
    idx, _ = ops.furthest_point_sample(xyz, npoint, _draw_start(B, N, xyz.device))
    return idx.long()


def query_ball_point(radius, nsample, xyz, new_xyz):
    """-> [B,S,min(nsample,N)] long: the first nsample indices in ascending order with d <= r^2, short rows
    padded with their first hit, empty balls filled with the nearest point (reference :92-136)."""
    return ops.ball_query(radius, nsample, xyz, new_xyz).long()


def _sample_and_group_i32(npoint, radius, nsample, xyz, points, xyz_last=False):
    """Fused core shared with blocks.py: int32 indices end to end, centroids come out of the FPS kernel."""
    B, N, _ = xyz.shape
    fps_idx, new_xyz = ops.furthest_point_sample(xyz, npoint, _draw_start(B, N, xyz.device))
    idx = ops.ball_query(radius, nsample, xyz, new_xyz)
    new_points = ops.GroupPoints.apply(xyz, new_xyz, points, idx, xyz_last)
    return new_xyz, new_points, idx, fps_idx


def sample_and_group(npoint, radius, nsample, xyz, points, returnfps=False):
    """xyz [B,N,3], points [B,N,D] or None -> new_xyz [B,npoint,3], new_points [B,npoint,nsample,3+D] with
    channels [xyz - centroid, points] (reference :139-167)."""
    new_xyz, new_points, idx, fps_idx = _sample_and_group_i32(npoint, radius, nsample, xyz, points)
    if returnfps:
        grouped_xyz = index_points(xyz, idx)
        return new_xyz, new_points, grouped_xyz, fps_idx.long()
    return new_xyz, new_points


def sample_and_group_all(xyz, points):
    """One group holding the whole cloud (reference :170-187); unused by every depth table."""
    B, N, C = xyz.shape
    new_xyz = torch.zeros(B, 1, C, device=xyz.device)
    grouped = xyz.reshape(B, 1, N, C)
    if points is not None:
        grouped = torch.cat([grouped, points.reshape(B, 1, N, -1)], dim=-1)
    return new_xyz, grouped


def three_nn(xyz1, xyz2):
    """xyz1 [B,N,3], xyz2 [B,S,3] -> (idx [B,N,3] long, weight [B,N,3]): the three nearest sampled points and
    their normalised inverse-distance weights (locals of reference blocks.py:194-203).  Equal distances keep
    the lower index first."""
    idx, w = ops.three_nn(xyz1, xyz2)
    return idx.long(), w


def three_interpolate(points2, idx, weight):
    """points2 [B,S,D] -> [B,N,D] = sum_k weight_k * points2[idx_k] (reference blocks.py:204)."""
    return ops.ThreeInterpolateConcat.apply(None, points2, idx, weight)
