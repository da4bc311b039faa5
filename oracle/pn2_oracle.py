"""ctypes/numpy front end of the C oracle (oracle/pn2_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; the product package never imports anything from oracle/.  Parity status: pinned by the golden
vectors in tests/golden/ (generated from the imported reference, see tests/golden/make_golden.py).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpn2oracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i64p = ctypes.POINTER(ctypes.c_int64)


def build(force=False):
    """Compile the oracle with gcc (idempotent)."""
    src = os.path.join(_HERE, "pn2_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
    return _lib


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int64)
    return a, a.ctypes.data_as(_i64p)


def r2_of(radius):
    """The reference compares fp32 distances with float32(double(radius)**2) (pointnet2_utils.py:107)."""
    return np.float32(float(radius) ** 2)


def square_distance(src, dst):
    src, ps = _f(src)
    dst, pd = _f(dst)
    B, N, _ = src.shape
    M = dst.shape[1]
    out = np.empty((B, N, M), np.float32)
    lib().pn2o_square_distance(ps, pd, B, N, M, out.ctypes.data_as(_f32p))
    return out


def farthest_point_sample(xyz, npoint, start):
    xyz, px = _f(xyz)
    start, pst = _i(start)
    B, N, _ = xyz.shape
    out = np.empty((B, npoint), np.int64)
    lib().pn2o_fps(px, B, N, int(npoint), pst, out.ctypes.data_as(_i64p))
    return out


def query_ball_point(radius, nsample, xyz, new_xyz):
    xyz, px = _f(xyz)
    new_xyz, pq = _f(new_xyz)
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    keff = min(int(nsample), N)
    out = np.empty((B, S, keff), np.int64)
    lib().pn2o_ball_query(px, pq, B, N, S, ctypes.c_float(r2_of(radius)), int(nsample),
                          out.ctypes.data_as(_i64p))
    return out


def index_points(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, N, C = points.shape
    S = int(np.prod(idx.shape[1:]))
    out = np.empty((B, S, C), np.float32)
    lib().pn2o_gather(pp, pi, B, N, C, S, out.ctypes.data_as(_f32p))
    return out.reshape(*idx.shape, C)


def index_points_grad(dout, idx, N):
    idx, pi = _i(idx)
    B = idx.shape[0]
    S = int(np.prod(idx.shape[1:]))
    dout, pd = _f(np.asarray(dout).reshape(B, S, -1))
    C = dout.shape[-1]
    out = np.empty((B, N, C), np.float32)
    lib().pn2o_gather_grad(pd, pi, B, N, C, S, out.ctypes.data_as(_f32p))
    return out


def group(xyz, new_xyz, feats, idx, xyz_last=False):
    xyz, px = _f(xyz)
    new_xyz, pq = _f(new_xyz)
    idx, pi = _i(idx)
    B, N, _ = xyz.shape
    _, S, K = idx.shape
    if feats is None:
        D, pf = 0, None
    else:
        feats, pf = _f(feats)
        D = feats.shape[-1]
    out = np.empty((B, S, K, 3 + D), np.float32)
    lib().pn2o_group(px, pq, pf, pi, B, N, S, K, D, int(bool(xyz_last)), out.ctypes.data_as(_f32p))
    return out


def three_nn(xyz1, xyz2):
    xyz1, p1 = _f(xyz1)
    xyz2, p2 = _f(xyz2)
    B, N, _ = xyz1.shape
    S = xyz2.shape[1]
    kk = min(3, S)
    idx = np.empty((B, N, kk), np.int64)
    dist = np.empty((B, N, kk), np.float32)
    lib().pn2o_three_nn(p1, p2, B, N, S, idx.ctypes.data_as(_i64p), dist.ctypes.data_as(_f32p))
    return dist, idx


def three_weights(dist):
    dist, pd = _f(dist)
    assert dist.shape[-1] == 3
    w = np.empty_like(dist)
    lib().pn2o_three_weights(pd, ctypes.c_size_t(dist.size // 3), w.ctypes.data_as(_f32p))
    return w


def three_interpolate(points2, idx, w):
    points2, pp = _f(points2)
    idx, pi = _i(idx)
    w, pw = _f(w)
    B, S, D = points2.shape
    N = idx.shape[1]
    out = np.empty((B, N, D), np.float32)
    lib().pn2o_three_interpolate(pp, pi, pw, B, N, S, D, out.ctypes.data_as(_f32p))
    return out


def three_interpolate_grad(dout, idx, w, S):
    dout, pd = _f(dout)
    idx, pi = _i(idx)
    w, pw = _f(w)
    B, N, D = dout.shape
    out = np.empty((B, S, D), np.float32)
    lib().pn2o_three_interpolate_grad(pd, pi, pw, B, N, S, D, out.ctypes.data_as(_f32p))
    return out


# ------------------------------------------------------------------------------------------- kNN feature helpers
_f64p = ctypes.POINTER(ctypes.c_double)


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_f64p)


def knn_radius(points, k, radius=None):
    """-> (nn_idx int64 [N,k], nn_d2 [N,k], counts int64 [N] or None): Modules/Features.py:120-124, 170-172."""
    pts, pp = _d(np.asarray(points)[:, :3])
    n = pts.shape[0]
    idx = np.empty((n, k), np.int64)
    d2 = np.empty((n, k), np.float64)
    cnt = np.empty(n, np.int64) if radius is not None else None
    lib().pn2o_knn_radius_f64(pp, n, k, ctypes.c_double(float(radius) ** 2 if radius is not None else -1.0),
                              idx.ctypes.data_as(_i64p), d2.ctypes.data_as(_f64p),
                              cnt.ctypes.data_as(_i64p) if cnt is not None else None)
    return idx, d2, cnt


def cov_eig(points, nn_idx, k):
    """-> (evals [N,3] ascending, evecs [N,3,3] rows): Modules/Features.py:126-130, 151-156."""
    pts, pp = _d(np.asarray(points)[:, :3])
    idx, pi = _i(nn_idx)
    n = pts.shape[0]
    evals = np.empty((n, 3), np.float64)
    evecs = np.empty((n, 3, 3), np.float64)
    lib().pn2o_cov_eig_f64(pp, n, pi, idx.shape[1], k, evals.ctypes.data_as(_f64p), evecs.ctypes.data_as(_f64p))
    return evals, evecs


def add_features(labeled_cloud):
    """Modules/Features.py:178-229 with every feature on; the normals' signs follow this oracle's convention."""
    pts = np.asarray(labeled_cloud)[:, :3]
    idx, _, cnt = knn_radius(pts, 15, 0.1)
    _, evecs15 = cov_eig(pts, idx, 15)
    evals10, _ = cov_eig(pts, idx, 10)
    normals = np.stack([evecs15[:, 2, 2], evecs15[:, 1, 2], evecs15[:, 0, 2]], axis=1)
    curvature = evals10[:, 0] / (evals10.sum(1) + 1e-6)
    height = (pts[:, 2] - pts[:, 2].min()) / (pts[:, 2].max() - pts[:, 2].min())
    vert = np.abs(normals @ np.array([0, 0, 1]))
    dist = np.linalg.norm(pts[:, :2] - pts[:, :2].mean(0), axis=1)
    return np.concatenate([labeled_cloud, normals, curvature[:, None], cnt[:, None].astype(np.float64), height[:, None],
                           vert[:, None], dist[:, None]], axis=1)


# --------------------------------------------------------------------------------------------- round 2 (parity unpinned)
def cylinder_axes(start, end):
    """generate_offset_cloud_cuda_batched's preparation (Modules/Projection.py:126-132), float32: axis = end - start,
    length = |axis|, unit = axis / max(length, 1e-8)."""
    start = np.ascontiguousarray(start, np.float32)
    axis = np.ascontiguousarray(end, np.float32) - start
    length = np.sqrt((axis[:, 0] * axis[:, 0] + axis[:, 1] * axis[:, 1]) + axis[:, 2] * axis[:, 2]).astype(np.float32)
    safe = np.where(length < np.float32(1e-8), np.float32(1e-8), length).astype(np.float32)
    return start, (axis / safe[:, None]).astype(np.float32), length


def cylinder_project(points, start, unit, length, radius, ids=None, mantle=True):
    """closest_cylinder_cuda_batch (Modules/Projection.py:19-114) -> (ids int32 [N], distances f32 [N], offsets f32 [N,3])."""
    points, pp = _f(points)
    start, ps = _f(start)
    unit, pu = _f(unit)
    length, pl = _f(length)
    radius, pr = _f(radius)
    N, M = points.shape[0], start.shape[0]
    out_id = np.empty(N, np.int32)
    out_d = np.empty(N, np.float32)
    out_o = np.empty((N, 3), np.float32)
    i32p = ctypes.POINTER(ctypes.c_int32)
    idp = None
    if ids is not None:
        ids = np.ascontiguousarray(ids, np.int32)
        idp = ids.ctypes.data_as(i32p)
    lib().pn2o_cylinder_project(pp, N, ps, pu, pl, pr, idp, M, int(bool(mantle)), out_id.ctypes.data_as(i32p),
                                out_d.ctypes.data_as(_f32p), out_o.ctypes.data_as(_f32p))
    return out_id, out_d, out_o
