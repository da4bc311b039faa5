"""TEST INFRASTRUCTURE ONLY (oracle): a numpy restatement of the reference's point-cloud serialization codes,
Modules/PointTransformerV3/serialization -- only tests/ and benchmark baselines may import it.

Pinned: tests/golden/serialization.npz holds inputs and the codes the reference's own module produces (generated in this
container by tests/golden/make_serialization.py, which imports the three serialization files on the CPU).

  z_order   z_order.py:45-55 (bit i of x, y, z -> bit 3i+2, 3i+1, 3i), batch at bit 3*depth (default.py:21-23)
  hilbert   hilbert.py:91-198: Skilling's axes -> transpose walk (:157-177), interleave axis 0 first (:180), Gray-decode the
            3*depth-bit string by prefix xor (:183 / :66-88)
  decode    hilbert.py:201-302, z_order.py:106-125
"""
import numpy as np

ORDERS = ("z", "z-trans", "hilbert", "hilbert-trans")


def _interleave(a, b, c, depth):
    code = np.zeros(a.shape, np.uint64)
    for i in range(depth):
        bit = np.uint64(1) << np.uint64(i)
        code |= ((a & bit) << np.uint64(2 * i + 2)) | ((b & bit) << np.uint64(2 * i + 1)) | ((c & bit) << np.uint64(2 * i))
    return code


def _deinterleave(code, depth):
    out = [np.zeros(code.shape, np.uint64) for _ in range(3)]
    for i in range(depth):
        for d in range(3):
            out[d] |= ((code >> np.uint64(3 * i + 2 - d)) & np.uint64(1)) << np.uint64(i)
    return out


def _skilling(X, depth, forward):
    X = [x.copy() for x in X]
    bits = range(depth - 1, -1, -1) if forward else range(depth)
    dims = range(3) if forward else range(2, -1, -1)
    for k in bits:
        q = np.uint64(1) << np.uint64(k)
        p = q - np.uint64(1)
        for i in dims:
            on = (X[i] & q) != 0
            t = np.where(on, np.uint64(0), (X[0] ^ X[i]) & p)
            X[0] = np.where(on, X[0] ^ p, X[0] ^ t)
            if i:
                X[i] = X[i] ^ t
    return X


def encode(grid_coord, batch=None, depth=16, order="z"):
    g = np.asarray(grid_coord).astype(np.int64).astype(np.uint64) & np.uint64((1 << depth) - 1)
    a, b, c = (g[:, 1], g[:, 0], g[:, 2]) if order.endswith("-trans") else (g[:, 0], g[:, 1], g[:, 2])
    if order.startswith("z"):
        code = _interleave(a, b, c, depth)
    else:
        X = _skilling([a, b, c], depth, True)
        code = _interleave(X[0], X[1], X[2], depth)
        for s in (1, 2, 4, 8, 16, 32):
            code ^= code >> np.uint64(s)
    if batch is not None:
        code |= np.asarray(batch).astype(np.int64).astype(np.uint64) << np.uint64(3 * depth)
    return code.astype(np.int64)


def decode(code, depth=16, order="z"):
    code = np.asarray(code).astype(np.int64)
    batch = code >> (3 * depth)
    key = code.astype(np.uint64) & np.uint64((1 << (3 * depth)) - 1)
    if order == "hilbert":
        key = key ^ (key >> np.uint64(1))
        X = _skilling(_deinterleave(key, depth), depth, False)
    else:
        X = _deinterleave(key, depth)
    return np.stack(X, -1).astype(np.int64), batch
