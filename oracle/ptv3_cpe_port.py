"""TEST INFRASTRUCTURE (oracle): CPU restatement of the submanifold 3 x 3 x 3 sparse convolution PointTransformerV3 uses as its
conditional positional encoding (/root/reference/Modules/PointTransformerV3/blocks.py:561-568: spconv.SubMConv3d(channels,
channels, kernel_size=3, bias=True) on the voxels of Point.sparsify :153-190, indices = [batch, grid_coord]).
PARITY UNPINNED against spconv (a CUDA-only dependency that exists neither here nor on the GPU box; the reference holds no
fixture of it).  The restatement follows the operator's definition and is PINNED to torch.nn.functional.conv3d on the densified
grid (dense_reference below; tests/test_ptv3_cpe.py).  Only tests/ and tools' cpu-baseline legs may import this module."""
import numpy as np
import torch


def subm_neighbors(batch, grid_coord, kernel_size=3):
    """-> int32 [N, k^3]: column ((dx + r) * k + (dy + r)) * k + (dz + r), r = k // 2, = index of the active voxel of the same cloud
    at grid_coord[i] + (dx, dy, dz), or -1.  Duplicate voxels are represented by their lowest index."""
    grid = np.asarray(grid_coord, dtype=np.int64)
    N, k = len(grid), int(kernel_size)
    r = k // 2
    b = np.zeros(N, np.int64) if batch is None else np.asarray(batch, dtype=np.int64)
    table = {}
    for i in range(N - 1, -1, -1):                      # lowest index wins
        table[(int(b[i]), int(grid[i, 0]), int(grid[i, 1]), int(grid[i, 2]))] = i
    nbr = np.full((N, k ** 3), -1, np.int32)
    for i in range(N):
        bi, x, y, z = int(b[i]), int(grid[i, 0]), int(grid[i, 1]), int(grid[i, 2])
        for dx in range(-r, r + 1):
            for dy in range(-r, r + 1):
                for dz in range(-r, r + 1):
                    j = table.get((bi, x + dx, y + dy, z + dz))
                    if j is not None:
                        nbr[i, ((dx + r) * k + (dy + r)) * k + (dz + r)] = j
    return nbr


def subm_conv(feat, nbr, weight, bias):
    """feat [N, C_in], nbr [N, k^3], weight [C_out, k, k, k, C_in] (spconv 2.x), bias [C_out] or None -> [N, C_out] float64:
    out[i] = bias + sum_d W[:, d, :] feat[nbr[i, d]]."""
    feat = torch.as_tensor(feat, dtype=torch.float64)
    noff = nbr.shape[1]
    w = torch.as_tensor(weight, dtype=torch.float64).reshape(weight.shape[0], noff, weight.shape[-1])   # [C_out, k^3, C_in]
    nbr = torch.as_tensor(np.asarray(nbr), dtype=torch.long)
    out = torch.zeros(feat.shape[0], w.shape[0], dtype=torch.float64)
    for d in range(noff):
        j = nbr[:, d]
        ok = j >= 0
        out[ok] += feat[j[ok]] @ w[:, d, :].t()
    if bias is not None:
        out += torch.as_tensor(bias, dtype=torch.float64)
    return out


def dense_reference(batch, grid_coord, feat, weight, bias):
    """The definition the restatement is pinned to: scatter the voxels into a dense [B, C_in, X, Y, Z] grid (zeros elsewhere),
    torch.nn.functional.conv3d with padding 1 (cross-correlation), read the result back at the active voxels."""
    grid = torch.as_tensor(np.asarray(grid_coord), dtype=torch.long)
    N = grid.shape[0]
    b = torch.zeros(N, dtype=torch.long) if batch is None else torch.as_tensor(np.asarray(batch), dtype=torch.long)
    feat = torch.as_tensor(feat, dtype=torch.float64)
    B = int(b.max()) + 1
    X, Y, Z = (int(grid[:, a].max()) + 1 for a in range(3))
    dense = torch.zeros(B, feat.shape[1], X, Y, Z, dtype=torch.float64)
    dense[b, :, grid[:, 0], grid[:, 1], grid[:, 2]] = feat
    w = torch.as_tensor(weight, dtype=torch.float64).permute(0, 4, 1, 2, 3).contiguous()      # [C_out, C_in, k, k, k]
    out = torch.nn.functional.conv3d(dense, w, None if bias is None else torch.as_tensor(bias, dtype=torch.float64),
                                     padding=weight.shape[1] // 2)
    return out[b, :, grid[:, 0], grid[:, 1], grid[:, 2]]
