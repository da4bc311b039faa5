"""torch-CPU restatement of the reference's PointNet++ path -- TEST INFRASTRUCTURE ONLY.

Purpose: (1) the "PyTorch-CPU reference path" column of the benchmark (bench.py's cpu_baseline leg, kind
"port"): the reference's Python cannot travel to the GPU box, so its algorithmic structure is restated here
with plain torch CPU ops -- materialised [B,S,N] distance matrix + full sort for the ball query
(pointnet2_utils.py:105-108), a Python loop of npoint passes for FPS (:81-87), full sort for the 3-NN
(blocks.py:195), Conv/BatchNorm/ReLU modules and autograd for the MLPs; (2) a float reference for the MLP
kernels in tests.  It is never imported by the product package.

Parity status: pinned.  tests/test_torch_port.py checks this file against the golden vectors of the imported
reference (indices bit-exact, model outputs/grads to fp32 rounding) in the build container.  Its bmm-based
distances depend on the host BLAS exactly as the reference's do; the machine-independent statement of the
numerics is oracle/pn2_oracle.c.
"""
import functools

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---------------------------------------------------------------------------------------------------- L0 ops
def square_distance(src, dst):
    d = -2 * torch.matmul(src, dst.transpose(1, 2))            # pointnet2_utils.py:39
    d += (src ** 2).sum(-1)[:, :, None]                        # :40
    d += (dst ** 2).sum(-1)[:, None, :]                        # :41
    return d


def index_points(points, idx):
    B = points.shape[0]
    b = torch.arange(B).view(B, *([1] * (idx.dim() - 1))).expand_as(idx)
    return points[b, idx, :]                                    # :62


def farthest_point_sample(xyz, npoint, start=None):
    B, N, _ = xyz.shape
    out = torch.zeros(B, npoint, dtype=torch.long)
    mind = torch.full((B, N), 1e10)
    far = torch.randint(0, N, (B,), dtype=torch.long) if start is None else start.clone()
    rows = torch.arange(B)
    for i in range(npoint):                                     # :81-87, npoint sequential passes
        out[:, i] = far
        c = xyz[rows, far, :].view(B, 1, 3)
        d = ((xyz - c) ** 2).sum(-1)
        closer = d < mind
        mind[closer] = d[closer]
        far = mind.max(-1)[1]
    return out


def query_ball_point(radius, nsample, xyz, new_xyz):
    B, N, _ = xyz.shape
    S = new_xyz.shape[1]
    d = square_distance(new_xyz, xyz)
    idx = torch.arange(N).view(1, 1, N).repeat(B, S, 1)
    idx[d > radius ** 2] = N                                    # :107
    idx = idx.sort(dim=-1)[0][:, :, :nsample]                   # :108, the S*N int64 sort
    first = idx[:, :, 0].clone()
    empty = first == N
    if empty.any():
        first[empty] = d.argmin(-1)[empty]                      # :113-122
    return torch.where(idx == N, first[:, :, None].expand_as(idx), idx)


def sample_and_group(npoint, radius, nsample, xyz, points, start=None, xyz_last=False):
    fps = farthest_point_sample(xyz, npoint, start)
    new_xyz = index_points(xyz, fps)
    idx = query_ball_point(radius, nsample, xyz, new_xyz)
    rel = index_points(xyz, idx) - new_xyz[:, :, None, :]
    if points is None:
        return new_xyz, rel
    grouped = index_points(points, idx)
    return new_xyz, torch.cat([grouped, rel] if xyz_last else [rel, grouped], dim=-1)


STABLE_SORT = False   # tests set this to compare against the product's documented three-NN tie rule (lower index first)


def three_nn_interpolate(xyz1, xyz2, points2):
    B, N, _ = xyz1.shape
    d, idx = square_distance(xyz1, xyz2).sort(dim=-1, stable=STABLE_SORT)   # blocks.py:195, full sort over S
    d, idx = d[:, :, :3], idx[:, :, :3]
    rec = 1.0 / torch.clamp(d, min=1e-6)
    w = rec / rec.sum(dim=2, keepdim=True)
    return (index_points(points2, idx) * w.view(B, N, 3, 1)).sum(dim=2)


# --------------------------------------------------------------------------------------------------- L1 / L2
class _SA(nn.Module):
    def __init__(self, npoint, radius, nsample, cin, widths):
        super().__init__()
        self.cfg = (npoint, radius, nsample)
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        for c in widths:
            self.mlp_convs.append(nn.Conv2d(cin, c, 1))
            self.mlp_bns.append(nn.BatchNorm2d(c))
            cin = c

    def forward(self, xyz, pts):
        new_xyz, g = sample_and_group(*self.cfg, xyz.transpose(1, 2), None if pts is None else pts.transpose(1, 2))
        g = g.permute(0, 3, 2, 1)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            g = F.relu(bn(conv(g)))
        return new_xyz.transpose(1, 2), g.max(2)[0]


class _SAMsg(nn.Module):
    def __init__(self, npoint, radii, nsamples, cin, mlp_list):
        super().__init__()
        self.npoint, self.radii, self.nsamples = npoint, radii, nsamples
        self.conv_blocks, self.bn_blocks = nn.ModuleList(), nn.ModuleList()
        for widths in mlp_list:
            convs, bns, c = nn.ModuleList(), nn.ModuleList(), cin
            for w in widths:
                convs.append(nn.Conv2d(c, w, 1))
                bns.append(nn.BatchNorm2d(w))
                c = w
            self.conv_blocks.append(convs)
            self.bn_blocks.append(bns)

    def forward(self, xyz, pts):
        x, p = xyz.transpose(1, 2), None if pts is None else pts.transpose(1, 2)
        new_xyz = index_points(x, farthest_point_sample(x, self.npoint))
        outs = []
        for r, k, convs, bns in zip(self.radii, self.nsamples, self.conv_blocks, self.bn_blocks):
            idx = query_ball_point(r, k, x, new_xyz)
            rel = index_points(x, idx) - new_xyz[:, :, None, :]
            g = rel if p is None else torch.cat([index_points(p, idx), rel], dim=-1)
            g = g.permute(0, 3, 2, 1)
            for conv, bn in zip(convs, bns):
                g = F.relu(bn(conv(g)))
            outs.append(g.max(2)[0])
        return new_xyz.transpose(1, 2), torch.cat(outs, dim=1)


class _FP(nn.Module):
    def __init__(self, cin, widths):
        super().__init__()
        self.mlp_convs, self.mlp_bns = nn.ModuleList(), nn.ModuleList()
        for c in widths:
            self.mlp_convs.append(nn.Conv1d(cin, c, 1))
            self.mlp_bns.append(nn.BatchNorm1d(c))
            cin = c

    def forward(self, xyz1, xyz2, p1, p2):
        x1, x2, q2 = xyz1.transpose(1, 2), xyz2.transpose(1, 2), p2.transpose(1, 2)
        N = x1.shape[1]
        y = q2.repeat(1, N, 1) if x2.shape[1] == 1 else three_nn_interpolate(x1, x2, q2)
        if p1 is not None:
            y = torch.cat([p1.transpose(1, 2), y], dim=-1)
        y = y.transpose(1, 2)
        for conv, bn in zip(self.mlp_convs, self.mlp_bns):
            y = F.relu(bn(conv(y)))
        return y


class _Head(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.net = nn.Sequential(nn.Conv1d(cin, cin, 1), nn.BatchNorm1d(cin, eps=1e-4, momentum=0.1), nn.ReLU(inplace=True),
                                 nn.Conv1d(cin, cout, 1))

    def forward(self, x):
        return self.net(x)


_TABLE = {  # depth -> (SA rows, FP rows fpN..fp1); PointNet2.py:38-97
    4: ([(1024, .1, 32, [32, 32, 64]), (256, .2, 32, [64, 64, 128]), (64, .4, 32, [128, 128, 256]), (16, .8, 32, [256, 256, 512])],
        [(768, [256, 256]), (384, [256, 256]), (320, [256, 128]), (128, [128, 128, 128])]),
    5: ([(100, .1, 32, [32, 32, 64]), (50, .2, 32, [64, 64, 128]), (20, .4, 32, [128, 128, 256]), (8, .8, 32, [256, 256, 512])],
        [(768, [256, 256]), (384, [256, 256]), (320, [256, 128]), (128, [128, 128, 128])]),
    3: ([(1024, .1, 32, [32, 32, 64]), (256, .3, 32, [64, 64, 128]), (64, .6, 32, [128, 128, 256])],
        [(384, [256, 256]), (320, [256, 128]), (128, [128, 128, 128])]),
    2: ([(1024, .02, 32, [32, 32, 64]), (256, .2, 32, [64, 64, 128])],
        [(192, [128, 128, 128]), (128, [128, 128, 128])]),
}


class PortPointNet2(nn.Module):
    """Same state-dict keys as the reference model; forward(coords [B,3,N], feats [B,4,N]) -> (sem, off)."""

    def __init__(self, depth=4, dim_feat=4):
        super().__init__()
        self.depth = depth
        if depth == 6:
            self.sa1 = _SAMsg(500, [.02, .04, .08], [16, 32, 32], 3 + dim_feat, [[16, 16, 32], [32, 32, 64], [64, 64, 64]])
            sa = [None, (100, .2, 32, [64, 64, 128]), (50, .4, 32, [128, 128, 256]), (20, .8, 32, [256, 256, 512])]
            fp = [(768, [256, 256]), (384, [256, 256]), (416, [256, 128]), (128, [128, 128, 128])]
            prev = 160
        else:
            sa, fp = _TABLE[depth]
            prev = None
        for lvl, row in enumerate(sa, start=1):
            if row is None:
                continue
            cin = 3 + dim_feat if prev is None else prev + 3
            setattr(self, f"sa{lvl}", _SA(row[0], row[1], row[2], cin, row[3]))
            prev = row[3][-1]
        self.levels = len(sa)
        for k, (cin, widths) in enumerate(fp):
            setattr(self, f"fp{self.levels - k}", _FP(cin, widths))
        self.semantic_linear, self.offset_linear = _Head(128, 2), _Head(128, 3)

    def backbone(self, coords, feats):
        xyz, pts = [coords], [feats]
        for lvl in range(1, self.levels + 1):
            nx, np_ = getattr(self, f"sa{lvl}")(xyz[-1], pts[-1])
            xyz.append(nx)
            pts.append(np_)
        for lvl in range(self.levels, 1, -1):
            pts[lvl - 1] = getattr(self, f"fp{lvl}")(xyz[lvl - 1], xyz[lvl], pts[lvl - 1], pts[lvl])
        return self.fp1(xyz[0], xyz[1], None, pts[1])

    def forward(self, coords, feats):
        f = self.backbone(coords, feats)
        return self.semantic_linear(f), self.offset_linear(f)


def point_wise_loss(sem_logits, off_preds, sem_labels, off_labels):
    """Loss.py:6-36 without the subsampling branch."""
    sem = F.cross_entropy(sem_logits, sem_labels, reduction="sum") / len(sem_logits) if len(sem_logits) else 0 * sem_labels.sum()
    sq = (off_preds - off_labels).pow(2).sum(1)
    off = torch.sqrt(torch.clamp(sq, min=1e-8)).mean() if len(off_preds) else 0 * off_preds.sum()
    return sem, off


def loss_from_batch(model, batch, mult_sem=1.0, mult_off=1.0):
    """PointNet2.forward(return_loss=True) + get_loss (PointNet2.py:118-134, 180-207)."""
    sem, off = model(batch["coords"], batch["feats"])
    keep = batch["masks_pad"].reshape(-1)
    sem_v = sem.permute(0, 2, 1).reshape(-1, 2)[keep]
    off_v = off.permute(0, 2, 1).reshape(-1, 3)[keep][batch["masks_off"]]
    ls, lo = point_wise_loss(sem_v.float(), off_v.float(), batch["semantic_labels"], batch["offset_labels"])
    return ls * mult_sem + lo * mult_off, {"semantic_loss": ls * mult_sem, "offset_loss": lo * mult_off}, sem, off
