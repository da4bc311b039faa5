"""Whole-tree execution of the reference's raster modes (Modules/PointNet2/PointNet2.py:210-394).

The reference feeds a tree to the network as a stream of mini-batches of rasters -- ~40 forward (+ backward) passes of
~10 small zero-padded clouds each for a 262 144-point tree -- and every pass is ~460 kernel launches a few microseconds
long: on a 256-CU GPU that is launch latency, not work.  Nothing couples the passes except (a) the weights, which do not
change inside a tree (one optimizer step per tree, train_utils.py:57-61), (b) the BatchNorm running statistics and (c) the
global CPU RNG that draws the FPS start indices.  So all mini-batches of a tree are run here as ONE pass over a ragged
super-batch:

  * level 0 keeps every raster at its mini-batch's padded length (padding is real data to FPS / ball query / BatchNorm,
    SURVEY Q1): the per-mini-batch tensors are laid end to end in flat channel-first buffers (ops.RaggedClouds) and the
    level-0 kernels index them through per-cloud offsets; sampled levels are regular [C, S, *] batches of all C rasters;
  * every MLP chain runs once over all rows with the mini-batches as row SEGMENTS: train-mode BatchNorm statistics,
    normalisation and backward are per segment, running statistics advance segment after segment (mlp.chain_rows);
  * FPS start indices are drawn up front in the reference's order (per mini-batch: sa1, sa2, ... -- one torch.randint
    each, pointnet2_utils.py:79), so a seeded run samples the same centroids;
  * the per-mini-batch losses are segment means of one per-row loss tensor, their sum is back-propagated once
    (= the sum of the reference's per-mini-batch backward calls), predictions are scatter-averaged per point id with
    index_add (no boolean indexing, no host synchronisation before the final loss read-back).

Same numbers as the sequential loop up to fp32 summation order (tests/test_streaming.py compares both with the
reference's fixture).

Overlapping rasters (stride < size: the training default, train_PointNet2.py:84-85,109).  A point id then occurs more than
once inside ONE mini-batch, and the reference's `avg[point_ids] += x; count[point_ids] += 1` (PointNet2.py:272-276) is an
index_put WITHOUT accumulation: per mini-batch ONE of the duplicates is kept (the last one on the CPU, an arbitrary one on
a GPU) and the count goes up by one.  Default here (PN2_OVERLAP=reference): exactly that, deterministically -- the last
occurrence inside the mini-batch contributes, count + 1 per (id, mini-batch) -- including, in forward_hierarchical, the
gradient the reference's expression implies (RefScatter below).  PN2_OVERLAP=average (opt-in) averages every occurrence
instead (round 2's behaviour).  Pinned by tests/golden/streaming_overlap_d5.npz / hierarchical_overlap_d5.npz.
"""
import os

import numpy as np
import torch
import torch.nn.functional as F

from . import _hip, ops
from .mlp import batched_counters, chain_pair_rows, chain_rows, group_hoist_ok, hoist_ok, hoisted_conv, interp_bn_rows
from .PointNet2.blocks import PointNetSetAbstractionMsg, _group_mlp_max, _hoisted_group_mlp_max

MAX_ROWS_PER_PASS = 6_000_000       # level-0 rows (padded points) per pass: ~60 GB of activations at depth 5


class TreeLayout:
    """Host-side description of one pass: which mini-batches, how many rasters each, their padded lengths."""

    def __init__(self, mini_batches, flat=None):
        self.mbs = mini_batches
        self.flat = flat                  # rasters.RasterStream.flat: the pass's tensors already exist as flat buffers
        self.clouds = [int(mb["coords"].shape[0]) for mb in mini_batches]            # B_j
        self.length = [int(mb["coords"].shape[2]) for mb in mini_batches]            # N_j
        self.n_valid = [int(mb["point_ids"].shape[0]) for mb in mini_batches]
        self.M = len(mini_batches)
        self.cum_clouds = np.concatenate([[0], np.cumsum(self.clouds)]).astype(np.int64)
        self.C = int(self.cum_clouds[-1])
        self.rows0 = np.concatenate([[0], np.cumsum(np.multiply(self.clouds, self.length))]).astype(np.int64)

    def seg_rows(self, rows_per_cloud):
        """row offsets of the segments for a regular level with `rows_per_cloud` rows per raster"""
        return (self.cum_clouds * int(rows_per_cloud)).tolist()


def supported(mini_batches, model):
    """The fused path needs: every padded raster at least as long as the first level's neighbourhood size (so that
    K = nsample everywhere), no longer than one FPS workgroup holds, and at most MAX_SEGMENTS mini-batches per pass
    (longer streams are cut into several passes by the caller)."""
    sa1 = model.sa1
    k1 = max(sa1.nsample_list) if isinstance(sa1, PointNetSetAbstractionMsg) else sa1.nsample
    for mb in mini_batches:
        n = int(mb["coords"].shape[2])
        if n < max(k1, 3) or n > 16384 or mb["coords"].shape[0] < 1:
            return False
    return len(mini_batches) > 0


def _device_cat(tensors, device, dtype=None):
    """cat on whichever side the pieces live, one upload if they are host tensors (the reference's collate output)."""
    t = torch.cat([x.reshape(-1) for x in tensors])
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t.to(device, non_blocking=True)


def draw_starts(layout, model):
    """FPS start indices for every SA level, drawn in the reference's order: mini-batch after mini-batch, level after
    level (each SA forward consumes one torch.randint from the global CPU generator, pointnet2_utils.py:79)."""
    sas = _sa_modules(model)
    per_level = [[] for _ in sas]
    for b, n in zip(layout.clouds, layout.length):
        upper = n
        for lvl, sa in enumerate(sas):
            per_level[lvl].append(torch.randint(0, upper, (b,), dtype=torch.long))
            upper = sa.npoint
    return [torch.cat(p) for p in per_level]


def _sa_modules(model):
    out, lvl = [], 1
    while hasattr(model, f"sa{lvl}"):
        out.append(getattr(model, f"sa{lvl}"))
        lvl += 1
    return out


def _sa_level(sa, xyz_t, pts_t, start, layout, rc=None):
    """One set-abstraction level over all rasters of the pass.  rc: ragged level-0 clouds (first level) or None
    (xyz_t [C,N,3], pts_t [C,N,D] regular).  -> new_xyz [C,S,3], pooled [C,S,C_out] (channels-last)."""
    dev = start.device
    if rc is not None:
        fps_idx, new_xyz = ops.fps_ragged(rc, sa.npoint, start)
    else:
        fps_idx, new_xyz = ops.furthest_point_sample(xyz_t, sa.npoint, start)
    msg = isinstance(sa, PointNetSetAbstractionMsg)
    specs = list(zip(sa.radius_list, sa.nsample_list, sa.conv_blocks, sa.bn_blocks)) if msg else \
        [(sa.radius, sa.nsample, sa.mlp_convs, sa.mlp_bns)]
    outs = []
    for radius, K, convs, bns in specs:
        if rc is not None:
            idx = ops.ball_query_ragged(radius, K, rc, new_xyz)
            grouped = ops.group_ragged(rc, new_xyz, idx, xyz_last=msg)
        else:
            idx = ops.ball_query(radius, K, xyz_t, new_xyz)
            Cn, Nn = xyz_t.shape[0], xyz_t.shape[1]
            if pts_t is not None and group_hoist_ok(convs[0], bns[0], len(convs), Cn, Nn, idx.shape[1], idx.shape[2], pts_t.shape[2],
                                                    xyz_t.device):
                outs.append(_hoisted_group_mlp_max(xyz_t, new_xyz, pts_t, idx, convs, bns, msg,
                                                   seg_off=layout.seg_rows(idx.shape[1] * idx.shape[2])))
                continue
            grouped = ops.GroupPoints.apply(xyz_t, new_xyz, pts_t, idx, msg)
        rows_per_cloud = grouped.shape[1] * grouped.shape[2]
        outs.append(_group_mlp_max(grouped, convs, bns, seg_off=layout.seg_rows(rows_per_cloud),
                                   coords_first=0 if (msg or rc is not None or pts_t is None) else 3))
    return new_xyz, (outs[0] if len(outs) == 1 else torch.cat(outs, dim=-1))


def _fp_level(fp, x1, x2, p1, p2, layout, rc=None, lazy=False):
    """One feature-propagation level.  rc given: the dense side is the ragged level 0 (x1, p1 unused) and the result is
    packed rows [rows, C_out]; otherwise x1 [C,N,3], p1 [C,N,D1] or None -> [C,N,C_out]."""
    layers = [(c, b, True) for c, b in zip(fp.mlp_convs, fp.mlp_bns)]
    C, S, _ = x2.shape
    if rc is not None:
        if S == 1:
            raise RuntimeError("feature propagation from a single sampled point onto ragged clouds is not supported")
        idx, w = ops.three_nn_ragged(rc, x2)
        seg0 = layout.rows0.tolist()
        if hoist_ok(fp.mlp_convs[0], fp.mlp_bns[0], len(layers), x2.device):
            # no skip connection at level 0: the first conv runs on the C * S sampled rows (mlp.interp_bn_rows)
            q = hoisted_conv(p2.reshape(C * S, -1), fp.mlp_convs[0])
            feats = interp_bn_rows(q.view(C, S, -1), idx, w, fp.mlp_bns[0], seg_off=seg0, bias=fp.mlp_convs[0].bias, rc=rc)
            layers = layers[1:]
        else:
            feats = ops.ThreeInterpolateRagged.apply(p2, idx, w, rc)
        return chain_rows(feats, layers, seg_off=seg0, lazy_out=lazy)   # lazy: a mlp.LazyRows for the heads
    N = x1.shape[1]
    if S == 1:
        feats = p2.repeat(1, N, 1)
        if p1 is not None:
            feats = torch.cat([p1, feats], dim=-1)
    else:
        idx, w = ops.three_nn(x1, x2)
        feats = ops.ThreeInterpolateConcat.apply(p1, p2, idx, w)
    y = chain_rows(feats.reshape(C * N, -1), layers, seg_off=layout.seg_rows(N))
    return y.view(C, N, -1)


def backbone_and_heads(model, layout, device):
    """-> (semantic logits [rows,2], offsets [rows,3]) for the packed padded rows of the pass (row = raster by raster,
    point by point, i.e. the order of `masks_pad.reshape(-1)` mini-batch after mini-batch)."""
    mbs = layout.mbs
    lengths = np.repeat(layout.length, layout.clouds).tolist()
    feats_cf, dim_feat = None, 0
    if layout.flat is not None:           # built on the device by rasters.build_stream: nothing to concatenate
        xyz_cf = layout.flat["xyz_cf"]
        if model.use_features:
            dim_feat, feats_cf = int(mbs[0]["feats"].shape[1]), layout.flat["feats_cf"]
    else:
        xyz_cf = _device_cat([mb["coords"] for mb in mbs], device, torch.float32)
        if model.use_features:
            dim_feat = int(mbs[0]["feats"].shape[1])
            feats_cf = _device_cat([mb["feats"] for mb in mbs], device, torch.float32)
    rc = ops.RaggedClouds(xyz_cf, feats_cf, dim_feat, lengths)
    starts = [s.pin_memory().to(device, non_blocking=True) for s in draw_starts(layout, model)]
    sas = _sa_modules(model)
    n = len(sas)
    with torch.amp.autocast("cuda", enabled=False), batched_counters():
        xyz, pts = [None], [None]
        for lvl, sa in enumerate(sas):
            nx, npts = _sa_level(sa, xyz[-1], pts[-1], starts[lvl], layout, rc=rc if lvl == 0 else None)
            xyz.append(nx)
            pts.append(npts)
        for level in range(n, 1, -1):
            pts[level - 1] = _fp_level(getattr(model, f"fp{level}"), xyz[level - 1], xyz[level], pts[level - 1], pts[level],
                                       layout)
        feats = _fp_level(model.fp1, None, xyz[1], None, pts[1], layout, rc=rc, lazy=torch.is_grad_enabled())
        seg0 = layout.rows0.tolist()
        sem, off = chain_pair_rows(feats, model.semantic_linear._layers(), model.offset_linear._layers(), seg_off=seg0)
    return sem, off


def _valid_rows(layout, device):
    """Packed row index of every real point (the reference's x[masks_pad], mini-batch after mini-batch) -- the count
    is known on the host (point_ids), so no synchronisation -- plus the global point id, the offset mask and the
    mini-batch number of each."""
    n_valid = int(sum(layout.n_valid))
    if layout.flat is not None:
        pad, ids, moff = layout.flat["masks_pad"], layout.flat["point_ids"], layout.flat["masks_off"]
    else:
        pad = _device_cat([mb["masks_pad"] for mb in layout.mbs], device)
        ids = _device_cat([mb["point_ids"] for mb in layout.mbs], device, torch.long)
        moff = _device_cat([mb["masks_off"] for mb in layout.mbs], device)
    rows = torch.nonzero_static(pad, size=n_valid).squeeze(1) if hasattr(torch, "nonzero_static") else pad.nonzero().squeeze(1)
    counts = torch.tensor(layout.n_valid).to(device, non_blocking=True)
    seg = torch.repeat_interleave(torch.arange(layout.M, device=device), counts, output_size=n_valid)   # no sync
    return rows, ids, moff, seg


def overlap_mode():
    mode = os.environ.get("PN2_OVERLAP", "reference")
    if mode not in ("reference", "average"):
        raise ValueError(f"PN2_OVERLAP={mode!r}: 'reference' (one contribution per id and mini-batch, PointNet2.py:272-276) or 'average'")
    return mode


def last_occurrence(keys, size, valid=None):
    """-> (keep, count): keep[r] = row r is the LAST row (highest r) that carries keys[r] -- among the rows with valid[r]
    when a mask is given -- which is what the reference's CPU `t[ids] += x` keeps of duplicate ids; count [size] int64 =
    number of (valid) rows per key.  keys int64 in [0, size)."""
    dev = keys.device
    pos = torch.arange(keys.numel(), device=dev)
    src = pos if valid is None else torch.where(valid, pos, torch.full_like(pos, -1))
    last = torch.full((size,), -1, dtype=torch.long, device=dev).scatter_reduce_(0, keys, src, "amax", include_self=True)
    keep = last.index_select(0, keys) == pos
    ones = torch.ones_like(keys) if valid is None else valid.to(torch.long)
    count = torch.zeros(size, dtype=torch.long, device=dev).index_add_(0, keys, ones)
    return keep, count


class RefScatter(torch.autograd.Function):
    """Contribution rows of the reference's `t[ids] += v` over a stream of mini-batches (PointNet2.py:272-276, 376-380) as one
    differentiable op: forward -> v * keep (the last duplicate of an id inside a mini-batch is the one that lands).
    Backward = what autograd makes of the reference's expression (gather, add, NON-accumulating index_put per mini-batch):
    every duplicate row receives the gradient of its id's accumulator entry, and the gradient that flows on to the
    accumulator of EARLIER mini-batches is multiplied by the number of duplicates (the gather's backward adds once per
    occurrence): row r of mini-batch j gets G[id] * prod over later mini-batches j' of max(1, occurrences of id in j')."""

    @staticmethod
    def forward(ctx, v, keep, mult):
        ctx.save_for_backward(mult)
        return v * keep.to(v.dtype).unsqueeze(1)

    @staticmethod
    def backward(ctx, g):
        (mult,) = ctx.saved_tensors
        return g * mult.unsqueeze(1), None, None


class RefPut(torch.autograd.Function):
    """ONE mini-batch of the reference's `t[ids] += v` with autograd history (forward_hierarchical, PointNet2.py:376-380),
    mini-batch-by-mini-batch execution: forward t + scatter(v * keep); backward exactly what autograd derives for
    `t.index_put_(ids, t[ids] + v)`: dv = G[ids] for EVERY occurrence, dt = G off the indexed entries and
    (occurrences of the id) * G on them.  count: [n] occurrences per id (streaming.last_occurrence)."""

    @staticmethod
    def forward(ctx, t, ids, v, keep, count):
        ctx.save_for_backward(ids, count)
        return t.index_add(0, ids, v * keep.to(v.dtype).unsqueeze(1))

    @staticmethod
    def backward(ctx, g):
        ids, count = ctx.saved_tensors
        return g * count.clamp_min(1).to(g.dtype).unsqueeze(1), None, g.index_select(0, ids), None, None


def _later_duplicates(count, seg, ids, M, n):
    """mult[r] = prod_{j' > seg[r]} max(1, count[j', ids[r]]) (float32); count: [M * n] occurrences per (mini-batch, id)."""
    d = count.view(M, n).clamp_min(1).to(torch.float32)
    suffix = torch.flip(torch.cumprod(torch.flip(d, [0]), 0), [0])                # prod_{j' >= j}
    after = torch.cat([suffix[1:], torch.ones(1, n, dtype=torch.float32, device=d.device)])
    return after.view(-1).index_select(0, seg * n + ids)


class _Accumulators:
    def __init__(self, n, device):
        self.n = n
        self.sem = torch.zeros(n, 2, dtype=torch.float, device=device)
        self.off = torch.zeros(n, 3, dtype=torch.float, device=device)
        self.sem_cnt = torch.zeros(n, 1, dtype=torch.float, device=device)
        self.off_cnt = torch.zeros(n, 1, dtype=torch.float, device=device)

    def add(self, ids, sem, off, moff, differentiable, seg=None, M=1, disjoint=False):
        """ids / sem / off / moff: the valid rows of the pass, mini-batch after mini-batch; seg: their mini-batch number
        (None: one mini-batch); disjoint: the caller knows that no id repeats inside a mini-batch (no overlap handling)."""
        m = moff.to(torch.float).unsqueeze(1)
        if not disjoint and overlap_mode() == "reference" and ids.numel():
            n = self.n
            if M * n >= 2 ** 31:
                raise RuntimeError("overlap handling: too many (mini-batch, point) pairs for one pass")
            segv = torch.zeros_like(ids) if seg is None else seg
            keys = segv * n + ids
            ks, cs = last_occurrence(keys, M * n)
            ko, co = last_occurrence(keys, M * n, moff)
            if differentiable:
                sem = RefScatter.apply(sem, ks, _later_duplicates(cs, segv, ids, M, n))
                off = RefScatter.apply(off, ko, _later_duplicates(co, segv, ids, M, n) * m.squeeze(1))   # unmasked rows never enter
            else:
                sem, off = sem * ks.to(sem.dtype).unsqueeze(1), off * ko.to(off.dtype).unsqueeze(1)
            ones, m_cnt = ks.to(torch.float).unsqueeze(1), ko.to(torch.float).unsqueeze(1)
        else:
            off = off * m
            ones, m_cnt = torch.ones_like(m), m
        if differentiable:                  # forward_hierarchical keeps autograd history through the average
            self.sem = self.sem.index_add(0, ids, sem)
            self.off = self.off.index_add(0, ids, off)
        else:
            self.sem.index_add_(0, ids, sem.detach())
            self.off.index_add_(0, ids, off.detach())
        self.sem_cnt.index_add_(0, ids, ones)
        self.off_cnt.index_add_(0, ids, m_cnt)

    def average(self):
        return {"semantic_prediction_logits": self.sem / self.sem_cnt.clamp_min(1.0),
                "offset_predictions": self.off / self.off_cnt.clamp_min(1.0)}


def _passes(mini_batches):
    """cut a stream into passes of at most MAX_SEGMENTS mini-batches and MAX_ROWS_PER_PASS padded points"""
    cur, rows = [], 0
    for mb in mini_batches:
        r = int(mb["coords"].shape[0]) * int(mb["coords"].shape[2])
        if cur and (len(cur) == _hip.MAX_SEGMENTS or rows + r > MAX_ROWS_PER_PASS):
            yield cur
            cur, rows = [], 0
        cur.append(mb)
        rows += r
    if cur:
        yield cur


def run_tree(model, batch, return_loss, scaler=None, streaming=True):
    """The fused counterpart of PointNet2.forward_hierarchical_streaming (streaming=True: per-mini-batch losses,
    back-propagated here, scaled by 50 through `scaler`) and forward_hierarchical (streaming=False: one loss on the
    averaged predictions, returned with its autograd history).  `batch["mini_batches"]` must be a list accepted by
    `supported`."""
    device = torch.device("cuda", torch.cuda.current_device())
    acc = _Accumulators(batch["cloud_length"], device)
    want_stream_loss = return_loss and streaming
    if want_stream_loss:
        sem_all = batch["semantic_labels"].squeeze().to(device, non_blocking=True)
        off_all = batch["offset_labels"].to(device, non_blocking=True)
        total = torch.zeros((), dtype=torch.float64, device=device)
        sums = torch.zeros(2, dtype=torch.float32, device=device)
    n_mb = 0
    stream = batch["mini_batches"]
    for chunk in _passes(stream):
        whole = len(chunk) == len(stream)
        layout = TreeLayout(chunk, flat=getattr(stream, "flat", None) if whole else None)
        sem_rows, off_rows = backbone_and_heads(model, layout, device)
        rows, ids, moff, seg = _valid_rows(layout, device)
        sem, off = sem_rows.index_select(0, rows), off_rows.index_select(0, rows)
        acc.add(ids, sem, off, moff, differentiable=return_loss and not streaming, seg=seg, M=layout.M,
                disjoint=bool(getattr(stream, "disjoint", False)))
        if want_stream_loss:
            # per-mini-batch point_wise_loss (Loss.py:6-36) as segment means of per-row terms
            M = layout.M
            m = moff.to(torch.float32)
            ce = F.cross_entropy(sem.float(), sem_all.index_select(0, ids), reduction="none")
            dist = torch.sqrt(torch.clamp((off.float() - off_all.index_select(0, ids)).pow(2).sum(1), min=1e-8)) * m
            n_sem = torch.tensor(layout.n_valid, dtype=torch.float32).to(device, non_blocking=True)
            sem_loss = torch.zeros(M, device=device).index_add(0, seg, ce) / n_sem.clamp_min(1.0)
            n_off = torch.zeros(M, device=device).index_add_(0, seg, m)
            off_loss = torch.zeros(M, device=device).index_add(0, seg, dist) / n_off.clamp_min(1.0)
            sem_loss, off_loss = sem_loss * model.loss_multiplier_semantic, off_loss * model.loss_multiplier_offset
            mini = sem_loss + off_loss
            if scaler:
                scaler.scale(mini.sum() * 50).backward()        # = the reference's per-mini-batch backward calls, summed
            total += mini.detach().double().sum()
            sums += torch.stack([sem_loss.detach().sum(), off_loss.detach().sum()])
            del sem_loss, off_loss, mini, ce, dist
        n_mb += layout.M
        del sem_rows, off_rows, sem, off
    output = acc.average()
    if not return_loss:
        return output
    if not streaming:
        return model.get_loss_hierarchical(output, batch["semantic_labels"].squeeze(), batch["offset_labels"])
    total_loss = float(total)                 # the one host synchronisation of the tree
    ops.check_status(device)
    loss_dict = {"semantic_loss": sums[0] / max(n_mb, 1), "offset_loss": sums[1] / max(n_mb, 1)}
    return (total_loss / n_mb if n_mb else 0.0), loss_dict
