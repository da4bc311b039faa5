"""PointNet2 model wrapper with the reference's constructor, attribute names, batch-dict keys and return values
(Modules/PointNet2/PointNet2.py).  This is the caller of the hot path: its forward + loss + backward is the
unit the benchmark times.

Layer tables per ``depth`` follow reference lines 38-97: SA(npoint, radius, nsample, in_channel, mlp) and
FP(in_channel, mlp); heads are ConvHead(128 -> 128 -> {2,3}) with BatchNorm1d(eps=1e-4) (lines 22, 103-104).
"""
import functools
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops, streaming
from ..Loss import MaskedPointLoss, mask_ranks, point_wise_loss
from ..mlp import batched_counters, chain_pair_rows
from ..Utils import cuda_cast
from .pointnet2_utils import start_batch
from .blocks import (MLP, ConvHead, PointNetFeaturePropagation, PointNetSetAbstraction,
                     PointNetSetAbstractionMsg)

# depth -> ([SA rows], [FP rows in the order fpN ... fp1]); in_channel of sa1 is filled in at build time
_SA = {
    4: [(1024, 0.1, 32, None, [32, 32, 64]), (256, 0.2, 32, 64 + 3, [64, 64, 128]),
        (64, 0.4, 32, 128 + 3, [128, 128, 256]), (16, 0.8, 32, 256 + 3, [256, 256, 512])],
    5: [(100, 0.1, 32, None, [32, 32, 64]), (50, 0.2, 32, 64 + 3, [64, 64, 128]),
        (20, 0.4, 32, 128 + 3, [128, 128, 256]), (8, 0.8, 32, 256 + 3, [256, 256, 512])],
    6: [None, (100, 0.2, 32, 160 + 3, [64, 64, 128]), (50, 0.4, 32, 128 + 3, [128, 128, 256]),
        (20, 0.8, 32, 256 + 3, [256, 256, 512])],
    3: [(1024, 0.1, 32, None, [32, 32, 64]), (256, 0.3, 32, 64 + 3, [64, 64, 128]),
        (64, 0.6, 32, 128 + 3, [128, 128, 256])],
    2: [(1024, 0.02, 32, None, [32, 32, 64]), (256, 0.2, 32, 64 + 3, [64, 64, 128])],
}
_FP = {
    4: [(768, [256, 256]), (384, [256, 256]), (320, [256, 128]), (128, [128, 128, 128])],
    5: [(768, [256, 256]), (384, [256, 256]), (320, [256, 128]), (128, [128, 128, 128])],
    6: [(768, [256, 256]), (384, [256, 256]), (416, [256, 128]), (128, [128, 128, 128])],
    3: [(128 + 256, [256, 256]), (64 + 256, [256, 128]), (128, [128, 128, 128])],
    2: [(64 + 128, [128, 128, 128]), (128, [128, 128, 128])],
}


class PointNet2(nn.Module):
    def __init__(self, input_nc=3, loss_multiplier_semantic=1, loss_multiplier_offset=1, dim_feat=4,
                 use_coords=True, use_features=True, depth=4, **kwargs):
        super().__init__()
        if depth not in _SA:
            raise ValueError("Unsupported depth value. Please use depth=2, 3, or 4.")
        self.loss_multiplier_semantic = loss_multiplier_semantic
        self.loss_multiplier_offset = loss_multiplier_offset
        self.use_coords, self.use_features, self.depth = use_coords, use_features, depth

        input_dim = (3 if use_coords else 0) + (dim_feat if use_features else 0)
        for level, row in enumerate(_SA[depth], start=1):
            if row is None:  # depth 6: multi-scale first level (reference lines 64-70)
                sa = PointNetSetAbstractionMsg(npoint=500, radius_list=[0.02, 0.04, 0.08], nsample_list=[16, 32, 32],
                                               in_channel=input_dim,
                                               mlp_list=[[16, 16, 32], [32, 32, 64], [64, 64, 64]])
            else:
                npoint, radius, nsample, cin, widths = row
                sa = PointNetSetAbstraction(npoint, radius, nsample, input_dim if cin is None else cin, widths, False)
            setattr(self, f"sa{level}", sa)
        n_levels = len(_SA[depth])
        for k, (cin, widths) in enumerate(_FP[depth]):
            setattr(self, f"fp{n_levels - k}", PointNetFeaturePropagation(cin, widths))

        norm_fn = functools.partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        self.semantic_linear = ConvHead(128, 2, norm_fn=norm_fn, num_layers=2)
        self.offset_linear = ConvHead(128, 3, norm_fn=norm_fn, num_layers=2)
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, MLP):
                m.init_weights()

    # ------------------------------------------------------------------------------------------ flattened mode
    def forward(self, batch, return_loss):
        """batch: coords [B,3,N], feats [B,F,N] (+ masks/labels when return_loss).  Reference lines 118-134."""
        output = dict()
        with batched_counters():        # one launch for all BatchNorm step counters of the pass
            # with a loss to compute nobody sees the backbone features: fp1 leaves its last BatchNorm + ReLU to the heads
            feats = self.forward_backbone(coords=batch["coords"], feats=batch["feats"], lazy_rows=bool(return_loss))
            if not isinstance(feats, tuple):
                output["backbone_feats"] = feats
            output["semantic_prediction_logits"], output["offset_predictions"] = self._heads(feats)
        if return_loss:
            output = self.get_loss(model_output=output, **batch)
        return output

    @cuda_cast
    def forward_backbone(self, coords, feats, lazy_rows=False, **kwargs):
        """SA x L then FP x L; always fp32 (the reference disables autocast here, lines 136-178)."""
        n = len(_SA[self.depth])
        # the FPS start indices of all levels in one host-to-device copy (drawn in the levels' order: same numbers)
        sizes = [coords.shape[2]] + [getattr(self, f"sa{level}").npoint for level in range(1, n)]
        with torch.amp.autocast("cuda", enabled=False), start_batch(coords.shape[0], sizes, coords.device):
            xyz = [coords]
            pts = [feats if self.use_features else None]
            for level in range(1, n + 1):
                nx, npts = getattr(self, f"sa{level}")(xyz[-1], pts[-1])
                xyz.append(nx)
                pts.append(npts)
            for level in range(n, 1, -1):
                pts[level - 1] = getattr(self, f"fp{level}")(xyz[level - 1], xyz[level], pts[level - 1], pts[level])
            if lazy_rows:     # -> (rows or mlp.LazyRows, B, N): the heads apply fp1's last BatchNorm + ReLU themselves
                return self.fp1(xyz[0], xyz[1], None, pts[1], lazy_rows=True)
            return self.fp1(xyz[0], xyz[1], None, pts[1])

    def _heads(self, feats):
        """feats [B,128,N] -- or the (rows, B, N) of forward_backbone(lazy_rows=True) -- -> (semantic logits [B,2,N], offsets
        [B,3,N]): the two ConvHeads (reference lines 128-129) as one autograd node, so that their common input gets ONE
        gradient tensor instead of two and a sum."""
        if isinstance(feats, tuple):
            rows, B, N = feats
        else:
            B, C, N = feats.shape
            rows = feats.permute(0, 2, 1).reshape(B * N, C)
        sem, off = chain_pair_rows(rows, self.semantic_linear._layers(), self.offset_linear._layers())
        return sem.view(B, N, -1).permute(0, 2, 1), off.view(B, N, -1).permute(0, 2, 1)

    @staticmethod
    def _valid_rows(sem_logits, off_preds, masks_pad, masks_off):
        """[B,C,N] predictions -> rows of real points; the offset rows are additionally filtered by masks_off.
        Same result as the reference's boolean indexing ``x[mask]`` (lines 188-196), spelled as nonzero +
        index_select: the backward of boolean indexing is an accumulating index_put_ whose host-side set-up costs
        ~16 ms per call on ROCm (5 calls per step), index_select's backward is a plain index_add_."""
        sem = sem_logits.permute(0, 2, 1).reshape(-1, 2)
        off = off_preds.permute(0, 2, 1).reshape(-1, 3)
        keep = masks_pad.reshape(-1).nonzero().squeeze(1)
        keep_off = keep.index_select(0, masks_off.nonzero().squeeze(1))
        return sem.index_select(0, keep), off.index_select(0, keep_off)

    def get_loss(self, model_output, semantic_labels, offset_labels, masks_off, masks_pad, **kwargs):
        """Flattened-mode loss (reference lines 180-207): padding mask, then offset mask, then point_wise_loss.

        The reference compacts the predictions with boolean indexing, which needs the number of selected rows on
        the host (a device->host sync in the middle of every step) and an index_put_ in backward.  The same two
        means are computed here WITHOUT compaction: the compacted label arrays are expanded to the padded rows with
        a cumulative-sum rank, and the per-row losses are summed under the masks and divided by the mask counts
        (all on the device, no sync).  Values agree with point_wise_loss on the compacted rows to fp32 rounding
        (tests/test_hip_parity.py::test_get_loss_matches_compacted_form)."""
        sem = model_output["semantic_prediction_logits"].permute(0, 2, 1).reshape(-1, 2).float()
        off = model_output["offset_predictions"].permute(0, 2, 1).reshape(-1, 3).float()
        if semantic_labels.numel() == 0 or offset_labels.numel() == 0 or masks_off.numel() == 0:
            sem_v, off_v = self._valid_rows(model_output["semantic_prediction_logits"],
                                            model_output["offset_predictions"], masks_pad, masks_off)
            return self.get_loss_hierarchical({"semantic_prediction_logits": sem_v, "offset_predictions": off_v},
                                              semantic_labels, offset_labels)
        pad = masks_pad.reshape(-1)
        fused = sem.is_cuda and semantic_labels.dtype == torch.long
        if fused and pad.dtype == torch.bool and masks_off.dtype == torch.bool and masks_off.numel():
            cum_pad, off_mask, cum_off = mask_ranks(pad, masks_off)
        else:
            cum_pad = torch.cumsum(pad, 0)                   # cum - 1 = index of a real row among the real rows
            off_mask = pad & masks_off.index_select(0, (cum_pad - 1).clamp(0, masks_off.numel() - 1))
            cum_off = torch.cumsum(off_mask, 0)
        if fused:
            # both multipliers and the total inside the loss kernels: the scalar arithmetic of the generic path below (two
            # selects, two multiplies, two adds and their backward nodes) is ten launches of a few microseconds each
            total, parts = MaskedPointLoss.apply(sem, off, pad, off_mask, cum_pad, cum_off, semantic_labels.reshape(-1),
                                                 offset_labels, self._loss_weights(sem.device))
            return total, {"semantic_loss": parts[0], "offset_loss": parts[1]}
        else:
            rank = (cum_pad - 1).clamp(0, semantic_labels.numel() - 1)
            n_valid, n_off = cum_pad[-1].clamp_min(1), cum_off[-1].clamp_min(1)
            sem_labels = semantic_labels.reshape(-1).index_select(0, rank)
            off_labels = offset_labels.index_select(0, (cum_off - 1).clamp(0, offset_labels.shape[0] - 1))
            ce = F.cross_entropy(sem, sem_labels, reduction="none")
            semantic_loss = (ce * pad).sum() / n_valid
            dist = torch.sqrt(torch.clamp((off - off_labels).pow(2).sum(1), min=1e-8))
            offset_loss = (dist * off_mask).sum() / n_off
        loss_dict = {"semantic_loss": semantic_loss * self.loss_multiplier_semantic,
                     "offset_loss": offset_loss * self.loss_multiplier_offset}
        return sum(loss_dict.values()), loss_dict

    def _loss_weights(self, device):
        w = getattr(self, "_loss_w", None)
        key = (self.loss_multiplier_semantic, self.loss_multiplier_offset, device)
        if w is None or w[0] != key:
            w = (key, torch.tensor([float(key[0]), float(key[1])], dtype=torch.float32, device=device))
            self._loss_w = w
        return w[1]

    def get_loss_hierarchical(self, model_output, semantic_labels, offset_labels, **kwargs):
        semantic_loss, offset_loss = point_wise_loss(model_output["semantic_prediction_logits"].float(),
                                                     model_output["offset_predictions"].float(),
                                                     semantic_labels, offset_labels)
        loss_dict = {"semantic_loss": semantic_loss * self.loss_multiplier_semantic,
                     "offset_loss": offset_loss * self.loss_multiplier_offset}
        return sum(loss_dict.values()), loss_dict

    # --------------------------------------------------------------------------------------- hierarchical modes
    def _predict_minibatch(self, mini_batch):
        """Backbone + heads (autocast off, reference line 251) on one padded mini-batch of rasters; returns the
        valid semantic rows, the valid+masked offset rows and the global point ids of both."""
        feats = self.forward_backbone(coords=mini_batch["coords"], feats=mini_batch["feats"])
        with torch.amp.autocast("cuda", enabled=False):
            sem_logits, off_preds = self._heads(feats)
        dev = feats.device          # the reference's collate hands over host tensors; its x[mask] indexing accepts them
        masks_pad, masks_off = mini_batch["masks_pad"].to(dev), mini_batch["masks_off"].to(dev)
        sem, off = self._valid_rows(sem_logits, off_preds, masks_pad, masks_off)
        ids = mini_batch["point_ids"].to(dev)
        return sem, off, ids, ids[masks_off]

    @staticmethod
    def _accumulators(n, device):
        z = functools.partial(torch.zeros, dtype=torch.float, device=device)
        return z((n, 2)), z((n, 3)), z((n, 1)), z((n, 1))

    @staticmethod
    def _scatter_minibatch(total, count, ids, values, differentiable=False):
        """The reference's `total[ids] += values; count[ids] += 1` (lines 272-276, 376-380) for ONE mini-batch.  With
        overlapping rasters an id occurs more than once in `ids`, and that expression is an index_put WITHOUT accumulation:
        one duplicate lands (the last one on the CPU; on a GPU whichever store comes last) and the count goes up by one.
        Done here deterministically: the last occurrence lands (streaming.last_occurrence) -- unless PN2_OVERLAP=average.
        differentiable: returns the new `total` with the autograd history the reference's expression has."""
        if streaming.overlap_mode() == "average" or ids.numel() == 0:
            if differentiable:
                total = total.index_add(0, ids, values)
            else:
                total.index_add_(0, ids, values)
            count.index_add_(0, ids, torch.ones(ids.numel(), 1, dtype=count.dtype, device=count.device))
            return total
        keep, cnt = streaming.last_occurrence(ids, total.shape[0])
        count.index_add_(0, ids, keep.to(count.dtype).unsqueeze(1))
        if not differentiable:
            total.index_add_(0, ids, values * keep.to(values.dtype).unsqueeze(1))
            return total
        return streaming.RefPut.apply(total, ids, values, keep, cnt)

    @staticmethod
    def _average(total, count):
        seen = count.squeeze(1) > 0
        total[seen] /= count[seen]
        return total

    def forward_hierarchical_streaming(self, batch, return_loss, scaler=None):
        """One tree given as a stream of raster mini-batches (reference lines 210-327).  Predictions of
        overlapping rasters are scatter-averaged per original point id; with ``return_loss`` every mini-batch's
        loss is back-propagated (scaled by 50 through ``scaler``) so gradients accumulate over the tree.
        Returns (avg_loss, loss_dict) when return_loss else the averaged prediction dict.

        Default: all mini-batches of the tree run as ONE ragged pass with per-mini-batch BatchNorm segments
        (streaming.run_tree: same numbers, ~40x fewer launches).  PN2_STREAMING=sequential, or a stream the fused
        path does not take (a padded raster shorter than the neighbourhood size or longer than 16384 points), runs
        the reference's loop mini-batch by mini-batch."""
        if os.environ.get("PN2_STREAMING", "fused") != "sequential":
            mbs = self._materialise(batch)
            if streaming.supported(mbs, self):
                return streaming.run_tree(self, batch, return_loss, scaler=scaler, streaming=True)
        return self._streaming_sequential(batch, return_loss, scaler)

    @staticmethod
    def _materialise(batch):
        """The reference's collate hands over a one-shot generator; its prediction scripts then run two models on the
        same batch and the second sees nothing.  The stream is turned into a list IN the caller's dict, so it
        can be walked again."""
        mbs = batch["mini_batches"]
        if not isinstance(mbs, list):
            mbs = list(mbs)
            batch["mini_batches"] = mbs
        return mbs

    def _streaming_sequential(self, batch, return_loss, scaler=None):
        """The reference's loop: one forward (+ backward) per mini-batch."""
        device = "cuda"
        sem_sum, off_sum, sem_cnt, off_cnt = self._accumulators(batch["cloud_length"], device)
        total_loss, n_mb = 0.0, 0
        loss_dict = {"offset_loss": 0, "semantic_loss": 0}
        if return_loss:
            # The reference gathers the labels on the host per mini-batch (`labels[ids.cpu()].to(device)`, lines
            # 279-280): a device->host sync plus a CPU advanced-indexing op each time (on a many-core host the OpenMP
            # team that op wakes then competes with the autograd thread: 20+ ms per mini-batch measured).  Same
            # values from one upload per tree and a device-side index_select; the running loss likewise stays on the
            # device (float64, like the reference's Python float) and is read back once.
            sem_all = batch["semantic_labels"].squeeze().to(device)
            off_all = batch["offset_labels"].to(device)
            total_dev = torch.zeros((), dtype=torch.float64, device=device)
        for mini_batch in batch["mini_batches"]:
            sem, off, ids, ids_off = self._predict_minibatch(mini_batch)
            self._scatter_minibatch(sem_sum, sem_cnt, ids, sem.detach())
            self._scatter_minibatch(off_sum, off_cnt, ids_off, off.detach())
            if return_loss:
                sem_lab = sem_all.index_select(0, ids)
                off_lab = off_all.index_select(0, ids_off)
                mini_loss, mini_dict = self.get_loss_hierarchical(
                    {"semantic_prediction_logits": sem, "offset_predictions": off}, sem_lab, off_lab, n_points=None)
                if scaler:
                    scaler.scale(mini_loss * 50).backward()
                loss_dict["offset_loss"] += mini_dict["offset_loss"]
                loss_dict["semantic_loss"] += mini_dict["semantic_loss"]
                total_dev += mini_loss.detach().double()
                n_mb += 1
            del sem, off, mini_batch
        if return_loss:
            total_loss = float(total_dev)
            ops.check_status(total_dev.device)   # at the sync the read-back just made: did any kernel give up?
        output = {"semantic_prediction_logits": self._average(sem_sum, sem_cnt),
                  "offset_predictions": self._average(off_sum, off_cnt)}
        if not return_loss:
            return output
        if n_mb > 0:
            loss_dict["offset_loss"] /= n_mb
            loss_dict["semantic_loss"] /= n_mb
            return total_loss / n_mb, loss_dict
        # the reference divides by 0.0 here (lines 321-322); an empty stream is reported as zero loss instead
        return 0.0, loss_dict

    def forward_hierarchical(self, batch, return_loss):
        """Non-streaming variant (reference lines 329-394): accumulate WITH autograd history, average, then
        compute one loss on the averaged predictions.  Fused like the streaming mode unless PN2_STREAMING=sequential."""
        if os.environ.get("PN2_STREAMING", "fused") != "sequential":
            mbs = self._materialise(batch)
            if streaming.supported(mbs, self):
                return streaming.run_tree(self, batch, return_loss, streaming=False)
        sem_sum, off_sum, sem_cnt, off_cnt = self._accumulators(batch["cloud_length"], "cuda")
        for mini_batch in batch["mini_batches"]:
            sem, off, ids, ids_off = self._predict_minibatch(mini_batch)
            sem_sum = self._scatter_minibatch(sem_sum, sem_cnt, ids, sem, differentiable=True)
            off_sum = self._scatter_minibatch(off_sum, off_cnt, ids_off, off, differentiable=True)
        output = {"semantic_prediction_logits": self._average(sem_sum, sem_cnt),
                  "offset_predictions": self._average(off_sum, off_cnt)}
        if return_loss:
            output = self.get_loss_hierarchical(output, batch["semantic_labels"].squeeze(), batch["offset_labels"])
        return output
