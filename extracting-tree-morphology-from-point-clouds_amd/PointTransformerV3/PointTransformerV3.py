"""The PointTransformerV3 backbone with the reference's constructor, module tree and parameter names
(Modules/PointTransformerV3/PointTransformerV3.py:261-460), so that a state dict of the reference's `backbone` loads: embedding
(5 x 5 x 5 stem) -> five encoder stages (serialized pooling, Blocks) -> four decoder stages (serialized unpooling, Blocks).
Built from blocks.py (which says what is this library's kernel and what is a plain library call).  Inference and training,
PARITY UNPINNED (see blocks.py); the repository's configuration -- enable_flash = enable_rpe = False, no PDNorm -- is what is built,
the other options raise."""
from functools import partial

import torch
import torch.nn as nn

from .linear import LayerNorm, Linear
from .blocks import Block, Embedding, Point, PointModule, PointSequential, SerializedPooling, SerializedUnpooling


class PointTransformerV3(PointModule):
    def __init__(self, in_channels=6, order=("z", "z-trans", "hilbert", "hilbert-trans"), stride=(2, 2, 2, 2),
                 enc_depths=(2, 2, 2, 6, 2), enc_channels=(32, 64, 128, 256, 512), enc_num_head=(2, 4, 8, 16, 32),
                 enc_patch_size=(1024, 1024, 1024, 1024, 1024), dec_depths=(2, 2, 2, 2), dec_channels=(64, 64, 128, 256),
                 dec_num_head=(4, 4, 8, 16), dec_patch_size=(1024, 1024, 1024, 1024), mlp_ratio=4, qkv_bias=True, qk_scale=None,
                 attn_drop=0.0, proj_drop=0.0, drop_path=0.3, pre_norm=True, shuffle_orders=True, enable_rpe=False,
                 enable_flash=False, upcast_attention=False, upcast_softmax=False, cls_mode=False, pdnorm_bn=False,
                 pdnorm_ln=False, pdnorm_decouple=True, pdnorm_adaptive=False, pdnorm_affine=True,
                 pdnorm_conditions=("ScanNet", "S3DIS", "Structured3D")):
        super().__init__()
        if pdnorm_bn or pdnorm_ln:
            raise NotImplementedError("PDNorm is not built (the repository's model uses plain BatchNorm1d / LayerNorm)")
        self.num_stages = len(enc_depths)
        self.order = [order] if isinstance(order, str) else order
        self.cls_mode, self.shuffle_orders = cls_mode, shuffle_orders
        assert self.num_stages == len(stride) + 1 == len(enc_channels) == len(enc_num_head) == len(enc_patch_size)
        assert self.cls_mode or self.num_stages == len(dec_depths) + 1 == len(dec_channels) + 1 == len(dec_num_head) + 1 \
            == len(dec_patch_size) + 1
        bn_layer = partial(nn.BatchNorm1d, eps=1e-3, momentum=0.01)
        ln_layer = LayerNorm
        act_layer = nn.GELU
        block = partial(Block, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale, attn_drop=attn_drop, proj_drop=proj_drop,
                        norm_layer=ln_layer, act_layer=act_layer, pre_norm=pre_norm, enable_rpe=enable_rpe, enable_flash=enable_flash,
                        upcast_attention=upcast_attention, upcast_softmax=upcast_softmax)
        self.embedding = Embedding(in_channels=in_channels, embed_channels=enc_channels[0], norm_layer=bn_layer, act_layer=act_layer)

        enc_drop_path = [x.item() for x in torch.linspace(0, drop_path, sum(enc_depths))]
        self.enc = PointSequential()
        for s in range(self.num_stages):
            dp = enc_drop_path[sum(enc_depths[:s]):sum(enc_depths[:s + 1])]
            enc = PointSequential()
            if s > 0:
                enc.add(SerializedPooling(in_channels=enc_channels[s - 1], out_channels=enc_channels[s], stride=stride[s - 1],
                                          norm_layer=bn_layer, act_layer=act_layer), name="down")
            for i in range(enc_depths[s]):
                enc.add(block(channels=enc_channels[s], num_heads=enc_num_head[s], patch_size=enc_patch_size[s], drop_path=dp[i],
                              order_index=i % len(self.order), cpe_indice_key=f"stage{s}"), name=f"block{i}")
            if len(enc) != 0:
                self.enc.add(module=enc, name=f"enc{s}")

        if not self.cls_mode:
            dec_drop_path = [x.item() for x in torch.linspace(0, drop_path, sum(dec_depths))]
            self.dec = PointSequential()
            dec_channels = list(dec_channels) + [enc_channels[-1]]
            for s in reversed(range(self.num_stages - 1)):
                dp = dec_drop_path[sum(dec_depths[:s]):sum(dec_depths[:s + 1])]
                dp.reverse()
                dec = PointSequential()
                dec.add(SerializedUnpooling(in_channels=dec_channels[s + 1], skip_channels=enc_channels[s],
                                            out_channels=dec_channels[s], norm_layer=bn_layer, act_layer=act_layer), name="up")
                for i in range(dec_depths[s]):
                    dec.add(block(channels=dec_channels[s], num_heads=dec_num_head[s], patch_size=dec_patch_size[s], drop_path=dp[i],
                                  order_index=i % len(self.order), cpe_indice_key=f"stage{s}"), name=f"block{i}")
                self.dec.add(module=dec, name=f"dec{s}")

    def forward(self, data_dict):
        """data_dict: "feat" [N, C], "grid_coord" [N, 3] (or "coord" + "grid_size"), "coord" [N, 3], "offset" or "batch" -> the
        Point of the finest stage with its new "feat" [N, dec_channels[0]] (:445-460)."""
        point = Point(data_dict)
        point.serialization(order=self.order, shuffle_orders=self.shuffle_orders)
        point.sparsify()
        point = self.embedding(point)
        point = self.enc(point)
        if not self.cls_mode:
            point = self.dec(point)
        return point


class MLP_Head(nn.Sequential):
    """blocks.py:42-62: Linear -> norm -> ReLU (num_layers - 1 times) -> Linear."""

    def __init__(self, in_channels, out_channels, norm_fn=None, num_layers=2):
        modules = []
        for _ in range(num_layers - 1):
            modules.append(Linear(in_channels, in_channels))
            if norm_fn:
                modules.append(norm_fn(in_channels))
            modules.append(nn.ReLU())
        modules.append(Linear(in_channels, out_channels))
        super().__init__(*modules)

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight)
                nn.init.constant_(m.bias, 0)
        nn.init.normal_(self[-1].weight, 0, 0.01)
        nn.init.constant_(self[-1].bias, 0)


class PointTransformerWithHeads(nn.Module):
    """PointTransformerV3.py:19-258 at inference: the backbone on the voxels of `coords` at `voxel_size` (several points may share a
    voxel: a duplicate voxel is represented by its lowest-index point in the neighbour tables), the semantic and the offset head on
    its 64 output features, `forward(batch, return_loss)` and the prediction-averaging `forward_hierarchical_streaming(batch,
    return_loss=False)`.  `forward(batch, return_loss=True)` is the training step's forward (loss, loss_dict): every stage is
    differentiable (attention and submanifold-conv backward kernels of this library, torch autograd for the dense layers)."""

    def __init__(self, dim_feat=4, use_feats=False, voxel_size=0.02, loss_multiplier_semantic=1, loss_multiplier_offset=1,
                 enable_flash=False, optional_autocast=True, **kwargs):
        super().__init__()
        self.voxel_size, self.use_feats, self.optional_autocast = voxel_size, use_feats, optional_autocast
        self.loss_multiplier_semantic, self.loss_multiplier_offset = loss_multiplier_semantic, loss_multiplier_offset
        self.backbone = PointTransformerV3(in_channels=dim_feat, enable_flash=enable_flash)
        norm_fn = partial(nn.BatchNorm1d, eps=1e-4, momentum=0.1)
        self.semantic_linear = MLP_Head(64, 2, norm_fn=norm_fn, num_layers=2)
        self.offset_linear = MLP_Head(64, 3, norm_fn=norm_fn, num_layers=2)
        self.init_weights()

    def init_weights(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm1d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
            elif isinstance(m, MLP_Head):
                m.init_weights()

    def _point_dict(self, batch):
        feats = batch["feats"]
        if not self.use_feats:
            feats = torch.ones_like(feats)
        return {"coord": batch["coords"].to("cuda"), "feat": feats.to("cuda"), "grid_size": self.voxel_size,
                "batch": batch["batch_ids"].to("cuda")}

    def forward_backbone(self, point_dict, **kwargs):
        return self.backbone(point_dict)

    def forward_head(self, backbone_output, **kwargs):
        feats = backbone_output["feat"]
        return {"backbone_feats": feats, "semantic_prediction_logits": self.semantic_linear(feats),
                "offset_predictions": self.offset_linear(feats)}

    def forward(self, batch, return_loss, **kwargs):
        output = self.forward_head(self.forward_backbone(self._point_dict(batch)))
        if return_loss:
            output = self.get_loss(model_output=output, **batch)
        return output

    def get_loss(self, model_output, semantic_labels, offset_labels, masks_off, **kwargs):
        from ..Loss import point_wise_loss
        sem, off = point_wise_loss(model_output["semantic_prediction_logits"].float(), model_output["offset_predictions"][masks_off].float(),
                                   semantic_labels, offset_labels[masks_off])
        loss_dict = {"semantic_loss": sem * self.loss_multiplier_semantic, "offset_loss": off * self.loss_multiplier_offset}
        return sum(loss_dict.values()), loss_dict

    @torch.no_grad()
    def forward_hierarchical_streaming(self, batch, return_loss=False, scaler=None):
        """:117-243 for return_loss=False: per-mini-batch predictions averaged over the tree (`avg[ids] += x`, the reference's
        index_put: one contribution per id and mini-batch)."""
        if return_loss:
            raise NotImplementedError("PointTransformerWithHeads.forward_hierarchical_streaming(return_loss=True): the reference's own "
                                      "code for this pass cannot run (it permutes the backbone's Point, PointTransformerV3.py:161); train "
                                      "through forward(batch, return_loss=True)")
        n, dev = batch["cloud_length"], "cuda"
        avg_off, avg_sem = torch.zeros((n, 3), device=dev), torch.zeros((n, 2), device=dev)
        cnt_sem, cnt_off = torch.zeros((n, 1), device=dev), torch.zeros((n, 1), device=dev)
        for mb in batch["mini_batches"]:
            out = self.forward_head(self.forward_backbone(self._point_dict(mb)))
            ids, mask_off = mb["point_ids"].to(dev), mb["masks_off"].to(dev)
            sem, off = out["semantic_prediction_logits"], out["offset_predictions"]
            if "masks_pad" in mb and sem.shape[0] == mb["masks_pad"].numel():
                keep = mb["masks_pad"].reshape(-1).to(dev)
                sem, off = sem[keep], off[keep]
            off, ids_off = off[mask_off], ids[mask_off]
            avg_sem[ids] += sem
            avg_off[ids_off] += off
            cnt_sem[ids] += 1
            cnt_off[ids_off] += 1
        ok_s, ok_o = cnt_sem.squeeze(1) > 0, cnt_off.squeeze(1) > 0
        avg_sem[ok_s] /= cnt_sem[ok_s]
        avg_off[ok_o] /= cnt_off[ok_o]
        return {"semantic_prediction_logits": avg_sem, "offset_predictions": avg_off}
