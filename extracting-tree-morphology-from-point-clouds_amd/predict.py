"""Per-tree prediction driver sharded by tree (BASELINE configs[4]; reference: ModelTestingScripts/
predict_all_trees_PointNet2.py:29-108 and Modules/Pipeline/ModelPredicting.py:167-260).

For every tree of this rank's shard: rasterise on the device (rasters.build_stream), classify noise with one model and
predict offsets with another through forward_hierarchical_streaming (whole-tree pass), build the reference's outputs --
`pred_full` = [xyz | offset | noise flag], the executed (shifted, de-noised) cloud -- and, when a QSM is given, project the
executed cloud onto its cylinders (Projection.cylinder_project: the kNN-to-QSM step).  Trees are independent: ranks take
contiguous shards (parallel.shard_range) and there is NO collective on the data path; results are gathered as files or by
the caller.
"""
import numpy as np
import torch

from . import Projection, parallel, rasters
from .DataLoading.RasterizedTreeSet import cloud_columns


def predict_tree(model_offset, model_noise, cloud, raster_size=1.0, stride=1.0, minibatch_size=60, cylinders=None):
    """cloud: float array [N, 3] or [N, >=8] (xyz, then the label / feature columns of the 11-column label files; columns
    7: are the network features).  An xyz-only cloud gets the reference dataset's padding -- eight ZERO columns
    (RasterizedTreeSet.py:207-211), i.e. zero features -- through the same code as the dataset path
    (DataLoading.RasterizedTreeSet.cloud_columns), so a cloud predicts the same either way.
    -> dict of numpy arrays."""
    dev = torch.device("cuda", torch.cuda.current_device())
    data = np.asarray(cloud)
    if data.shape[1] != 3 and data.shape[1] < 8:
        raise ValueError(f"predict_tree: a cloud has 3 columns (xyz) or at least 8 (xyz | offset | id | features...), got {data.shape[1]}")
    xyz64 = np.ascontiguousarray(data[:, :3], dtype=np.float64)
    n = len(data)
    pts, _, feats = cloud_columns(data)                                 # the one definition of "features of a cloud"
    pts, feats = pts.contiguous(), feats.contiguous()
    stream = rasters.build_stream(pts, feats, None, raster_size, stride, minibatch_size)
    batch = {"mini_batches": stream, "cloud_length": n}
    with torch.no_grad():
        noise_logits = model_noise.forward_hierarchical_streaming(batch, return_loss=False)["semantic_prediction_logits"]
        offsets = model_offset.forward_hierarchical_streaming(batch, return_loss=False)["offset_predictions"]
    noise_flag = torch.argmax(noise_logits, dim=1)
    executed = pts + offsets                                            # predict_all_trees_PointNet2.py:96
    keep = noise_flag == 0
    out = {"pred_full": np.concatenate([xyz64, offsets.cpu().numpy().astype(np.float64),
                                        noise_flag.cpu().numpy().reshape(-1, 1).astype(np.float64)], axis=1),
           "executed_cloud": executed[keep].cpu().numpy()}
    if cylinders is not None:
        start, radius, axis_length, axis_unit, ids = Projection.cylinder_tensors(cylinders, dev)
        cid, dist, off = Projection.cylinder_project(executed[keep], start, axis_unit, axis_length, radius, ids)
        out["qsm_ids"], out["qsm_distance"], out["qsm_offsets"] = cid.cpu().numpy(), dist.cpu().numpy(), off.cpu().numpy()
    return out


def predict_forest(model_offset, model_noise, trees, cylinders=None, rank=0, world=1, seed=None, **kw):
    """trees: list of clouds (arrays); cylinders: optional list of QSM tables, one per tree.  Returns {tree index: result}
    for this rank's contiguous shard.  seed: when given, the FPS start draws of tree i come from torch.manual_seed(seed + i),
    so a tree's prediction does not depend on which rank processes it or on the trees before it."""
    lo, hi = parallel.shard_range(len(trees), rank, world)
    out = {}
    for i in range(lo, hi):
        if seed is not None:
            torch.manual_seed(seed + i)
        out[i] = predict_tree(model_offset, model_noise, trees[i], cylinders=None if cylinders is None else cylinders[i], **kw)
    return out
