"""Hierarchical raster dataset with the reference's class name, constructor and batch layout
(Modules/DataLoading/RasterizedTreeSet.py:150-459), with the rasterisation and the collate done on the device
(rasters.py, csrc/raster.hip) instead of O(#rasters x N) boolean masks and per-mini-batch host padding.

Differences, all documented in rasters.py: the box grid comes from the boxes stored in the JSON when it carries them (like
the reference, RasterizedTreeSet.py:228-238) and is computed from the cloud with the constructor's raster_size / stride
otherwise, and
`collate_fn_streaming` returns a re-iterable list of mini-batches instead of a one-shot generator (the
reference's second model of predict_*_PointNet2.py sees an exhausted generator).
"""
import json
import os

import numpy as np
import torch
from torch.utils.data import DataLoader, Dataset

from .. import rasters


def load_cloud(path):
    """The .npy / .txt branches of Modules/Utils.py:190-218 (laz/las need laspy, which the reference treats as optional)."""
    ext = os.path.splitext(path)[1].lower()
    if ext == ".npy":
        pts = np.load(path)
        return pts.reshape(-1, 3) if pts.ndim == 1 else pts
    if ext == ".txt":
        try:
            return np.loadtxt(path, delimiter=" ")
        except ValueError:
            return np.loadtxt(path, delimiter=",")
    raise ValueError(f"unsupported cloud format {ext}")


def rasterize_clouds(data_paths, json_path, raster_size, stride, store_metadata):
    """Reference signature (ModelPredicting.py:98): writes {tree_id: {"rasters": [{raster_id, bounds}], "path"}}."""
    meta, n_rasters = {}, 0
    for cloud_path in data_paths:
        name = os.path.splitext(os.path.basename(cloud_path))[0]
        plot, tree = name.split("_")[:2]
        cloud = load_cloud(cloud_path)
        pts = torch.from_numpy(np.ascontiguousarray(cloud[:, :3], dtype=np.float32)).cuda()
        lo, hi = cloud[:, :3].min(axis=0), cloud[:, :3].max(axis=0)
        bounds = rasters.grid_bounds(lo, hi, raster_size, stride)
        _, lengths, boxes, dims, bounds = rasters.rasterize_points(pts, raster_size, stride, bounds=bounds)
        n_rasters += len(lengths)
        if store_metadata:
            meta.setdefault(f"{plot}_{tree}", {"rasters": [], "path": cloud_path})["rasters"] += \
                rasters.raster_bounds_metadata(boxes, dims, bounds, raster_size)
    if store_metadata:
        with open(json_path, "w") as f:
            json.dump(meta, f, indent=4)
    return n_rasters


def cloud_columns(data):
    """The column convention of the reference's __getitem__ (:206-218) for a cloud in memory: an xyz-only cloud is padded
    with eight ZERO columns first, then columns are xyz | offset(3) | cylinder id | features...
    -> (points [N,3], offsets [N,3], features [N,F]) float32 on the current device.  The ONE definition of "features of a
    cloud": the dataset path and predict.predict_tree both go through here."""
    data = np.asarray(data)
    if data.shape[1] == 3:
        data = np.hstack((data, np.zeros((data.shape[0], 8), dtype=data.dtype)))
    dev = torch.device("cuda", torch.cuda.current_device())
    full = torch.from_numpy(np.ascontiguousarray(data, dtype=np.float32)).to(dev)
    return full[:, :3], full[:, 3:6], full[:, 7:]


class RasterizedTreeSet_Hierarchical(Dataset):
    def __init__(self, paths, training=True, logger=None, data_augmentations=None, noise_distance=0.05, minibatch_size=20,
                 single_sample=False, raster_size=1.0, stride=1.0):
        """paths: JSON file(s) as written by rasterize_clouds ({tree_id: {"path": ...}}), or a list of cloud files."""
        self.data = {}
        if isinstance(paths, str):
            paths = [paths]
        for p in paths:
            if p.endswith(".json"):
                with open(p) as f:
                    for key, value in json.load(f).items():
                        self.data.setdefault(key, value)
            else:
                self.data[os.path.splitext(os.path.basename(p))[0]] = {"path": p}
        self.tree_keys = list(self.data.keys())
        if single_sample and self.tree_keys:
            self.tree_keys = self.tree_keys[:1]
        self.training, self.logger, self.data_augmentations = training, logger, data_augmentations
        self.noise_distance, self.minibatch_size = noise_distance, minibatch_size
        self.raster_size, self.stride = raster_size, stride

    def __len__(self):
        return len(self.tree_keys)

    def __getitem__(self, idx):
        entry = self.data[self.tree_keys[idx]]
        item = self.from_array(load_cloud(entry["path"]))
        if entry.get("rasters"):
            # the JSON's own boxes decide (the reference reads them, RasterizedTreeSet.py:228-238) -- not the constructor's
            # raster_size / stride, which only serve clouds that come without metadata
            bounds, size, stride = rasters.bounds_from_metadata(entry["rasters"])
            item["raster_grid"] = (bounds, size, stride if stride is not None else size)
        return item

    def from_array(self, data):
        """The body of the reference's __getitem__ (:201-268) for a cloud that is already in memory: columns
        xyz | offset(3) | cylinder id | features..."""
        points, offsets, features = cloud_columns(data)
        norms = offsets.norm(dim=1)
        return {"points": points, "features": features, "offset_mask": norms <= self.noise_distance,
                "cloud_length": len(data), "offset_labels": offsets.contiguous(), "semantic_labels": (norms > self.noise_distance).long()}

    def collate_fn_streaming(self, batch):
        tree = batch[0]
        bounds, size, stride = tree.get("raster_grid", (None, self.raster_size, self.stride))
        stream = rasters.build_stream(tree["points"], tree["features"], tree["offset_mask"], size, stride, self.minibatch_size,
                                      bounds=bounds)
        return {"mini_batches": stream, "cloud_length": tree["cloud_length"], "offset_labels": tree["offset_labels"],
                "semantic_labels": tree["semantic_labels"]}

    collate_fn = collate_fn_streaming


def get_dataloader(dataset, batch_size, num_workers, training, collate_fn):
    # the items already live on the device: no worker processes, no pinning
    return DataLoader(dataset, batch_size=batch_size, shuffle=training, num_workers=0, collate_fn=collate_fn)
