"""GPU diagnostic: where does a streaming-mode mini-batch spend its time (host vs kernels)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd import parallel, _hip
from pn2_amd.PointNet2.PointNet2 import PointNet2
from pn2_amd.synthetic import gaussian_branch_tree, rasterize
N, MBS = 262144, 10
xyz, off, _ = gaussian_branch_tree(N, seed=0)
rasters = rasterize(xyz, 1.0, 1.0)
dev = torch.device("cuda:0")
group = rasters[100:110]
nmax = max(len(r) for r in group)
coords = np.zeros((len(group), 3, nmax), np.float32); mpad = np.zeros((len(group), nmax), bool)
for i, r in enumerate(group):
    coords[i, :, :len(r)] = xyz[r].T; mpad[i, :len(r)] = True
ids = np.concatenate(group)
mb = {"coords": torch.from_numpy(coords).to(dev), "feats": torch.ones(len(group), 4, nmax, device=dev), "masks_pad": torch.from_numpy(mpad).to(dev),
      "masks_off": torch.ones(len(ids), dtype=torch.bool, device=dev), "point_ids": torch.from_numpy(ids).to(dev)}
print("mini-batch", coords.shape)
torch.manual_seed(0)
model = PointNet2(depth=5, loss_multiplier_semantic=0).to(dev).train()
off_lab = torch.from_numpy(off[ids]).to(dev); sem_lab = torch.zeros(len(ids), dtype=torch.long, device=dev)
def one():
    sem, offp, i1, i2 = model._predict_minibatch(mb)
    loss, _ = model.get_loss_hierarchical({"semantic_prediction_logits": sem, "offset_predictions": offp}, sem_lab, off_lab)
    (loss * 50).backward()
for _ in range(3): one()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): one()
torch.cuda.synchronize(); print("fwd+bwd per mini-batch ms:", (time.perf_counter() - t0) / 5 * 1e3)
groups = _hip.kernel_profile(lambda: (one(), torch.cuda.synchronize()))
print("library kernel time ms:", sum(g["ms"] for g in groups), "launches", sum(g["calls"] for g in groups))
for g in sorted(groups, key=lambda g: -g["ms"])[:8]: print("  ", g["name"], g["calls"], round(g["ms"], 3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    one(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=50))

# ---- the streaming loop itself over 4 mini-batches
class Scaler:
    def scale(self, x):
        return x
mbs = []
for k in range(100, 140, 10):
    group = rasters[k:k + 10]
    nmax = max(len(r) for r in group)
    coords = np.zeros((len(group), 3, nmax), np.float32); mpad = np.zeros((len(group), nmax), bool)
    for i, r in enumerate(group):
        coords[i, :, :len(r)] = xyz[r].T; mpad[i, :len(r)] = True
    ids = np.concatenate(group)
    mbs.append({"coords": torch.from_numpy(coords).to(dev), "feats": torch.ones(len(group), 4, nmax, device=dev),
                "masks_pad": torch.from_numpy(mpad).to(dev), "masks_off": torch.ones(len(ids), dtype=torch.bool, device=dev),
                "point_ids": torch.from_numpy(ids).to(dev)})
labels = {"cloud_length": N, "semantic_labels": torch.zeros(N, 1, dtype=torch.long), "offset_labels": torch.from_numpy(off)}
def stream():
    model.zero_grad()
    return model.forward_hierarchical_streaming(dict(labels, mini_batches=iter(mbs)), return_loss=True, scaler=Scaler())
stream(); torch.cuda.synchronize(); t0 = time.perf_counter(); stream(); torch.cuda.synchronize()
print("streaming, 4 mini-batches, ms:", (time.perf_counter() - t0) * 1e3)
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    stream(); torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cpu_time_total", row_limit=22, max_name_column_width=60))

s0 = torch.cuda.memory_stats()
stream(); torch.cuda.synchronize()
s1 = torch.cuda.memory_stats()
for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams", "segment.all.allocated", "segment.all.freed", "allocation.all.allocated"):
    print(k, s1.get(k, 0) - s0.get(k, 0))
print("reserved MB", torch.cuda.memory_reserved() / 1e6, "allocated MB", torch.cuda.memory_allocated() / 1e6)
import torch.utils.benchmark as tb
w = next(model.parameters())
t0 = time.perf_counter()
for _ in range(200): z = torch.zeros_like(w)
torch.cuda.synchronize(); print("zeros_like small, us each:", (time.perf_counter() - t0) / 200 * 1e6)
print("threads", torch.get_num_threads())
lab = labels["semantic_labels"].squeeze(); idc = mbs[0]["point_ids"].cpu()
t0 = time.perf_counter()
for _ in range(20): v = lab[idc]
print("cpu index us each:", (time.perf_counter() - t0) / 20 * 1e6)
