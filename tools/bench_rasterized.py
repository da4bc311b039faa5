"""GPU: the reference's own whole-tree training mode (train_PointNet2.py --hierarchical --streaming, SURVEY 8d
config 2(i)): the 262144-point tree cut into 1 m rasters, mini-batches of 10 rasters zero-padded to the group maximum,
depth 5, forward + loss + backward per mini-batch with gradient accumulation, one optimizer step per tree.
Mini-batches are resident on the device before the timed region (the reference builds them on the host).
    python tools/bench_rasterized.py [steps]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402

load_pkg()
from pn2_amd import parallel  # noqa: E402
from pn2_amd.PointNet2.PointNet2 import PointNet2  # noqa: E402
from pn2_amd.synthetic import gaussian_branch_tree, rasterize  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3
N, MBS = 262144, 10
xyz, off, _ = gaussian_branch_tree(N, seed=0)
rasters = rasterize(xyz, 1.0, 1.0)
dev = torch.device("cuda:0")


class Scaler:                       # the reference passes a GradScaler; fp32 here, so scaling is the identity
    def scale(self, x):
        return x


def build_minibatches():
    out = []
    for k in range(0, len(rasters), MBS):
        group = rasters[k:k + MBS]
        nmax = max(len(r) for r in group)
        coords = np.zeros((len(group), 3, nmax), np.float32)
        mpad = np.zeros((len(group), nmax), bool)
        for i, r in enumerate(group):
            coords[i, :, :len(r)] = xyz[r].T                       # raw metres, like RasterizedTreeSet.py:427
            mpad[i, :len(r)] = True
        ids = np.concatenate(group)
        out.append({"coords": torch.from_numpy(coords).to(dev), "feats": torch.ones(len(group), 4, nmax, device=dev) *
                    torch.from_numpy(mpad).to(dev)[:, None, :], "masks_pad": torch.from_numpy(mpad).to(dev),
                    "masks_off": torch.ones(len(ids), dtype=torch.bool, device=dev), "point_ids": torch.from_numpy(ids).to(dev)})
    return out


mbs = build_minibatches()
padded = sum(int(m["coords"].shape[0] * m["coords"].shape[2]) for m in mbs)
torch.manual_seed(0)
model = PointNet2(depth=5, loss_multiplier_semantic=0).to(dev).train()
grads = parallel.FlatGradAllReduce(model)
opt = torch.optim.AdamW(model.parameters(), lr=0.01, weight_decay=1e-3, fused=True)
labels = {"cloud_length": N, "semantic_labels": torch.zeros(N, 1, dtype=torch.long), "offset_labels": torch.from_numpy(off)}


def step():
    grads.zero()
    batch = dict(labels, mini_batches=iter(mbs))
    loss, _ = model.forward_hierarchical_streaming(batch, return_loss=True, scaler=Scaler())
    opt.step()
    return loss


step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / STEPS
print(json.dumps({"metric": "points/sec fwd+bwd, PointNet2 depth 5, rasterised streaming mode (1 m rasters, mini-batches of 10)",
                  "value": N / dt, "unit": "points/s", "ms_per_tree": 1e3 * dt, "rasters": len(rasters), "mini_batches": len(mbs),
                  "padded_points": padded, "padding_factor": padded / N, "ms_per_mini_batch": 1e3 * dt / len(mbs),
                  "loss": float(loss), "dtype": "f32", "data": "synthetic"}))
