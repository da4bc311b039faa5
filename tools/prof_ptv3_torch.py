"""GPU: which torch (non-library) kernels the PTv3 backbone forward spends its time in (torch.profiler, 1 M voxels)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from __graft_entry__ import load_pkg
load_pkg()
from bench_ptv3_model import plot_voxels
from pn2_amd.PointTransformerV3.PointTransformerV3 import PointTransformerV3
train = "--train" in sys.argv
torch.manual_seed(0)
model = PointTransformerV3(in_channels=4).cuda()
model.train() if train else model.eval()
g = plot_voxels(1 << 20, 0.02)
N = len(g)
grid = torch.from_numpy(g.astype(np.int32)).cuda()
coord = grid.float() * 0.02
feat = torch.randn(N, 4, device="cuda")
batch = torch.zeros(N, dtype=torch.int64, device="cuda")
def run():
    d = {"feat": feat, "coord": coord, "grid_coord": grid, "batch": batch}
    if train:
        model.zero_grad(set_to_none=True)
        model(d).feat.square().mean().backward()
    else:
        with torch.no_grad():
            model(d)
for _ in range(2):
    run()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    run()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if t >= 200:
        rows.append((t, e.key, e.count))
tot = sum(r[0] for r in rows)
print(f"device time in listed entries: {tot / 1e3:.1f} ms")
for t, k, c in sorted(rows, reverse=True)[:45]:
    print(f"{t / 1e3:9.2f} ms x{c:<4d} {k[:110]}")
