"""GPU: stage timing of rasters.build_stream (wall clock with a sync after each stage)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd import rasters
from pn2_amd.synthetic import gaussian_branch_tree
import cProfile, pstats
xyz, _, _ = gaussian_branch_tree(262144, seed=0)
dev = torch.device("cuda")
pts = torch.from_numpy(xyz).to(dev); feats = torch.ones(len(xyz), 4, device=dev); om = torch.ones(len(xyz), dtype=torch.bool, device=dev)
for _ in range(3):
    rasters.build_stream(pts, feats, om, 1.0, 1.0, 10)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    rasters.rasterize_points(pts, 1.0, 1.0)
torch.cuda.synchronize()
print("rasterize_points", (time.perf_counter() - t0) / 5 * 1e3, "ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    rasters.build_stream(pts, feats, om, 1.0, 1.0, 10)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
