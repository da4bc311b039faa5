"""Tuning aid (GPU): time farthest_point_sample for the candidate (PPT, T) configurations.
    python tools/bench_fps.py [N] [npoint] [B]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402

load_pkg()
from pn2_amd import ops  # noqa: E402
from pn2_amd.synthetic import gaussian_branch_tree  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
npoint = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
xyz = np.stack([gaussian_branch_tree(N, seed=s)[0] for s in range(B)])
x = torch.from_numpy(xyz.transpose(0, 2, 1).copy()).cuda().permute(0, 2, 1)
start = torch.zeros(B, dtype=torch.long, device="cuda")
ref = None
for cfg in ["", "1,256", "2,256", "4,256", "8,256", "16,256", "4,1024", "8,1024", "16,512", "32,512"]:
    if cfg:
        os.environ["PN2_FPS_CFG"] = cfg
    else:
        os.environ.pop("PN2_FPS_CFG", None)
    try:
        idx, _ = ops.furthest_point_sample(x, npoint, start)
    except RuntimeError as e:
        print(f"cfg {cfg or 'auto':8s}: n/a ({e})")
        continue
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        idx, _ = ops.furthest_point_sample(x, npoint, start)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    if ref is None:
        ref = idx.clone()
    same = bool(torch.equal(ref, idx))
    print(f"cfg {cfg or 'auto':8s}: {1e3 * dt:8.3f} ms  {1e6 * dt / npoint:6.2f} us/step  same_result={same}")
