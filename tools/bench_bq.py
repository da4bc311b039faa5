"""Tuning aid (GPU): time ball_query for candidate (Q, nseg) plans.
    python tools/bench_bq.py [N] [S] [B] [radius] [nsample]"""
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
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
radius = float(sys.argv[4]) if len(sys.argv) > 4 else 0.1
K = int(sys.argv[5]) if len(sys.argv) > 5 else 32
xyz = np.stack([gaussian_branch_tree(N, seed=s)[0] for s in range(B)])
x = torch.from_numpy(xyz.transpose(0, 2, 1).copy()).cuda().permute(0, 2, 1)
start = torch.zeros(B, dtype=torch.long, device="cuda")
_, new_xyz = ops.furthest_point_sample(x, S, start)
ref = None
plans = [""] + [f"{q},{g}" for q in (2, 4, 8) for g in (1, 8, 32, 64)]
for cfg in plans:
    if cfg:
        os.environ["PN2_BQ_PLAN"] = cfg
    else:
        os.environ.pop("PN2_BQ_PLAN", None)
    idx = ops.ball_query(radius, K, x, new_xyz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        idx = ops.ball_query(radius, K, x, new_xyz)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    if ref is None:
        ref = idx.clone()
    print(f"plan {cfg or 'auto':6s}: {1e6 * dt:8.1f} us  same_result={bool(torch.equal(ref, idx))}", flush=True)
