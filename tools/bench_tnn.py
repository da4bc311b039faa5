"""Tuning aid (GPU): time three_nn for P = 1, 2, 4 dense points per thread.
    python tools/bench_tnn.py [N] [S] [B]"""
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
xyz = np.stack([gaussian_branch_tree(N, seed=s)[0] for s in range(B)])
x = torch.from_numpy(xyz.transpose(0, 2, 1).copy()).cuda().permute(0, 2, 1)
_, sparse = ops.furthest_point_sample(x, S, torch.zeros(B, dtype=torch.long, device="cuda"))
ref = None
for cfg in ["", "1", "2", "4"]:
    if cfg:
        os.environ["PN2_TNN_P"] = cfg
    else:
        os.environ.pop("PN2_TNN_P", None)
    idx, w = ops.three_nn(x, sparse)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        idx, w = ops.three_nn(x, sparse)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    if ref is None:
        ref = (idx.clone(), w.clone())
    print(f"P {cfg or 'auto':5s}: {1e6 * dt:8.1f} us  same_result={bool(torch.equal(ref[0], idx) and torch.equal(ref[1], w))}", flush=True)
