"""Tuning aid: where does a cooperative forward chain spend its time?  PN2_COOP_DIAG hands the kernel a stamp buffer
([workgroup][phase][8] 100 MHz counter values); prints, per phase, the median over the workgroups that worked of:
take | finalize | tile(s) | done+take | finish, in microseconds."""
import os
import sys

import numpy as np
import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg  # noqa: E402
from tools.bench_chain import SHAPES  # noqa: E402

load_pkg()
from pn2_amd import mlp  # noqa: E402

for name, rows, cin, widths, K, skip in SHAPES:
    if len(sys.argv) > 1 and name not in sys.argv[1].split(","):
        continue
    torch.manual_seed(0)
    layers, c = [], cin
    for w in widths:
        layers.append((nn.Conv2d(c, w, 1).cuda(), nn.BatchNorm2d(w).cuda().train(), True))
        c = w
    x = torch.randn(rows, cin, device="cuda")
    buf = torch.zeros(256 * 8 * 8, dtype=torch.int64, device="cuda")
    for _ in range(3):
        mlp.chain_rows(x, layers, pool_k=K)
    torch.cuda.synchronize()
    os.environ["PN2_COOP_DIAG"] = hex(buf.data_ptr())
    mlp.chain_rows(x, layers, pool_k=K)
    torch.cuda.synchronize()
    os.environ.pop("PN2_COOP_DIAG")
    d = buf.cpu().numpy().reshape(256, 8, 8)
    t0 = d[:, 0, 0][d[:, 0, 0] > 0].min()
    print(f"== {name}: rows {rows} {cin}->{widths} K={K}")
    for ph in range(len(widths) + 1):
        s = d[:, ph, :].astype(np.float64)
        ran = s[:, 0] > 0
        worked = ran & (s[:, 2] > 0)
        us = lambda a: 0.01 * np.median(a) if len(a) else float("nan")
        w = s[worked]
        last = ph == len(widths)
        print(f"  phase {ph}: {int(ran.sum())} wgs ({int(worked.sum())} worked)  start +{0.01 * (s[ran, 0].min() - t0):6.1f} us | take {us(w[:, 1] - w[:, 0]):5.1f} "
              f"| finalize {us(w[:, 2] - w[:, 1]):5.1f} | last tile {us(w[:, 3] - w[:, 2]):5.1f} | done+take {us(w[:, 4] - w[:, 3]):5.1f} "
              f"| finish {float('nan') if last else us(w[:, 6] - w[:, 5]):5.1f} | phase span {0.01 * (s[ran, 5 if last else 6].max() - s[ran, 0].min()):6.1f} us")
