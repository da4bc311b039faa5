"""GPU: run the FP1-shaped MLP chain (262144 rows, 128 -> 128 -> 128 -> 128) forward + backward a few times
(target for rocprofv3 --pmc passes)."""
import os, sys
import torch, torch.nn as nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd.mlp import chain_rows
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
torch.manual_seed(0)
layers = [(nn.Conv1d(128, 128, 1).cuda(), nn.BatchNorm1d(128).cuda().train(), True) for _ in range(3)]
x = torch.randn(rows, 128, device="cuda", requires_grad=True)
g = torch.randn(rows, 128, device="cuda")
for _ in range(4):
    y = chain_rows(x, layers)
    y.backward(g)
torch.cuda.synchronize()
print("done")
