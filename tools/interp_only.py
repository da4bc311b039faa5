"""GPU: three_interpolate forward/backward at the FP1 shape (N=262144, S=1024, D=128) a few times (PMC target)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd import ops
N, S, D = 262144, 1024, 128
torch.manual_seed(0)
idx = torch.randint(0, S, (1, N, 3), device="cuda", dtype=torch.int32)
w = torch.rand(1, N, 3, device="cuda")
p2 = torch.randn(1, S, D, device="cuda", requires_grad=True)
g = torch.randn(1, N, D, device="cuda")
for _ in range(4):
    out = ops.ThreeInterpolateConcat.apply(None, p2, idx, w)
    out.backward(g)
torch.cuda.synchronize()
print("done")
