"""GPU: the submanifold conv's weight-gradient kernel by layer shape (voxels of the 1 M-voxel plot, pooled like the backbone)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from __graft_entry__ import load_pkg
load_pkg()
from bench_ptv3_model import plot_voxels
from pn2_amd import _hip
from pn2_amd.PointTransformerV3.cpe import SubMConv3d, subm_neighbors
g = plot_voxels(1 << 20, 0.02)
for shift, C in ((0, 32), (1, 64), (2, 128), (3, 256), (4, 512)):
    gg = np.unique(g >> shift, axis=0)
    N = len(gg)
    grid = torch.from_numpy(gg.astype(np.int32)).cuda()
    nbr = subm_neighbors(None, grid, 3)
    valid = float((nbr >= 0).float().mean()) * 27
    conv = SubMConv3d(C, C).cuda()
    x = torch.randn(N, C, device="cuda", requires_grad=True)
    y = conv(x, nbr)
    go = torch.randn_like(y)
    for _ in range(2):
        y.backward(go, retain_graph=True)
    rows = _hip.kernel_profile(lambda: (y.backward(go, retain_graph=True), torch.cuda.synchronize()))
    t = {r["name"]: r["ms"] for r in rows}
    print(f"N={N:8d} C={C:4d} neighbours {valid:4.1f}: wgrad {t.get('ptv3_subm_wgrad', 0):7.3f} ms, dgrad (conv kernel) {t.get('ptv3_subm_conv', 0):7.3f} ms")
