"""GPU: run-to-run spread of the PTv3 backbone's parameter gradients (same model, same input, 8 passes) -- which parameters
differ between passes, in backward order."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_pkg
load_pkg()
from pn2_amd.PointTransformerV3.PointTransformerV3 import PointTransformerV3
from test_ptv3_train import _small_cfg, _voxels
cfg = _small_cfg()
torch.manual_seed(0)
model = PointTransformerV3(**cfg).cuda().eval()
for m in model.modules():
    if hasattr(m, "shuffle_orders"):
        m.shuffle_orders = False
clouds = []
for b in range(2):
    grid = _voxels(5000, seed=20 + b)
    clouds.append(np.concatenate([np.full((len(grid), 1), b), grid], 1))
vox = np.concatenate(clouds)
batch, grid = vox[:, 0].copy(), vox[:, 1:].copy()
N = len(grid)
rng = np.random.default_rng(2)
feat = torch.from_numpy(rng.standard_normal((N, 4)).astype(np.float32)).cuda()
coord = torch.from_numpy((grid * 0.05).astype(np.float32)).cuda()
R = torch.from_numpy(rng.standard_normal((N, 64)).astype(np.float32)).cuda()
g_t, b_t = torch.from_numpy(grid).cuda().int(), torch.from_numpy(batch).cuda()
# poison the allocator's free blocks so that uninitialised reads show
junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(64)]
del junk
import warnings
if os.environ.get("DET"):
    torch.use_deterministic_algorithms(True, warn_only=os.environ["DET"] == "warn")
runs, outs = [], []
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    model.zero_grad(set_to_none=True)
    point = model({"feat": feat, "coord": coord, "grid_coord": g_t, "batch": b_t})
    (point.feat * R).sum().backward()
    runs.append({k: p.grad.detach().clone() for k, p in model.named_parameters()})
    outs.append(point.feat.detach().clone())
names = [k for k, _ in model.named_parameters()]
print("forward spread:", max(float((o - outs[0]).abs().max()) for o in outs) / float(outs[0].abs().max()))
for it in range(1, len(runs)):
    d = {k: float((runs[it][k] - runs[it - 1][k]).norm() / runs[0][k].norm().clamp_min(1e-30)) for k in names}
    bad = [(k, v) for k, v in d.items() if not v < 1e-5]
    print(f"pass {it} vs pass {it - 1}: {len(bad)} of {len(names)} parameters differ by >= 1e-5; nan: {sum(1 for k in names if not torch.isfinite(runs[it][k]).all())}")
    for k, v in bad[-3:]:
        print(f"    {k:50s} {v:.2e}")
