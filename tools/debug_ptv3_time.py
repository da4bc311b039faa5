"""GPU: where the PTv3 backbone's forward time goes by module type and stage (HIP events around forward hooks), at the
bench_ptv3_model.py workload."""
import os, sys, collections
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_pkg
load_pkg()
from tools.bench_ptv3_model import plot_voxels
from pn2_amd.PointTransformerV3.PointTransformerV3 import PointTransformerV3
from pn2_amd.PointTransformerV3.cpe import SubMConv3d
from pn2_amd.PointTransformerV3.attention import SerializedAttention
from pn2_amd.PointTransformerV3.blocks import MLP, SerializedPooling, SerializedUnpooling
torch.manual_seed(0)
model = PointTransformerV3(in_channels=4).cuda().eval()
g = plot_voxels(1 << 20, 0.02)
N = len(g)
grid = torch.from_numpy(g.astype(np.int32)).cuda()
data = lambda: {"feat": torch.randn(N, 4, device="cuda"), "coord": grid.float() * 0.02, "grid_coord": grid, "batch": torch.zeros(N, dtype=torch.int64, device="cuda")}
rec = []
def pre(name):
    def f(mod, inp):
        e = torch.cuda.Event(enable_timing=True); e.record(); mod._t0 = e
    return f
def post(name, kind):
    def f(mod, inp, out):
        e = torch.cuda.Event(enable_timing=True); e.record()
        rows = (inp[0].shape[0] if torch.is_tensor(inp[0]) else inp[0]["feat"].shape[0])
        rec.append((name, kind, mod._t0, e, rows))
    return f
for name, m in model.named_modules():
    kind = "conv" if isinstance(m, SubMConv3d) else "attn" if isinstance(m, SerializedAttention) else "mlp" if isinstance(m, MLP) else \
        "pool" if isinstance(m, SerializedPooling) else "unpool" if isinstance(m, SerializedUnpooling) else None
    if kind:
        m.register_forward_pre_hook(pre(name)); m.register_forward_hook(post(name, kind))
with torch.no_grad():
    model(data()); rec.clear(); model(data())
torch.cuda.synchronize()
tot = collections.defaultdict(float)
for name, kind, a, b, rows in rec:
    ms = a.elapsed_time(b)
    stage = ".".join(name.split(".")[:2])
    tot[(stage, kind, rows)] += ms
for k, v in sorted(tot.items()):
    print(f"{k[0]:12s} {k[1]:7s} rows {k[2]:8d}  {v:8.2f} ms")
by = collections.defaultdict(float)
for k, v in tot.items():
    by[k[1]] += v
print({k: round(v, 1) for k, v in by.items()})
