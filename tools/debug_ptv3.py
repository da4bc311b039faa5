"""GPU: stage-by-stage comparison of the PTv3 backbone mirror with the float64 restatement (same set-up as
tests/test_ptv3_model.py::test_backbone_forward_matches_float64_restatement)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from __graft_entry__ import load_pkg
load_pkg()
from test_ptv3_model import _cfg
from oracle import ptv3_model_port as P
from pn2_amd.PointTransformerV3.PointTransformerV3 import PointTransformerV3
from pn2_amd.PointTransformerV3.blocks import Block, SerializedPooling, SerializedUnpooling, Embedding
from pn2_amd.synthetic import gaussian_branch_tree
cfg = _cfg()
torch.manual_seed(0)
model = PointTransformerV3(**cfg).cuda().eval()
for m in model.modules():
    if hasattr(m, "shuffle_orders"):
        m.shuffle_orders = False
clouds = []
for b in range(2):
    xyz = gaussian_branch_tree(6000, seed=10 + b)[0]
    grid = np.unique(np.floor((xyz - xyz.min(0)) / 0.05).astype(np.int64), axis=0)
    clouds.append(np.concatenate([np.full((len(grid), 1), b), grid], 1))
vox = np.concatenate(clouds)
batch, grid = vox[:, 0].copy(), vox[:, 1:].copy()
N = len(grid)
feat = np.random.default_rng(2).standard_normal((N, 4)).astype(np.float32)
coord = (grid * 0.05).astype(np.float32)
got = []
def hook(name):
    def f(mod, inp, out):
        got.append((name, out.feat.detach().cpu().double().clone(), out.offset.cpu().numpy().copy()))
    return f
for name, m in model.named_modules():
    if isinstance(m, (Block, SerializedPooling, SerializedUnpooling, Embedding)):
        m.register_forward_hook(hook(name))
    if name.startswith("enc.") and (name.endswith(".cpe") or name.endswith(".attn")):
        m.register_forward_hook(hook(name))
with torch.no_grad():
    model({"feat": torch.from_numpy(feat).cuda(), "coord": torch.from_numpy(coord).cuda(), "grid_coord": torch.from_numpy(grid).cuda().int(),
           "batch": torch.from_numpy(batch).cuda()})
trace = []
P.backbone_forward(model.state_dict(), cfg, feat, coord, grid, batch, trace=trace)
gd = {n: a for n, a, _ in got}
for n2, b in trace:
    if n2 in gd and ("cpe" in n2 or "attn" in n2):
        a = gd[n2]
        print(f"   part {n2:26s} rel err {float((a - b).abs().max()) / float(b.abs().max()):.3e}")
got = [g for g in got if not (g[0].endswith(".cpe") or g[0].endswith(".attn"))]
trace = [t for t in trace if not (t[0].endswith(".cpe") or t[0].endswith(".attn"))]
for (n1, a, off), (n2, b) in zip(got, trace):
    ok = a.shape == b.shape
    err = float((a - b).abs().max()) / float(b.abs().max()) if ok else float("nan")
    print(f"{n1:22s} {n2:22s} {tuple(a.shape)} {tuple(b.shape)} offsets {off.tolist()} rel err {err:.3e}")
