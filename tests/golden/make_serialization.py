"""Generates tests/golden/serialization.npz by importing the reference's serialization files (pure torch, CPU) from
/root/reference -- run in the build container only; the fixture is data (inputs + the reference's outputs).
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_serialization.py"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

BASE = "/root/reference/Modules/PointTransformerV3/serialization"
pkg = types.ModuleType("refser")       # the sub-package alone: its parent's __init__ needs spconv / torch_scatter
pkg.__path__ = [BASE]
sys.modules["refser"] = pkg
for name in ("z_order", "hilbert", "default"):
    spec = importlib.util.spec_from_file_location("refser." + name, f"{BASE}/{name}.py")
    mod = importlib.util.module_from_spec(spec)
    sys.modules["refser." + name] = mod
    spec.loader.exec_module(mod)
ref = sys.modules["refser.default"]

g = torch.Generator().manual_seed(20251004)
out = {}
for depth, n in ((16, 4096), (10, 2048), (8, 1024), (5, 512), (1, 64)):
    grid = torch.randint(0, 1 << depth, (n, 3), generator=g, dtype=torch.int64)
    grid[:8] = torch.tensor([[0, 0, 0], [(1 << depth) - 1] * 3, [1, 0, 0], [0, 1, 0], [0, 0, 1], [(1 << depth) - 1, 0, 0],
                             [0, (1 << depth) - 1, 0], [0, 0, (1 << depth) - 1]]) & ((1 << depth) - 1)
    batch = torch.randint(0, 7, (n,), generator=g, dtype=torch.int64)
    out[f"grid_d{depth}"] = grid.numpy().astype(np.int32)
    out[f"batch_d{depth}"] = batch.numpy()
    for order in ("z", "z-trans", "hilbert", "hilbert-trans"):
        out[f"code_d{depth}_{order}"] = ref.encode(grid.int(), batch, depth, order).numpy()
        out[f"code_nobatch_d{depth}_{order}"] = ref.encode(grid.int(), None, depth, order).numpy()
    for order in ("z", "hilbert"):
        code = torch.from_numpy(out[f"code_d{depth}_{order}"])
        if order == "z":
            # default.decode(order="z") raises in the reference (z_order_decode unpacks three values, key2xyz returns
            # four: default.py:49 vs z_order.py:125) -- the fixture takes key2xyz itself, on the masked key like decode does
            b = code >> depth * 3
            x, y, z, _ = sys.modules["refser.z_order"].key2xyz(code & ((1 << depth * 3) - 1), depth)
            gc = torch.stack([x, y, z], dim=-1)
        else:
            gc, b = ref.decode(code, depth, order)
        out[f"decoded_grid_d{depth}_{order}"] = gc.reshape(-1, 3).numpy().astype(np.int64)
        out[f"decoded_batch_d{depth}_{order}"] = b.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "serialization.npz"), **out)
print({k: v.shape for k, v in out.items() if k.startswith("code_d16")})
